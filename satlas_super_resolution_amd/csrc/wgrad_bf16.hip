// bf16 weight gradient on v_mfma_f32_32x32x16_bf16 with LDS transpose reads.
//   dW[co][ci][ky][kx] += alpha * sum_pixels dY[p][co] * X[p*S + (ky,kx) - pad][ci]     (autograd of nn.Conv2d at
//   /root/reference/ssr/archs/rrdbnet_arch.py:30-34,104-113 and discriminator_arch.py:28-40, executed by
//   l_g_total.backward() / l_d_real.backward() / l_d_fake.backward(): ssr/models/ssr_esrgan_model.py:188,219,227)
//
// wgrad contracts over PIXELS, the strided axis of NHWC, while the bf16 MFMA wants 8 consecutive
// k-values per lane.  gfx950's ds_read_b64_tr_b16 does the transpose in the LDS read path: within a
// 16-lane group, source lane 4j+q supplies 4 contiguous bf16 (channels 4q..4q+3 of pixel j) and output
// lane c receives channel c of pixels j = 0..3 (semantics measured with tools/probe_tr16.hip).  So a
// dense [pixel][32 ch] LDS image — exactly what coalesced 16-byte NHWC loads produce — feeds both MFMA
// operands directly:   A[i = co][k = pixel] <- dY tile,   B[k = pixel][j = ci] <- X halo patch + tap shift
// (the tap shift is just a different per-lane pixel address; no alignment constraint).  A 64-byte row
// (32 bf16) makes the 4 pixel rows of a 32-lane read tile the 64 banks exactly: conflict-free.
//
// Work decomposition: device tables, one (layer, 32co, 32ci, pixel-tile range) item per workgroup, all taps in
// registers.  Structure (r01, tools/wgrad_probe.hip): when the four MFMA waves also fetched their own tiles, an
// iteration took 2580 cycles for 640 cycles of MFMAs — 1330 of them the wave sitting in the ISSUE of its five
// 16-byte loads: a wave gets only ~6.4 B/clk from L2/HBM (tools/l2_probe.hip) and the tile is 4.9 KB per wave.
// So the workgroup is 8 waves:
//   * waves 4..7 are LOADERS: global -> registers (one tile ahead) -> ds_write_b128 into an LDS ring of NST stages;
//   * waves 0..3 only transpose-read and issue MFMAs, operand reads WG_PF ahead of their MFMAs;
//   * no s_barrier in the loop: per-stage `ready` counters (loaders -> MFMA waves) and per-wave `done` words
//     (MFMA waves -> loaders) in LDS.
// Write-out: partial sums of the 4 pixel-split waves are reduced through the (now idle) ring and transposed in LDS
// to the OIHW order so that the fp32 read-modify-write of dW is contiguous (it was 20 barriers and stride-36-byte
// accesses: 25k cycles per item).
#include "wgrad_common.h"

#ifdef SSR_PROBE   // tools/wgrad_probe.hip
#define GPROBE(k) do { if (threadIdx.x == 0) g_probe[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define GPROBE_K(kk) do { if (threadIdx.x == 0 && k == 20) g_probe[blockIdx.x * 16 + (kk)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GPROBE(k)
#define GPROBE_K(kk)
#endif

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_pair(const __bf16* lo, const __bf16* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// LDS flag helpers (the loaders poll through inline asm: for a volatile / atomic LDS read hipcc first drains
// vmcnt(0), i.e. waits for the prefetch loads issued a moment ago)
__device__ __forceinline__ int wg_ld(const int* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
__device__ __forceinline__ void wg_sync4(int* cnt, int& phase, int lane) {   // barrier of the four MFMA waves
    phase += 4;
    if (lane == 0) __atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED);
    while (wg_ld(cnt) < phase) {}
}

template <int KH, int KW, int S, bool SPLIT_TAPS>
struct WgCfg {
    static constexpr int TH = wgrad_bf16_th(KH);             // pixel-tile rows: 16 for 3x3, 8 for 4x4
    static constexpr int PH = (TH - 1) * S + KH, PW = (WG_TW - 1) * S + KW;
    static constexpr int NTAP = SPLIT_TAPS ? KW : KH * KW;   // accumulators per wave
    static constexpr int ROW = 32;                           // bf16 per pixel row (64 B)
    static constexpr int NDYV = TH * WG_TW * 4, NXV = PH * PW * 4;   // 16-B vectors of a stage: dY tile, X patch
    static constexpr int NLV = (NDYV + NXV + 255) / 256;     // vectors per loader thread
    static constexpr int STAGE = (NDYV + NXV) * 16;          // bytes
    static constexpr int NST = (156 * 1024) / STAGE < 4 ? (156 * 1024) / STAGE : 4;
    static constexpr int CTL = NST * STAGE;                  // control words behind the ring
    static constexpr int LDS = CTL + 256;
    static_assert(NST >= 2 && LDS <= 160 * 1024, "LDS budget");
};
constexpr int WGC_READY = 0;    // [NST] loader waves that have stored their part of the stage's current tile
constexpr int WGC_DONE = 8;     // [4] tiles MFMA wave w is finished with
constexpr int WGC_SYNC = 12;    // write-out barrier counter
constexpr int WGC_LSYNC = 13;   // loader-wave barrier counter
constexpr int WGC_BIAS = 16;    // [32] floats: bias-gradient partial sums of the loader threads

template <int KH, int KW, int S, bool SPLIT_TAPS>
__global__ __launch_bounds__(512) void wgrad_bf16_kernel(const ssr_wgrad_layer* __restrict__ layers,
                                                         const ssr_wgrad_item* __restrict__ items) {
    using C = WgCfg<KH, KW, S, SPLIT_TAPS>;
    constexpr int TH = C::TH, PW = C::PW, NTAP = C::NTAP, ROW = C::ROW, NST = C::NST;
    static_assert(!SPLIT_TAPS || KH == 4, "tap-row split assumes 4 waves = KH");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* ctl = reinterpret_cast<int*>(smem + C::CTL);

    const ssr_wgrad_item it = items[blockIdx.x];
    const ssr_wgrad_layer L = layers[it.layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (L.Gw + WG_TW - 1) / WG_TW, tiles_y = (L.Gh + TH - 1) / TH;
    const int ntile = it.tile_end - it.tile_begin;
    if (tid < 64) ctl[tid] = 0;
    GPROBE(0);
    __syncthreads();   // the only barrier

    if (wave >= 4) {
        // =============================== loader waves ===============================
        const int lt = tid - 256;
        const int upshift = L.up == 2 ? 1 : 0;
        const int LH = L.Hi << upshift, LW = L.Wi << upshift;
        const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(L.x.p);
        const __bf16* __restrict__ dyg = reinterpret_cast<const __bf16*>(L.dy.p);
        // Everything that does not depend on the tile is computed once per thread: the loaders must sustain a 37 KB tile
        // per ~1300 cycles and were otherwise bound by their own address arithmetic (two divisions per vector).
        static_assert(C::NDYV % 256 == 0, "vector q of a loader thread is a dY vector iff q < NDYV / 256");
        constexpr int QDY = C::NDYV / 256;
        int rel[C::NLV], yx[C::NLV];                           // global offset relative to the tile origin; (y, x) in the tile / patch
#pragma unroll
        for (int q = 0; q < C::NLV; ++q) {
            const int v = lt + q * 256;
            if (q < QDY) {
                const int pix = v >> 2, part = v & 3;
                const int y = pix >> 4, x = pix & 15;
                rel[q] = (y * L.Gw + x) * L.dy.cs + part * 8;
                yx[q] = (it.co0 + part * 8 < L.Cout) ? (y | (x << 16)) : 0x7fff7fff;   // never inside
            } else {
                const int vx = v - C::NDYV;
                const int pix = vx >> 2, part = vx & 3;
                const int py = pix / PW, px = pix - py * PW;
                const int y = py - L.pad_y, x = px - L.pad_x;
                rel[q] = ((y >> upshift) * L.Wi + (x >> upshift)) * L.x.cs + part * 8;
                yx[q] = (vx < C::NXV && it.ci0 + part * 8 < L.Cin) ? ((y & 0xffff) | (x << 16)) : 0x7fff7fff;
            }
        }
        // bias gradient db[co] = sum over pixels of dY: the loaders see every dY vector anyway (8 channels, always the
        // same 8 for a given thread), so the MFMA waves carry no `ones` product and no 17th accumulator
        const bool do_bias = L.db != nullptr && it.ci0 == 0;
        float bacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        u32x4 ra[C::NLV], rb[C::NLV];
        auto load_tile = [&](int k, u32x4 (&r)[C::NLV]) {
            int b = it.tile_begin + k;
            const int tx_i = b % tiles_x; b /= tiles_x;
            const int ty_i = b % tiles_y;
            const int n = b / tiles_y;
            const int gy0 = ty_i * TH, gx0 = tx_i * WG_TW;
            const __bf16* dyb = dyg + ((size_t)(n * L.Gh + gy0) * L.Gw + gx0) * L.dy.cs + L.dy.coff + it.co0;
            const __bf16* xb = xg + ((size_t)(n * L.Hi + ((gy0 * S) >> upshift)) * L.Wi + ((gx0 * S) >> upshift)) * L.x.cs +
                               L.x.coff + it.ci0;
#pragma unroll
            for (int q = 0; q < C::NLV; ++q) {
                const int y = (int)(short)(yx[q] & 0xffff), x = yx[q] >> 16;
                u32x4 val = {0u, 0u, 0u, 0u};
                if (q < QDY) {
#ifndef WG_X_NOLOAD
                    if (gy0 + y < L.Gh && gx0 + x < L.Gw) val = *reinterpret_cast<const u32x4*>(dyb + rel[q]);
#endif
                } else {
#ifndef WG_X_NOLOAD
                    if ((unsigned)(gy0 * S + y) < (unsigned)LH && (unsigned)(gx0 * S + x) < (unsigned)LW)
                        val = *reinterpret_cast<const u32x4*>(xb + rel[q]);
#endif
                }
                r[q] = val;
            }
        };
        // tile k -> stage k % NST once every MFMA wave is finished with tile k - NST; LDS operations of a wave execute
        // in order, so the counter increment follows the data
        auto put = [&](int k, int st, const u32x4 (&r)[C::NLV]) {
            if (k >= NST) {
                for (;;) {
                    u32x4 dn;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dn) : "v"((int)(C::CTL + 4 * WGC_DONE)) : "memory");
                    if ((int)min(min(dn[0], dn[1]), min(dn[2], dn[3])) >= k - NST + 1) break;
                }
            }
            char* base = smem + st * C::STAGE;
#pragma unroll
            for (int q = 0; q < C::NLV; ++q) {
                const int v = lt + q * 256;
#ifndef WG_X_NOWRITE
                if (v < C::NDYV + C::NXV) *reinterpret_cast<u32x4*>(base + v * 16) = r[q];
#else
                if (r[q][0] == 0x12345u) *reinterpret_cast<u32x4*>(base + v * 16) = r[q];
#endif
            }
            if (do_bias) {
#pragma unroll
                for (int q = 0; q < QDY; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bacc[2 * e] += __builtin_bit_cast(float, r[q][e] << 16);
                        bacc[2 * e + 1] += __builtin_bit_cast(float, r[q][e] & 0xffff0000u);
                    }
            }
            if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"((int)(C::CTL + 4 * (WGC_READY + st))), "v"(1) : "memory");
        };
        if (ntile > 0) load_tile(0, ra);
        int st = 0;
        for (int k = 0; k < ntile; k += 2) {
            if (k + 1 < ntile) load_tile(k + 1, rb);
            put(k, st, ra);
            st = st + 1 == NST ? 0 : st + 1;
            if (k + 2 < ntile) load_tile(k + 2, ra);
            if (k + 1 < ntile) {
                put(k + 1, st, rb);
                st = st + 1 == NST ? 0 : st + 1;
            }
        }
        if (do_bias) {   // 64 threads share each channel octet: LDS float atomics, then one global atomic per channel
            float* bl = reinterpret_cast<float*>(ctl + WGC_BIAS);
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(bl + (lt & 3) * 8 + e, bacc[e]);
            if (lane == 0) __atomic_fetch_add(ctl + WGC_LSYNC, 1, __ATOMIC_RELAXED);
            while (wg_ld(ctl + WGC_LSYNC) < 4) {}
            if (lt < 32 && it.co0 + lt < L.Cout) atomicAdd(L.db + it.co0 + lt, L.alpha * bl[lt]);
        }
        return;
    }

    // =============================== MFMA waves ===============================
    const int g = lane >> 5;
    // transpose-read source role of this lane: pixel j (0..3) of the 4-pixel group, channel quad
    const int t16 = lane & 15;
    const int src_px = 8 * g + (t16 >> 2);                    // + 4 for the second read of the pair
    const int src_ch = 16 * ((lane >> 4) & 1) + 4 * (t16 & 3);

    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    int st = 0, target = 4;                                    // stage of tile k and its ready target 4 * (uses + 1)
    for (int k = 0; k < ntile; ++k) {
        GPROBE_K(2);
        while (wg_ld(ctl + WGC_READY + st) < target) {}
        GPROBE_K(3);
        const __bf16* ldy = reinterpret_cast<const __bf16*>(smem + st * C::STAGE);
        const __bf16* lx = ldy + TH * WG_TW * ROW;
        constexpr int NROW = SPLIT_TAPS ? TH : TH / 4;        // tile rows (16-pixel k-steps) per wave
        // Operand stream of this tile: per k-step the dY fragment, then one X fragment per tap.  Reads run WG_PF
        // operands ahead of the MFMAs; the sched_barrier fences pin that order.
        constexpr int NOP = NROW * (1 + NTAP), WG_PF = 8;   // LDS latency under load is several MFMAs long
        bf16x8 op[NOP];
        auto issue = [&](auto nc) {
            constexpr int n = decltype(nc)::value, s = n / (1 + NTAP), r = n % (1 + NTAP);
            const int ty = SPLIT_TAPS ? s : NROW * wave + s;
#ifdef WG_X_NOREAD
            if constexpr (true) { op[n] = __builtin_bit_cast(bf16x8, u32x4{0x3c003c00u + (unsigned)lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}); (void)ty; } else
#endif
            if constexpr (r == 0) {
                const __bf16* ap = ldy + (ty * WG_TW + src_px) * ROW + src_ch;
                op[n] = tr_pair(ap, ap + 4 * ROW);
            } else {
                constexpr int t = r - 1;
                const int ky = SPLIT_TAPS ? wave : t / KW, kx = SPLIT_TAPS ? t : t % KW;
                const __bf16* bp = lx + ((ty * S + ky) * PW + src_px * S + kx) * ROW + src_ch;
                op[n] = tr_pair(bp, bp + 4 * S * ROW);
            }
        };
        static_for<0, WG_PF>([&](auto nc) { issue(nc); });
        static_for<0, NOP>([&](auto nc) {
            constexpr int n = decltype(nc)::value, s = n / (1 + NTAP), r = n % (1 + NTAP);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (n + WG_PF < NOP) issue(std::integral_constant<int, n + WG_PF>{});
            if constexpr (n + WG_PF == NOP - 1) {
                // every read of this stage has been issued: hand it back (the LDS executes this wave's operations in order)
                if (lane == 0) __atomic_store_n(ctl + WGC_DONE + wave, k + 1, __ATOMIC_RELAXED);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef WG_X_NOMFMA
            if constexpr (r == 1)
#else
            if constexpr (r != 0)
#endif
                acc[r - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[s * (1 + NTAP)], op[n], acc[r - 1], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        GPROBE_K(4);
        if (st + 1 == NST) { st = 0; target += 4; } else ++st;
    }
    GPROBE(7);

    // =============================== write-out ===============================
    // partial sums -> LDS [part][tap][16 regs][64 lanes] -> per-thread sums -> LDS out tile [co][ci][tap] -> contiguous
    // fp32 read-modify-write of dW rows (each co row of the tile is 32 * KH*KW consecutive floats)
    constexpr int KK = KH * KW, NPART = SPLIT_TAPS ? 1 : 4;
    constexpr int NOUT = 32 * 32 * KK, PER_T = NOUT / 256;
    static_assert(NPART * KK * 16 * 64 * 4 <= C::CTL && NOUT * 4 <= C::CTL, "reduction scratch fits in the ring");
    float* red = reinterpret_cast<float*>(smem);
    int phase = 0;
    wg_sync4(ctl + WGC_SYNC, phase, lane);   // all four waves are finished reading the ring
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int slot = SPLIT_TAPS ? wave * KW + t : wave * KK + t;   // tap-split: slot = tap; pixel-split: part*KK + tap
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(slot * 16 + r) * 64 + lane] = acc[t][r];
    }
    wg_sync4(ctl + WGC_SYNC, phase, lane);
    float sum[PER_T];
#pragma unroll
    for (int q = 0; q < PER_T; ++q) {
        const int e = tid + q * 256;                           // ci fastest: conflict-free reads
        const int ci = e & 31, tap = (e >> 5) % KK, co = e / (32 * KK);
        const int r = (co & 3) + 4 * (co >> 3), ln = ((co >> 2) & 1) * 32 + ci;
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < NPART; ++p) s += red[((p * KK + tap) * 16 + r) * 64 + ln];
        sum[q] = L.alpha * s;
    }
    wg_sync4(ctl + WGC_SYNC, phase, lane);   // every partial has been read
#pragma unroll
    for (int q = 0; q < PER_T; ++q) {
        const int e = tid + q * 256;
        const int ci = e & 31, tap = (e >> 5) % KK, co = e / (32 * KK);
        red[(co * 32 + ci) * KK + tap] = sum[q];               // stride KK (odd or 16) over ci
    }
    wg_sync4(ctl + WGC_SYNC, phase, lane);
    float* __restrict__ dw = L.dw;
    const int nci = min(32, L.Cin_w - it.ci0);                 // valid ci of this tile
    // fp32 atomic adds throughout (returnless: fire and forget).  A plain `dw[idx] += v` made every element a
    // serialized load -> add -> store round trip (hipcc waits vmcnt(0) between them): 30k cycles per item.
#pragma unroll
    for (int q = 0; q < PER_T; ++q) {
        const int e = tid + q * 256;                           // (co, ci*KK + tap): contiguous in dW per co row
        const int co = e / (32 * KK), rem = e - co * (32 * KK);
        if (it.co0 + co < L.Cout && rem < nci * KK)
            atomicAdd(dw + ((size_t)(it.co0 + co) * L.Cin_w + it.ci0) * KK + rem, red[e]);
    }
    GPROBE(8);
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 stride 1: 32 co x 64 ci per workgroup, the 18 (tap, 32-channel half) units dealt to the four MFMA waves.
// Why (tools/wgrad_probe.hip): with 32 x 32 tiles the loaders had to deliver a 37 KB tile per 1280 MFMA cycles and got
// 16.8 B/clk — they fetch 64-byte slices of 384-byte NHWC pixels (half of every 128-byte line is wasted) — so the
// MFMA waves idled half the time.  A 64-channel X patch is whole lines, the dY tile is shared by twice the MFMAs
// (22 B/clk needed), every wave sees ALL pixels of the tile for its 4..5 units (80 accumulator registers, no cross-wave
// reduction at the end), and the X patch sits in LDS as two dense 64-byte-row planes (conflict-free transpose reads).
// ------------------------------------------------------------------------------------------------------------------
struct Wg3 {
    static constexpr int TH = 16, PH = 18, PW = 18, ROW = 32;
    static constexpr int NDYV = TH * WG_TW * 4;              // 1024 vectors: dY tile [256 px][32 co]
    static constexpr int NXV = PH * PW * 8;                  // 2592 vectors: X patch [324 px][64 ci]
    static constexpr int NLV = (NDYV + NXV + 255) / 256;     // 15 per loader thread
    static constexpr int XPLANE = PH * PW * 64;              // bytes of one 32-channel plane
    static constexpr int STAGE = NDYV * 16 + 2 * XPLANE;     // 57,856 B
    static constexpr int NST = 2;
    static constexpr int CTL = NST * STAGE;
    static constexpr int LDS = CTL + 256;
    static_assert(LDS <= 160 * 1024 && 32 * 64 * 9 * 4 <= CTL, "LDS budget / write-out tile fits in the ring");
};

template <int NU>
__device__ __forceinline__ void wg3_contract(f32x16 (&acc)[5], const __bf16* ldy, const int (&uoff)[5], int src_px, int src_ch,
                                             int* done_word, int k, int lane) {
    constexpr int NOP = Wg3::TH * (1 + NU), PF = 8;
    bf16x8 op[NOP];
    const __bf16* lxb = ldy + Wg3::NDYV * 8;                  // X planes behind the dY tile
    auto issue = [&](auto nc) {
        constexpr int n = decltype(nc)::value, s = n / (1 + NU), r = n % (1 + NU);
        if constexpr (r == 0) {
            const __bf16* ap = ldy + (s * WG_TW + src_px) * Wg3::ROW + src_ch;
            op[n] = tr_pair(ap, ap + 4 * Wg3::ROW);
        } else {
            const __bf16* bp = lxb + uoff[r - 1] + (s * Wg3::PW + src_px) * Wg3::ROW + src_ch;
            op[n] = tr_pair(bp, bp + 4 * Wg3::ROW);
        }
    };
    static_for<0, PF>([&](auto nc) { issue(nc); });
    static_for<0, NOP>([&](auto nc) {
        constexpr int n = decltype(nc)::value, s = n / (1 + NU), r = n % (1 + NU);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF < NOP) issue(std::integral_constant<int, n + PF>{});
        if constexpr (n + PF == NOP - 1) {
            if (lane == 0) __atomic_store_n(done_word, k + 1, __ATOMIC_RELAXED);   // every read of the stage is issued
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (r != 0)
            acc[r - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[s * (1 + NU)], op[n], acc[r - 1], 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512) void wgrad_bf16_k3_kernel(const ssr_wgrad_layer* __restrict__ layers,
                                                            const ssr_wgrad_item* __restrict__ items) {
    using C = Wg3;
    constexpr int TH = C::TH, PW = C::PW, NST = C::NST;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* ctl = reinterpret_cast<int*>(smem + C::CTL);
    const ssr_wgrad_item it = items[blockIdx.x];
    const ssr_wgrad_layer L = layers[it.layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (L.Gw + WG_TW - 1) / WG_TW, tiles_y = (L.Gh + TH - 1) / TH;
    const int ntile = it.tile_end - it.tile_begin;
    if (tid < 64) ctl[tid] = 0;
    GPROBE(0);
    __syncthreads();   // the only barrier

    if (wave >= 4) {
        // =============================== loader waves ===============================
        const int lt = tid - 256;
        const int upshift = L.up == 2 ? 1 : 0;
        const int LH = L.Hi << upshift, LW = L.Wi << upshift;
        const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(L.x.p);
        const __bf16* __restrict__ dyg = reinterpret_cast<const __bf16*>(L.dy.p);
        constexpr int QDY = C::NDYV / 256;                     // vector q of a thread is a dY vector iff q < QDY
        int rel[C::NLV], yx[C::NLV], lo[C::NLV];               // global offset rel. to the tile origin, (y, x), LDS offset
#pragma unroll
        for (int q = 0; q < C::NLV; ++q) {
            const int v = lt + q * 256;
            if (q < QDY) {
                const int pix = v >> 2, part = v & 3;
                const int y = pix >> 4, x = pix & 15;
                rel[q] = (y * L.Gw + x) * L.dy.cs + part * 8;
                yx[q] = (it.co0 + part * 8 < L.Cout) ? (y | (x << 16)) : 0x7fff7fff;   // never inside
                lo[q] = v * 16;
            } else {
                const int vx = v - C::NDYV;
                const int pix = vx >> 3, part = vx & 7;        // 8 x 16 B = the 64 channels of one pixel: a whole line
                const int py = pix / PW, px = pix - py * PW;
                const int y = py - L.pad_y, x = px - L.pad_x;
                rel[q] = ((y >> upshift) * L.Wi + (x >> upshift)) * L.x.cs + part * 8;
                yx[q] = (vx < C::NXV && it.ci0 + part * 8 < L.Cin) ? ((y & 0xffff) | (x << 16)) : 0x7fff7fff;
                lo[q] = vx < C::NXV ? C::NDYV * 16 + (part >> 2) * C::XPLANE + pix * 64 + (part & 3) * 16 : -1;
            }
        }
        const bool do_bias = L.db != nullptr && it.ci0 == 0;
        float bacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        u32x4 ra[C::NLV], rb[C::NLV];
        auto load_tile = [&](int k, u32x4 (&r)[C::NLV]) {
            int b = it.tile_begin + k;
            const int tx_i = b % tiles_x; b /= tiles_x;
            const int ty_i = b % tiles_y;
            const int n = b / tiles_y;
            const int gy0 = ty_i * TH, gx0 = tx_i * WG_TW;
            const __bf16* dyb = dyg + ((size_t)(n * L.Gh + gy0) * L.Gw + gx0) * L.dy.cs + L.dy.coff + it.co0;
            const __bf16* xb = xg + ((size_t)(n * L.Hi + (gy0 >> upshift)) * L.Wi + (gx0 >> upshift)) * L.x.cs + L.x.coff + it.ci0;
#pragma unroll
            for (int q = 0; q < C::NLV; ++q) {
                const int y = (int)(short)(yx[q] & 0xffff), x = yx[q] >> 16;
                u32x4 val = {0u, 0u, 0u, 0u};
                if (q < QDY) {
                    if (gy0 + y < L.Gh && gx0 + x < L.Gw) val = *reinterpret_cast<const u32x4*>(dyb + rel[q]);
                } else {
                    if ((unsigned)(gy0 + y) < (unsigned)LH && (unsigned)(gx0 + x) < (unsigned)LW)
                        val = *reinterpret_cast<const u32x4*>(xb + rel[q]);
                }
                r[q] = val;
            }
        };
        auto put = [&](int k, int st, const u32x4 (&r)[C::NLV]) {
            if (k >= NST) {
                for (;;) {
                    u32x4 dn;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dn) : "v"((int)(C::CTL + 4 * WGC_DONE)) : "memory");
                    if ((int)min(min(dn[0], dn[1]), min(dn[2], dn[3])) >= k - NST + 1) break;
                }
            }
            char* base = smem + st * C::STAGE;
#pragma unroll
            for (int q = 0; q < C::NLV; ++q)
                if (lo[q] >= 0) *reinterpret_cast<u32x4*>(base + lo[q]) = r[q];
            if (do_bias) {
#pragma unroll
                for (int q = 0; q < QDY; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bacc[2 * e] += __builtin_bit_cast(float, r[q][e] << 16);
                        bacc[2 * e + 1] += __builtin_bit_cast(float, r[q][e] & 0xffff0000u);
                    }
            }
            if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"((int)(C::CTL + 4 * (WGC_READY + st))), "v"(1) : "memory");
        };
        if (ntile > 0) load_tile(0, ra);
        int st = 0;
        for (int k = 0; k < ntile; k += 2) {
            if (k + 1 < ntile) load_tile(k + 1, rb);
            put(k, st, ra);
            st ^= 1;
            if (k + 2 < ntile) load_tile(k + 2, ra);
            if (k + 1 < ntile) {
                put(k + 1, st, rb);
                st ^= 1;
            }
        }
        if (do_bias) {
            float* bl = reinterpret_cast<float*>(ctl + WGC_BIAS);
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(bl + (lt & 3) * 8 + e, bacc[e]);
            if (lane == 0) __atomic_fetch_add(ctl + WGC_LSYNC, 1, __ATOMIC_RELAXED);
            while (wg_ld(ctl + WGC_LSYNC) < 4) {}
            if (lt < 32 && it.co0 + lt < L.Cout) atomicAdd(L.db + it.co0 + lt, L.alpha * bl[lt]);
        }
        return;
    }

    // =============================== MFMA waves ===============================
    const int g = lane >> 5;
    const int t16 = lane & 15;
    const int src_px = 8 * g + (t16 >> 2);
    const int src_ch = 16 * ((lane >> 4) & 1) + 4 * (t16 & 3);
    // units u = tap + 9 * half; wave w owns u = w, w + 4, ...; a tile with <= 32 valid input channels has no second half
    const int nci = min(64, L.Cin_w - it.ci0);
    const int nunit = nci > 32 ? 18 : 9;
    int uoff[5], utap[5], usub[5];
    int nu = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int u = wave + 4 * j;
        const bool ok = u < nunit;
        const int uu = ok ? u : wave;                          // harmless duplicate address for unused slots
        utap[j] = uu % 9; usub[j] = uu / 9;
        uoff[j] = usub[j] * (C::XPLANE / 2) + ((utap[j] / 3) * PW + utap[j] % 3) * C::ROW;   // bf16 elements
        nu += ok ? 1 : 0;
    }
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int st = 0, target = 4;
    for (int k = 0; k < ntile; ++k) {
        GPROBE_K(2);
        while (wg_ld(ctl + WGC_READY + st) < target) {}
        GPROBE_K(3);
        const __bf16* ldy = reinterpret_cast<const __bf16*>(smem + st * C::STAGE);
        int* dw_ = ctl + WGC_DONE + wave;
        if (nu == 5) wg3_contract<5>(acc, ldy, uoff, src_px, src_ch, dw_, k, lane);
        else if (nu == 4) wg3_contract<4>(acc, ldy, uoff, src_px, src_ch, dw_, k, lane);
        else if (nu == 3) wg3_contract<3>(acc, ldy, uoff, src_px, src_ch, dw_, k, lane);
        else wg3_contract<2>(acc, ldy, uoff, src_px, src_ch, dw_, k, lane);
        GPROBE_K(4);
        if (st == 1) target += 4;
        st ^= 1;
    }
    GPROBE(7);
    // =============================== write-out ===============================
    // every wave holds complete sums for its units: D[row = co][col = ci] -> LDS tile [co][64 ci][9 taps] (stride 9 floats
    // between lanes: conflict-free) -> contiguous fp32 atomic adds (each co row of the tile is 64 * 9 consecutive floats)
    float* red = reinterpret_cast<float*>(smem);
    int phase = 0;
    wg_sync4(ctl + WGC_SYNC, phase, lane);   // all four waves are finished reading the ring
    const int i = lane & 31;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        if (j < nu) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[(mfma32_row(r, g) * 64 + usub[j] * 32 + i) * 9 + utap[j]] = L.alpha * acc[j][r];
        }
    }
    wg_sync4(ctl + WGC_SYNC, phase, lane);
    float* __restrict__ dw = L.dw;
    constexpr int NOUT = 32 * 64 * 9;
#pragma unroll 8
    for (int q = 0; q < NOUT / 256; ++q) {
        const int e = tid + q * 256;
        const int co = e / 576, rem = e - co * 576;
        if (it.co0 + co < L.Cout && rem < nci * 9)
            atomicAdd(dw + ((size_t)(it.co0 + co) * L.Cin_w + it.ci0) * 9 + rem, red[e]);
    }
    GPROBE(8);
}

int launch_k3(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16_k3_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wg3::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(wgrad_bf16_k3_kernel, dim3(n_items), dim3(512), Wg3::LDS, st, layers, items);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

template <int KH, int KW, int S, bool SPLIT>
int launch(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, hipStream_t st) {
    using C = WgCfg<KH, KW, S, SPLIT>;
    auto kern = wgrad_bf16_kernel<KH, KW, S, SPLIT>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_items), dim3(512), C::LDS, st, layers, items);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

}  // namespace

int ssr_wgrad_bf16_dispatch(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, int KH, int KW,
                            int S, hipStream_t st) {
    if (KH == 3 && KW == 3 && S == 1) return launch_k3(layers, items, n_items, st);
    if (KH == 4 && KW == 4 && S == 2) return launch<4, 4, 2, true>(layers, items, n_items, st);
    return SSR_EUNSUP;
}
