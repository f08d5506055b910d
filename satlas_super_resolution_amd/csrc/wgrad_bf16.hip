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
// Work decomposition (generic kernel below; the 3x3 stride-1 kernel further down has its own): device tables, one (layer, 32co,
// 32ci, pixel-tile range) item per workgroup, all taps in registers.  Structure (r01, tools/wgrad_probe.hip): when the four MFMA waves also fetched their own tiles, an
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
// device-wide 100-MHz clock + where the workgroup ran (s_memtime is per XCD): the launch's timeline across CUs
#define GPROBE_RT(kk) do { if (threadIdx.x == 0) { g_probe[blockIdx.x * 16 + (kk)] = __builtin_amdgcn_s_memrealtime(); \
    g_probe[blockIdx.x * 16 + 11] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); } } while (0)
#else
#define GPROBE_RT(kk)
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

typedef const __attribute__((address_space(1))) char* wg_gptr;          // global memory, explicitly: never a flat access
typedef const __attribute__((address_space(1))) u32x4* wg_gvec;
__device__ const u32x4 g_wg_zero16 = {0u, 0u, 0u, 0u};                  // where the loads of lanes outside the image go

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
        // hand-issued, hand-waited global loads, every lane loads (see wgrad_bf16_k3_kernel): one tile stays in flight across the
        // LDS stores of the previous one
        const wg_gptr zero16 = (wg_gptr)&g_wg_zero16;
        const wg_gptr dyg1 = (wg_gptr)dyg, xg1 = (wg_gptr)xg;
        auto load_tile = [&](int k, u32x4 (&r)[C::NLV]) {
            int b = it.tile_begin + k;
            const int tx_i = b % tiles_x; b /= tiles_x;
            const int ty_i = b % tiles_y;
            const int n = b / tiles_y;
            const int gy0 = ty_i * TH, gx0 = tx_i * WG_TW;
            const wg_gptr dyb = dyg1 + (((size_t)(n * L.Gh + gy0) * L.Gw + gx0) * L.dy.cs + L.dy.coff + it.co0) * 2;
            const wg_gptr xb = xg1 + (((size_t)(n * L.Hi + ((gy0 * S) >> upshift)) * L.Wi + ((gx0 * S) >> upshift)) * L.x.cs +
                                      L.x.coff + it.ci0) * 2;
#pragma unroll
            for (int q = 0; q < C::NLV; ++q) {
                const int y = (int)(short)(yx[q] & 0xffff), x = yx[q] >> 16;
                wg_gptr src;
                if (q < QDY) src = (gy0 + y < L.Gh && gx0 + x < L.Gw) ? dyb + rel[q] * 2 : zero16;
                else src = ((unsigned)(gy0 * S + y) < (unsigned)LH && (unsigned)(gx0 * S + x) < (unsigned)LW) ? xb + rel[q] * 2 : zero16;
#ifndef WG_X_NOLOAD
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[q]) : "v"(src) : "memory");
#else
                r[q] = u32x4{(unsigned)(size_t)src, 0u, 0u, 0u};
#endif
            }
        };
        auto wait_tile = [&](u32x4 (&r)[C::NLV]) {   // every load older than the newest NLV has landed
            static_assert(C::NLV == 12, "operand list and vmcnt below");
#ifndef WG_X_NOLOAD
            asm volatile("s_waitcnt vmcnt(12)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]),
                         "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]) :: "memory");
#endif
        };
        // tile k -> stage k % NST once every MFMA wave is finished with tile k - NST; LDS operations of a wave execute
        // in order, so the counter increment follows the data
        auto put = [&](int k, int st, u32x4 (&r)[C::NLV]) {
            wait_tile(r);
            if (k >= NST) {
                for (;;) {
                    u32x4 dn;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dn) : "v"((int)(C::CTL + 4 * WGC_DONE)) : "memory");
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
        // two tiles in flight, no branch around a load (past the end the last tile is loaded again and never stored)
        const int last = ntile - 1;
        if (ntile > 0) {
            load_tile(0, ra);
            load_tile(min(1, last), rb);
        }
        int st = 0, k = 0;
        for (; k + 1 < ntile; k += 2) {
            put(k, st, ra);
            st = st + 1 == NST ? 0 : st + 1;
            load_tile(min(k + 2, last), ra);
            put(k + 1, st, rb);
            st = st + 1 == NST ? 0 : st + 1;
            load_tile(min(k + 3, last), rb);
        }
        if (k < ntile) put(k, st, ra);
        {   // the loads past the end: their registers stay allocated until they have landed
            auto hold = [&](u32x4 (&r)[C::NLV]) {
                asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                             "+v"(r[9]), "+v"(r[10]), "+v"(r[11]));
            };
            if (ntile > 0) {
                hold(ra); hold(rb);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                hold(ra); hold(rb);
            }
        }
        if (do_bias) {
            // 64 threads share each channel octet (lane & 3).  A FIXED-ORDER sum (round 4; LDS float atomics before: arrival order):
            // xor-shuffle tree over the 16 lanes of a wave that hold the octet, then the four loader waves add their sums one after
            // the other (LSYNC is the turn counter), then one global add per channel (one item per db element and launch unless
            // the layer is split over pixel ranges - the deterministic mode gives every split its own db)
            volatile float* bl = reinterpret_cast<volatile float*>(ctl + WGC_BIAS);
#pragma unroll
            for (int m = 4; m < 64; m <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) bacc[e] += __shfl_xor(bacc[e], m);
            const int lw = lt >> 6;
            while (wg_ld(ctl + WGC_LSYNC) < lw) {}
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lane < 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bl[lane * 8 + e] = lw == 0 ? bacc[e] : bl[lane * 8 + e] + bacc[e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __atomic_fetch_add(ctl + WGC_LSYNC, 1, __ATOMIC_RELAXED);
            while (wg_ld(ctl + WGC_LSYNC) < 4) {}
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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
// 3x3 stride 1: (32 or 64) co x 64 ci per workgroup; an MFMA wave owns one 32-co plane of dY, one 32-ci half of the X patch and
// all nine taps (wg3_rows).  How it got there, each step measured with tools/wgrad_probe.hip / tools/wgrad_body_probe.hip:
//  * 64 ci: with 32 x 32 tiles the loaders had to deliver a 37 KB tile per 1280 MFMA cycles and got 16.8 B/clk - they fetch
//    64-byte slices of 384-byte NHWC pixels (half of every 128-byte line is wasted).  A 64-channel X patch is whole lines and
//    sits in LDS as two dense 64-byte-row planes.
//  * 64 co (round 3): at 32 co the kernel took ~4400 cycles per tile with EITHER the global loads or 4/5 of the MFMAs removed
//    (3980 / 3940): the LDS pipe was the bound - a wave read 1 dY + 5 X fragments (1 KB each as two ds_read_b64_tr_b16) for 5
//    MFMAs.  With a SECOND 32-channel dY plane in the item every X fragment feeds two MFMAs.  The two planes may belong to
//    different layers that read the same input (a dense block's conv1..conv4 share x and write dpre1..4; conv5's 64 outputs
//    are the two halves of one layer): the host pairs items (engine.WgradBatch._pair).
//  * rolling rows (round 3): one X fragment serves three k-steps - 4 LDS reads per 9 MFMAs instead of 7 per 10 (wg3_rows).
//  * hand-waited loads (round 3): the loaders had no load in flight while they stored (see the loader branch).
//  Paired item: 6600 -> 5900 cycles per 36-product tile; a dense block's 14 items became 6 pairs + 2 singles; 3x3 weight
//  gradients 2.56 -> 1.95 ms per step.  What remains: one MFMA wave per SIMD issues an MFMA per ~35 cycles (5040 per tile).
// ------------------------------------------------------------------------------------------------------------------
#ifndef WG3_PF
#define WG3_PF 6      // operand fragments read ahead of their MFMAs
#endif
struct Wg3 {
    static constexpr int TH = 16, PH = 18, PW = 18, ROW = 32;
    static constexpr int DYPLANE = TH * WG_TW * 64;          // bytes of one dY plane [256 px][32 co]
    static constexpr int NDYV = TH * WG_TW * 8;              // 2048 vectors: two planes
    static constexpr int NXV = PH * PW * 8;                  // 2592 vectors: X patch [324 px][64 ci]
    static constexpr int NLV = (NDYV + NXV + 255) / 256;     // 19 per loader thread
    static constexpr int XPLANE = PH * PW * 64;              // bytes of one 32-channel plane
    static constexpr int STAGE = 2 * DYPLANE + 2 * XPLANE;   // 74,240 B
    static constexpr int NST = 2;
    static constexpr int CTL = NST * STAGE;
    static constexpr int LDS = CTL + 512;
    static_assert(LDS <= 160 * 1024 && 32 * 64 * 9 * 4 <= CTL, "LDS budget / write-out tile fits in the ring");
};
constexpr int W3C_BIAS = 16;
    // [64] floats: bias-gradient partial sums of the loader threads, both planes

// One wave's share of a tile: NR tile rows (k-steps of 16 pixels), ONE dY plane, ONE 32-channel half of the X patch, all nine
// taps.  The X fragment of patch row r and column shift kx serves three k-steps (tile row r with ky = 0, r - 1 with ky = 1,
// r - 2 with ky = 2), so the wave keeps a rolling window of three patch rows in registers and reads per k-step only the
// new row (3 fragments) and the dY fragment: 4 reads for 9 MFMAs.  (Units dealt by (tap, half) with all rows per wave needed
// 7 reads for 9-10 MFMAs and the LDS pipe — 128 B/clk for ds_read_b64_tr_b16 — was as busy as the MFMA pipe: 6600 cycles
// per 36-product tile for 4600 of MFMAs; tools/wgrad_body_probe.hip.)
//   lap: this lane's source address in the wave's dY plane at its first row; lbp: same in its X half plane.
template <int NR>
__device__ __forceinline__ void wg3_rows(f32x16 (&acc)[9], const __bf16* lap, const __bf16* lbp, int* done_word, int k, int lane) {
    constexpr int ROW = Wg3::ROW, PW = Wg3::PW;
    constexpr int NOP = 6 + 4 * NR, PF = WG3_PF;              // read stream: patch rows 0, 1; then per k-step dY, patch row i + 2
    static_assert(NOP - 1 - PF >= 0, "prefetch distance");
    bf16x8 op[NOP];
    auto issue = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        if constexpr (n < 6) {
            const __bf16* bp = lbp + ((n / 3) * PW + n % 3) * ROW;
            op[n] = tr_pair(bp, bp + 4 * ROW);
        } else if constexpr ((n - 6) % 4 == 0) {
            const __bf16* ap = lap + ((n - 6) / 4) * WG_TW * ROW;
            op[n] = tr_pair(ap, ap + 4 * ROW);
        } else {
            const __bf16* bp = lbp + (((n - 6) / 4 + 2) * PW + (n - 6) % 4 - 1) * ROW;
            op[n] = tr_pair(bp, bp + 4 * ROW);
        }
    };
    static_for<0, PF>([&](auto nc) { issue(nc); });
    static_for<0, NOP>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF < NOP) issue(std::integral_constant<int, n + PF>{});
        if constexpr (n + PF == NOP - 1) {
            if (lane == 0) __atomic_store_n(done_word, k + 1, __ATOMIC_RELAXED);   // every read of the stage is issued
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n >= 6) {
            constexpr int i = (n - 6) / 4, r = (n - 6) % 4, a = 6 + 4 * i;
            // r = 0: dY(i) has arrived -> ky = 0 (patch row i); r = 1: ky = 1; r = 3: the new patch row is complete -> ky = 2
            constexpr int ky = r == 3 ? 2 : r;
            if constexpr (r != 2) {
                constexpr int prow = i + ky;                   // patch row relative to the wave's first
                static_for<0, 3>([&](auto kc) {
                    constexpr int kx = decltype(kc)::value;
                    constexpr int bi = prow < 2 ? prow * 3 + kx : 6 + 4 * (prow - 2) + 1 + kx;
#ifdef WG_X_NOMFMA
                    if constexpr (kx == 0 && ky == 0)
#endif
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[a], op[bi], acc[ky * 3 + kx], 0, 0, 0);
                });
            }
        }
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
    const bool pair = it.nco == 2;
    const ssr_wgrad_layer L = layers[it.layer];
    const ssr_wgrad_layer LB = layers[pair ? it.layer_b : it.layer];   // layer of the second dY plane
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: scalar unit tables
    const int tiles_x = (L.Gw + WG_TW - 1) / WG_TW, tiles_y = (L.Gh + TH - 1) / TH;
    const int ntile = it.tile_end - it.tile_begin;
    if (tid < 128) ctl[tid] = 0;
    GPROBE(0); GPROBE_RT(9);
    __syncthreads();   // the only barrier

    if (wave >= 4) {
        // =============================== loader waves ===============================
        const int lt = tid - 256;
        const int upshift = L.up == 2 ? 1 : 0;
        const int LH = L.Hi << upshift, LW = L.Wi << upshift;
        const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(L.x.p);
        // the dY vectors of a thread: vector v = lt + 256 q (q < QDY) is pixel v >> 3, part v & 7 = lt & 7 — always the same
        // plane (part >> 2) and channel octet (part & 3)
        constexpr int QDY = C::NDYV / 256;
        const int hb = (lt >> 2) & 1, oct = lt & 3;
        const ssr_view dyv = hb ? LB.dy : L.dy;
        const int dy_co0 = hb ? it.co0_b : it.co0;
        const bool dy_ok = (hb == 0 || pair) && dy_co0 + oct * 8 < (hb ? LB.Cout : L.Cout);
        const __bf16* __restrict__ dyg = reinterpret_cast<const __bf16*>(dyv.p) + dyv.coff + dy_co0;
        // per vector only (y, x) relative to the tile origin is kept (yx; 0x7fff7fff = never inside): the global offset follows
        // from it and the LDS offset is lo0 + q * 2048 (vector v = lt + 256 q: pixel (lt >> 3) + 32 q, part lt & 7).  Three
        // tables of 19 registers next to 2 x 19 x 4 data registers spilled.
        int yx[C::NLV];
#pragma unroll
        for (int q = 0; q < C::NLV; ++q) {
            const int v = lt + q * 256;
            if (q < QDY) {
                const int pix = v >> 3;
                yx[q] = dy_ok ? ((pix >> 4) | ((pix & 15) << 16)) : 0x7fff7fff;
            } else {
                const int vx = v - C::NDYV;
                const int pix = vx >> 3, part = vx & 7;        // 8 x 16 B = the 64 channels of one pixel: a whole line
                const int py = pix / PW, px = pix - py * PW;
                const int y = py - L.pad_y, x = px - L.pad_x;
                yx[q] = (vx < C::NXV && it.ci0 + part * 8 < L.Cin) ? ((y & 0xffff) | (x << 16)) : 0x7fff7fff;
            }
        }
        const int lo_dy = hb * C::DYPLANE + (lt >> 3) * 64 + oct * 16;
        const int lo_x = 2 * C::DYPLANE + ((lt >> 2) & 1) * C::XPLANE + (lt >> 3) * 64 + oct * 16;
        const bool wr_dy = hb == 0 || pair;
        const int x_c8 = (lt & 7) * 8;
        const bool bias_a = L.db != nullptr && it.ci0 == 0, bias_b = pair && LB.db != nullptr && it.ci0 == 0;
        const bool do_bias = hb ? bias_b : bias_a;
        float bacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        u32x4 ra[C::NLV], rb[C::NLV];
        // The loads are issued through inline asm and waited for by hand (wait_tile), every lane loads (outside the image: a
        // zero block).  Left to the compiler — a branch around each load, flat loads (pointers that come out of the layer table
        // are generic), and even plain global loads in a straight-line loop body — every store of a tile into LDS was preceded
        // by s_waitcnt vmcnt(0): the loaders had NO load in flight while they stored and then waited a full memory latency,
        // 4700 cycles per tile whatever its size (tools/wgrad_body_probe.hip: the same with 64 or 256 CUs busy; 2200 with the
        // loads removed).  Now exactly one tile (NLV loads per lane) stays in flight across every store.
        const wg_gptr zero16 = (wg_gptr)&g_wg_zero16;
        const wg_gptr dyg1 = (wg_gptr)dyg, xg1 = (wg_gptr)xg;
        auto load_tile = [&](int k, u32x4 (&r)[C::NLV]) {
            int b = it.tile_begin + k;
            const int tx_i = b % tiles_x; b /= tiles_x;
            const int ty_i = b % tiles_y;
            const int n = b / tiles_y;
            const int gy0 = ty_i * TH, gx0 = tx_i * WG_TW;
            const wg_gptr dyb = dyg1 + ((size_t)(n * L.Gh + gy0) * L.Gw + gx0) * dyv.cs * 2;
            const wg_gptr xb = xg1 + (((size_t)(n * L.Hi + (gy0 >> upshift)) * L.Wi + (gx0 >> upshift)) * L.x.cs + L.x.coff + it.ci0) * 2;
#pragma unroll
            for (int q = 0; q < C::NLV; ++q) {
                int yxq = yx[q];
                asm volatile("" : "+v"(yxq));                  // keeps the offsets below from being hoisted into 19 more registers (spills)
                const int y = (int)(short)(yxq & 0xffff), x = yxq >> 16;
                wg_gptr src;
                if (q < QDY) src = (gy0 + y < L.Gh && gx0 + x < L.Gw) ? dyb + ((y * L.Gw + x) * dyv.cs + oct * 8) * 2 : zero16;
                else src = ((unsigned)(gy0 + y) < (unsigned)LH && (unsigned)(gx0 + x) < (unsigned)LW)
                               ? xb + (((y >> upshift) * L.Wi + (x >> upshift)) * L.x.cs + x_c8) * 2 : zero16;
#ifndef WG_X_NOLOAD
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[q]) : "v"(src) : "memory");
#else
                r[q] = u32x4{(unsigned)(size_t)src, 0u, 0u, 0u};
#endif
            }
        };
        // every load older than the newest NLV has landed; the registers pass through the asm so that no use can move above it
        auto wait_tile = [&](u32x4 (&r)[C::NLV]) {
            static_assert(C::NLV == 19, "operand lists below");
#ifndef WG_X_NOLOAD
            asm volatile("s_waitcnt vmcnt(19)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]),
                         "+v"(r[7]), "+v"(r[8]), "+v"(r[9]) :: "memory");
            asm volatile("" : "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(r[16]), "+v"(r[17]),
                         "+v"(r[18]) :: "memory");
#endif
        };
        auto put = [&](int k, int st, u32x4 (&r)[C::NLV]) {
            wait_tile(r);
            if (k >= NST) {
                for (;;) {
                    u32x4 dn;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dn) : "v"((int)(C::CTL + 4 * WGC_DONE)) : "memory");
                    if ((int)min(min(dn[0], dn[1]), min(dn[2], dn[3])) >= k - NST + 1) break;
                }
            }
            char* base = smem + st * C::STAGE;
#pragma unroll
            for (int q = 0; q < C::NLV; ++q) {
                if (q < QDY) {
                    if (wr_dy) *reinterpret_cast<u32x4*>(base + lo_dy + q * 2048) = r[q];
                } else if (lt + (q - QDY) * 256 < C::NXV) {
                    *reinterpret_cast<u32x4*>(base + lo_x + (q - QDY) * 2048) = r[q];
                }
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
        // two tiles in flight, no branch around a load (past the end the last tile is loaded again and never stored)
        const int last = ntile - 1;
        if (ntile > 0) {
            load_tile(0, ra);
            load_tile(min(1, last), rb);
        }
        int k = 0;
        for (; k + 1 < ntile; k += 2) {                        // even tiles -> stage 0 from ra, odd tiles -> stage 1 from rb
            put(k, 0, ra);
            load_tile(min(k + 2, last), ra);
            put(k + 1, 1, rb);
            load_tile(min(k + 3, last), rb);
        }
        if (k < ntile) put(k, 0, ra);
        // the loads past the end: their registers stay allocated until they have landed
        auto hold = [&](u32x4 (&r)[C::NLV]) {
            asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]));
            asm volatile("" : "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(r[16]), "+v"(r[17]), "+v"(r[18]));
        };
        hold(ra); hold(rb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        hold(ra); hold(rb);
        if (bias_a || bias_b) {
            // 32 threads share each channel octet of a plane ((hb, oct) = lane & 7).  Fixed-order sum as in wgrad_bf16_kernel: xor-
            // shuffle tree over the 8 lanes of a wave with the same (hb, oct), then the four loader waves in turn, then one global
            // add per channel
            volatile float* bl = reinterpret_cast<volatile float*>(ctl + W3C_BIAS);
#pragma unroll
            for (int e = 0; e < 8; ++e) bacc[e] = do_bias ? bacc[e] : 0.f;
#pragma unroll
            for (int m = 8; m < 64; m <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) bacc[e] += __shfl_xor(bacc[e], m);
            const int lw = lt >> 6;
            while (wg_ld(ctl + WGC_LSYNC) < lw) {}
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lane < 8) {          // lane = hb * 4 + oct
#pragma unroll
                for (int e = 0; e < 8; ++e) bl[hb * 32 + oct * 8 + e] = lw == 0 ? bacc[e] : bl[hb * 32 + oct * 8 + e] + bacc[e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __atomic_fetch_add(ctl + WGC_LSYNC, 1, __ATOMIC_RELAXED);
            while (wg_ld(ctl + WGC_LSYNC) < 4) {}
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lt < 64) {
                const int h = lt >> 5, c = lt & 31;
                const ssr_wgrad_layer& LC = h ? LB : L;
                const int co = (h ? it.co0_b : it.co0) + c;
                if ((h ? bias_b : bias_a) && co < LC.Cout) atomicAdd(LC.db + co, LC.alpha * bl[lt]);
            }
        }
        return;
    }

    // =============================== MFMA waves ===============================
    const int g = lane >> 5;
    const int t16 = lane & 15;
    const int src_px = 8 * g + (t16 >> 2);
    const int src_ch = 16 * ((lane >> 4) & 1) + 4 * (t16 & 3);
    // wave -> (dY plane p, 32-channel half h of the X patch, share q of nq of the tile's 16 rows): always nine taps = nine
    // accumulators.  paired, 64 ci: (w & 1, w >> 1, all rows); paired, <= 32 ci: (w & 1, 0, rows halved); single, 64 ci:
    // (0, w & 1, rows halved); single, <= 32 ci: (0, 0, rows quartered).  Row shares are summed in the write-out.
    const int nci_a = min(64, L.Cin_w - it.ci0), nci_b = min(64, LB.Cin_w - it.ci0);
    const bool full = max(nci_a, nci_b) > 32;
    const int wp = pair ? (wave & 1) : 0;
    const int wh = !full ? 0 : pair ? (wave >> 1) : (wave & 1);
    const int nq = pair ? (full ? 1 : 2) : (full ? 2 : 4);
    const int wq = nq == 1 ? 0 : nq == 2 ? (wave >> 1) : wave;
    const int r0 = wq * (TH / nq);
    const int la_off = wp * (C::DYPLANE / 2) + (r0 * WG_TW + src_px) * C::ROW + src_ch;                  // bf16 elements from the stage base
    const int lb_off = C::DYPLANE + wh * (C::XPLANE / 2) + (r0 * PW + src_px) * C::ROW + src_ch;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // one tile loop per instance (with the instances inside ONE loop the register allocator spilled accumulators at every join)
    auto run = [&](auto nrc) {
        int st = 0, target = 4;
        int* dw_ = ctl + WGC_DONE + wave;
        for (int k = 0; k < ntile; ++k) {
            GPROBE_K(2);
            while (wg_ld(ctl + WGC_READY + st) < target) {}
            GPROBE_K(3);
            const __bf16* ldy = reinterpret_cast<const __bf16*>(smem + st * C::STAGE);
            wg3_rows<decltype(nrc)::value>(acc, ldy + la_off, ldy + lb_off, dw_, k, lane);
            GPROBE_K(4);
            if (st == 1) target += 4;
            st ^= 1;
        }
    };
    if (nq == 1) run(std::integral_constant<int, 16>{});
    else if (nq == 2) run(std::integral_constant<int, 8>{});
    else run(std::integral_constant<int, 4>{});
    GPROBE(7);
    // =============================== write-out ===============================
    // D[row = co][col = ci] of the nine taps -> LDS tile [32 co][64 ci][9 taps] (stride 9 floats between lanes: conflict-free),
    // the row shares one after the other (the first stores, the others add) -> contiguous fp32 atomic adds (each co row of
    // the tile is 64 * 9 consecutive floats); one dY plane after the other through the same 73.7 KB
    float* red = reinterpret_cast<float*>(smem);
    int phase = 0;
    wg_sync4(ctl + WGC_SYNC, phase, lane);   // all four waves are finished reading the ring
    const int i = lane & 31;
    constexpr int NOUT = 32 * 64 * 9;
    static_for<0, 2>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (c == 0 || pair) {
            const ssr_wgrad_layer& LC = c ? LB : L;
            const int co0 = c ? it.co0_b : it.co0, nci = c ? nci_b : nci_a;
            const float alpha = c ? LB.alpha : L.alpha;
            for (int qq = 0; qq < nq; ++qq) {
                if (wp == c && wq == qq) {
                    float al = alpha;
                    asm volatile("" : "+v"(al));               // the 144 products are not loop invariants to be kept (and spilled)
#pragma unroll
                    for (int t = 0; t < 9; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float* dst = red + (mfma32_row(r, g) * 64 + wh * 32 + i) * 9 + t;
                            *dst = qq == 0 ? al * acc[t][r] : *dst + al * acc[t][r];
                        }
                }
                wg_sync4(ctl + WGC_SYNC, phase, lane);
            }
            float* __restrict__ dw = LC.dw;
#pragma unroll 8
            for (int q = 0; q < NOUT / 256; ++q) {
                const int e = tid + q * 256;
                const int co = e / 576, rem = e - co * 576;
                if (co0 + co < LC.Cout && rem < nci * 9) {
                    float* dst = dw + ((size_t)(co0 + co) * LC.Cin_w + it.ci0) * 9 + rem;
#ifdef WG_X_PLAINOUT
                    *dst += red[e];
#else
                    atomicAdd(dst, red[e]);
#endif
                }
            }
            if (c == 0 && pair) wg_sync4(ctl + WGC_SYNC, phase, lane);   // the tile is read before the second plane overwrites it
        }
    });
    GPROBE(8); GPROBE_RT(10);
}

int launch_k3(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, hipStream_t st) {
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16_k3_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wg3::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    hipLaunchKernelGGL(wgrad_bf16_k3_kernel, dim3(n_items), dim3(512), Wg3::LDS, st, layers, items);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

template <int KH, int KW, int S, bool SPLIT>
int launch(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, hipStream_t st) {
    using C = WgCfg<KH, KW, S, SPLIT>;
    auto kern = wgrad_bf16_kernel<KH, KW, S, SPLIT>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
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
