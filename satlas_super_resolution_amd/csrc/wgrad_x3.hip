// fp32x3 (split-bf16) weight gradient of the 3x3 stride-1 layers (round 5) and of the 4x4 stride-2 layers (round 6, further down) in ONE pass
// over the fp32 buffers.
//   dW[co][ci][ky][kx] += alpha * sum_pixels dY[p][co] * X[p + (ky,kx) - pad][ci]      (autograd of nn.Conv2d at
//   /root/reference/ssr/archs/rrdbnet_arch.py:30-34,104-113 and discriminator_arch.py:28-40, executed by
//   l_g_total.backward() / l_d_real.backward() / l_d_fake.backward(): ssr/models/ssr_esrgan_model.py:188,219,227)
//
// Until round 5 the mode ran `ssr_split_bf16` over every activation and gradient buffer (7 GB of HBM traffic per step, 2.8 ms) and
// then the bf16 kernel of wgrad_bf16.hip three times over the planes (x_hi, dy_hi), (x_hi, dy_lo), (x_lo, dy_hi): 28 bytes moved
// per (x, dy) element pair, three write-outs per item.  Here the LOADER waves read the fp32 tiles themselves (8 bytes per pair,
// once) and split them on the way into LDS (hi = bf16(v), lo = bf16(v - hi), the same arithmetic as ssr_split_bf16); the MFMA waves
// issue the three products of every (dY, X) fragment pair into ONE accumulator set.  Everything else is wgrad_bf16_k3_kernel's
// design: ds_read_b64_tr_b16 transpose reads of dense [pixel][32 ch] planes, 4 loader + 4 MFMA waves, two-stage LDS ring with flag
// hand-over, rolling three-row window of X fragments, paired items (two 32-co dY planes against one 64-ci patch), write-out
// transposed in LDS to OIHW rows.  A stage holds hi AND lo planes, so the pixel tile is 8 x 16 instead of 16 x 16:
//   stage = 4 dY planes [128 px][32 co] (plane A hi, lo, plane B hi, lo) + 4 X planes [10 x 18 px][32 ci] (half 0 hi, lo, half 1 hi, lo)
//         = 78,848 B; two stages + the control words = 158,208 B of the 160 KB.
// Per k-step (one tile row of 16 pixels) a wave reads 2 dY fragments and the 6 fragments of the new patch row for 27 MFMAs
// (bf16 kernel: 4 reads per 9 MFMAs).
#include "wgrad_common.h"

namespace {

typedef short wx_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) wx_s16x4 wx_lds_s16x4;
typedef const __attribute__((address_space(1))) char* wx_gptr;          // global memory, explicitly: never a flat access

__device__ __forceinline__ bf16x8 wx_tr_pair(const __bf16* lo, const __bf16* hi) {
    const wx_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wx_lds_s16x4*)(lo));
    const wx_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wx_lds_s16x4*)(hi));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ int wx_ld(const int* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
__device__ __forceinline__ void wx_sync4(int* cnt, int& phase, int lane) {   // barrier of the four MFMA waves
    phase += 4;
    if (lane == 0) __atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED);
    while (wx_ld(cnt) < phase) {}
}
__device__ const u32x4 g_wx_zero32[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};   // where the loads of lanes outside the image go

#ifndef WX3_PF
#define WX3_PF 4      // operand fragments read ahead of their MFMAs (register budget: 144 accumulators + three patch rows of hi / lo fragments)
#endif
struct Wx3 {
    static constexpr int TH = 8, PH = 10, PW = 18, ROW = 32;
    static constexpr int DYP = TH * WG_TW * 64;              // bytes of one dY plane [128 px][32 co] bf16
    static constexpr int XP = PH * PW * 64;                  // bytes of one X plane [180 px][32 ci] bf16
    static constexpr int XBASE = 4 * DYP;
    static constexpr int STAGE = 4 * DYP + 4 * XP;           // 78,848 B
    static constexpr int NDYU = TH * WG_TW * 8;              // loader units (8 fp32 channels of a pixel = 32 B): 1024 of dY (two planes)
    static constexpr int NXU = PH * PW * 8;                  // 1440 of X (64 channels)
    static constexpr int NLU = (NDYU + NXU + 255) / 256;     // 10 per loader thread
    static constexpr int QDY = NDYU / 256;                   // a thread's first 4 units are dY units
    static constexpr int NST = 2;
    static constexpr int CTL = NST * STAGE;
    static constexpr int LDS = CTL + 512;
    static_assert(NDYU % 256 == 0 && NLU == 10, "operand lists of wait_tile / hold");
    static_assert(LDS <= 160 * 1024 && 32 * 64 * 9 * 4 <= CTL, "LDS budget / write-out tile fits in the ring");
};
constexpr int WXC_READY = 0;    // [NST] loader waves that have stored their part of the stage's current tile
constexpr int WXC_DONE = 8;     // [4] tiles MFMA wave w is finished with
constexpr int WXC_SYNC = 12;    // write-out barrier counter
constexpr int WXC_LSYNC = 13;   // loader-wave barrier counter
constexpr int WXC_BIAS = 16;    // [64] floats: bias-gradient partial sums of the loader threads, both planes

// One wave's share of a tile: NR tile rows (k-steps of 16 pixels), ONE dY plane (hi + lo), ONE 32-channel half of the X patch
// (hi + lo), all nine taps.  Read stream: patch rows 0, 1 (3 column shifts x hi / lo = 6 fragments each), then per k-step i:
// dY hi, dY lo, the six fragments of patch row i + 2.  The 27 MFMAs of a k-step are dealt over the eight read slots so that
// every group runs when its newest operand is PF reads old.
//   lap / lbp: this lane's source address in the wave's dY (X) hi plane at its first row; the lo plane is DYP (XP) bytes behind.
template <int NR>
__device__ __forceinline__ void wx3_rows(f32x16 (&acc)[9], const __bf16* lap, const __bf16* lbp, int* done_word, int k, int lane) {
    constexpr int ROW = Wx3::ROW, PW = Wx3::PW;
    constexpr int NOP = 12 + 8 * NR, PF = WX3_PF;
    constexpr int DYLO = Wx3::DYP / 2, XLO = Wx3::XP / 2;    // bf16 elements between a hi plane and its lo plane
    bf16x8 op[NOP];
    auto issue = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        if constexpr (n < 12) {
            const __bf16* bp = lbp + (n & 1) * XLO + ((n / 6) * PW + (n % 6) / 2) * ROW;
            op[n] = wx_tr_pair(bp, bp + 4 * ROW);
        } else if constexpr ((n - 12) % 8 < 2) {
            const __bf16* ap = lap + ((n - 12) % 8) * DYLO + ((n - 12) / 8) * WG_TW * ROW;
            op[n] = wx_tr_pair(ap, ap + 4 * ROW);
        } else {
            constexpr int r = (n - 12) % 8 - 2;
            const __bf16* bp = lbp + (r & 1) * XLO + (((n - 12) / 8 + 2) * PW + r / 2) * ROW;
            op[n] = wx_tr_pair(bp, bp + 4 * ROW);
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
        if constexpr (n >= 12) {
            constexpr int i = (n - 12) / 8, r = (n - 12) % 8, a = 12 + 8 * i;
            auto group = [&](auto kyc, auto kxc) {
                constexpr int ky = decltype(kyc)::value, kx = decltype(kxc)::value, prow = i + ky;
                constexpr int xh = prow < 2 ? prow * 6 + kx * 2 : 12 + 8 * (prow - 2) + 2 + kx * 2;
                constexpr int t = ky * 3 + kx;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[a + 1], op[xh], acc[t], 0, 0, 0);       // dy_lo . x_hi
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[a], op[xh + 1], acc[t], 0, 0, 0);       // dy_hi . x_lo
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[a], op[xh], acc[t], 0, 0, 0);           // dy_hi . x_hi
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
            if constexpr (r == 1) { group(I0{}, I0{}); group(I0{}, I1{}); }
            if constexpr (r == 2) group(I0{}, I2{});
            if constexpr (r == 3) group(I1{}, I0{});
            if constexpr (r == 4) group(I1{}, I1{});
            if constexpr (r == 5) { group(I1{}, I2{}); group(I2{}, I0{}); }
            if constexpr (r == 6) group(I2{}, I1{});
            if constexpr (r == 7) group(I2{}, I2{});
        }
    });
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512) void wgrad_x3_k3_kernel(const ssr_wgrad_layer* __restrict__ layers,
                                                          const ssr_wgrad_item* __restrict__ items) {
    using C = Wx3;
    constexpr int TH = C::TH, PW = C::PW, NST = C::NST, NLU = C::NLU, QDY = C::QDY;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* ctl = reinterpret_cast<int*>(smem + C::CTL);
    const ssr_wgrad_item it = items[blockIdx.x];
    const bool pair = it.nco == 2;
    const ssr_wgrad_layer L = layers[it.layer];
    const ssr_wgrad_layer LB = layers[pair ? it.layer_b : it.layer];   // layer of the second dY plane
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (L.Gw + WG_TW - 1) / WG_TW, tiles_y = (L.Gh + TH - 1) / TH;
    const int ntile = it.tile_end - it.tile_begin;
    if (tid < 128) ctl[tid] = 0;
    __syncthreads();   // the only barrier

    if (wave >= 4) {
        // =============================== loader waves ===============================
        // unit u = lt + 256 q: q < 4: dY pixel (lt >> 3) + 32 q of the tile, part lt & 7 = (plane, channel octet); q >= 4: X patch
        // pixel (lt >> 3) + 32 (q - 4), part lt & 7 = (32-channel half, octet).  A unit is 8 fp32 channels = two 16-byte loads,
        // and becomes one 16-byte store into the hi plane and one into the lo plane.
        const int lt = tid - 256;
        const int upshift = L.up == 2 ? 1 : 0;
        const int LH = L.Hi << upshift, LW = L.Wi << upshift;
        const int hb = (lt >> 2) & 1, oct = lt & 3;
        const ssr_view dyv = hb ? LB.dy : L.dy;
        const int dy_co0 = hb ? it.co0_b : it.co0;
        const bool dy_ok = (hb == 0 || pair) && dy_co0 + oct * 8 < (hb ? LB.Cout : L.Cout);
        int yx[NLU];                                           // (y, x) relative to the tile origin; 0x7fff7fff = never inside
#pragma unroll
        for (int q = 0; q < NLU; ++q) {
            if (q < QDY) {
                const int pix = (lt >> 3) + 32 * q;
                yx[q] = dy_ok ? ((pix >> 4) | ((pix & 15) << 16)) : 0x7fff7fff;
            } else {
                const int vx = lt + (q - QDY) * 256;
                const int pix = vx >> 3, part = vx & 7;
                const int py = pix / PW, px = pix - py * PW;
                const int y = py - L.pad_y, x = px - L.pad_x;
                yx[q] = (vx < C::NXU && it.ci0 + part * 8 < L.Cin) ? ((y & 0xffff) | (x << 16)) : 0x7fff7fff;
            }
        }
        const int lo_dy = hb * 2 * C::DYP + (lt >> 3) * 64 + oct * 16;
        const int lo_x = C::XBASE + hb * 2 * C::XP + (lt >> 3) * 64 + oct * 16;
        const bool wr_dy = hb == 0 || pair;
        const int x_c8 = (lt & 7) * 8;
        const bool bias_a = L.db != nullptr && it.ci0 == 0, bias_b = pair && LB.db != nullptr && it.ci0 == 0;
        const bool do_bias = hb ? bias_b : bias_a;
        float bacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        u32x4 ra[2 * NLU], rb[2 * NLU];
        // hand-issued loads, hand-waited (wgrad_bf16.hip, lesson 32): every lane loads (outside the image: a zero block), exactly
        // one tile (2 NLU loads per lane) stays in flight across every store
        const wx_gptr zero32 = (wx_gptr)&g_wx_zero32[0];
        const wx_gptr dyg1 = (wx_gptr)(reinterpret_cast<const float*>(dyv.p) + dyv.coff + dy_co0);
        const wx_gptr xg1 = (wx_gptr)reinterpret_cast<const float*>(L.x.p);
        auto load_tile = [&](int k, u32x4 (&r)[2 * NLU]) {
            int b = it.tile_begin + k;
            const int tx_i = b % tiles_x; b /= tiles_x;
            const int ty_i = b % tiles_y;
            const int n = b / tiles_y;
            const int gy0 = ty_i * TH, gx0 = tx_i * WG_TW;
            const wx_gptr dyb = dyg1 + ((size_t)(n * L.Gh + gy0) * L.Gw + gx0) * dyv.cs * 4;
            const wx_gptr xb = xg1 + (((size_t)(n * L.Hi + (gy0 >> upshift)) * L.Wi + (gx0 >> upshift)) * L.x.cs + L.x.coff + it.ci0) * 4;
#pragma unroll
            for (int q = 0; q < NLU; ++q) {
                int yxq = yx[q];
                asm volatile("" : "+v"(yxq));                  // keeps the offsets below from being hoisted into more registers
                const int y = (int)(short)(yxq & 0xffff), x = yxq >> 16;
                wx_gptr src;
                if (q < QDY) src = (gy0 + y < L.Gh && gx0 + x < L.Gw) ? dyb + ((y * L.Gw + x) * dyv.cs + oct * 8) * 4 : zero32;
                else src = ((unsigned)(gy0 + y) < (unsigned)LH && (unsigned)(gx0 + x) < (unsigned)LW)
                               ? xb + (((y >> upshift) * L.Wi + (x >> upshift)) * L.x.cs + x_c8) * 4 : zero32;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[2 * q]) : "v"(src) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r[2 * q + 1]) : "v"(src) : "memory");
            }
        };
        // every load older than the newest 2 NLU has landed; the registers pass through the asm so that no use can move above it
        auto pass = [&](u32x4 (&r)[2 * NLU]) {
            asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]) :: "memory");
            asm volatile("" : "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(r[16]), "+v"(r[17]), "+v"(r[18]), "+v"(r[19]) :: "memory");
        };
        auto wait_tile = [&](u32x4 (&r)[2 * NLU]) {
            pass(r);
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            pass(r);
        };
        auto put = [&](int k, int st, u32x4 (&r)[2 * NLU]) {
            wait_tile(r);
            if (k >= NST) {
                for (;;) {
                    u32x4 dn;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dn) : "v"((int)(C::CTL + 4 * WXC_DONE)) : "memory");
                    if ((int)min(min(dn[0], dn[1]), min(dn[2], dn[3])) >= k - NST + 1) break;
                }
            }
            char* base = smem + st * C::STAGE;
#pragma unroll
            for (int q = 0; q < NLU; ++q) {
                const f32x4 v0 = __builtin_bit_cast(f32x4, r[2 * q]), v1 = __builtin_bit_cast(f32x4, r[2 * q + 1]);
                bf16x8 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = (__bf16)v0[e];
                    l[e] = (__bf16)(v0[e] - (float)h[e]);
                    h[4 + e] = (__bf16)v1[e];
                    l[4 + e] = (__bf16)(v1[e] - (float)h[4 + e]);
                }
                if (q < QDY) {
                    if (wr_dy) {
                        *reinterpret_cast<u32x4*>(base + lo_dy + q * 2048) = __builtin_bit_cast(u32x4, h);
                        *reinterpret_cast<u32x4*>(base + lo_dy + C::DYP + q * 2048) = __builtin_bit_cast(u32x4, l);
                    }
                    if (do_bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            bacc[e] += v0[e];
                            bacc[4 + e] += v1[e];
                        }
                    }
                } else if (lt + (q - QDY) * 256 < C::NXU) {
                    *reinterpret_cast<u32x4*>(base + lo_x + (q - QDY) * 2048) = __builtin_bit_cast(u32x4, h);
                    *reinterpret_cast<u32x4*>(base + lo_x + C::XP + (q - QDY) * 2048) = __builtin_bit_cast(u32x4, l);
                }
            }
            if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"((int)(C::CTL + 4 * (WXC_READY + st))), "v"(1) : "memory");
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
        pass(ra); pass(rb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pass(ra); pass(rb);
        if (bias_a || bias_b) {
            // 32 threads share each channel octet of a plane ((hb, oct) = lane & 7): xor-shuffle tree over the 8 lanes of a wave
            // with the same (hb, oct), then the four loader waves in turn, then one global add per channel (fixed order)
            volatile float* bl = reinterpret_cast<volatile float*>(ctl + WXC_BIAS);
#pragma unroll
            for (int e = 0; e < 8; ++e) bacc[e] = do_bias ? bacc[e] : 0.f;
#pragma unroll
            for (int m = 8; m < 64; m <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) bacc[e] += __shfl_xor(bacc[e], m);
            const int lw = lt >> 6;
            while (wx_ld(ctl + WXC_LSYNC) < lw) {}
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lane < 8) {          // lane = hb * 4 + oct
#pragma unroll
                for (int e = 0; e < 8; ++e) bl[hb * 32 + oct * 8 + e] = lw == 0 ? bacc[e] : bl[hb * 32 + oct * 8 + e] + bacc[e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __atomic_fetch_add(ctl + WXC_LSYNC, 1, __ATOMIC_RELAXED);
            while (wx_ld(ctl + WXC_LSYNC) < 4) {}
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
    // wave -> (dY plane p, 32-channel half h of the X patch, share q of nq of the tile's 8 rows): always nine taps = nine
    // accumulators.  paired, 64 ci: (w & 1, w >> 1, all rows); paired, <= 32 ci: (w & 1, 0, rows halved); single, 64 ci:
    // (0, w & 1, rows halved); single, <= 32 ci: (0, 0, rows quartered).  Row shares are summed in the write-out.
    const int nci_a = min(64, L.Cin_w - it.ci0), nci_b = min(64, LB.Cin_w - it.ci0);
    const bool full = max(nci_a, nci_b) > 32;
    const int wp = pair ? (wave & 1) : 0;
    const int wh = !full ? 0 : pair ? (wave >> 1) : (wave & 1);
    const int nq = pair ? (full ? 1 : 2) : (full ? 2 : 4);
    const int wq = nq == 1 ? 0 : nq == 2 ? (wave >> 1) : wave;
    const int r0 = wq * (TH / nq);
    const int la_off = wp * C::DYP + (r0 * WG_TW + src_px) * C::ROW + src_ch;                  // bf16 elements from the stage base
    const int lb_off = C::XBASE / 2 + wh * C::XP + (r0 * PW + src_px) * C::ROW + src_ch;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // one tile loop per instance (with the instances inside ONE loop the register allocator spills accumulators at every join)
    auto run = [&](auto nrc) {
        int st = 0, target = 4;
        int* dw_ = ctl + WXC_DONE + wave;
        for (int k = 0; k < ntile; ++k) {
            while (wx_ld(ctl + WXC_READY + st) < target) {}
            const __bf16* ldy = reinterpret_cast<const __bf16*>(smem + st * C::STAGE);
            wx3_rows<decltype(nrc)::value>(acc, ldy + la_off, ldy + lb_off, dw_, k, lane);
            if (st == 1) target += 4;
            st ^= 1;
        }
    };
    if (nq == 1) run(std::integral_constant<int, 8>{});
    else if (nq == 2) run(std::integral_constant<int, 4>{});
    else run(std::integral_constant<int, 2>{});
    // =============================== write-out ===============================
    // D[row = co][col = ci] of the nine taps -> LDS tile [32 co][64 ci][9 taps] (stride 9 floats between lanes: conflict-free),
    // the row shares one after the other (the first stores, the others add) -> contiguous fp32 atomic adds (each co row of
    // the tile is 64 * 9 consecutive floats); one dY plane after the other through the same 73.7 KB
    float* red = reinterpret_cast<float*>(smem);
    int phase = 0;
    wx_sync4(ctl + WXC_SYNC, phase, lane);   // all four waves are finished reading the ring
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
                wx_sync4(ctl + WXC_SYNC, phase, lane);
            }
            float* __restrict__ dw = LC.dw;
#pragma unroll 8
            for (int q = 0; q < NOUT / 256; ++q) {
                const int e = tid + q * 256;
                const int co = e / 576, rem = e - co * 576;
                if (co0 + co < LC.Cout && rem < nci * 9) {
                    float* dst = dw + ((size_t)(co0 + co) * LC.Cin_w + it.ci0) * 9 + rem;
                    atomicAdd(dst, red[e]);
                }
            }
            if (c == 0 && pair) wx_sync4(ctl + WXC_SYNC, phase, lane);   // the tile is read before the second plane overwrites it
        }
    });
}


// ------------------------------------------------------------------------------------------------------------------
// 4x4 stride 2 (the discriminator's conv1..conv3, discriminator_arch.py:31-33), round 6: the same one-pass idea on the GENERIC kernel's
// work decomposition (wgrad_bf16.hip, wgrad_bf16_kernel<4, 4, 2, true>: MFMA wave w owns tap row ky = w with four accumulators kx = 0..3
// per dY plane).  Until now these layers took ssr_split_bf16_multi over their buffers and three launches of the bf16 kernel over the planes
// (1.1 + 0.2 ms per step, six write-outs per item).  Here the loader waves read the fp32 tiles (8 channels of a pixel = two 16-byte
// loads), split them on the way into LDS (hi plane, lo plane) and the MFMA waves issue the three products of every (dY, X) fragment pair
// into ONE accumulator set.  The X patch of a stride-2 layer is 5.3 input pixels per output pixel (10 x 34 for a 4 x 16 tile): a PAIRED
// item contracts TWO 32-channel blocks of dY (it.co0, it.co0_b; the engine pairs the blocks of one layer) with the same patch - per
// k-step (one tile row of 16 output pixels) 4 dY + 8 X fragment reads for 24 MFMAs (the first cut, one block per item, was loader-bound:
// 51 KB of fp32 per tile for 48 MFMAs per wave, 0.27 MFMA-busy).  A stage holds hi AND lo planes, so the pixel tile is 4 x 16:
//   stage = dY A hi, A lo, B hi, B lo [64 px][32 co] + X hi, lo [10 x 34 px][32 ci] bf16 = 59,904 B; two stages + the control words = 120,320 B.
struct Wx4 {
    static constexpr int TH = 4, S = 2, KH = 4, KW = 4, PH = (TH - 1) * S + KH, PW = (WG_TW - 1) * S + KW, ROW = 32;   // patch 10 x 34
    static constexpr int DYP = TH * WG_TW * 64;              // bytes of one dY plane [64 px][32 co] bf16
    static constexpr int XP = PH * PW * 64;                  // bytes of one X plane [340 px][32 ci] bf16
    static constexpr int XBASE = 4 * DYP;
    static constexpr int STAGE = 4 * DYP + 2 * XP;           // 59,904 B
    static constexpr int NDYU = TH * WG_TW * 4;              // loader units (8 fp32 channels of a pixel = 32 B): 256 per dY plane
    static constexpr int NXU = PH * PW * 4;                  // 1360 of X
    static constexpr int NLU = 2 + (NXU + 255) / 256;        // 8 per loader thread: unit 0 / 1 = dY plane A / B, units 2..7 are X units
    static constexpr int NST = 2;
    static constexpr int CTL = NST * STAGE;
    static constexpr int LDS = CTL + 512;
    static_assert(NDYU == 256 && NLU == 8, "operand lists of wait_tile / hold");
    static_assert(LDS <= 160 * 1024 && 32 * 32 * 16 * 4 <= CTL, "LDS budget / write-out tile fits in the ring");
};

// one MFMA wave's share of a tile: tap row ky = wave, the four taps kx, NP dY planes (1: single item, 2: paired)
template <int NP>
__device__ __forceinline__ void wx4_rows(f32x16 (&acc)[2][4], const __bf16* ldy, const __bf16* lx, int* done_word, int k, int wave, int lane,
                                         int src_px, int src_ch) {
    using C = Wx4;
    constexpr int TH = C::TH, S = C::S, KW = C::KW, PW = C::PW, ROW = C::ROW;
    constexpr int DYLO = C::DYP / 2, XLO = C::XP / 2;         // bf16 elements between a hi plane and its lo plane
    // operand stream: per k-step s (tile row) the dY fragments (plane A hi, lo[, plane B hi, lo]), then (X hi, X lo) of the four taps
    constexpr int NDY = 2 * NP, PER = NDY + 2 * KW, NOP = TH * PER, PF = 8;
    bf16x8 op[NOP];
    auto issue = [&](auto nc) {
        constexpr int n = decltype(nc)::value, s = n / PER, r = n % PER;
        if constexpr (r < NDY) {
            const __bf16* ap = ldy + r * DYLO + (s * WG_TW + src_px) * ROW + src_ch;      // planes lie A hi, A lo, B hi, B lo
            op[n] = wx_tr_pair(ap, ap + 4 * ROW);
        } else {
            constexpr int t = (r - NDY) >> 1, lo = (r - NDY) & 1;
            const __bf16* bp = lx + lo * XLO + ((s * S + wave) * PW + src_px * S + t) * ROW + src_ch;
            op[n] = wx_tr_pair(bp, bp + 4 * S * ROW);
        }
    };
    static_for<0, PF>([&](auto nc) { issue(nc); });
    static_for<0, NOP>([&](auto nc) {
        constexpr int n = decltype(nc)::value, s = n / PER, r = n % PER;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF < NOP) issue(std::integral_constant<int, n + PF>{});
        if constexpr (n + PF == NOP - 1) {
            if (lane == 0) __atomic_store_n(done_word, k + 1, __ATOMIC_RELAXED);   // every read of the stage is issued
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (r >= NDY && ((r - NDY) & 1) == 1) {      // the lo fragment of tap t has arrived: its three products per plane
            constexpr int t = (r - NDY) >> 1, a = s * PER;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[a + 2 * p + 1], op[n - 1], acc[p][t], 0, 0, 0);   // dy_lo . x_hi
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[a + 2 * p], op[n], acc[p][t], 0, 0, 0);           // dy_hi . x_lo
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[a + 2 * p], op[n - 1], acc[p][t], 0, 0, 0);       // dy_hi . x_hi
            }
        }
    });
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512) void wgrad_x3_k4_kernel(const ssr_wgrad_layer* __restrict__ layers,
                                                          const ssr_wgrad_item* __restrict__ items) {
    using C = Wx4;
    constexpr int TH = C::TH, S = C::S, KW = C::KW, PW = C::PW, NST = C::NST, NLU = C::NLU;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* ctl = reinterpret_cast<int*>(smem + C::CTL);
    const ssr_wgrad_item it = items[blockIdx.x];
    const bool pair = it.nco == 2;
    const ssr_wgrad_layer L = layers[it.layer];
    const ssr_wgrad_layer LB = layers[pair ? it.layer_b : it.layer];   // layer of the second dY plane (the engine pairs blocks of ONE layer; any layer over the same x works)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (L.Gw + WG_TW - 1) / WG_TW, tiles_y = (L.Gh + TH - 1) / TH;
    const int ntile = it.tile_end - it.tile_begin;
    if (tid < 128) ctl[tid] = 0;
    __syncthreads();   // the only barrier

    if (wave >= 4) {
        // =============================== loader waves ===============================
        // unit q of a thread: q = 0 / 1: dY pixel lt >> 2 of the tile in plane A / B, channel octet lt & 3; q >= 2: X patch pixel (lt + 256 (q - 2)) >> 2, octet lt & 3
        const int lt = tid - 256;
        const int upshift = L.up == 2 ? 1 : 0;
        const int LH = L.Hi << upshift, LW = L.Wi << upshift;
        const int oct = lt & 3;
        int yx[NLU];                                           // (y, x) relative to the tile / patch origin; 0x7fff7fff = never inside
#pragma unroll
        for (int q = 0; q < NLU; ++q) {
            if (q < 2) {
                const int pix = lt >> 2;
                const bool ok = q == 0 ? it.co0 + oct * 8 < L.Cout : (pair && it.co0_b + oct * 8 < LB.Cout);
                yx[q] = ok ? ((pix >> 4) | ((pix & 15) << 16)) : 0x7fff7fff;
            } else {
                const int vx = lt + (q - 2) * 256;
                const int pix = vx >> 2;
                const int py = pix / PW, px = pix - py * PW;
                const int y = py - L.pad_y, x = px - L.pad_x;
                yx[q] = (vx < C::NXU && it.ci0 + oct * 8 < L.Cin) ? ((y & 0xffff) | (x << 16)) : 0x7fff7fff;
            }
        }
        const int lo_dy = (lt >> 2) * 64 + oct * 16;
        const int lo_x = C::XBASE + (lt >> 2) * 64 + oct * 16;
        const bool bias_a = L.db != nullptr && it.ci0 == 0, bias_b = pair && LB.db != nullptr && it.ci0 == 0;
        float bacc[2][8] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
        u32x4 ra[2 * NLU], rb[2 * NLU];
        const wx_gptr zero32 = (wx_gptr)&g_wx_zero32[0];
        const wx_gptr dyga = (wx_gptr)(reinterpret_cast<const float*>(L.dy.p) + L.dy.coff + it.co0);
        const wx_gptr dygb = (wx_gptr)(reinterpret_cast<const float*>(LB.dy.p) + LB.dy.coff + it.co0_b);
        const wx_gptr xg1 = (wx_gptr)(reinterpret_cast<const float*>(L.x.p) + L.x.coff + it.ci0);
        auto load_tile = [&](int k, u32x4 (&r)[2 * NLU]) {
            int b = it.tile_begin + k;
            const int tx_i = b % tiles_x; b /= tiles_x;
            const int ty_i = b % tiles_y;
            const int n = b / tiles_y;
            const int gy0 = ty_i * TH, gx0 = tx_i * WG_TW;
            const size_t pix0 = (size_t)(n * L.Gh + gy0) * L.Gw + gx0;
            const wx_gptr dyba = dyga + pix0 * L.dy.cs * 4, dybb = dygb + pix0 * LB.dy.cs * 4;
            const wx_gptr xb = xg1 + ((size_t)(n * L.Hi + ((gy0 * S) >> upshift)) * L.Wi + ((gx0 * S) >> upshift)) * L.x.cs * 4;
#pragma unroll
            for (int q = 0; q < NLU; ++q) {
                int yxq = yx[q];
                asm volatile("" : "+v"(yxq));                  // keeps the offsets below from being hoisted into more registers
                const int y = (int)(short)(yxq & 0xffff), x = yxq >> 16;
                wx_gptr src;
                if (q == 0) src = (gy0 + y < L.Gh && gx0 + x < L.Gw) ? dyba + ((y * L.Gw + x) * L.dy.cs + oct * 8) * 4 : zero32;
                else if (q == 1) src = (gy0 + y < L.Gh && gx0 + x < L.Gw) ? dybb + ((y * L.Gw + x) * LB.dy.cs + oct * 8) * 4 : zero32;
                else src = ((unsigned)(gy0 * S + y) < (unsigned)LH && (unsigned)(gx0 * S + x) < (unsigned)LW)
                               ? xb + (((y >> upshift) * L.Wi + (x >> upshift)) * L.x.cs + oct * 8) * 4 : zero32;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[2 * q]) : "v"(src) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r[2 * q + 1]) : "v"(src) : "memory");
            }
        };
        auto pass = [&](u32x4 (&r)[2 * NLU]) {
            asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) :: "memory");
            asm volatile("" : "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) :: "memory");
        };
        auto wait_tile = [&](u32x4 (&r)[2 * NLU]) {            // every load older than the newest 2 NLU has landed
            pass(r);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            pass(r);
        };
        auto put = [&](int k, int st, u32x4 (&r)[2 * NLU]) {
            wait_tile(r);
            if (k >= NST) {
                for (;;) {
                    u32x4 dn;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dn) : "v"((int)(C::CTL + 4 * WXC_DONE)) : "memory");
                    if ((int)min(min(dn[0], dn[1]), min(dn[2], dn[3])) >= k - NST + 1) break;
                }
            }
            char* base = smem + st * C::STAGE;
#pragma unroll
            for (int q = 0; q < NLU; ++q) {
                const f32x4 v0 = __builtin_bit_cast(f32x4, r[2 * q]), v1 = __builtin_bit_cast(f32x4, r[2 * q + 1]);
                bf16x8 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = (__bf16)v0[e];
                    l[e] = (__bf16)(v0[e] - (float)h[e]);
                    h[4 + e] = (__bf16)v1[e];
                    l[4 + e] = (__bf16)(v1[e] - (float)h[4 + e]);
                }
                if (q < 2) {
                    if (q == 0 || pair) {
                        *reinterpret_cast<u32x4*>(base + q * 2 * C::DYP + lo_dy) = __builtin_bit_cast(u32x4, h);
                        *reinterpret_cast<u32x4*>(base + q * 2 * C::DYP + C::DYP + lo_dy) = __builtin_bit_cast(u32x4, l);
                    }
                    if (q == 0 ? bias_a : bias_b) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            bacc[q][e] += v0[e];
                            bacc[q][4 + e] += v1[e];
                        }
                    }
                } else if (lt + (q - 2) * 256 < C::NXU) {
                    *reinterpret_cast<u32x4*>(base + lo_x + (q - 2) * 4096) = __builtin_bit_cast(u32x4, h);
                    *reinterpret_cast<u32x4*>(base + lo_x + C::XP + (q - 2) * 4096) = __builtin_bit_cast(u32x4, l);
                }
            }
            if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"((int)(C::CTL + 4 * (WXC_READY + st))), "v"(1) : "memory");
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
        if (ntile > 0) {                                       // the loads past the end: their registers stay allocated until they have landed
            pass(ra); pass(rb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pass(ra); pass(rb);
        }
        if (bias_a || bias_b) {
            // 64 threads share each channel octet (lane & 3) of a plane: xor-shuffle tree over the 16 lanes of a wave that hold the octet, then
            // the four loader waves in turn, then one global add per channel (fixed order)
            volatile float* bl = reinterpret_cast<volatile float*>(ctl + WXC_BIAS);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int m = 4; m < 64; m <<= 1)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bacc[p][e] += __shfl_xor(bacc[p][e], m);
            const int lw = lt >> 6;
            while (wx_ld(ctl + WXC_LSYNC) < lw) {}
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lane < 4) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bl[p * 32 + lane * 8 + e] = lw == 0 ? bacc[p][e] : bl[p * 32 + lane * 8 + e] + bacc[p][e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __atomic_fetch_add(ctl + WXC_LSYNC, 1, __ATOMIC_RELAXED);
            while (wx_ld(ctl + WXC_LSYNC) < 4) {}
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

    // =============================== MFMA waves: tap row ky = wave, accumulators [dY plane][kx] ===============================
    const int g = lane >> 5;
    const int t16 = lane & 15;
    const int src_px = 8 * g + (t16 >> 2);                    // + 4 for the second read of the pair
    const int src_ch = 16 * ((lane >> 4) & 1) + 4 * (t16 & 3);
    f32x16 acc[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < KW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][t][r] = 0.f;
    // one tile loop per instance (with the instances inside ONE loop the register allocator spills accumulators at every join)
    auto run = [&](auto npc) {
        int st = 0, target = 4;                                // stage of tile k and its ready target 4 * (uses + 1)
        int* dw_ = ctl + WXC_DONE + wave;
        for (int k = 0; k < ntile; ++k) {
            while (wx_ld(ctl + WXC_READY + st) < target) {}
            const __bf16* ldy = reinterpret_cast<const __bf16*>(smem + st * C::STAGE);
            wx4_rows<decltype(npc)::value>(acc, ldy, ldy + C::XBASE / 2, dw_, k, wave, lane, src_px, src_ch);
            if (st == 1) target += 4;
            st ^= 1;
        }
    };
    if (pair) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});

    // =============================== write-out, one dY plane after the other ===============================
    // D[row = co][col = ci] of this wave's four taps -> LDS tile [32 co][32 ci][16 taps] (stride 17 floats between lanes: conflict-free) ->
    // contiguous fp32 atomic adds (each co row of the tile is 32 * 16 consecutive floats of dW)
    constexpr int KK = 16, KP = KK + 1, NOUT = 32 * 32 * KK;
    static_assert(32 * 32 * KP * 4 <= C::CTL, "write-out tile fits in the ring");
    float* red = reinterpret_cast<float*>(smem);
    int phase = 0;
    wx_sync4(ctl + WXC_SYNC, phase, lane);   // all four waves are finished reading the ring
    const int i = lane & 31;
    static_for<0, 2>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (c == 0 || pair) {
            const ssr_wgrad_layer& LC = c ? LB : L;
            const int co0 = c ? it.co0_b : it.co0;
            float al = LC.alpha;
            asm volatile("" : "+v"(al));                       // the products are not loop invariants to be kept (and spilled)
#pragma unroll
            for (int t = 0; t < KW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(mfma32_row(r, g) * 32 + i) * KP + wave * KW + t] = al * acc[c][t][r];
            wx_sync4(ctl + WXC_SYNC, phase, lane);
            float* __restrict__ dw = LC.dw;
            const int nci = min(32, LC.Cin_w - it.ci0);
#pragma unroll 8
            for (int q = 0; q < NOUT / 256; ++q) {
                const int e = tid + q * 256;                   // (co, ci * 16 + tap): contiguous in dW per co row
                const int co = e / (32 * KK), rem = e - co * (32 * KK);
                if (co0 + co < LC.Cout && rem < nci * KK)
                    atomicAdd(dw + ((size_t)(co0 + co) * LC.Cin_w + it.ci0) * KK + rem, red[(co * 32 + (rem >> 4)) * KP + (rem & 15)]);
            }
            if (c == 0 && pair) wx_sync4(ctl + WXC_SYNC, phase, lane);   // the tile is read before the second plane overwrites it
        }
    });
}

}  // namespace

int ssr_wgrad_x3_dispatch(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, int KH, int KW, int S,
                          hipStream_t st) {
    if (KH == 4 && KW == 4 && S == 2) {                        // the discriminator's stride-2 layers (round 6)
        static bool attr4_done[SSR_MAX_DEVICES] = {};
        const int dev4 = ssr_device_ordinal();
        if (!attr4_done[dev4]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_k4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wx4::LDS);
            if (e != hipSuccess) return (int)e;
            attr4_done[dev4] = true;
        }
        hipLaunchKernelGGL(wgrad_x3_k4_kernel, dim3(n_items), dim3(512), Wx4::LDS, st, layers, items);
        SSR_LAUNCH_CHECK();
        return SSR_OK;
    }
    if (!(KH == 3 && KW == 3 && S == 1)) return SSR_EUNSUP;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_k3_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wx3::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    hipLaunchKernelGGL(wgrad_x3_k3_kernel, dim3(n_items), dim3(512), Wx3::LDS, st, layers, items);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}
