// Weight-stationary persistent 3x3 convolution for the large-spatial layers (bf16, Cin <= 64).
//
// The U-Net discriminator's and the generator tail's 3x3 layers at 128x128 / 64x64 have a huge pixel
// count (B*16384) but tiny weights (<= 64x64x9).  In the generic kernel (conv.hip) every 128-pixel
// workgroup re-stages the weight slab through LDS, runs 18..36 MFMAs per wave and pays a full
// prologue + split-K reduce + epilogue: rocprofv3 shows 250..560 TFLOP/s (10..22 % of peak), one
// workgroup per CU.  Here the weights never touch LDS:
//   * a workgroup (4 waves) owns ONE 32-channel output tile and keeps its whole [9 taps][Cin][32] weight
//     slice in REGISTERS (36 x 16-byte MFMA A-fragments per lane for Cin = 64) for its entire lifetime;
//   * it is persistent: it walks 8x16-pixel tiles with stride gridDim.x; per tile only the input halo
//     patch moves (global -> registers -> padded LDS rows, double buffered, ONE barrier per tile);
//   * MFMA operands are swapped (A = weights, B = pixels) so each lane ends with 16 consecutive-ish
//     channels of ONE pixel: bias / LeakyReLU / residual / mask / store are 8-byte vector ops, no
//     transpose and no split-K reduce;
//   * 2 workgroups per CU (57 KB LDS, <= 256 VGPRs): one runs MFMAs while the other is in its epilogue.
// Same descriptor and epilogue contract as conv.hip (ssr_conv_desc).
//
// Replaces nn.Conv2d 3x3 (and its dgrad) at /root/reference/ssr/archs/discriminator_arch.py:28,38-40,44,67-69
// and rrdbnet_arch.py:104,111-112,128,136 (conv0, conv7, conv8, conv9; conv_up2, conv_hr, conv_last).
#include "common.h"
#include <cstdlib>

#ifdef SSR_PROBE
#define WPROBE(k) do { if (threadIdx.x == 0 && tile == (int)blockIdx.x + (int)gridDim.x) g_probe[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WPROBE(k)
#endif

namespace {

constexpr int WS_TH = 8, WS_TW = 16, WS_PH = WS_TH + 2, WS_PW = WS_TW + 2, WS_PIX = WS_PH * WS_PW;   // 180
constexpr int WS_AROW = 80;                                   // bytes per patch row: 32 bf16 + 16 pad
typedef __bf16 bf16x4w __attribute__((ext_vector_type(4)));

// lane-slot s (0..31) of a wave's 2 x 16 pixel tile -> column: the second row is rotated by 14 so that with the patch
// pitch of 18 pixels the 16 lanes the LDS serves together read rows that are distinct mod 16 (80-byte rows: bank = 20 r mod 64)
// -> conflict-free ds_read_b128 for every tap (same map as conv_big.hip)
__device__ __forceinline__ int ws_col(int s) { return s < 16 ? s : ((s + 14) & 15); }

#define WS_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Epilogue variants.  The epilogue is VALU / issue bound (tools/ws_probe.hip: with every feature a run-time
// flag it cost 6100 of the 12400 cycles of a 64->64 tile, ~800 instructions of selects and branches), so the
// feature set is a template parameter: WS_GENERIC keeps the full ssr_conv_desc contract, the others are
// branch-free straight-line code for the combinations the ESRGAN step actually launches.
constexpr int WS_LRELU = 1, WS_MASK = 2, WS_R1 = 4, WS_ACC = 8, WS_Y1 = 16, WS_GENERIC = -1;   // Y1: second output = value before the mask

__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int NPL, int NT, int EP>   // 32-channel input planes (Cin <= 32*NPL), 32-channel output tiles per workgroup
__global__ __launch_bounds__(256, (NPL * NT > 2 ? 1 : 2)) void conv_ws_kernel(const ssr_conv_desc d) {
    constexpr int PLANE = WS_PIX * WS_AROW;                   // 14,400 B
    constexpr int BUF = NPL * PLANE;
    constexpr int NV = NPL * WS_PIX * 4;                      // 16-B vectors per patch
    constexpr int NPV = (NV + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x BUF

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int co0 = blockIdx.y * 32 * NT;
    const int tiles_x = (d.Gw + WS_TW - 1) / WS_TW, tiles_y = (d.Gh + WS_TH - 1) / WS_TH;
    const int ntiles = d.N * tiles_y * tiles_x;
    const int upshift = d.up == 2 ? 1 : 0;
    const int LH = d.Hi << upshift, LW = d.Wi << upshift;
    const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(d.x.p);

    // ---- the stationary operand: [tap][plane][kk] 16-byte fragments of W[co0 + i][.] in registers ----
    u32x4 wf[NT][9 * NPL * 2];
    {
        const __bf16* __restrict__ wg = reinterpret_cast<const __bf16*>(d.w);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
                        wf[t][(tap * NPL + pl) * 2 + kk] = *reinterpret_cast<const u32x4*>(
                            wg + ((size_t)(pl * 9 + tap) * d.CoutPad + co0 + t * 32 + i) * 32 + kk * 16 + g * 8);
    }
    const int co_l = 4 * g;                                   // + 8*q4 + e: this lane's 16 output channels
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * BUF);   // [32*NT] behind the two patch buffers
    if (tid < 32 * NT) bias_lds[tid] = (d.bias && co0 + tid < d.Cout) ? d.bias[co0 + tid] : 0.f;

    // ---- patch staging: everything that does not depend on the tile is computed once ----
    // vector v = tid + 256 q of the patch: consecutive lanes read consecutive 16-B parts of one pixel (all planes)
    // and then the next pixel: a 64-channel NHWC row is one 128-B line -> a wave instruction touches 8 full lines
    // Round 4: each patch vector is ONE unconditional buffer load.  The lane part of the address (tile-invariant, bytes) is kept
    // relative to a resource whose base lies one row and one pixel BELOW the tensor, so that the halo's (-1, -1) offsets stay
    // non-negative; the tile is a scalar offset; a vector outside the image (or past the last channel / the table) gets an
    // offset beyond num_records and reads zeros.  (A branch around each load, a zeroed destination and 64-bit pointer sums were
    // ~90 of the ~520 instructions a wave issues per tile for its 36 MFMAs: the loop is issue-bound.)
    constexpr int WS_OOB = 0x7ffffff0;
    const long xbias = (long)(d.Wi + 1) * d.x.cs * 2;
    const long xbytes = (long)d.N * d.Hi * d.Wi * d.x.cs * 2 + xbias;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(xg)) - xbias, 0,
                                                                         (int)(xbytes > 0x7fffff00L ? 0x7fffff00L : xbytes), 0x00020000);
    int p_rel[NPV], p_yx[NPV];
#pragma unroll
    for (int q = 0; q < NPV; ++q) {
        const int v = tid + q * 256;
        const int pix = v / (NPL * 4), pp8 = v - pix * (NPL * 4);
        const int pl = pp8 >> 2, part = pp8 & 3;
        const int py = pix / WS_PW, px = pix - py * WS_PW;
        const int c = pl * 32 + part * 8;
        const bool ok = v < NV && c < d.Cin;
        p_rel[q] = ((((py - 1) >> upshift) * d.Wi + ((px - 1) >> upshift)) * d.x.cs + d.x.coff + c) * 2 + (int)xbias;
        p_yx[q] = ok ? ((py - 1) & 0xffff) | ((px - 1) << 16) : 0x7fff7fff;   // never inside
    }
    u32x4 rp[NPV];
    const int x_cs2 = d.x.cs * 2, x_Hi = d.Hi, x_Wi = d.Wi;
    auto load_patch = [&](int n, int gy0, int gx0) {
        const int soff = ((n * x_Hi + (gy0 >> upshift)) * x_Wi + (gx0 >> upshift)) * x_cs2;
#pragma unroll
        for (int q = 0; q < NPV; ++q) {
            const int ly = gy0 + (int)(short)(p_yx[q] & 0xffff), lx = gx0 + (p_yx[q] >> 16);
            const bool in = (unsigned)ly < (unsigned)LH && (unsigned)lx < (unsigned)LW;
            rp[q] = __builtin_amdgcn_raw_buffer_load_b128(rsx, in ? p_rel[q] : WS_OOB, soff, 0);
        }
    };
    int p_lds[NPV];                                            // (tile-invariant LDS offsets: not recomputed per tile)
#pragma unroll
    for (int q = 0; q < NPV; ++q) {
        const int v = tid + q * 256;
        const int pix = v / (NPL * 4), pp8 = v - pix * (NPL * 4);
        p_lds[q] = (pp8 >> 2) * PLANE + pix * WS_AROW + (pp8 & 3) * 16;
    }
    auto store_patch = [&](int buf) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int q = 0; q < NPV; ++q) {
            if ((q + 1) * 256 <= NV || tid + q * 256 < NV) *reinterpret_cast<u32x4*>(base + p_lds[q]) = rp[q];
        }
    };
    // tile -> (image, tile row, tile column): decoded once by division, then advanced by the grid stride with carries (two
    // divisions by run-time values per tile were ~40 scalar / vector instructions in a loop that is issue-bound)
    struct TilePos { int n, ty, tx; };
    auto decode = [&](int tile) {
        TilePos t;
        int b = tile;
        t.tx = b % tiles_x; b /= tiles_x;
        t.ty = b % tiles_y;
        t.n = b / tiles_y;
        return t;
    };
    const TilePos stride_pos = decode((int)gridDim.x);
    auto advance = [&](TilePos t) {
        t.tx += stride_pos.tx;
        int cy = 0;
        if (t.tx >= tiles_x) { t.tx -= tiles_x; cy = 1; }
        t.ty += stride_pos.ty + cy;
        int cn = 0;
        if (t.ty >= tiles_y) { t.ty -= tiles_y; cn = 1; }
        t.n += stride_pos.n + cn;
        return t;
    };
    __bf16* __restrict__ yp = reinterpret_cast<__bf16*>(d.y.p);
    __bf16* __restrict__ y0p = reinterpret_cast<__bf16*>(d.y0.p);
    __bf16* __restrict__ y1p = reinterpret_cast<__bf16*>(d.y1.p);
    const __bf16* __restrict__ r1p = reinterpret_cast<const __bf16*>(d.r1.p);
    const __bf16* __restrict__ r2p = reinterpret_cast<const __bf16*>(d.r2.p);
    const __bf16* __restrict__ mp = reinterpret_cast<const __bf16*>(d.m.p);
    constexpr int SROW = 32 * NT + 8;                         // slab row pitch in elements
    const int a_off = ((2 * wave + (i >> 4)) * WS_PW + ws_col(i)) * WS_AROW + g * 16;   // this lane's pixel
    __bf16* slab = reinterpret_cast<__bf16*>(smem + 2 * BUF + 256) + wave * (32 * SROW);   // [32 px][32*NT co], rows padded by 16 B (bank spread)
    __bf16* slab1 = slab + 4 * (32 * SROW);                // second output (lean Y1 variants)

    int tile = blockIdx.x;
    int n = 0, gy0 = 0, gx0 = 0;
    TilePos tp = decode(tile);
    if (tile < ntiles) {
        n = tp.n; gy0 = tp.ty * WS_TH; gx0 = tp.tx * WS_TW;
        load_patch(n, gy0, gx0);
        store_patch(0);
    }
    WS_BAR();
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        WPROBE(0);
        const int next = tile + gridDim.x;
        tp = advance(tp);
        // past the last tile the patch of the current tile is requested again (valid addresses, never stored): no branch
        const bool has_next = next < ntiles;
        const int n_n = has_next ? tp.n : n, gy0_n = has_next ? tp.ty * WS_TH : gy0, gx0_n = has_next ? tp.tx * WS_TW : gx0;
#ifndef WS_NO_LOAD
        load_patch(n_n, gy0_n, gx0_n);
#endif
        WPROBE(1);
        // ---- this lane's output pixel ----
        const int gy = gy0 + 2 * wave + (i >> 4), gx = gx0 + ws_col(i);
        const bool pvalid = gy < d.Gh && gx < d.Gw;
        const int cy = pvalid ? gy : d.Gh - 1, cx = pvalid ? gx : d.Gw - 1;
        const size_t pp = (size_t)(n * d.Ho + cy * d.oys + d.oyo) * d.Wo + cx * d.oxs + d.oxo;
        // lean variants: the epilogue operands of this lane's pixel.  With one wave per SIMD (64-channel tiles) they are
        // requested now and land under the MFMAs; the two-workgroups-per-CU variants request them after the MFMA loop to
        // stay inside 256 registers (the other workgroup covers the latency).
        constexpr bool EARLY_OPS = NPL * NT > 2;
        u32x2 q1[EP >= 0 && (EP & WS_R1) ? NT * 4 : 1], qa[EP >= 0 && (EP & WS_ACC) ? NT * 4 : 1],
            qm[EP >= 0 && (EP & WS_MASK) ? NT * 4 : 1];
        auto load_ops = [&]() {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int c = co0 + t * 32 + 8 * q4 + co_l;
                    const int cc = c < d.Cout ? c : 0;            // clamped: always addressable
                    if constexpr (EP >= 0 && (EP & WS_R1) != 0)
                        q1[t * 4 + q4] = *reinterpret_cast<const u32x2*>(r1p + pp * d.r1.cs + d.r1.coff + cc);
                    if constexpr (EP >= 0 && (EP & WS_ACC) != 0)
                        qa[t * 4 + q4] = *reinterpret_cast<const u32x2*>(yp + pp * d.y.cs + d.y.coff + cc);
                    if constexpr (EP >= 0 && (EP & WS_MASK) != 0)
                        qm[t * 4 + q4] = *reinterpret_cast<const u32x2*>(mp + pp * d.m.cs + d.m.coff + cc);
                }
        };
        if constexpr (EP >= 0 && EARLY_OPS) load_ops();
        // ---- 9 taps x NPL planes x 2 k-substeps; accumulators start at the bias ----
        WPROBE(2);
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + t * 32 + 8 * q4 + co_l);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][4 * q4 + e] = EP >= 0 ? bq[e] : 0.f;
            }
        const char* ab = smem + buf * BUF + a_off;
        // explicit software pipeline: the pixel fragment of step n + PF is read before the MFMAs of step n.  The
        // sched_barrier fences pin that order (one wave per SIMD in the 64-channel variant: nothing else hides
        // the LDS latency).
        constexpr int NSTEP = 18 * NPL, PF = 3;
        u32x4 pxr[NSTEP];
        auto issue = [&](auto n_c) {   // step n = ((pl*3 + ky)*3 + kx)*2 + kk
            constexpr int s_ = decltype(n_c)::value;
            constexpr int kk = s_ & 1, tap = (s_ >> 1) % 9, pl = (s_ >> 1) / 9;
            pxr[s_] = *reinterpret_cast<const u32x4*>(ab + pl * PLANE + ((tap / 3) * WS_PW + (tap % 3)) * WS_AROW + kk * 32);
        };
        static_for<0, PF>([&](auto n_c) { issue(n_c); });
        static_for<0, NSTEP>([&](auto n_c) {
            constexpr int m = decltype(n_c)::value;
            constexpr int kk = m & 1, tap = (m >> 1) % 9, pl = (m >> 1) / 9;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m + PF < NSTEP) issue(std::integral_constant<int, m + PF>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) mma16<__bf16>(acc[t], wf[t][(tap * NPL + pl) * 2 + kk], pxr[m]);
        });
        __builtin_amdgcn_sched_barrier(0);
        WPROBE(3);
        if constexpr (EP >= 0) {
            if constexpr (!EARLY_OPS) load_ops();
            // ---- lean epilogue: this lane = one pixel x 16*NT channels (rows 8*q4 + 4*g + e of the C fragments) ----
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[t][4 * q4 + e];
                        if constexpr ((EP & WS_LRELU) != 0) v[e] = lrelu_max(v[e]);   // == lrelu()
                    }
                    if constexpr ((EP & WS_R1) != 0) {
                        const u32x2 r = q1[t * 4 + q4];
                        v[0] += d.beta1 * bf16_lo(r[0]); v[1] += d.beta1 * bf16_hi(r[0]);
                        v[2] += d.beta1 * bf16_lo(r[1]); v[3] += d.beta1 * bf16_hi(r[1]);
                    }
                    if constexpr ((EP & WS_ACC) != 0) {
                        const u32x2 r = qa[t * 4 + q4];
                        v[0] += bf16_lo(r[0]); v[1] += bf16_hi(r[0]); v[2] += bf16_lo(r[1]); v[3] += bf16_hi(r[1]);
                    }
                    if constexpr ((EP & WS_Y1) != 0) {   // y1 = value before the LeakyReLU-backward mask
                        bf16x4w o1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o1[e] = (__bf16)v[e];
                        *reinterpret_cast<bf16x4w*>(slab1 + i * SROW + t * 32 + 8 * q4 + co_l) = o1;
                    }
                    if constexpr ((EP & WS_MASK) != 0) {
                        const u32x2 r = qm[t * 4 + q4];
                        v[0] = lrelu_mask_lo(v[0], r[0]); v[1] = lrelu_mask_hi(v[1], r[0]);
                        v[2] = lrelu_mask_lo(v[2], r[1]); v[3] = lrelu_mask_hi(v[3], r[1]);
                    }
                    bf16x4w o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x4w*>(slab + i * SROW + t * 32 + 8 * q4 + co_l) = o;
                }
        } else {
        // ---- generic epilogue (full ssr_conv_desc contract); its operands are loaded after the MFMA loop to stay
        //      inside the register budget ----
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bf16x4w q1g[4], q2g[4], qag[4], qmg[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c = co0 + t * 32 + 8 * q4 + co_l;
                const int cc = c < d.Cout ? c : 0;                // clamped: always addressable
                if (r1p) q1g[q4] = *reinterpret_cast<const bf16x4w*>(r1p + pp * d.r1.cs + d.r1.coff + cc);
                if (r2p) q2g[q4] = *reinterpret_cast<const bf16x4w*>(r2p + pp * d.r2.cs + d.r2.coff + cc);
                if (d.accumulate) qag[q4] = *reinterpret_cast<const bf16x4w*>(yp + pp * d.y.cs + d.y.coff + cc);
                if (mp) qmg[q4] = *reinterpret_cast<const bf16x4w*>(mp + pp * d.m.cs + d.m.coff + cc);
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c = co0 + t * 32 + 8 * q4 + co_l;
                const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + t * 32 + 8 * q4 + co_l);
                bf16x4w o0, o1, o2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[t][4 * q4 + e] + bq[e];
                    if (d.act == SSR_ACT_LRELU) v = lrelu(v);
                    v *= d.alpha;
                    o0[e] = (__bf16)v;
                    if (r1p) v += d.beta1 * (float)q1g[q4][e];
                    if (r2p) v += d.beta2 * (float)q2g[q4][e];
                    if (d.accumulate) v += (float)qag[q4][e];
                    o1[e] = (__bf16)v;
                    if (mp) v *= lrelu_grad_from_out((float)qmg[q4][e]);
                    o2[e] = (__bf16)v;
                }
                if (pvalid && c < d.Cout) {
                    if (y0p) *reinterpret_cast<bf16x4w*>(y0p + pp * d.y0.cs + d.y0.coff + c) = o0;
                    if (y1p) *reinterpret_cast<bf16x4w*>(y1p + pp * d.y1.cs + d.y1.coff + c) = o1;
                }
                // final output: transposed through the wave's slab so that a store instruction writes whole lines
                *reinterpret_cast<bf16x4w*>(slab + i * SROW + t * 32 + 8 * q4 + co_l) = o2;
            }
        }
        }
        auto flush = [&](const __bf16* sl, __bf16* __restrict__ out, const ssr_view& vw) {
            constexpr int PARTS = 4 * NT;                         // 16-B parts per pixel row of the tile
            const size_t row0 = (size_t)(n * d.Ho + (gy0 + 2 * wave) * d.oys + d.oyo) * d.Wo + gx0 * d.oxs + d.oxo;
#pragma unroll
            for (int h = 0; h < 32 * PARTS / 64; ++h) {
                const int v = h * 64 + lane;
                const int pix = v / PARTS, part = v - pix * PARTS;
                const int dy = pix >> 4, dx = ws_col(pix);
                const int c = co0 + part * 8;
                const u32x4 val = *reinterpret_cast<const u32x4*>(sl + pix * SROW + part * 8);
#ifdef WS_NO_STORE
                if (val[0] == 0x12345678u)
#else
                if (gy0 + 2 * wave + dy < d.Gh && gx0 + dx < d.Gw && c < d.Cout)
#endif
                    *reinterpret_cast<u32x4*>(out + (row0 + (size_t)dy * d.oys * d.Wo + dx * d.oxs) * vw.cs + vw.coff + c) = val;
            }
        };
        flush(slab, yp, d.y);
        if constexpr (EP >= 0 && (EP & WS_Y1) != 0) flush(slab1, y1p, d.y1);
        WPROBE(4);
        if (next < ntiles) store_patch(buf ^ 1);
        WPROBE(5);
        WS_BAR();
        WPROBE(6);   // next patch visible; everyone is done reading the current one
        buf ^= 1;
        n = n_n; gy0 = gy0_n; gx0 = gx0_n;
    }
}

template <int NPL, int NT, int EP>
int launch_ws_ep(const ssr_conv_desc& d, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)NPL * WS_PIX * WS_AROW + 256 + (EP >= 0 && (EP & WS_Y1) ? 2 : 1) * 4 * 32 * (32 * NT + 8) * 2;
    auto kern = conv_ws_kernel<NPL, NT, EP>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    const int ntiles = d.N * ((d.Gh + WS_TH - 1) / WS_TH) * ((d.Gw + WS_TW - 1) / WS_TW);
    const int ny = d.CoutPad / (32 * NT);
    int gx = (NPL * NT > 2 ? 256 : 512) / ny;   // 1 or 2 workgroups per CU x 256 CUs, shared among the output tiles
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(kern, dim3(gx, ny, 1), dim3(256), lds, st, d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

template <int NPL, int NT>
int launch_ws(const ssr_conv_desc& d, hipStream_t st) {
    // lean epilogue variants: single output, alpha == 1, no second residual
    auto al16 = [](const ssr_view& v) { return (v.cs % 8) == 0 && (v.coff % 8) == 0 && ((uintptr_t)v.p % 16) == 0; };
    if (!d.y0.p && (!d.y1.p || al16(d.y1)) && !d.r2.p && d.alpha == 1.f) {
        const int ep = (d.act == SSR_ACT_LRELU ? WS_LRELU : 0) | (d.m.p ? WS_MASK : 0) | (d.r1.p ? WS_R1 : 0) |
                       (d.accumulate ? WS_ACC : 0) | (d.y1.p ? WS_Y1 : 0);
        switch (ep) {
            case 0: return launch_ws_ep<NPL, NT, 0>(d, st);
            case WS_LRELU: return launch_ws_ep<NPL, NT, WS_LRELU>(d, st);
            case WS_MASK: return launch_ws_ep<NPL, NT, WS_MASK>(d, st);
            case WS_MASK | WS_ACC: return launch_ws_ep<NPL, NT, WS_MASK | WS_ACC>(d, st);
            case WS_MASK | WS_R1: return launch_ws_ep<NPL, NT, WS_MASK | WS_R1>(d, st);
            case WS_MASK | WS_Y1: return launch_ws_ep<NPL, NT, WS_MASK | WS_Y1>(d, st);
            default: break;
        }
    }
    return launch_ws_ep<NPL, NT, WS_GENERIC>(d, st);
}

}  // namespace

// 3x3 stride 1 bf16, Cin <= 64, large grid, single input, full-range residual / mask channel windows
bool ssr_conv_ws_shape_ok(const ssr_conv_desc& d);
bool ssr_conv_ws_qualifies(const ssr_conv_desc& d) {
    static const bool off = [] { const char* e = getenv("SSR_CONV_WS"); return e && e[0] == '0'; }();
    if (off || d.dtype != SSR_BF16) return false;
    if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad_y == 1 && d.pad_x == 1) || d.x2.p) return false;
    if (d.Cin > 64 || (d.Cout % 8) != 0) return false;
    if (d.Gh != (d.Hi << (d.up == 2)) || d.Gw != (d.Wi << (d.up == 2))) return false;
    const long ntiles = (long)d.N * ((d.Gh + WS_TH - 1) / WS_TH) * ((d.Gw + WS_TW - 1) / WS_TW);
    if (ntiles * (d.CoutPad / 32) < 2048) return false;                 // small grids: resident / fused kernels
    // r01 measurements (tools/ws_probe.hip): Cin <= 32 beats the pipelined kernel by 1.3-3x; the 64-channel variant
    // (one wave per SIMD, 288 weight registers) by 1.5x once its epilogue is a branch-free instantiation
    return ssr_conv_ws_shape_ok(d);
}

// everything except the grid-size heuristic (ssr_conv2d_impl can force this kernel on small shapes)
bool ssr_conv_ws_shape_ok(const ssr_conv_desc& d) {
    if (d.dtype != SSR_BF16) return false;
    if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad_y == 1 && d.pad_x == 1) || d.x2.p) return false;
    if (d.Cin > 64 || (d.Cout % 8) != 0) return false;
    if (d.Gh != (d.Hi << (d.up == 2)) || d.Gw != (d.Wi << (d.up == 2))) return false;
    if (d.r1.p && d.r1_nc < d.Cout) return false;
    if (d.r2.p && d.r2_nc < d.Cout) return false;
    if (d.m.p && !(d.m_c0 == 0 && d.m_c1 >= d.Cout)) return false;
    // the patch loads address x through a 32-bit byte offset from a buffer resource: larger inputs go to the generic kernel
    if (((long)d.N * d.Hi * d.Wi + d.Wi + 1) * d.x.cs * 2 > 0x7fffff00L) return false;
    auto al = [](const ssr_view& v) { return !v.p || ((v.cs % 4) == 0 && (v.coff % 4) == 0 && ((uintptr_t)v.p % 8) == 0); };
    return al(d.y) && al(d.y0) && al(d.y1) && al(d.r1) && al(d.r2) && al(d.m);
}

bool ssr_conv_ws_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force) {
    if (force ? !ssr_conv_ws_shape_ok(d) : !ssr_conv_ws_qualifies(d)) return false;
    // Cin <= 32: 32-channel output tiles, 2 workgroups per CU; Cin <= 64: 64-channel tiles held by one wave per SIMD
    // (288 weight registers) when the padded output width allows it
    // Cin <= 64: 32-channel tiles with TWO workgroups per CU (one's epilogue overlaps the other's MFMAs: 27.7 vs 36.0 us on
    // a 64->64 128x128 layer) for the epilogue variants that fit 256 registers without spilling (plain / LeakyReLU);
    // the others keep 64-channel tiles on one wave per SIMD.  SSR_CONV_WS21=0 / 2: never / always 32-channel tiles.
    static const int v21 = [] { const char* e = getenv("SSR_CONV_WS21"); return e ? atoi(e) : 1; }();
    const bool light = !d.y0.p && !d.y1.p && !d.r2.p && !d.r1.p && !d.m.p && !d.accumulate && d.alpha == 1.f;
    const bool mask_only = !d.y0.p && !d.y1.p && !d.r2.p && !d.r1.p && d.m.p && !d.accumulate && d.alpha == 1.f &&
                           d.act != SSR_ACT_LRELU;
    if (d.Cin <= 32) *rc = launch_ws<1, 1>(d, st);
    else if (v21 == 2 || (v21 >= 1 && light) || (v21 == 3 && mask_only)) *rc = launch_ws<2, 1>(d, st);
    else if (d.CoutPad % 64 == 0) *rc = launch_ws<2, 2>(d, st);
    else if (force) *rc = launch_ws<2, 1>(d, st);             // spills with the heavier epilogues: only on request
    else return false;
    return true;
}
