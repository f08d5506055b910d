// Big-tile convolution of the split-bf16 mode (SSR_F32X3: fp32 tensors in HBM, three bf16 MFMAs per product) for the wide
// and the large-spatial layers: the U-Net discriminator's 3x3 layers and their dgrads, the generator's 64x64 / 128x128 tail,
// the four 2x2 parity classes of the stride-2 dgrads, and the 4x4 stride-2 forward layers through the space-to-depth view.
//
// Round 5.  Until now these layers ran on the pipelined split kernel (conv.hip, conv_x3_kernel): 128 pixels x 64 channels per
// workgroup, a workgroup barrier per 16-channel chunk (~27 MFMAs of a wave), the whole 46-KB weight slab of a chunk re-staged for
// every 128 pixels - 0.26 of the mode's 833 TFLOP/s ceiling, 9.5 ms of the 35.5-ms step (+ 3.2 ms for the stride-2 forward layers on
// the exact fp32 MFMA and 1.8 ms for the parity classes; profiles/r05a_*).  This is conv_big.hip's structure with split operands:
//   * a workgroup (4 waves, one per SIMD) owns 32 x 16 pixels x 64 output channels, 4 pixel tiles x 2 channel tiles per wave;
//   * an LDS chunk is 16 input channels: 80-byte rows [16 hi bf16 | 16 lo bf16 | pad] for the 34 x 18 patch and the 9 x 64
//     weight rows (the packed weights arrive pre-split, misc.hip put_packed: copied as they are; a staged fp32 activation is
//     split ONCE, when the register-staged chunk is written to LDS: hi = bf16(x), lo = bf16(x - hi));
//   * a k-step is one tap: 4 weight + 8 pixel fragment reads feed 8 x 3 MFMAs (w_lo p_hi + w_hi p_lo + w_hi p_hi into the same
//     fp32 accumulator; the dropped lo lo term is 2^-16 relative): 0.5 LDS reads per MFMA; the reads of step s + 1 and the staging
//     loads of the next chunk are issued BETWEEN the MFMAs of step s, one memory instruction per MFMA;
//   * workgroups are persistent over images (same tile position, the image is a scalar offset): the chunk stream crosses image
//     boundaries; every staging load is one unconditional buffer load (an out-of-range offset reads zeros);
//   * operands swapped (A = weights, B = pixels): a lane ends with ONE pixel x 16 channels per channel tile, i.e. four 16-byte
//     fp32 vectors; the epilogue (full ssr_conv_desc contract) is straight-line code in four variants picked per image and sends
//     each pixel tile through a wave-private LDS slab, so that a load / store instruction covers the whole 256-byte rows of four
//     pixels (see `epilogue_impl`: the first form tested its feature flags per element and cost as much as the MFMAs).
// Same descriptor, packed-weight layout ([chunk16][tap][CoutPad][16 hi | 16 lo]) and results (to fp32 summation order) as
// conv_x3_kernel.
//
// Replaces nn.Conv2d 3x3 / 4x4-stride-2 forward and dgrad at /root/reference/ssr/archs/discriminator_arch.py:28-40,45-69 and
// rrdbnet_arch.py:109-112,127-136 in the fp32x3 arithmetic mode.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int XB_TH = 32, XB_TW = 16;
template <int KT> struct XbT {
    static constexpr int AROW = 80;                            // bytes per LDS row: 16 hi + 16 lo bf16 + 16 pad (bank = 20 r: conflict-free
                                                               // for rows distinct mod 16)
    static constexpr int VPP = 4;                              // 16-byte fp32 vectors per pixel and chunk
    static constexpr int PH = XB_TH + KT - 1, PW = XB_TW + KT - 1, NPIX = PH * PW;     // 612 / 561
    static constexpr int PATCH = NPIX * AROW;                  // 48,960 / 44,880
    static constexpr int WROWS = KT * KT * 64;
    static constexpr int WBYTES = WROWS * AROW;                // 46,080 / 20,480
    static constexpr int BUF = PATCH + WBYTES;
    static constexpr int BIAS = BUF;                           // 64 floats
    static constexpr int SLABS = BIAS + 256;                   // four wave-private epilogue slabs [32 px][64 co] fp32, 272-byte rows
    static constexpr int LDS = SLABS + 4 * 32 * 272;
    static constexpr int NPV = (NPIX * VPP + 255) / 256;       // patch vectors per thread: 10 / 9
    static constexpr int NWV = (WROWS * VPP + 255) / 256;      // weight vectors per thread: 9 / 4
    static constexpr int NSTEP = KT * KT;                      // k-steps (taps) per chunk, 24 MFMAs each
    static constexpr int TAIL = KT == 3 ? 2 : 1;               // steps at the end of a chunk without staging loads (the store does not wait)
    static constexpr int LPS = (NPV + NWV + NSTEP - TAIL - 1) / (NSTEP - TAIL);   // staging loads per step: 3 / 5
    static_assert(LDS <= 160 * 1024 && LPS <= 6, "LDS budget / load slots of a step");
};

#define XB_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef SSR_PROBE   // tools/bigx3_probe.hip: phase sums of wave 0 (s_memtime ticks): [0] whole, [1] prologue, [2] wait for everyone to leave the previous
                   // chunk, [3] store (incl. the wait for the staging loads), [4] barrier behind the store, [5] k-steps, [6] epilogues, [7] chunks
#define XB_T() (__builtin_amdgcn_s_memtime())
#define XB_ACC(k, t0) do { const unsigned long long t1_ = XB_T(); xb_t[k] += t1_ - (t0); (t0) = t1_; } while (0)
#else
#define XB_T() 0ull
#define XB_ACC(k, t0) do { } while (0)
#endif
constexpr int XB_OOB = 0x7ffffff0;
constexpr int XB_SLAB_PITCH = 272, XB_SLAB = 32 * XB_SLAB_PITCH;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xb_rsrc(const void* p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffff00L ? 0x7fffff00L : (bytes < 0 ? 0 : bytes)), 0x00020000);
}
// pixel of lane-slot s (0..31) of pixel tile m of wave w: tile rows 8w + 2m, 8w + 2m + 1; the second row is rotated so that the
// rows the 16-lane read groups touch are distinct mod 16 for every tap (as conv_big.hip)
template <int KT>
__device__ __forceinline__ void xb_pixel(int w, int m, int s, int& row, int& col) {
    row = 8 * w + 2 * m + (s >> 4);
    col = s < 16 ? s : ((s + 32 - (XB_TW + KT - 1)) & 15);
}
__device__ __forceinline__ void xb_split4(const u32x4& v, uint2& hi, uint2& lo) {
    const f32x4 f = __builtin_bit_cast(f32x4, v);
    bf16x4 h, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h[k] = (__bf16)f[k];
        l[k] = (__bf16)(f[k] - (float)h[k]);
    }
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

// H: the fp16-split FORWARD arithmetic (SSR_F32H, include/ssr_hip.h) on the same data path: fp16 pieces, v_mfma_f32_32x32x16_f16; the
// accumulators start at 2^SSR_F32H_WSHIFT x bias and leave through the epilogue multiplied by 2^-SSR_F32H_WSHIFT (both exact)
template <int KT, bool H = false>
__device__ __forceinline__ void conv_bigx3_body(const ssr_conv_desc& d) {
    using T = XbT<KT>;
    constexpr int AROW = T::AROW, VPP = T::VPP, PW = T::PW, NPIX = T::NPIX, PATCH = T::PATCH, NPV = T::NPV, NWV = T::NWV, NSTEP = T::NSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int tiles_x = (d.Gw + XB_TW - 1) / XB_TW, tiles_y = (d.Gh + XB_TH - 1) / XB_TH;
    // persistent over images: tile position (ty_i, tx_i) of images n0, n0 + G, n0 + 2G, ... as ONE chunk stream
    const int tpi = tiles_x * tiles_y, G = gridDim.x / tpi;
    const int tp = blockIdx.x % tpi, n0 = blockIdx.x / tpi;
    const int tx_i = tp % tiles_x, ty_i = tp / tiles_x;
    const int nimg = (d.N - n0 + G - 1) / G;
    const int gy0 = ty_i * XB_TH, gx0 = tx_i * XB_TW;
    const int co0 = blockIdx.y * 64;
    const int upshift = d.up == 2 ? 1 : 0;
    const int LH = d.Hi << upshift, LW = d.Wi << upshift;
    const int nchunks = (d.Cin + 15) / 16;
    const int wchunk = KT * KT * d.CoutPad * 64;               // packed bytes per 16-channel chunk
    const int T_ = nimg * nchunks;                             // length of this workgroup's chunk stream
    const int ximg = d.Hi * d.Wi * d.x.cs * 4, oimg = d.Ho * d.Wo;   // per-image strides (bytes of x / output pixels)

    unsigned long long xb_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xb_t0 = XB_T();
    const unsigned long long xb_start = xb_t0;
    (void)xb_t; (void)xb_start;
    float* bias_lds = reinterpret_cast<float*>(smem + T::BIAS);
    if (tid < 64) bias_lds[tid] = (d.bias && co0 + tid < d.Cout) ? d.bias[co0 + tid] * (H ? SSR_F32H_WSCALE : 1.f) : 0.f;

    // ---- staging descriptors (independent of the chunk) ----
    // space-to-depth view (ssr_conv_desc.s2d, the 4x4 stride-2 forward layers as 2x2 layers over 4 x C channels): chunk c is parity
    // class q = c / (C/16) of source channels [16 (c mod C/16), +16); patch pixel (Y, X) reads source pixel (2Y-1 + (q>>1), 2X-1 + (q&1))
    const bool s2d = KT == 2 && d.s2d != 0;
    const int cpc_shift = s2d ? 31 - __builtin_clz((unsigned)(d.Cin >> 6)) : 0;   // log2(source channels / 16)
    const int x_cs = d.x.cs, x_Wi = d.Wi, x_Cin = d.Cin;
    // vector j of a thread = lane-slot tid + 256 j: pixel (or weight row) tid / 4 + 64 j, 16-byte part tid % 4 - so the LDS offsets
    // and the weight offsets of the vectors differ by constants (64 rows; one tap = CoutPad rows of 64 bytes) and only the patch
    // pixels' global offsets need a register each
    static_assert((T::WROWS * VPP) % 256 == 0, "no partial weight vector");
    const int p4 = tid >> 2, part4 = tid & 3;
    const int plo0 = p4 * AROW + part4 * 8;                    // hi half of the row; the lo half lies 32 bytes further
    const int wlo0 = PATCH + p4 * AROW + part4 * 16;
    const int wgo0 = (co0 + p4) * 64 + part4 * 16, wstep = d.CoutPad * 64;
    constexpr int PTAIL = NPIX * VPP - (NPV - 1) * 256;        // threads that own a last patch vector
    int pgo[NPV];
    unsigned pmk[KT == 2 ? NPV : 1];
    const int climit = s2d ? 0x7fffffff : x_Cin - (tid % VPP) * 4;   // chunk start c0 is inside this lane's 4 channels iff c0 < climit
#pragma unroll
    for (int q = 0; q < NPV; ++q) {
        const int v = tid + q * 256;
        const int pix = v / VPP, part = v % VPP;
        const int py = pix / PW, px = pix - py * PW;
        const int ly = gy0 + py - d.pad_y, lx = gx0 + px - d.pad_x;
        if constexpr (KT == 2) {
            if (s2d) {
                const int sy = 2 * ly - 1, sx = 2 * lx - 1;
                const bool y0 = sy >= 0 && sy < d.Hi, y1 = sy + 1 >= 0 && sy + 1 < d.Hi;
                const bool x0 = sx >= 0 && sx < d.Wi, x1 = sx + 1 >= 0 && sx + 1 < d.Wi;
                pmk[q] = v < NPIX * VPP ? (unsigned)(y0 && x0) | (unsigned)(y0 && x1) << 1 | (unsigned)(y1 && x0) << 2 | (unsigned)(y1 && x1) << 3 : 0u;
                // the q = 0 source pixel may lie one row / column outside the image: the resource base is one row and one pixel
                // BELOW the tensor (xbias), so that this lane offset is never negative
                pgo[q] = ((sy * d.Wi + sx) * d.x.cs + d.x.coff + part * 4) * 4 + (d.Wi + 1) * d.x.cs * 4;
            } else {
                const bool ok = v < NPIX * VPP && ly >= 0 && ly < LH && lx >= 0 && lx < LW;
                pmk[q] = ok ? 1u : 0u;
                pgo[q] = ok ? (((ly >> upshift) * d.Wi + (lx >> upshift)) * d.x.cs + d.x.coff + part * 4) * 4 : XB_OOB;
            }
        } else {
            const bool ok = v < NPIX * VPP && ly >= 0 && ly < LH && lx >= 0 && lx < LW;
            pgo[q] = ok ? (((ly >> upshift) * d.Wi + (lx >> upshift)) * d.x.cs + d.x.coff + part * 4) * 4 : XB_OOB;
        }
    }
    const long xbias = s2d ? (long)(d.Wi + 1) * d.x.cs * 4 : 0;
    const __amdgpu_buffer_rsrc_t rsx = xb_rsrc(reinterpret_cast<const char*>(d.x.p) - xbias, (long)d.N * ximg + xbias),
                                 rsw = xb_rsrc(d.w, (long)nchunks * wchunk);
    u32x4 rp[NPV], rw[NWV];
    struct LoadPos { int q, c0, xs, ws; };
    auto load_pos = [&](int nn, int c) {                        // the wave-uniform part of a chunk's loads
        LoadPos p;
        p.c0 = c * 16;
        if constexpr (KT == 2) {
            p.q = s2d ? c >> cpc_shift : 0;
            p.xs = nn * ximg + (s2d ? (((p.q >> 1) * x_Wi + (p.q & 1)) * x_cs + ((c - (p.q << cpc_shift)) << 4)) * 4 : p.c0 * 4);
        } else {
            p.q = 0;
            p.xs = nn * ximg + p.c0 * 4;
        }
        p.ws = c * wchunk;
        return p;
    };
    auto load_one = [&](const LoadPos& lp, auto jc) {           // vector j of the chunk at lp: ONE unconditional buffer load
        constexpr int j = decltype(jc)::value;
        if constexpr (j < NPV) {
            bool ok = lp.c0 < climit;
            if constexpr (KT == 2) ok = ok && ((pmk[j] >> lp.q) & 1u);
            rp[j] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? pgo[j] : XB_OOB, lp.xs, 0);
        } else {
            rw[j - NPV] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wgo0, lp.ws + (j - NPV) * wstep, 0);   // weight row tap (j - NPV), co p4
        }
    };
    auto load_chunk = [&](int nn, int c) { const LoadPos lp = load_pos(nn, c); static_for<0, NPV + NWV>([&](auto jc) { load_one(lp, jc); }); };
    auto store_chunk = [&]() {
        static_for<0, NPV>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            uint2 hi, lo;
            split_f32x4<H>(rp[j], hi, lo);
            if (j + 1 < NPV || tid < PTAIL) {
                *reinterpret_cast<uint2*>(smem + plo0 + j * 64 * AROW) = hi;
                *reinterpret_cast<uint2*>(smem + plo0 + j * 64 * AROW + 32) = lo;
            }
        });
        static_for<0, NWV>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            *reinterpret_cast<u32x4*>(smem + wlo0 + j * 64 * AROW) = rw[j];
        });
    };

    // ---- this lane's pixels (one per pixel tile) ----
    int a_off[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        int row, col;
        xb_pixel<KT>(wave, m, i, row, col);
        a_off[m] = (row * PW + col) * AROW + g * 16;
    }
    const int b_off = PATCH + i * AROW + g * 16;
    const int co_l = 4 * g;                                    // + t*32 + 8*q4 + e: this lane's 16 channels of a channel tile

    load_chunk(n0, 0);
    __syncthreads();                                           // bias table
    XB_ACC(1, xb_t0);
    f32x16 acc[4][2];
    auto acc_init = [&]() {                                    // accumulators start at the bias
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + t * 32 + 8 * q4 + co_l);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m][t][4 * q4 + e] = bq[e];
            }
    };

    // ---- epilogue of image nn: the whole ssr_conv_desc contract on 16-byte fp32 vectors.  The accumulators of a pixel tile (lane =
    //      one pixel x 16 channels per channel tile) go through a wave-private LDS slab [32 px][64 co] (272-byte rows: conflict-free
    //      both ways) and come back as lane = (pixel 4h + lane / 16, channels 4 (lane % 16) ..): a wave's load / store instruction
    //      then covers the whole 256-byte rows of 4 pixels.  (Round 5, tools/bigx3_probe: stored straight from the MFMA layout - 32
    //      pixels x 32 bytes per instruction, every 128-byte line written in four pieces - the epilogues took 32 k ticks per image,
    //      as long as the 64-channel layers' MFMAs; 59 k with a residual and a second output.) ----
    char* slab = smem + T::SLABS + wave * XB_SLAB;
    // Straight-line code: the feature set is two template flags picked by two uniform branches per image (OPS: any of r1 / r2 /
    // accumulate / mask - absent ones are loaded through an out-of-range offset, i.e. as zeros, and enter with a zero factor; Y01: the
    // second / third outputs).  With run-time `if (has_r1) ...` inside the unrolled element loops the epilogue was 2,500 branches
    // of code and 23 k of its 32 k ticks per image were branch issue (lesson 2 of round 1, relearned).
    auto epilogue_impl = [&](int nn, auto opsc, auto y01c) __attribute__((always_inline)) {
        constexpr bool OPS = decltype(opsc)::value, Y01 = decltype(y01c)::value;
        // (resources are set up HERE, not ahead of the chunk stream: six descriptors live across the main loop cost 100 - 600 spilled SGPRs)
        const long obytes = (long)d.N * oimg * 4;
        const __amdgpu_buffer_rsrc_t rs_r1 = xb_rsrc(d.r1.p, d.r1.p ? obytes * d.r1.cs : 0), rs_r2 = xb_rsrc(d.r2.p, d.r2.p ? obytes * d.r2.cs : 0),
                                     rs_m = xb_rsrc(d.m.p, d.m.p ? obytes * d.m.cs : 0), rs_y = xb_rsrc(d.y.p, obytes * d.y.cs),
                                     rs_y0 = xb_rsrc(d.y0.p, d.y0.p ? obytes * d.y0.cs : 0), rs_y1 = xb_rsrc(d.y1.p, d.y1.p ? obytes * d.y1.cs : 0);
        const bool has_r1 = d.r1.p != nullptr, has_r2 = d.r2.p != nullptr, has_m = d.m.p != nullptr, has_acc = d.accumulate != 0,
                   has_y0 = d.y0.p != nullptr, has_y1 = d.y1.p != nullptr, m_relu = d.m_relu != 0;
        // act(x) = max(x, slope x): LeakyReLU 0.2, ReLU 0, none 1
        const float slope = d.act == SSR_ACT_LRELU ? 0.2f : d.act == SSR_ACT_RELU ? 0.f : 1.f;
        const float alpha = d.alpha, beta1 = has_r1 ? d.beta1 : 0.f, beta2 = has_r2 ? d.beta2 : 0.f, gacc = has_acc ? 1.f : 0.f;
        const int y_cs = d.y.cs, y_co = d.y.coff, y0_cs = d.y0.cs, y0_co = d.y0.coff, y1_cs = d.y1.cs, y1_co = d.y1.coff;
        const int r1_cs = d.r1.cs, r1_co = d.r1.coff, r2_cs = d.r2.cs, r2_co = d.r2.coff, m_cs = d.m.cs, m_co = d.m.coff;
        const int Gh = d.Gh, Gw = d.Gw, oys = d.oys, oyo = d.oyo, oxs = d.oxs, oxo = d.oxo, Wo = d.Wo;
        const int pimg = nn * oimg;
        const int hp = lane >> 4, c = co0 + (lane & 15) * 4;
        const bool cok = c < d.Cout;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            int po[8];                                          // output pixel of slot 4h + hp (clamped into the grid), sign bit = outside
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                int row, col;
                xb_pixel<KT>(wave, m, 4 * h + hp, row, col);
                const int gy = gy0 + row, gx = gx0 + col;
                const bool ok = gy < Gh && gx < Gw;
                const int cy = gy < Gh ? gy : Gh - 1, cx = gx < Gw ? gx : Gw - 1;
                po[h] = (pimg + (cy * oys + oyo) * Wo + cx * oxs + oxo) | (ok && cok ? 0 : (int)0x80000000);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 v = {acc[m][t][4 * q4], acc[m][t][4 * q4 + 1], acc[m][t][4 * q4 + 2], acc[m][t][4 * q4 + 3]};
                    *reinterpret_cast<f32x4*>(slab + i * XB_SLAB_PITCH + (t * 32 + 8 * q4 + co_l) * 4) = v;
                }
            // (the operands of four pixel slots at a time: 64 registers in flight, not 128)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                u32x4 q1[OPS ? 4 : 1], q2[OPS ? 4 : 1], qa[OPS ? 4 : 1], qm[OPS ? 4 : 1];
                if constexpr (OPS) {
#pragma unroll
                    for (int h4 = 0; h4 < 4; ++h4) {
                        const int pp = po[4 * hh + h4] & 0x7fffffff;
                        q1[h4] = __builtin_amdgcn_raw_buffer_load_b128(rs_r1, cok && has_r1 ? (pp * r1_cs + r1_co + c) * 4 : XB_OOB, 0, 0);
                        q2[h4] = __builtin_amdgcn_raw_buffer_load_b128(rs_r2, cok && has_r2 ? (pp * r2_cs + r2_co + c) * 4 : XB_OOB, 0, 0);
                        qa[h4] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, cok && has_acc ? (pp * y_cs + y_co + c) * 4 : XB_OOB, 0, 0);
                        qm[h4] = __builtin_amdgcn_raw_buffer_load_b128(rs_m, cok && has_m ? (pp * m_cs + m_co + c) * 4 : XB_OOB, 0, 0);
                    }
                }
#pragma unroll
                for (int h4 = 0; h4 < 4; ++h4) {
                    const int h = 4 * hh + h4;
                    f32x4 a = *reinterpret_cast<const f32x4*>(slab + (4 * h + hp) * XB_SLAB_PITCH + (lane & 15) * 16);
                    if constexpr (H) a *= SSR_F32H_UNSCALE;
                    const bool ok = po[h] >= 0;
                    const int pp = po[h] & 0x7fffffff;
                    f32x4 v, s0, s1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = fmaxf(a[e], slope * a[e]) * alpha;
                        s0[e] = x;
                        if constexpr (OPS) {
                            x += beta1 * __builtin_bit_cast(f32x4, q1[h4])[e] + beta2 * __builtin_bit_cast(f32x4, q2[h4])[e] + gacc * __builtin_bit_cast(f32x4, qa[h4])[e];
                            s1[e] = x;
                            const float mv = __builtin_bit_cast(f32x4, qm[h4])[e];
                            const float f = m_relu ? (mv > 0.f ? 1.f : 0.f) : lrelu_grad_from_out(mv);
                            x *= has_m ? f : 1.f;
                        } else {
                            s1[e] = x;
                        }
                        v[e] = x;
                    }
                    if constexpr (Y01) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s0), rs_y0, ok && has_y0 ? (pp * y0_cs + y0_co + c) * 4 : XB_OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s1), rs_y1, ok && has_y1 ? (pp * y1_cs + y1_co + c) * 4 : XB_OOB, 0, 0);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_y, ok ? (pp * y_cs + y_co + c) * 4 : XB_OOB, 0, 0);
                }
            }
        }
    };
    auto epilogue = [&](int nn) __attribute__((always_inline)) {
        const bool ops = d.r1.p != nullptr || d.r2.p != nullptr || d.m.p != nullptr || d.accumulate != 0;
        const bool y01 = d.y0.p != nullptr || d.y1.p != nullptr;
        if (ops) {
            if (y01) epilogue_impl(nn, std::true_type{}, std::true_type{});
            else epilogue_impl(nn, std::true_type{}, std::false_type{});
        } else {
            if (y01) epilogue_impl(nn, std::false_type{}, std::true_type{});
            else epilogue_impl(nn, std::false_type{}, std::false_type{});
        }
    };

    // ---- one chunk: (store) - barrier - 9 / 4 k-steps of 24 MFMAs with the fragment reads of the next step and the staging loads of
    //      the next chunk between them ----
    auto chunk = [&](int c, bool stored, int nn1, int c1, auto hnc) __attribute__((always_inline)) {   // c = stream position; (nn1, c1) = position c + 1
        constexpr bool has_next = decltype(hnc)::value;
        xb_t0 = XB_T();
        if (!stored) {                                         // (an image's first chunk was stored before the previous epilogue)
            if (c > 0) XB_BAR();                               // everyone is finished reading the previous chunk
            XB_ACC(2, xb_t0);
            store_chunk();
            XB_ACC(3, xb_t0);
        }
        XB_BAR();
        XB_ACC(4, xb_t0);
        const LoadPos lp1 = load_pos(nn1, c1);
        bf16x8 wq[2][2][2], pq[2][4][2];                       // [step parity][tile][hi | lo]
        auto issue1 = [&](auto sc, auto kc) {                  // memory operation k (0..11) of step s: 4 weight fragments, 8 pixel fragments
            constexpr int s_ = decltype(sc)::value, k = decltype(kc)::value;
            if constexpr (k < 4)
                wq[s_ & 1][k >> 1][k & 1] = *reinterpret_cast<const bf16x8*>(smem + b_off + (s_ * 64 + (k >> 1) * 32) * AROW + (k & 1) * 32);
            else
                pq[s_ & 1][(k - 4) >> 1][k & 1] =
                    *reinterpret_cast<const bf16x8*>(smem + a_off[(k - 4) >> 1] + ((s_ / KT) * PW + s_ % KT) * AROW + (k & 1) * 32);
        };
        static_for<0, 12>([&](auto kc) { issue1(std::integral_constant<int, 0>{}, kc); });
        static_for<0, NSTEP>([&](auto sc) {
            constexpr int s_ = decltype(sc)::value;
            static_for<0, 24>([&](auto kc) {
                constexpr int k = decltype(kc)::value, m = k / 6, t = (k % 6) / 3, p = k % 3;
                // w_lo p_hi, w_hi p_lo, w_hi p_hi (the order of conv_x3_kernel: small terms first)
                acc[m][t] = split_mfma<H>(wq[s_ & 1][t][p == 0 ? 1 : 0], pq[s_ & 1][m][p == 1 ? 1 : 0], acc[m][t]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (k < 12) {
                    if constexpr (s_ + 1 < NSTEP) issue1(std::integral_constant<int, s_ + 1>{}, kc);
                } else if constexpr (has_next && (k & 1) == 0 && (k - 12) / 2 < T::LPS) {
                    constexpr int j = s_ * T::LPS + (k - 12) / 2;
                    if constexpr (j < NPV + NWV) load_one(lp1, std::integral_constant<int, j>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        __builtin_amdgcn_sched_barrier(0);
        XB_ACC(5, xb_t0);
        xb_t[7] += 1;
    };
    // every chunk but the stream's last one runs in the loops; the last chunk is peeled (no further loads)
    int gc = 0;
    for (int k = 0; k < nimg; ++k) {
        const int nn = n0 + k * G;
        const bool last_img = k + 1 == nimg;
        acc_init();
        for (int c = 0; c < (last_img ? nchunks - 1 : nchunks); ++c, ++gc) {
            const bool last_c = c + 1 == nchunks;
            chunk(gc, k > 0 && c == 0, last_c ? nn + G : nn, last_c ? 0 : c + 1, std::true_type{});
        }
        if (!last_img) {
            // image boundary: the next image's first chunk leaves the staging registers BEFORE the epilogue; its LDS stores, and the
            // epilogue's global stores, drain under the next MFMAs
            xb_t0 = XB_T();
            XB_BAR();
            XB_ACC(2, xb_t0);
            store_chunk();
            XB_ACC(3, xb_t0);
            epilogue(nn);
            XB_ACC(6, xb_t0);
        }
    }
    const int nl = n0 + (nimg - 1) * G;
    chunk(gc, nimg > 1 && nchunks == 1, nl, 0, std::false_type{});
    xb_t0 = XB_T();
    epilogue(nl);
    XB_ACC(6, xb_t0);
#ifdef SSR_PROBE
    if (tid == 0) {
        xb_t[0] = XB_T() - xb_start;
        const int b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (int k = 0; k < 8; ++k) g_probe[b * 8 + k] = xb_t[k];
    }
#endif
    (void)T_;
}

// up to four descriptors of identical geometry (the output-parity classes of a stride-2 transposed conv), blockIdx.z selects.  A single
// layer is launched through the same kernel with one descriptor: indexed by blockIdx.z the fields are fetched from the kernel-argument
// segment where they are used, while a by-value ssr_conv_desc is preloaded into ~100 SGPRs that then spill around the chunk loop
// (117 / 179 spilled SGPRs against 8 / 13)
struct ssr_conv_desc4x { ssr_conv_desc d[4]; };
template <int KT>
__global__ __launch_bounds__(256, 1) void conv_bigx3_kernel4(const ssr_conv_desc4x p) {
    conv_bigx3_body<KT>(p.d[blockIdx.z]);
}
template <int KT>
__global__ __launch_bounds__(256, 1) void conv_bigh3_kernel4(const ssr_conv_desc4x p) {      // the fp16-split form (SSR_F32H)
    conv_bigx3_body<KT, true>(p.d[blockIdx.z]);
}

template <int KT, bool H = false>
int launch_bigx3(const ssr_conv_desc* ds, int n, hipStream_t st) {
    constexpr int lds = XbT<KT>::LDS;
    const ssr_conv_desc& d = ds[0];
    // grid.x = (tile positions per image) x G image groups; G minimises rounds x images-per-workgroup on the device's CUs
    const int tpi = ((d.Gh + XB_TH - 1) / XB_TH) * ((d.Gw + XB_TW - 1) / XB_TW);
    static const int ncu = [] { int dev = 0, v = 256; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 256; return v > 0 ? v : 256; }();
    int G = d.N;
    {
        const long per_img = (long)tpi * (d.CoutPad / 64) * n;
        long best = -1;
        for (int g = 1; g <= d.N; ++g) {
            const long rounds = (per_img * g + ncu - 1) / ncu, cost = rounds * ((d.N + g - 1) / g);
            const long key = (cost << 24) | (rounds << 12) | (4095 - (g > 4095 ? 4095 : g));
            if (best < 0 || key < best) { best = key; G = g; }
        }
    }
    if (const char* e = getenv("SSR_CONV_BIG_G")) {           // test hook: force the number of image groups (1..N)
        const int g = atoi(e);
        if (g >= 1) G = g < d.N ? g : d.N;
    }
    const int tiles = tpi * G;
    auto kern = H ? conv_bigh3_kernel4<KT> : conv_bigx3_kernel4<KT>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    ssr_conv_desc4x p;
    for (int k = 0; k < 4; ++k) p.d[k] = ds[k < n ? k : 0];
    hipLaunchKernelGGL(kern, dim3(tiles, d.CoutPad / 64, n), dim3(256), lds, st, p);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

}  // namespace

// everything except the grid-size heuristic
bool ssr_conv_bigx3_shape_ok(const ssr_conv_desc& d) {
    if ((d.dtype != SSR_F32X3 && d.dtype != SSR_F32H) || d.fix_list) return false;
    const bool k3 = d.KH == 3 && d.KW == 3 && d.pad_y == 1 && d.pad_x == 1;
    const bool k2 = d.KH == 2 && d.KW == 2 && (d.pad_y == 0 || d.pad_y == 1) && (d.pad_x == 0 || d.pad_x == 1);
    if (!(k3 || k2) || d.stride != 1 || d.x2.p) return false;
    if (d.Cin < 8 || (d.CoutPad % 64) != 0 || (d.Cout % 4) != 0) return false;
    if (d.s2d) {   // internal form of a 4x4 stride-2 layer (conv.hip): 2x2, pad 0, Cin = 4 x source channels
        const int cpc = d.Cin / 64;
        if (!k2 || d.pad_y != 0 || d.pad_x != 0 || d.up != 1 || (d.Cin % 64) != 0 || cpc < 1 || (cpc & (cpc - 1)) != 0) return false;
        if ((d.Hi % 2) != 0 || (d.Wi % 2) != 0 || d.Gh != d.Hi / 2 || d.Gw != d.Wi / 2) return false;
    } else if (d.Gh != (d.Hi << (d.up == 2)) || d.Gw != (d.Wi << (d.up == 2))) return false;
    if (d.r1.p && d.r1_nc < d.Cout) return false;
    if (d.r2.p && d.r2_nc < d.Cout) return false;
    if (d.m.p && !(d.m_c0 == 0 && d.m_c1 >= d.Cout)) return false;
    // buffer addressing: every tensor is reached through a 32-bit byte offset from its resource base
    const long lim = 0x7fffff00L;
    const long xin = ((long)d.N * d.Hi * d.Wi + d.Wi + 1) * d.x.cs * 4, npo = (long)d.N * d.Ho * d.Wo * 4;
    if (xin > lim || npo * d.y.cs > lim || (d.y0.p && npo * d.y0.cs > lim) || (d.y1.p && npo * d.y1.cs > lim) || (d.r1.p && npo * d.r1.cs > lim) ||
        (d.r2.p && npo * d.r2.cs > lim) || (d.m.p && npo * d.m.cs > lim))
        return false;
    if ((long)((d.Cin + 15) / 16) * d.KH * d.KW * d.CoutPad * 64 > lim) return false;
    auto al16 = [](const ssr_view& v) { return !v.p || ((v.cs % 4) == 0 && (v.coff % 4) == 0 && ((uintptr_t)v.p % 16) == 0); };
    return d.x.p && d.y.p && al16(d.x) && al16(d.y) && al16(d.y0) && al16(d.y1) && al16(d.r1) && al16(d.r2) && al16(d.m);
}

bool ssr_conv_bigx3_qualifies(const ssr_conv_desc& d) {
    static const bool off = [] { const char* e = getenv("SSR_X3_BIGTILE"); return e && e[0] == '0'; }();
    if (off || !ssr_conv_bigx3_shape_ok(d) || d.KH != 3) return false;
    // The choice depends on the LAYER only, never on the batch: an image's result must not depend on how many images are launched
    // together (whole-tile inference deals chunks to ranks and batches; a last partial batch on another kernel would change bytes -
    // tests/test_gpu_baseline_shapes.py::test_infer_grid_tile_end_to_end_vs_oracle).  >= 4 tile x channel-group workgroups per image
    // (64 x 64 grids and up, or >= 256 output channels at 32 x 32), 32-row tiles not half empty; the 32 x 32 body goes to conv_x3q.hip.
    const long per_img = (long)((d.Gh + XB_TH - 1) / XB_TH) * ((d.Gw + XB_TW - 1) / XB_TW) * (d.CoutPad / 64);
    return per_img >= 4 && d.Gh >= 24;
}

bool ssr_conv_bigx3_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force) {
    if (force ? !ssr_conv_bigx3_shape_ok(d) : !ssr_conv_bigx3_qualifies(d)) return false;
    if (d.dtype == SSR_F32H) *rc = d.KH == 3 ? launch_bigx3<3, true>(&d, 1, st) : launch_bigx3<2, true>(&d, 1, st);
    else *rc = d.KH == 3 ? launch_bigx3<3>(&d, 1, st) : launch_bigx3<2>(&d, 1, st);
    return true;
}

// n <= 4 parity-class descriptors (2x2 stride 1, identical geometry) in one launch
bool ssr_conv_bigx3_batch_try(const ssr_conv_desc* ds, int n, hipStream_t st, int* rc) {
    const char* e = getenv("SSR_X3_BIGTILE2");               // 0: never, 2: always (tests), default: by grid size
    const bool off = e && e[0] == '0', always = e && e[0] == '2';
    if (off || n < 1 || n > 4) return false;
    for (int k = 0; k < n; ++k)
        if (!ssr_conv_bigx3_shape_ok(ds[k]) || ds[k].dtype != SSR_F32X3 || ds[k].KH != 2 || ds[k].s2d) return false;
    const ssr_conv_desc& d = ds[0];
    for (int k = 1; k < n; ++k)
        if (ds[k].N != d.N || ds[k].Gh != d.Gh || ds[k].Gw != d.Gw || ds[k].CoutPad != d.CoutPad || ds[k].Cin != d.Cin || ds[k].Hi != d.Hi || ds[k].Wi != d.Wi ||
            ds[k].up != d.up)
            return false;
    const long per_img = (long)((d.Gh + XB_TH - 1) / XB_TH) * ((d.Gw + XB_TW - 1) / XB_TW) * (d.CoutPad / 64) * n;
    if (!always && (per_img < 8 || d.Gh < 24)) return false;  // small grids / half-empty 32-row tiles: pipelined kernel (batch-independent rule)
    *rc = launch_bigx3<2>(ds, n, st);
    return true;
}
