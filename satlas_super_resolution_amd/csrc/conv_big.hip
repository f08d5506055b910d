// Big-tile 3x3 stride-1 convolution for the wide U-Net discriminator layers (bf16, Cin >= 64, CoutPad % 64 == 0).
//
// The generic kernel (conv.hip) gives a workgroup 128 pixels x 64 output channels: per 32-channel chunk it stages
// an 11.5 KB input patch and a 36.9 KB weight slab, i.e. every 128 pixels re-read the weights (conv6 forward at
// 128x128, B = 16: 300 MB of weight traffic against 94 MB of activations), each wave fetches 12 KB per 1152 cycles of
// MFMAs — twice what a wave can pull (~6.4 B/clk, tools/l2_probe.hip) — and needs 1.5 LDS operand reads per MFMA.
// rocprofv3 r01: 400..630 TFLOP/s on conv4/5/6 and their dgrads.
// Here a workgroup (4 waves, one per SIMD, the whole register file) owns 32 x 16 pixels x 64 output channels:
//   * register tiling 4 pixel tiles x 2 channel tiles per wave (8 accumulators): 6 operand reads feed 8 MFMAs
//     (0.75 per MFMA) and the weight slab is amortised over 512 pixels;
//   * per 32-channel chunk: 49 KB patch + 46 KB weights in ONE LDS stage (80-byte padded rows), the next chunk
//     waits in registers (19 x 16 B per lane) while 144 MFMAs per wave run: 4.1 B/clk per wave; inside a k-step every
//     LDS read / staging load is issued BETWEEN two MFMAs (bunched at the step boundary they idle the matrix pipe);
//   * 2x2 kernels (KT = 2: the parity classes of a stride-2 dgrad, or a 4x4 stride-2 forward layer through the
//     space-to-depth view of ssr_conv_desc.s2d) have only 64 MFMAs per chunk: two chunk buffers and one continuous
//     k-step stream (store chunk c+1 / request chunk c+2 under the MFMAs of chunk c, two bare barriers per chunk);
//   * workgroups are PERSISTENT over images (same tile position, image = scalar offset): the chunk stream crosses image
//     boundaries, so the first chunk's latency and the epilogue's stores hide under MFMAs; the stream's last chunk is
//     a separate instantiation whose (dead) staging registers hold the prefetched epilogue operands;
//   * lane i of a pixel tile owns pixel (row i >> 4, column i or (i + 14) & 15 for the second row): rows are distinct
//     mod 16 inside the hardware's 16-lane read groups for every tap -> conflict-free ds_read_b128 (pitch 18);
//   * operands swapped (A = weights, B = pixels): a lane ends with ONE pixel x 16 channels per channel tile; the
//     epilogue is pixel-per-lane, branch-free per feature set (template), outputs leave as 16-byte vectors through
//     144-byte-pitch transpose slabs; staging loads, epilogue operands and output stores use buffer addressing (SGPR
//     resource + 32-bit lane offset + scalar offset) — 64-bit pointers cost a register pair per vector.
// Same descriptor and epilogue contract as conv.hip (ssr_conv_desc).
//
// Replaces nn.Conv2d 3x3 forward / dgrad at /root/reference/ssr/archs/discriminator_arch.py:35-37,55,59,63
// (conv4, conv5, conv6 and the dgrads of conv4..conv6) when the pixel count fills the chip.
#include "common.h"
#include <cstdlib>

#ifdef SSR_PROBE   // tools/big_probe.hip
#define BPROBE(k) do { if (threadIdx.x == 0) g_probe[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define BPROBE_C(k) do { if (threadIdx.x == 0 && c == 1) g_probe[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BPROBE(k)
#define BPROBE_C(k)
#endif

namespace {

constexpr int CB_TH = 32, CB_TW = 16;
// KT x KT taps: 3 (the 3x3 layers) or 2 (one output-parity class of a 4x4 stride-2 transposed conv: conv1..conv3 dgrad,
// or a 4x4 stride-2 forward layer through the space-to-depth view).  CK = input channels per LDS chunk: 32, or 16 for the
// double-buffered 3x3 pipeline (two 32-channel 3x3 chunks do not fit the 160 KB of LDS, two 16-channel ones do).
// output transpose slabs: [32 px][64 co] with 144-byte rows (a 128-byte pitch puts the 32 lanes of a ds_write_b64 on two
// bank pairs: a 16-way conflict that cost 8k of the 11k epilogue cycles, tools/big_probe.hip)
constexpr int CB_SROW = 72, CB_SLAB = 32 * CB_SROW;
template <int KT, int CK> struct CbT {
    static constexpr int AROW = CK * 2 + 16;                   // bytes per LDS row: CK bf16 + 16 pad (80: bank = 20 r, 48: 12 r — both
                                                               // conflict-free for rows distinct mod 16)
    static constexpr int VPP = CK / 8, KSUB = CK / 16;         // 16-byte vectors per row; 16-channel k-substeps per tap
    static constexpr bool DB = KT == 2 || CK == 16;            // double-buffered continuous k-step stream (see conv_big_body)
    static constexpr int PH = CB_TH + KT - 1, PW = CB_TW + KT - 1, NPIX = PH * PW;     // 612 / 561
    static constexpr int PATCH = NPIX * AROW;                  // 48,960 / 44,880 (29,376 for 3x3 with CK = 16)
    static constexpr int WROWS = KT * KT * 64;
    static constexpr int WBYTES = WROWS * AROW;                // 46,080 / 20,480 (27,648)
    static constexpr int BUF = PATCH + WBYTES;                 // one chunk: 95,040 / 65,360 (57,024)
    static constexpr int NBUF = DB ? 2 : 1;
    static constexpr int BIAS = NBUF * BUF;                    // 64 floats
    // output transpose slabs: the single-buffered kernel has a dedicated area behind the bias table (the chunk area already
    // holds the NEXT image's first chunk when an epilogue runs); the double-buffered ones reuse the buffer just consumed
    static constexpr int SLABS = DB ? 0 : BIAS + 256, SLABB = DB ? BUF : 4 * 2 * 32 * 72 * 2;
    static constexpr int LDS = BIAS + 256 + (DB ? 0 : SLABB);
    static constexpr int NPV = (NPIX * VPP + 255) / 256;       // patch vectors per thread: 10 / 9 (5)
    static constexpr int NWV = (WROWS * VPP + 255) / 256;      // weight vectors per thread: 9 / 4 (5)
    static constexpr int NSTEP = KT * KT * KSUB;               // k-steps per chunk: 18 / 8 (9)
    static_assert(LDS <= 160 * 1024 && 4 * 2 * 32 * 72 * 2 <= SLABB, "LDS budget / two slabs per wave fit");
};

#ifndef CB_TAIL
#define CB_TAIL 4
#endif
#define CB_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
constexpr int CB_LRELU = 1, CB_MASK = 2, CB_R1 = 4, CB_ACC = 8, CB_Y0 = 16, CB_GENERIC = -1;   // Y0: second output = activation before the residual
typedef __bf16 bf16x4c __attribute__((ext_vector_type(4)));
typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
// Buffer addressing: SGPR resource + 32-bit per-lane byte offset + uniform (SGPR) byte offset.  With 64-bit global pointers
// hipcc keeps a loop-invariant pointer PAIR per staging vector / epilogue operand (38 + registers; the persistent 3x3
// kernel spilled them to scratch); here the per-lane part stays one register and the image / chunk offset is scalar.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cb_rsrc(const void* p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}
__device__ __forceinline__ float cb_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float cb_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// pixel of lane-slot s (0..31) of pixel tile m of wave w: tile rows 8w + 2m, 8w + 2m + 1; the second row is rotated by
// 32 - PW so that the rows the 16-lane read groups touch are distinct mod 16 for every tap
template <int KT>
__device__ __forceinline__ void cb_pixel(int w, int m, int s, int& row, int& col) {
    row = 8 * w + 2 * m + (s >> 4);
    col = s < 16 ? s : ((s + 32 - (CB_TW + KT - 1)) & 15);
}

template <int EP, int KT, int CK>
__device__ __forceinline__ void conv_big_body(const ssr_conv_desc& d) {
    using T = CbT<KT, CK>;
    constexpr int CB_AROW = T::AROW, VPP = T::VPP;
    constexpr int CB_PW = T::PW, CB_NPIX = T::NPIX, CB_PATCH = T::PATCH, CB_BIAS = T::BIAS, CB_NPV = T::NPV, CB_NWV = T::NWV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int tiles_x = (d.Gw + CB_TW - 1) / CB_TW, tiles_y = (d.Gh + CB_TH - 1) / CB_TH;
    // A workgroup is PERSISTENT over images: it owns tile position (ty_i, tx_i) of images n0, n0 + G, n0 + 2G, ... and
    // runs their chunks as one stream — while the last chunk of one image is contracted the first chunk of the next is
    // already on its way, and the epilogue's stores drain under the next image's MFMAs.  (One tile per workgroup paid
    // the first chunk's HBM latency and the epilogue with idle matrix cores: the 128x128 layers measured at compute time
    // PLUS memory time.)  Staging descriptors and output coordinates are image-relative; the image is a uniform offset.
    const int tpi = tiles_x * tiles_y, G = gridDim.x / tpi;
    const int tp = blockIdx.x % tpi, n0 = blockIdx.x / tpi;
    const int tx_i = tp % tiles_x, ty_i = tp / tiles_x;
    const int nimg = (d.N - n0 + G - 1) / G;
    const int gy0 = ty_i * CB_TH, gx0 = tx_i * CB_TW;
    const int co0 = blockIdx.y * 64;
    const int upshift = d.up == 2 ? 1 : 0;
    const int LH = d.Hi << upshift, LW = d.Wi << upshift;
    const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(d.x.p);
    const __bf16* __restrict__ wg = reinterpret_cast<const __bf16*>(d.w);
    const int nchunks = (d.Cin + CK - 1) / CK;
    const size_t wchunk = (size_t)KT * KT * d.CoutPad * 32;   // the packed weights stay in 32-channel chunks: a 16-channel
    constexpr int WSUB = 32 / CK;                              // LDS chunk reads half of each 64-byte row
    const int T_ = nimg * nchunks;                             // length of this workgroup's chunk stream
    const size_t ximg = (size_t)d.Hi * d.Wi * d.x.cs, oimg = (size_t)d.Ho * d.Wo;   // per-image strides (elements / pixels)

    float* bias_lds = reinterpret_cast<float*>(smem + CB_BIAS);
    if (tid < 64) bias_lds[tid] = (d.bias && co0 + tid < d.Cout) ? d.bias[co0 + tid] : 0.f;

    // ---- staging descriptors (independent of the chunk) ----
    // 2x2 kernels can read x through a space-to-depth view (ssr_conv_desc.s2d: the 4x4 stride-2 forward layers): chunk c
    // is parity class q = c / (C/32) of source channels [32 (c mod C/32), +32), patch pixel (Y, X) reads source pixel
    // (2Y-1 + (q>>1), 2X-1 + (q&1)).  pgo = offset of the q = 0 pixel, pmk = one validity bit per class (bit 0 only
    // without s2d), soff(c) = the chunk's uniform offset.
    const bool s2d = KT == 2 && d.s2d != 0;
    const int cpc_shift = s2d ? 31 - __builtin_clz((unsigned)(d.Cin >> 7)) : 0;   // log2(source channels / 32)
    // Round 4: every staging load is ONE unconditional buffer load.  pgo / wgo are per-lane BYTE offsets from the image / chunk
    // base; a vector that must read zeros (outside the image, past the last channel, past the end of the table) gets CB_OOB,
    // which is beyond num_records of every resource: the hardware returns 0.  Written with a branch around each load (and a
    // zero-initialised destination) the 2x2 kernels issued ~30 VALU / scalar instructions per MFMA - exec-mask save / restore,
    // four accumulator-file moves per load for the zeros, descriptor fields re-read from the kernel arguments with
    // s_waitcnt lgkmcnt(0) in the middle of the LDS read stream - and ran at 15 % MFMA-busy with SQ_WAIT_ANY at 27 %: issue-bound.
    constexpr int CB_OOB = 0x7ffffff0;
    const int x_cs = d.x.cs, x_Wi = d.Wi, x_Cin = d.Cin;
    int pgo[CB_NPV], plo[CB_NPV], wgo[CB_NWV], wlo[CB_NWV];
    unsigned pmk[KT == 2 ? CB_NPV : 1];
    const int climit = s2d ? 0x7fffffff : x_Cin - (tid % VPP) * 8;   // chunk start c0 is inside this lane's channels iff c0 < climit (s2d: whole chunks)
#pragma unroll
    for (int q = 0; q < CB_NPV; ++q) {
        const int v = tid + q * 256;
        const int pix = v / VPP, part = v % VPP;
        const int py = pix / CB_PW, px = pix - py * CB_PW;
        const int ly = gy0 + py - d.pad_y, lx = gx0 + px - d.pad_x;
        if constexpr (KT == 2) {
            if (s2d) {
                const int sy = 2 * ly - 1, sx = 2 * lx - 1;
                const bool y0 = sy >= 0 && sy < d.Hi, y1 = sy + 1 >= 0 && sy + 1 < d.Hi;
                const bool x0 = sx >= 0 && sx < d.Wi, x1 = sx + 1 >= 0 && sx + 1 < d.Wi;
                pmk[q] = v < CB_NPIX * VPP ? (unsigned)(y0 && x0) | (unsigned)(y0 && x1) << 1 | (unsigned)(y1 && x0) << 2 | (unsigned)(y1 && x1) << 3 : 0u;
                // the q = 0 source pixel may lie one row / column outside the image: the resource's base address is one row and
                // one pixel BELOW the tensor (xbias), so that this lane offset is never negative (valid classes land inside)
                pgo[q] = ((sy * d.Wi + sx) * d.x.cs + d.x.coff + part * 8) * 2 + (d.Wi + 1) * d.x.cs * 2;
            } else {
                const bool ok = v < CB_NPIX * VPP && ly >= 0 && ly < LH && lx >= 0 && lx < LW;
                pmk[q] = ok ? 1u : 0u;                          // (q = 0 without s2d)
                pgo[q] = ok ? (int)((((size_t)(ly >> upshift) * d.Wi + (lx >> upshift)) * d.x.cs + d.x.coff + part * 8) * 2) : CB_OOB;
            }
        } else {
            const bool ok = v < CB_NPIX * VPP && ly >= 0 && ly < LH && lx >= 0 && lx < LW;
            pgo[q] = ok ? (int)((((size_t)(ly >> upshift) * d.Wi + (lx >> upshift)) * d.x.cs + d.x.coff + part * 8) * 2) : CB_OOB;
        }
        plo[q] = v < CB_NPIX * VPP ? pix * CB_AROW + part * 16 : -1;
    }
#pragma unroll
    for (int q = 0; q < CB_NWV; ++q) {
        const int v = tid + q * 256;
        const int row = v / VPP, part = v % VPP;               // row = tap*64 + co
        wlo[q] = v < T::WROWS * VPP ? CB_PATCH + row * CB_AROW + part * 16 : -1;
        wgo[q] = wlo[q] >= 0 ? (((row >> 6) * d.CoutPad + co0 + (row & 63)) * 32 + part * 8) * 2 : CB_OOB;
    }
    auto cap = [](long b) { return b > 0x7fffff00L ? 0x7fffff00L : b; };
    const long xbias = s2d ? (long)(d.Wi + 1) * d.x.cs * 2 : 0;
    const __amdgpu_buffer_rsrc_t rsx = cb_rsrc(reinterpret_cast<const char*>(d.x.p) - xbias, cap((long)d.N * ximg * 2 + xbias)), rsw = cb_rsrc(d.w, cap((long)((nchunks + WSUB - 1) / WSUB) * wchunk * 2));
    u32x4 rp[CB_NPV], rw[CB_NWV];
    // one staging load (vector j of the 19 per thread).  A wave that issues its loads back to back sits in the issue
    // of each one until the previous has drained (~170 cycles per 1-KiB instruction: the ~6.4 B/clk per-wave limit of
    // tools/l2_probe.hip) and cannot issue MFMAs meanwhile, so the loads of chunk c+1 are sprinkled over the k-steps
    // of chunk c.
    // the wave-uniform part of a chunk's loads, computed ONCE per chunk (inside load_one the compiler recomputed these scalar
    // products for each of the 13 - 19 loads): class bit to test, byte offsets of the chunk in x and in the packed weights
    struct LoadPos { int q, c0, xs, ws; };
    auto load_pos = [&](int nn, int c) {
        LoadPos p;
        p.c0 = c * CK;
        if constexpr (KT == 2) {
            p.q = s2d ? c >> cpc_shift : 0;
            // s2d: class q reads source pixel (2Y - 1 + (q >> 1), 2X - 1 + (q & 1)); pgo is relative to (2Y - 1, 2X - 1)
            p.xs = (int)(nn * ximg * 2) + (s2d ? (((p.q >> 1) * x_Wi + (p.q & 1)) * x_cs + ((c - (p.q << cpc_shift)) << 5)) * 2 : p.c0 * 2);
            p.ws = (int)(c * wchunk * 2);
        } else {
            p.q = 0;
            p.xs = (int)((nn * ximg + p.c0) * 2);
            p.ws = (int)((c / WSUB) * wchunk * 2) + (c % WSUB) * CK * 2;
        }
        return p;
    };
    auto load_one = [&](const LoadPos& lp, auto jc) {       // vector j of the chunk at lp
        constexpr int j = decltype(jc)::value;
#ifdef CB_X_NOLOAD
        if (lp.ws != 0 || lp.xs != (int)(n0 * ximg * 2)) return;   // probe: only the stream's first chunk is loaded
#endif
        if constexpr (j < CB_NPV) {
            bool ok = lp.c0 < climit;                          // (s2d: climit covers every chunk)
            if constexpr (KT == 2) ok = ok && ((pmk[j] >> lp.q) & 1u);
            rp[j] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? pgo[j] : CB_OOB, lp.xs, 0);
        } else {
            rw[j - CB_NPV] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wgo[j - CB_NPV], lp.ws, 0);
        }
    };
    auto load_chunk = [&](int nn, int c) { const LoadPos lp = load_pos(nn, c); static_for<0, CB_NPV + CB_NWV>([&](auto jc) { load_one(lp, jc); }); };
    // (only the LAST vector of each table can be short of threads: the others are stored without a test)
    auto store_vec = [&](int base, auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < CB_NPV) {
            if constexpr ((j + 1) * 256 <= CB_NPIX * VPP) *reinterpret_cast<u32x4*>(smem + base + plo[j]) = rp[j];
            else if (plo[j] >= 0) *reinterpret_cast<u32x4*>(smem + base + plo[j]) = rp[j];
        } else {
            constexpr int jw = j - CB_NPV;
            if constexpr ((jw + 1) * 256 <= T::WROWS * VPP) *reinterpret_cast<u32x4*>(smem + base + wlo[jw]) = rw[jw];
            else if (wlo[jw] >= 0) *reinterpret_cast<u32x4*>(smem + base + wlo[jw]) = rw[jw];
        }
    };
    auto store_chunk = [&]() { static_for<0, CB_NPV + CB_NWV>([&](auto jc) { store_vec(0, jc); }); };

    // ---- this lane's pixels (one per pixel tile) ----
    int a_off[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        int row, col;
        cb_pixel<KT>(wave, m, i, row, col);
        a_off[m] = (row * CB_PW + col) * CB_AROW + g * 16;
    }
    const int b_off = CB_PATCH + i * CB_AROW + g * 16;
    const int co_l = 4 * g;                                   // + 8*q4 + e: this lane's 16 channels of a channel tile

    BPROBE(0);
    load_chunk(n0, 0);
    __syncthreads();                                          // bias table
    BPROBE(1);
    f32x16 acc[4][2];
    auto acc_init = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + t * 32 + 8 * q4 + co_l);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = EP >= 0 ? bq[e] : 0.f;   // lean variants: accumulators start at the bias
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m][t][4 * q4 + e] = v0;
                }
            }
    };

    // ---- lean epilogues: this lane's pixel of every pixel tile and its epilogue operands (residual / old value / mask).
    //      They are requested when the LAST chunk starts, so their latency hides under its MFMAs (fetched inside the
    //      epilogue they cost a global round trip per pixel tile: ~5 of the 11k epilogue cycles, tools/big_probe.hip). ----
    const __bf16* __restrict__ r1p = reinterpret_cast<const __bf16*>(d.r1.p);
    const __bf16* __restrict__ mp = reinterpret_cast<const __bf16*>(d.m.p);
    __bf16* __restrict__ yp = reinterpret_cast<__bf16*>(d.y.p);
    int ppx[4];                                               // image-relative output pixel of this lane per pixel tile
    bool pval[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        int row, col;
        cb_pixel<KT>(wave, m, i, row, col);
        const int gy = gy0 + row, gx = gx0 + col;
        pval[m] = gy < d.Gh && gx < d.Gw;
        const int cy = pval[m] ? gy : d.Gh - 1, cx = pval[m] ? gx : d.Gw - 1;
        ppx[m] = (cy * d.oys + d.oyo) * d.Wo + cx * d.oxs + d.oxo;   // + nn * oimg
    }
    constexpr bool HAS_R1 = EP >= 0 && (EP & CB_R1), HAS_ACC = EP >= 0 && (EP & CB_ACC), HAS_MASK = EP >= 0 && (EP & CB_MASK);
    typedef u32x2c OpsR1[HAS_R1 ? 4 : 1][8];
    typedef u32x2c OpsAcc[HAS_ACC ? 4 : 1][8];
    typedef u32x2c OpsMask[HAS_MASK ? 4 : 1][8];
    OpsR1 q1;
    OpsAcc qa;
    OpsMask qm;                                               // prefetched under the stream's last chunk
    const long obytes = (long)d.N * oimg * 2;                 // x view.cs = bytes of an output-side tensor (exact: out-of-range reads return 0)
    const __amdgpu_buffer_rsrc_t rs_r1 = cb_rsrc(d.r1.p, obytes * d.r1.cs), rs_m = cb_rsrc(d.m.p, obytes * d.m.cs),
                                 rs_y = cb_rsrc(d.y.p, obytes * d.y.cs), rs_y0 = cb_rsrc(d.y0.p, obytes * d.y0.cs);
    auto load_epi_ops_to = [&](int nn, OpsR1& q1, OpsAcc& qa, OpsMask& qm, int m0 = 0, int m1 = 4) {
        const int s1 = (int)(nn * oimg * d.r1.cs * 2), sa = (int)(nn * oimg * d.y.cs * 2), sm = (int)(nn * oimg * d.m.cs * 2);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < m0 || m >= m1) continue;                      // (compile-time after unrolling)
            const int c = co0 + co_l;                             // + t * 32 + 8 * q4: constant offsets of the load instructions
            const int v1 = (ppx[m] * d.r1.cs + d.r1.coff + c) * 2, va = (ppx[m] * d.y.cs + d.y.coff + c) * 2,
                      vm = (ppx[m] * d.m.cs + d.m.coff + c) * 2;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int o = (t * 32 + 8 * q4) * 2;
                    if constexpr (HAS_R1) q1[m][t * 4 + q4] = __builtin_amdgcn_raw_buffer_load_b64(rs_r1, v1 + o, s1, 0);
                    if constexpr (HAS_ACC) qa[m][t * 4 + q4] = __builtin_amdgcn_raw_buffer_load_b64(rs_y, va + o, sa, 0);
                    if constexpr (HAS_MASK) qm[m][t * 4 + q4] = __builtin_amdgcn_raw_buffer_load_b64(rs_m, vm + o, sm, 0);
                }
        }
    };
    auto load_epi_ops = [&](int nn) { load_epi_ops_to(nn, q1, qa, qm); };

    // ---- epilogue: per pixel tile m this lane = one pixel x (2 x 16) channels ----
    __bf16* __restrict__ y0p = reinterpret_cast<__bf16*>(d.y0.p);
    __bf16* __restrict__ y1p = reinterpret_cast<__bf16*>(d.y1.p);
    const __bf16* __restrict__ r2p = reinterpret_cast<const __bf16*>(d.r2.p);
    // the wave's [32 px][64 co] slab of pixel tile m -> 256 16-byte vectors, 4 per lane: whole 128-byte lines per pixel
    auto flush = [&](int nn, int m, const __bf16* sl, const __amdgpu_buffer_rsrc_t& rs, const ssr_view& vw) {
        const int so = (int)(nn * oimg * vw.cs * 2);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int v = h * 64 + lane;
            const int s_ = v >> 3, part = v & 7;
            int prow, pcol;
            cb_pixel<KT>(wave, m, s_, prow, pcol);
            const int oy = gy0 + prow, ox = gx0 + pcol, c = co0 + part * 8;
            const u32x4 val = *reinterpret_cast<const u32x4*>(sl + s_ * CB_SROW + part * 8);
#ifdef CB_X_NOSTORE
            if (val.x == 0x12345678u && oy < d.Gh && ox < d.Gw && c < d.Cout)
#else
            if (oy < d.Gh && ox < d.Gw && c < d.Cout)
#endif
                __builtin_amdgcn_raw_buffer_store_b128(val, rs, (((oy * d.oys + d.oyo) * d.Wo + ox * d.oxs + d.oxo) * vw.cs + vw.coff + c) * 2, so, 0);
        }
    };
    // Epilogue of image nn.  The (now idle) chunk area at byte offset sbase becomes the waves' output transpose slabs.  PRE: the
    // operands were requested under the last chunk's MFMAs (final image of the stream, where the staging registers are
    // free); otherwise they are requested here, with the next image's first chunk still waiting in the staging registers.
    auto epilogue_with = [&](int nn, int sbase, OpsR1& q1, OpsAcc& qa, OpsMask& qm, auto lazyc) {
        constexpr bool LAZY = decltype(lazyc)::value;         // request the operands of each round of pixel tiles at its start
        if constexpr (EP >= 0) {
            // branch-free variants: the slabs of MB pixel tiles are written back to back, then read and stored back to back
            // (wave-private slabs, LDS executes a wave's operations in order: no barrier, one LDS round trip per round)
            constexpr int NSL = (EP & CB_Y0) ? 2 : 1;
            constexpr int MB_ = T::SLABB / (4 * CB_SLAB * 2 * NSL);
            constexpr int MB = MB_ >= 4 ? 4 : MB_ >= 2 ? 2 : 1;
            __bf16* wslab = reinterpret_cast<__bf16*>(smem + sbase) + wave * (MB * NSL * CB_SLAB);
#pragma unroll
            for (int r0 = 0; r0 < 4; r0 += MB) {
                if constexpr (LAZY) load_epi_ops_to(nn, q1, qa, qm, r0, r0 + MB);
#pragma unroll
                for (int mm = 0; mm < MB; ++mm) {
                    const int m = r0 + mm;
                    __bf16* slab = wslab + mm * NSL * CB_SLAB;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = acc[m][t][4 * q4 + e];
                                if constexpr ((EP & CB_LRELU) != 0) v[e] = lrelu_max(v[e]);   // == lrelu()
                            }
                            if constexpr ((EP & CB_Y0) != 0) {   // y0 = act(conv + bias), before the residual
                                bf16x4c o0;
#pragma unroll
                                for (int e = 0; e < 4; ++e) o0[e] = (__bf16)v[e];
                                *reinterpret_cast<bf16x4c*>(slab + CB_SLAB + i * CB_SROW + t * 32 + 8 * q4 + co_l) = o0;
                            }
                            if constexpr (HAS_R1) {
                                const u32x2c r = q1[m][t * 4 + q4];
                                v[0] += d.beta1 * cb_lo(r[0]); v[1] += d.beta1 * cb_hi(r[0]);
                                v[2] += d.beta1 * cb_lo(r[1]); v[3] += d.beta1 * cb_hi(r[1]);
                            }
                            if constexpr (HAS_ACC) {
                                const u32x2c r = qa[m][t * 4 + q4];
                                v[0] += cb_lo(r[0]); v[1] += cb_hi(r[0]); v[2] += cb_lo(r[1]); v[3] += cb_hi(r[1]);
                            }
                            if constexpr (HAS_MASK) {
                                const u32x2c r = qm[m][t * 4 + q4];
                                v[0] = lrelu_mask_lo(v[0], r[0]); v[1] = lrelu_mask_hi(v[1], r[0]);
                                v[2] = lrelu_mask_lo(v[2], r[1]); v[3] = lrelu_mask_hi(v[3], r[1]);
                            }
                            bf16x4c o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                            *reinterpret_cast<bf16x4c*>(slab + i * CB_SROW + t * 32 + 8 * q4 + co_l) = o;
                        }
                }
#pragma unroll
                for (int mm = 0; mm < MB; ++mm) {
                    const __bf16* slab = wslab + mm * NSL * CB_SLAB;
                    flush(nn, r0 + mm, slab, rs_y, d.y);
                    if constexpr ((EP & CB_Y0) != 0) flush(nn, r0 + mm, slab + CB_SLAB, rs_y0, d.y0);
                }
            }
        } else {
            // generic epilogue: the full ssr_conv_desc contract with run-time flags, one pixel tile at a time
            __bf16* slab = reinterpret_cast<__bf16*>(smem + sbase) + wave * CB_SLAB;   // [32 lane-slots][64 co]
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const size_t pp = nn * oimg + ppx[m];
                const bool pvalid = pval[m];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    bf16x4c q1g[4], q2g[4], qag[4], qmg[4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int c = co0 + t * 32 + 8 * q4 + co_l;
                        const int cc = c < d.Cout ? c : 0;
                        if (r1p) q1g[q4] = *reinterpret_cast<const bf16x4c*>(r1p + pp * d.r1.cs + d.r1.coff + cc);
                        if (r2p) q2g[q4] = *reinterpret_cast<const bf16x4c*>(r2p + pp * d.r2.cs + d.r2.coff + cc);
                        if (d.accumulate) qag[q4] = *reinterpret_cast<const bf16x4c*>(yp + pp * d.y.cs + d.y.coff + cc);
                        if (mp) qmg[q4] = *reinterpret_cast<const bf16x4c*>(mp + pp * d.m.cs + d.m.coff + cc);
                    }
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int c = co0 + t * 32 + 8 * q4 + co_l;
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + t * 32 + 8 * q4 + co_l);
                        bf16x4c o0, o1, o2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[m][t][4 * q4 + e] + bq[e];
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
                            if (y0p) *reinterpret_cast<bf16x4c*>(y0p + pp * d.y0.cs + d.y0.coff + c) = o0;
                            if (y1p) *reinterpret_cast<bf16x4c*>(y1p + pp * d.y1.cs + d.y1.coff + c) = o1;
                        }
                        *reinterpret_cast<bf16x4c*>(slab + i * CB_SROW + t * 32 + 8 * q4 + co_l) = o2;
                    }
                }
                flush(nn, m, slab, rs_y, d.y);
            }
        }
    };

    auto epilogue = [&](int nn, int sbase, auto prec) {
        constexpr bool PRE = decltype(prec)::value;
        if constexpr (T::DB) CB_BAR();                        // in-place slabs: everyone is finished reading the buffer that becomes them
        if constexpr (EP >= 0 && !PRE) {                      // operands of an in-stream epilogue live only here
            OpsR1 l1;
            OpsAcc la;
            OpsMask lm;
            // two operand kinds: 128 registers for all four pixel tiles do not fit next to the accumulator copies
            constexpr int NOPS = (HAS_R1 ? 1 : 0) + (HAS_ACC ? 1 : 0) + (HAS_MASK ? 1 : 0);
            constexpr bool LAZY = NOPS > 1 || (T::DB && NOPS > 0);   // (double-buffered: staging + prefetched fragments stay live here)
            if constexpr (!LAZY) load_epi_ops_to(nn, l1, la, lm);
            epilogue_with(nn, sbase, l1, la, lm, std::bool_constant<LAZY>{});
        } else {
            epilogue_with(nn, sbase, q1, qa, qm, std::false_type{});
        }
    };

    if constexpr (T::DB) {
        // ---- 2x2 taps (and 3x3 in 16-channel chunks): only 8 k-steps (64 MFMAs per wave) per chunk, so a store -> barrier -> MFMA sequence per chunk left
        //      the matrix cores idle for half of the loop (tools/big_probe.hip: 1500 of 5600 cycles in barriers and the
        //      chunk store, 4100 for 2048 cycles of MFMAs).  Two chunk buffers and ONE continuous stream of k-steps instead:
        //      during chunk c   steps 0..3  store chunk c+1 (registers -> the other buffer), after barrier X (everyone is
        //                                   finished reading that buffer: chunk c-1)
        //                       step  4     barrier Y (chunk c+1 visible)
        //                       steps 4..7  request chunk c+2 (global -> the same registers)
        //                       steps 6..7  already read the first fragments of chunk c+1
        //      The barriers are bare s_barrier + lgkmcnt(0): global loads stay in flight across them. ----
        constexpr int NSTEP = T::NSTEP, CB_PF = 2, NV = CB_NPV + CB_NWV;
        static_assert(NSTEP >= 8 && NV <= 16 && (NSTEP - 4) * 4 >= NV, "store / load slots of the step schedule");
        auto store_one = [&](int base, auto jc) { store_vec(base, jc); };
        u32x4 wq[NSTEP][2], pq[NSTEP][4];
        // memory operation k (0..5) of k-step s_: the two weight fragments, then pixel fragments 0..3
        auto issue1 = [&](int base, auto sc, auto kc) {
            constexpr int s_ = decltype(sc)::value, tap = s_ / T::KSUB, kk = s_ % T::KSUB, k = decltype(kc)::value;
            if constexpr (k >= 2)
                pq[s_][k - 2] = *reinterpret_cast<const u32x4*>(smem + base + a_off[k - 2] + ((tap / KT) * CB_PW + tap % KT) * CB_AROW + kk * 32);
            else
                wq[s_][k] = *reinterpret_cast<const u32x4*>(smem + base + b_off + (tap * 64 + k * 32) * CB_AROW + kk * 32);
        };
        static_for<0, NV>([&](auto jc) { store_one(0, jc); });
        if (T_ > 1) load_chunk(n0 + (1 / nchunks) * G, 1 % nchunks);
        CB_BAR();
        static_for<0, CB_PF>([&](auto sc) { static_for<0, 6>([&](auto kc) { issue1(0, sc, kc); }); });
        // the last chunk is a separate instantiation: the staging registers are dead there and hold the epilogue operands
        auto chunk = [&](int c, int nn, int nn2, int c2, auto h1c) {   // c = stream position; (nn2, c2) = position c + 2
            constexpr bool has1 = decltype(h1c)::value;
            BPROBE_C(2);
            const int cur = (c & 1) * T::BUF, nxt = T::BUF - cur;
            const bool has2 = c + 2 < T_;
            // past the end of the stream the loads are issued all the same (of the current position: valid addresses, never stored):
            // no branch around a load
            const LoadPos lp2 = has2 ? load_pos(nn2, c2) : load_pos(nn, c % nchunks);
            if constexpr (!has1) load_epi_ops(nn);
            static_for<0, NSTEP>([&](auto sc) {
                constexpr int s_ = decltype(sc)::value;
                if constexpr (s_ == 0 || s_ == 4) {
                    if constexpr (has1) CB_BAR();                 // X / Y
                }
                // memory instructions go between the MFMAs (see the 3x3 loop): a fragment read after each of the first
                // six, a staging store (steps 0..3) or load (steps 4..7) after MFMAs 1, 3, 6 and 7
                static_for<0, 8>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    mma16<__bf16>(acc[k >> 1][k & 1], wq[s_][k & 1], pq[s_][k >> 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (k < 6) {
                        if constexpr (s_ + CB_PF < NSTEP) issue1(cur, std::integral_constant<int, s_ + CB_PF>{}, kc);
                        else if constexpr (has1) issue1(nxt, std::integral_constant<int, s_ + CB_PF - NSTEP>{}, kc);
                    }
                    constexpr int e = k == 1 ? 0 : k == 3 ? 1 : k == 6 ? 2 : k == 7 ? 3 : -1;
                    if constexpr (has1 && e >= 0) {
                        constexpr int j = (s_ < 4 ? s_ : s_ - 4) * 4 + e;   // NV vectors, four per step
                        if constexpr (j < NV) {
                            if constexpr (s_ < 4) store_one(nxt, std::integral_constant<int, j>{});
                            else load_one(lp2, std::integral_constant<int, j>{});
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
            BPROBE_C(6);
        };
        // stream position gc = (image k, chunk c); position gc + 2 is (image nn + ((c + 2) / nchunks) * G, chunk (c + 2) % nchunks).
        // At an image boundary nothing in the schedule changes: the next image's first chunk is already in the other
        // buffer, the one after it waits in the staging registers, and the epilogue's slabs use the buffer just consumed
        // (barrier X of the next chunk orders them before its stores).  The stream's last chunk is peeled.
        int gc = 0;
        for (int k = 0; k < nimg; ++k) {
            const int nn = n0 + k * G;
            const bool last_img = k + 1 == nimg;
            acc_init();
            for (int c = 0; c < (last_img ? nchunks - 1 : nchunks); ++c, ++gc)
                chunk(gc, nn, nn + ((c + 2) / nchunks) * G, (c + 2) % nchunks, std::true_type{});
            if (!last_img) epilogue(nn, ((gc - 1) & 1) * T::BUF, std::false_type{});
        }
        const int nl = n0 + (nimg - 1) * G;
        chunk(gc, nl, nl, 0, std::false_type{});
        BPROBE(7);
        epilogue(nl, (gc & 1) * T::BUF, std::true_type{});
    } else {
        auto chunk = [&](int c, bool stored, int nn, int nn1, int c1, auto hnc) {   // c = stream position; (nn1, c1) = position c + 1
            constexpr bool has_next = decltype(hnc)::value;            // the stream's last chunk is peeled: see above
            BPROBE_C(2);
            if (!stored) {                                        // (an image's first chunk was stored before the previous epilogue)
                if (c > 0) CB_BAR();                              // everyone is finished reading the previous chunk
                BPROBE_C(3);
                store_chunk();
            }
            BPROBE_C(4);
            CB_BAR();
            BPROBE_C(5);
            if constexpr (!has_next) load_epi_ops(nn);
            const LoadPos lp1 = load_pos(nn1, c1);
            // k-steps: 9 taps x 2 sixteen-channel halves; per step 2 weight fragments + 4 pixel fragments -> 8 MFMAs.
            // Reads run CB_PF steps ahead, pinned by sched_barrier fences (one wave per SIMD: nothing else hides LDS latency).
            constexpr int NSTEP = T::NSTEP, CB_PF = 2;
            u32x4 wq[NSTEP][2], pq[NSTEP][4];
            // memory operation k (0..5) of k-step s_: the two weight fragments, then pixel fragments 0..3 (the order the
            // MFMAs of that step need them in)
            auto issue1 = [&](auto sc, auto kc) {
                constexpr int s_ = decltype(sc)::value, tap = s_ / T::KSUB, kk = s_ % T::KSUB, k = decltype(kc)::value;
                if constexpr (k >= 2)
                    pq[s_][k - 2] = *reinterpret_cast<const u32x4*>(smem + a_off[k - 2] + ((tap / KT) * CB_PW + tap % KT) * CB_AROW + kk * 32);
                else
                    wq[s_][k] = *reinterpret_cast<const u32x4*>(smem + b_off + (tap * 64 + k * 32) * CB_AROW + kk * 32);
            };
            static_for<0, CB_PF>([&](auto sc) { static_for<0, 6>([&](auto kc) { issue1(sc, kc); }); });
            // Inside a step the memory instructions go BETWEEN the MFMAs, one per MFMA: bunched at the step boundary
            // (6 LDS reads + up to 2 global loads, ~100 cycles of issue) they left the matrix pipe idle at every one of
            // the 18 boundaries (tools/big_probe.hip: 6600 cycles for 4608 of MFMAs).
            static_for<0, NSTEP>([&](auto sc) {
                constexpr int s_ = decltype(sc)::value;
                static_for<0, 8>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    mma16<__bf16>(acc[k >> 1][k & 1], wq[s_][k & 1], pq[s_][k >> 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (k < 6) {
                        if constexpr (s_ + CB_PF < NSTEP) issue1(std::integral_constant<int, s_ + CB_PF>{}, kc);
                    } else if constexpr (has_next) {
                        // staging loads of the next chunk: two in each of the first CB_L2 steps, then one per step; the
                        // last one four steps before the end so that the chunk store does not wait for it
                        constexpr int NV = CB_NPV + CB_NWV, SPAN = NSTEP - CB_TAIL, CB_L2 = NV > SPAN ? NV - SPAN : 0;
                        constexpr int j = s_ < CB_L2 ? 2 * s_ + (k - 6) : (k == 6 ? CB_L2 + s_ : NV);
                        if constexpr (j < NV) load_one(lp1, std::integral_constant<int, j>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
            BPROBE_C(6);
        };
        // every chunk but the stream's last one runs in the loops; the last chunk and its epilogue are peeled (staging
        // registers dead: they hold the prefetched epilogue operands)
        int gc = 0;
        for (int k = 0; k < nimg; ++k) {
            const int nn = n0 + k * G;
            const bool last_img = k + 1 == nimg;
            acc_init();
            for (int c = 0; c < (last_img ? nchunks - 1 : nchunks); ++c, ++gc) {
                const bool last_c = c + 1 == nchunks;
                chunk(gc, k > 0 && c == 0, nn, last_c ? nn + G : nn, last_c ? 0 : c + 1, std::true_type{});
            }
            if (!last_img) {
                // image boundary: the next image's first chunk leaves the staging registers BEFORE the epilogue (which then
                // has the register file to itself); its stores, and the epilogue's global stores, drain under the next MFMAs
                CB_BAR();                                         // everyone is finished reading this image's last chunk
                store_chunk();
                epilogue(nn, T::SLABS, std::false_type{});
            }
        }
        const int nl = n0 + (nimg - 1) * G;
        chunk(gc, nimg > 1 && nchunks == 1, nl, nl, 0, std::false_type{});
        BPROBE(7);
        epilogue(nl, T::SLABS, std::true_type{});
    }
    BPROBE(8);
}

template <int EP, int KT, int CK>
__global__ __launch_bounds__(256, 1) void conv_big_kernel(const ssr_conv_desc d) {
    conv_big_body<EP, KT, CK>(d);
}
// up to four descriptors of identical geometry (the output-parity classes of a stride-2 transposed conv), blockIdx.z selects
struct ssr_conv_desc4b { ssr_conv_desc d[4]; };
template <int EP, int KT, int CK>
__global__ __launch_bounds__(256, 1) void conv_big_kernel4(const ssr_conv_desc4b p) {
    conv_big_body<EP, KT, CK>(p.d[blockIdx.z]);
}

template <int EP, int KT, int CK>
int launch_big(const ssr_conv_desc* ds, int n, hipStream_t st) {
    constexpr int lds = CbT<KT, CK>::LDS;
    const ssr_conv_desc& d = ds[0];
    // grid.x = (tile positions per image) x G image groups; a workgroup walks images n0, n0 + G, ... (conv_big_body).
    // G minimises rounds x images-per-workgroup on 256 CUs (one workgroup per CU), then rounds, then prefers more groups.
    const int tpi = ((d.Gh + CB_TH - 1) / CB_TH) * ((d.Gw + CB_TW - 1) / CB_TW);
    // SSR_CONV_BIG_PERSIST: 0 = one image per workgroup, 2 / 3 = persistent 2x2 / 3x3 kernels only, default both
    static const int persist = [] { const char* e = getenv("SSR_CONV_BIG_PERSIST"); return e ? atoi(e) : 1; }();
    int G = d.N;
    if (persist == 1 || persist == KT) {
        const long per_img = (long)tpi * (d.CoutPad / 64) * n;
        long best = -1;
        for (int g = 1; g <= d.N; ++g) {
            const long rounds = (per_img * g + 255) / 256, cost = rounds * ((d.N + g - 1) / g);
            const long key = (cost << 24) | (rounds << 12) | (4095 - (g > 4095 ? 4095 : g));
            if (best < 0 || key < best) { best = key; G = g; }
        }
    }
    if (const char* e = getenv("SSR_CONV_BIG_G")) {           // test hook: force the number of image groups (1..N)
        const int g = atoi(e);
        if (g >= 1) G = g < d.N ? g : d.N;
    }
    const int tiles = tpi * G;
    if (n == 1) {
        auto kern = conv_big_kernel<EP, KT, CK>;
        static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
        const int attr_dev = ssr_device_ordinal();
        if (!attr_done[attr_dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return (int)e;
            attr_done[attr_dev] = true;
        }
        hipLaunchKernelGGL(kern, dim3(tiles, d.CoutPad / 64, 1), dim3(256), lds, st, d);
    } else {
        auto kern = conv_big_kernel4<EP, KT, CK>;
        static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
        const int attr_dev = ssr_device_ordinal();
        if (!attr_done[attr_dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return (int)e;
            attr_done[attr_dev] = true;
        }
        ssr_conv_desc4b p;
        for (int k = 0; k < 4; ++k) p.d[k] = ds[k < n ? k : 0];
        hipLaunchKernelGGL(kern, dim3(tiles, d.CoutPad / 64, n), dim3(256), lds, st, p);
    }
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

// feature set of the branch-free epilogue instantiation that fits `d`, or CB_GENERIC
int cb_epilogue_of(const ssr_conv_desc& d) {
    auto al16 = [](const ssr_view& v) { return (v.cs % 8) == 0 && (v.coff % 8) == 0 && ((uintptr_t)v.p % 16) == 0; };
    if ((!d.y0.p || al16(d.y0)) && !d.y1.p && !d.r2.p && d.alpha == 1.f)
        return (d.act == SSR_ACT_LRELU ? CB_LRELU : 0) | (d.m.p ? CB_MASK : 0) | (d.r1.p ? CB_R1 : 0) |
               (d.accumulate ? CB_ACC : 0) | (d.y0.p ? CB_Y0 : 0);
    return CB_GENERIC;
}

}  // namespace

// everything except the grid-size heuristic (ssr_conv2d_impl(impl = 4) forces this kernel on small shapes)
bool ssr_conv_big_shape_ok(const ssr_conv_desc& d) {
    if (d.dtype != SSR_BF16) return false;
    const bool k3 = d.KH == 3 && d.KW == 3 && d.pad_y == 1 && d.pad_x == 1;
    const bool k2 = d.KH == 2 && d.KW == 2 && (d.pad_y == 0 || d.pad_y == 1) && (d.pad_x == 0 || d.pad_x == 1);
    if (!(k3 || k2) || d.stride != 1 || d.x2.p) return false;
    if (d.Cin < 32 || (d.CoutPad % 64) != 0 || (d.Cout % 8) != 0) return false;
    if (d.s2d) {   // internal form of a 4x4 stride-2 layer (conv.hip): 2x2, pad 0, Cin = 4 x source channels
        const int cpc = d.Cin / 128;
        if (!k2 || d.pad_y != 0 || d.pad_x != 0 || d.up != 1 || (d.Cin % 128) != 0 || (cpc & (cpc - 1)) != 0) return false;
        if ((d.Hi % 2) != 0 || (d.Wi % 2) != 0 || d.Gh != d.Hi / 2 || d.Gw != d.Wi / 2) return false;
    } else if (d.Gh != (d.Hi << (d.up == 2)) || d.Gw != (d.Wi << (d.up == 2))) return false;
    if (d.r1.p && d.r1_nc < d.Cout) return false;
    if (d.r2.p && d.r2_nc < d.Cout) return false;
    if (d.m.p && !(d.m_c0 == 0 && d.m_c1 >= d.Cout)) return false;
    // buffer addressing: every tensor is reached through a 32-bit byte offset from its resource base (the staging loads, the
    // epilogue operands, the output stores).  Larger tensors go to the generic kernel (64-bit pointers).
    const long lim = 0x7fffff00L;
    const long xin = ((long)d.N * d.Hi * d.Wi + d.Wi + 1) * d.x.cs * 2, npo = (long)d.N * d.Ho * d.Wo * 2;
    if (xin > lim || npo * d.y.cs > lim || (d.y0.p && npo * d.y0.cs > lim) || (d.r1.p && npo * d.r1.cs > lim) || (d.m.p && npo * d.m.cs > lim)) return false;
    auto al = [](const ssr_view& v) { return !v.p || ((v.cs % 4) == 0 && (v.coff % 4) == 0 && ((uintptr_t)v.p % 8) == 0); };
    auto al16 = [](const ssr_view& v) { return (v.cs % 8) == 0 && (v.coff % 8) == 0 && ((uintptr_t)v.p % 16) == 0; };
    return al16(d.y) && al(d.y0) && al(d.y1) && al(d.r1) && al(d.r2) && al(d.m);
}

bool ssr_conv_big_qualifies(const ssr_conv_desc& d) {
    static const bool off = [] { const char* e = getenv("SSR_CONV_BIGTILE"); return e && e[0] == '0'; }();
    if (off || !ssr_conv_big_shape_ok(d) || d.KH != 3) return false;
    if (d.Cin <= 64) return false;                            // weight-stationary kernel (conv_ws.hip) territory
    const long wgs = (long)d.N * ((d.Gh + CB_TH - 1) / CB_TH) * ((d.Gw + CB_TW - 1) / CB_TW) * (d.CoutPad / 64);
    return wgs >= 128;                                        // at least half the CUs get a 512-pixel tile
}

template <int KT, int CK>
static int cb_dispatch(const ssr_conv_desc* ds, int n, hipStream_t st) {
    switch (cb_epilogue_of(ds[0])) {
        case 0: return launch_big<0, KT, CK>(ds, n, st);
        case CB_LRELU: return launch_big<CB_LRELU, KT, CK>(ds, n, st);
        case CB_LRELU | CB_R1: return launch_big<CB_LRELU | CB_R1, KT, CK>(ds, n, st);
        case CB_LRELU | CB_R1 | CB_Y0: return launch_big<CB_LRELU | CB_R1 | CB_Y0, KT, CK>(ds, n, st);
        case CB_MASK: return launch_big<CB_MASK, KT, CK>(ds, n, st);
        case CB_MASK | CB_ACC: return launch_big<CB_MASK | CB_ACC, KT, CK>(ds, n, st);
        case CB_MASK | CB_R1: return launch_big<CB_MASK | CB_R1, KT, CK>(ds, n, st);
        default: return launch_big<CB_GENERIC, KT, CK>(ds, n, st);
    }
}

bool ssr_conv_big_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force) {
    if (force ? !ssr_conv_big_shape_ok(d) : !ssr_conv_big_qualifies(d)) return false;
    // (CbT<3, 16> — the 3x3 layers on the double-buffered stream in 16-channel chunks — builds and passes the parity tests but
    // is slower: 49.6 vs 42.5 us on tools/big_probe.hip, 13.86 vs 13.76 ms per step.  A 16-channel chunk reads 32-byte
    // pieces of every pixel / weight row, so each staging load touches twice the cache lines, and its 48-byte LDS rows are
    // one third padding.  Not instantiated.)
    *rc = d.KH == 3 ? cb_dispatch<3, 32>(&d, 1, st) : cb_dispatch<2, 32>(&d, 1, st);
    return true;
}

// n <= 4 parity-class descriptors (2x2 stride 1, identical geometry and epilogue features) in one launch
bool ssr_conv_big_batch_try(const ssr_conv_desc* ds, int n, hipStream_t st, int* rc) {
    const char* e = getenv("SSR_CONV_BIGTILE2");             // 0: never, 2: always (tests), default: by grid size
    const bool off = e && e[0] == '0', always = e && e[0] == '2';
    if (off || n < 1 || n > 4) return false;
    const int ep = cb_epilogue_of(ds[0]);
    for (int k = 0; k < n; ++k)
        if (!ssr_conv_big_shape_ok(ds[k]) || ds[k].KH != 2 || cb_epilogue_of(ds[k]) != ep) return false;
    const ssr_conv_desc& d = ds[0];
    const long wgs = (long)d.N * ((d.Gh + CB_TH - 1) / CB_TH) * ((d.Gw + CB_TW - 1) / CB_TW) * (d.CoutPad / 64) * n;
    if (!always && (wgs < 192 || d.Gh < 24)) return false;   // small grids / half-empty 32-row tiles: pipelined kernel
    *rc = cb_dispatch<2, 32>(ds, n, st);
    return true;
}
