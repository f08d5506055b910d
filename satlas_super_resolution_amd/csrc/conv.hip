// Direct (im2col-free) NHWC convolution on the gfx950 matrix cores — software-pipelined version.
//
// A workgroup of MW*KS wavefronts computes an output tile of (2*MW) x 16 grid positions times
// BN = 32*NT output channels.  Input channels are consumed in chunks of CK (32 bf16 / 16 fp32; half
// of that for the 4x4 stride-2 layers): per chunk the workgroup stages the input halo patch and the
// [tap][co][CK] weight slab in LDS ONCE and every tap reads its shifted window straight out of the patch.
//   * pipelining: LDS is double buffered and the global loads of chunk c+1 are issued into registers
//     before chunk c is contracted, so HBM/L2 latency overlaps the MFMAs; one barrier per chunk.
//   * waves: MW waves tile the pixels (32 grid positions = 2 rows x 16 each, NT 32x32 accumulators);
//     KS = 2 splits the reduction inside a chunk across two waves per pixel group (k-substep for 3x3/2x2,
//     tap-column parity for 4x4), which doubles the waves per CU for the small 32x32-pixel generator
//     layers; partial sums are combined through LDS once at the end.
//   * matrix core: one 16-byte LDS read per operand feeds 4 v_mfma_f32_32x32x2_f32 (fp32, exact) or
//     1 v_mfma_f32_32x32x16_bf16; rows are padded by 16 B (80 B / 48 B): conflict-free ds_read_b128.
//   * packed weights are chunk-major [chunk][tap][CoutPad][CK]: a stage's slab is contiguous in HBM.
//
// Replaces: every nn.Conv2d forward on the ESRGAN path and (with flipped weights) its dgrad —
// /root/reference/ssr/archs/rrdbnet_arch.py:26-30,99-112,123-136 and discriminator_arch.py:28-40,44-69 —
// with torch.cat (rrdbnet_arch.py:39-42), LeakyReLU, the 0.2-residuals (:44,:68), the trunk add (:125),
// nearest x2 upsampling (:127-128), the U-Net skip adds (discriminator_arch.py:53-64) and the
// corresponding backward masks / gradient fan-in sums folded into the load and the epilogue.
#include "conv_epilogue.h"
#include <cstdlib>
#include <cstdio>

namespace {

template <typename T, int KH, int KW, int S, int NT, int MW, int KS>
__device__ __forceinline__ void conv_body(const ssr_conv_desc& d) {
    constexpr int VEC = DT<T>::VEC;
    constexpr int KSTEPS = (KH == 4) ? 1 : 2;          // 16-byte k-reads per tap per chunk
    constexpr int CK = KSTEPS * 2 * VEC, VPR = 2 * KSTEPS, CKP = CK + VEC, BN = 32 * NT;
    constexpr int TH = 2 * MW, TW = 16;
    constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
    constexpr int NTHR = 64 * MW * KS;
    constexpr int PVEC = PH * PW * VPR, WVEC = KH * KW * BN * VPR;
    constexpr int NPV = (PVEC + NTHR - 1) / NTHR, NWV = (WVEC + NTHR - 1) / NTHR;
    constexpr int STAGE = (PH * PW + KH * KW * BN) * CKP;   // elements per LDS stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lds = reinterpret_cast<T*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % MW, kh = wave / MW;               // pixel group, k-split half
    const int tiles_x = (d.Gw + TW - 1) / TW, tiles_y = (d.Gh + TH - 1) / TH;
    int b = blockIdx.x;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty_i * TH, gx0 = tx_i * TW;
    const int co0 = blockIdx.y * BN;

    const T* __restrict__ xg = reinterpret_cast<const T*>(d.x.p);
    const T* __restrict__ wg = reinterpret_cast<const T*>(d.w);
    const int upshift = d.up == 2 ? 1 : 0;
    const int LH = d.Hi << upshift, LW = d.Wi << upshift;
    const T* __restrict__ x2g = reinterpret_cast<const T*>(d.x2.p);
    const int Ktot = d.Cin + d.Cin2;
    const int nchunks = (Ktot + CK - 1) / CK;
    const size_t wchunk = (size_t)KH * KW * d.CoutPad * CK;   // packed elements per chunk

    // ---- per-thread staging descriptors (independent of the chunk) ----
    int pgo[NPV], pgo2[NPV], plo[NPV], pch[NPV];   // global element offsets into x / x2 (-1: outside image),
                                                   // LDS offset, channel part
    int wgo[NWV], wlo[NWV];
#pragma unroll
    for (int q = 0; q < NPV; ++q) {
        const int v = tid + q * NTHR;
        const int pix = v / VPR, part = v - pix * VPR;
        const int py = pix / PW, px = pix - py * PW;
        const int ly = gy0 * S + py - d.pad_y, lx = gx0 * S + px - d.pad_x;
        const bool ok = v < PVEC && ly >= 0 && ly < LH && lx >= 0 && lx < LW;
        pgo[q] = ok ? (int)(((size_t)(n * d.Hi + (ly >> upshift)) * d.Wi + (lx >> upshift)) * d.x.cs + d.x.coff +
                            part * VEC)
                    : -1;
        pgo2[q] = (ok && x2g) ? (int)(((size_t)(n * d.Hi + (ly >> upshift)) * d.Wi + (lx >> upshift)) * d.x2.cs +
                                      d.x2.coff + part * VEC)
                              : -1;
        plo[q] = v < PVEC ? pix * CKP + part * VEC : -1;
        pch[q] = part * VEC;
    }
#pragma unroll
    for (int q = 0; q < NWV; ++q) {
        const int v = tid + q * NTHR;
        const int row = v / VPR, part = v - row * VPR;
        const int tap = row / BN, co = row - tap * BN;
        wgo[q] = v < WVEC ? (tap * d.CoutPad + co0 + co) * CK + part * VEC : -1;
        wlo[q] = PH * PW * CKP + row * CKP + part * VEC;
    }
    u32x4 rp[NPV], rw[NWV];
    auto load_chunk = [&](int c) {
        const int c0 = c * CK;
#pragma unroll
        for (int q = 0; q < NPV; ++q) {
            u32x4 val = {0u, 0u, 0u, 0u};
            const int k0 = c0 + pch[q];
            if (pgo[q] >= 0) {
                if (k0 < d.Cin) val = *reinterpret_cast<const u32x4*>(xg + (size_t)pgo[q] + c0);
                else if (k0 < Ktot) val = *reinterpret_cast<const u32x4*>(x2g + (size_t)pgo2[q] + (c0 - d.Cin));
            }
            rp[q] = val;
        }
#pragma unroll
        for (int q = 0; q < NWV; ++q)
            if (wgo[q] >= 0) rw[q] = *reinterpret_cast<const u32x4*>(wg + (size_t)c * wchunk + wgo[q]);
    };
    auto store_chunk = [&](int stage) {
        T* base = lds + stage * STAGE;
#pragma unroll
        for (int q = 0; q < NPV; ++q)
            if (plo[q] >= 0) *reinterpret_cast<u32x4*>(base + plo[q]) = rp[q];
#pragma unroll
        for (int q = 0; q < NWV; ++q)
            if (wgo[q] >= 0) *reinterpret_cast<u32x4*>(base + wlo[q]) = rw[q];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int i = lane & 31, g = lane >> 5;
    const int ty = 2 * wm + (i >> 4), tx = i & 15;
    // K-split offsets: k-substep (KSTEPS == 2) or tap-column parity (4x4)
    const int a_ks = (KS == 2) ? (KSTEPS == 2 ? kh * 2 * VEC : kh * CKP) : 0;
    const int b_ks = (KS == 2) ? (KSTEPS == 2 ? kh * 2 * VEC : kh * BN * CKP) : 0;
    const int a_off = ((ty * S) * PW + tx * S) * CKP + g * VEC + a_ks;
    const int b_off = PH * PW * CKP + i * CKP + g * VEC + b_ks;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool has_next = c + 1 < nchunks;
        if (has_next) load_chunk(c + 1);
        const T* ab = lds + (c & 1) * STAGE + a_off;
        const T* bb = lds + (c & 1) * STAGE + b_off;
        if (KSTEPS == 2) {
            constexpr int NKK = (KS == 2) ? 1 : 2;
#pragma unroll
            for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                    for (int kk = 0; kk < NKK; ++kk) {
                        const u32x4 a = *reinterpret_cast<const u32x4*>(ab + (ky * PW + kx) * CKP + kk * 2 * VEC);
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const u32x4 bv = *reinterpret_cast<const u32x4*>(
                                bb + ((ky * KW + kx) * BN + t * 32) * CKP + kk * 2 * VEC);
                            mma16<T>(acc[t], a, bv);
                        }
                    }
        } else {
            constexpr int NKX = (KS == 2) ? KW / 2 : KW, KXS = (KS == 2) ? 2 : 1;
#pragma unroll
            for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                for (int kxi = 0; kxi < NKX; ++kxi) {
                    const int kx = kxi * KXS;
                    const u32x4 a = *reinterpret_cast<const u32x4*>(ab + (ky * PW + kx) * CKP);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const u32x4 bv =
                            *reinterpret_cast<const u32x4*>(bb + ((ky * KW + kx) * BN + t * 32) * CKP);
                        mma16<T>(acc[t], a, bv);
                    }
                }
        }
        if (has_next) store_chunk((c + 1) & 1);
        __syncthreads();
    }

    // ---- fused epilogue for one 32-channel output tile ----
    char* slab = smem + (size_t)MW * 16 * 64 * sizeof(float) + (size_t)wave * EPI_STAGE_BYTES;   // behind `red`
    auto epilogue = [&](const f32x16& a, int t) {
        conv_epilogue<T>(d, a, co0 + t * 32, n, gy0 + 2 * wm, gx0, lane, slab);
    };

    if (KS == 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t) epilogue(acc[t], t);
    } else {
        // combine the two k-halves through LDS (stage buffers are free after the last barrier);
        // with NT == 2 each half finishes one of the two output tiles, otherwise half 0 finishes.
        float* red = reinterpret_cast<float*>(smem);          // [MW][16][64]
        float* mine = red + (wm * 16) * 64 + lane;
        if (NT == 2) {
            if (kh == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[r * 64] = acc[NT - 1][r];
            }
            __syncthreads();
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[NT - 1][r] += mine[r * 64];
            }
            __syncthreads();
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[r * 64] = acc[0][r];
            }
            __syncthreads();
            if (kh == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += mine[r * 64];
                epilogue(acc[0], 0);
            } else {
                epilogue(acc[NT - 1], NT - 1);
            }
        } else {
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[r * 64] = acc[0][r];
            }
            __syncthreads();
            if (kh == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += mine[r * 64];
                epilogue(acc[0], 0);
            }
        }
    }
}

template <typename T, int KH, int KW, int S, int NT, int MW, int KS>
__global__ __launch_bounds__(64 * MW * KS) void conv_kernel(const ssr_conv_desc d) {
    conv_body<T, KH, KW, S, NT, MW, KS>(d);
}
// up to four descriptors of identical geometry in one launch (blockIdx.z selects): the four output-parity classes of
// a stride-2 transposed convolution (r01 rocprofv3: 36 launches of 19..24 us per step, each a quarter of the chip)
struct ssr_conv_desc4 { ssr_conv_desc d[4]; };
template <typename T, int KH, int KW, int S, int NT, int MW, int KS>
__global__ __launch_bounds__(64 * MW * KS) void conv_kernel4(const ssr_conv_desc4 p) {
    conv_body<T, KH, KW, S, NT, MW, KS>(p.d[blockIdx.z]);
}

template <typename T, int KH, int KW, int S, int NT, int MW, int KS>
int launch_conv4(const ssr_conv_desc* ds, int n, hipStream_t st) {
    constexpr int VEC = DT<T>::VEC, KSTEPS = (KH == 4) ? 1 : 2, CK = KSTEPS * 2 * VEC, CKP = CK + VEC, BN = 32 * NT;
    constexpr int TH = 2 * MW, TW = 16, PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
    constexpr size_t stage = (size_t)(PH * PW + KH * KW * BN) * CKP * sizeof(T);
    constexpr size_t red = (size_t)MW * 16 * 64 * sizeof(float) + (size_t)MW * KS * EPI_STAGE_BYTES;
    constexpr size_t lds = 2 * stage > red ? 2 * stage : red;
    auto kern = conv_kernel4<T, KH, KW, S, NT, MW, KS>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    ssr_conv_desc4 p;
    for (int k = 0; k < 4; ++k) p.d[k] = ds[k < n ? k : 0];
    const ssr_conv_desc& d = ds[0];
    const int tiles = ((d.Gw + TW - 1) / TW) * ((d.Gh + TH - 1) / TH) * d.N;
    hipLaunchKernelGGL(kern, dim3(tiles, d.CoutPad / BN, n), dim3(64 * MW * KS), lds, st, p);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

template <typename T, int KH, int KW, int S, int NT, int MW, int KS>
int launch_conv(const ssr_conv_desc& d, hipStream_t st) {
    constexpr int VEC = DT<T>::VEC, KSTEPS = (KH == 4) ? 1 : 2, CK = KSTEPS * 2 * VEC, CKP = CK + VEC, BN = 32 * NT;
    constexpr int TH = 2 * MW, TW = 16, PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
    constexpr size_t stage = (size_t)(PH * PW + KH * KW * BN) * CKP * sizeof(T);
    constexpr size_t red = (size_t)MW * 16 * 64 * sizeof(float) + (size_t)MW * KS * EPI_STAGE_BYTES;
    constexpr size_t lds = 2 * stage > red ? 2 * stage : red;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv_kernel<T, KH, KW, S, NT, MW, KS>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    const int tiles = ((d.Gw + TW - 1) / TW) * ((d.Gh + TH - 1) / TH) * d.N;
    dim3 grid(tiles, d.CoutPad / BN, 1);
    hipLaunchKernelGGL(kern, grid, dim3(64 * MW * KS), lds, st, d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

// BN = 64 when the padded output width allows it; small problems use 4x16-pixel tiles (MW = 2) so
// that there are at least as many workgroups as CUs.
inline void pick_tile(const ssr_conv_desc& d, bool& nt2, bool& small) {
    nt2 = (d.CoutPad % 64) == 0;
    const long tiles4 = (long)((d.Gw + 15) / 16) * ((d.Gh + 7) / 8) * d.N * (d.CoutPad / (nt2 ? 64 : 32));
    small = tiles4 < 384;
}

template <typename T, int KH, int KW, int S>
int dispatch_tile(const ssr_conv_desc& d, hipStream_t st) {
    bool nt2, small;
    pick_tile(d, nt2, small);
    if (nt2) return small ? launch_conv<T, KH, KW, S, 2, 2, 2>(d, st) : launch_conv<T, KH, KW, S, 2, 4, 2>(d, st);
    return small ? launch_conv<T, KH, KW, S, 1, 2, 2>(d, st) : launch_conv<T, KH, KW, S, 1, 4, 2>(d, st);
}

template <typename T>
int dispatch_geom(const ssr_conv_desc& d, hipStream_t st) {
    if (d.KH == 3 && d.KW == 3 && d.stride == 1) return dispatch_tile<T, 3, 3, 1>(d, st);
    if (d.KH == 4 && d.KW == 4 && d.stride == 2) return dispatch_tile<T, 4, 4, 2>(d, st);
    if (d.KH == 2 && d.KW == 2 && d.stride == 1) return dispatch_tile<T, 2, 2, 1>(d, st);
    return SSR_EUNSUP;
}

// ------------------------------------------------------------------------------------------------------------------
// SSR_F32X3: fp32 tensors in HBM, bf16 matrix cores with split operands.  There is no TF32 on gfx950 and the fp32-input
// MFMA runs at 1/16 of the bf16 rate; an fp32 value x is split as x = hi + lo + O(2^-17 x) with hi = bf16(x), lo = bf16(x - hi)
// and a product is formed as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  (three v_mfma_f32_32x32x16_bf16 into the same fp32
// accumulator; the dropped lo*lo term is 2^-16 relative).  The split happens ONCE per staged element, when the register-staged
// chunk is written to LDS: a 16-channel chunk of a pixel (64 B of fp32) becomes 16 hi + 16 lo bf16 (the same 64 B), rows keep
// the conflict-free 80-byte pitch.  Everything else (pipelining, epilogue, packed-weight format [chunk][tap][co][16] fp32)
// is the fp32 kernel's.  KS = 2 splits the taps of a chunk by parity between the two waves of a pixel group.
// Stride-1 2x2 / 3x3 layers; the 4x4 stride-2 layers keep the exact fp32 kernel (two 20 KB-per-row-block stages of their
// 18x34 patch would not fit the LDS at 16 channels per chunk).
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4(const u32x4& v, uint2& hi, uint2& lo) {
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

// H: the fp16-split forward arithmetic (SSR_F32H, include/ssr_hip.h) on the same data path - fp16 pieces, v_mfma_f32_32x32x16_f16,
// accumulators x 2^-SSR_F32H_WSHIFT before the epilogue
template <int KH, int KW, int NT, int MW, int KS, bool H = false>
__device__ __forceinline__ void conv_body_x3(const ssr_conv_desc& d) {
    constexpr int VEC = 4, CK = 16, VPR = 4, ROWB = 80, BN = 32 * NT, S = 1;
    constexpr int TH = 2 * MW, TW = 16;
    constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
    constexpr int NTHR = 64 * MW * KS;
    constexpr int PVEC = PH * PW * VPR, WVEC = KH * KW * BN * VPR;
    constexpr int NPV = (PVEC + NTHR - 1) / NTHR, NWV = (WVEC + NTHR - 1) / NTHR;
    constexpr int STAGE = (PH * PW + KH * KW * BN) * ROWB;   // bytes per LDS stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % MW, kh = wave / MW;
    const int tiles_x = (d.Gw + TW - 1) / TW, tiles_y = (d.Gh + TH - 1) / TH;
    int b = blockIdx.x;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty_i * TH, gx0 = tx_i * TW;
    const int co0 = blockIdx.y * BN;

    const float* __restrict__ xg = reinterpret_cast<const float*>(d.x.p);
    const float* __restrict__ wg = reinterpret_cast<const float*>(d.w);
    const float* __restrict__ x2g = reinterpret_cast<const float*>(d.x2.p);
    const int upshift = d.up == 2 ? 1 : 0;
    const int LH = d.Hi << upshift, LW = d.Wi << upshift;
    const int Ktot = d.Cin + d.Cin2;
    const int nchunks = (Ktot + CK - 1) / CK;
    const size_t wchunk = (size_t)KH * KW * d.CoutPad * CK;

    int pgo[NPV], pgo2[NPV], plo[NPV], pch[NPV];
    int wgo[NWV], wlo[NWV];
#pragma unroll
    for (int q = 0; q < NPV; ++q) {
        const int v = tid + q * NTHR;
        const int pix = v / VPR, part = v - pix * VPR;
        const int py = pix / PW, px = pix - py * PW;
        const int ly = gy0 + py - d.pad_y, lx = gx0 + px - d.pad_x;
        const bool ok = v < PVEC && ly >= 0 && ly < LH && lx >= 0 && lx < LW;
        pgo[q] = ok ? (int)(((size_t)(n * d.Hi + (ly >> upshift)) * d.Wi + (lx >> upshift)) * d.x.cs + d.x.coff + part * VEC) : -1;
        pgo2[q] = (ok && x2g) ? (int)(((size_t)(n * d.Hi + (ly >> upshift)) * d.Wi + (lx >> upshift)) * d.x2.cs + d.x2.coff + part * VEC)
                              : -1;
        plo[q] = v < PVEC ? pix * ROWB + part * 8 : -1;          // hi half of the row; lo half 32 bytes further
        pch[q] = part * VEC;
    }
#pragma unroll
    for (int q = 0; q < NWV; ++q) {
        const int v = tid + q * NTHR;
        const int row = v / VPR, part = v - row * VPR;
        const int tap = row / BN, co = row - tap * BN;
        wgo[q] = v < WVEC ? (tap * d.CoutPad + co0 + co) * CK + part * VEC : -1;
        wlo[q] = (PH * PW + row) * ROWB + part * 16;            // packed rows arrive pre-split [16 hi | 16 lo] (misc.hip put_packed)
    }
    u32x4 rp[NPV], rw[NWV];
    auto load_chunk = [&](int c) {
        const int c0 = c * CK;
#pragma unroll
        for (int q = 0; q < NPV; ++q) {
            u32x4 val = {0u, 0u, 0u, 0u};
            const int k0 = c0 + pch[q];
            if (pgo[q] >= 0) {
                if (k0 < d.Cin) val = *reinterpret_cast<const u32x4*>(xg + (size_t)pgo[q] + c0);
                else if (k0 < Ktot) val = *reinterpret_cast<const u32x4*>(x2g + (size_t)pgo2[q] + (c0 - d.Cin));
            }
            rp[q] = val;
        }
#pragma unroll
        for (int q = 0; q < NWV; ++q)
            if (wgo[q] >= 0) rw[q] = *reinterpret_cast<const u32x4*>(wg + (size_t)c * wchunk + wgo[q]);
    };
    auto store_chunk = [&](int stage) {
        char* base = smem + stage * STAGE;
        uint2 hi, lo;
#pragma unroll
        for (int q = 0; q < NPV; ++q)
            if (plo[q] >= 0) {
                split_f32x4<H>(rp[q], hi, lo);
                *reinterpret_cast<uint2*>(base + plo[q]) = hi;
                *reinterpret_cast<uint2*>(base + plo[q] + 32) = lo;
            }
#pragma unroll
        for (int q = 0; q < NWV; ++q)
            if (wgo[q] >= 0) *reinterpret_cast<u32x4*>(base + wlo[q]) = rw[q];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int i = lane & 31, g = lane >> 5;
    const int ty = 2 * wm + (i >> 4), tx = i & 15;
    const int a_off = (ty * PW + tx) * ROWB + g * 16;               // lane (i, g): channels g*8 .. g*8+7 of pixel i
    const int b_off = (PH * PW + i) * ROWB + g * 16;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool has_next = c + 1 < nchunks;
        if (has_next) load_chunk(c + 1);
        const char* ab = smem + (c & 1) * STAGE + a_off;
        const char* bb = smem + (c & 1) * STAGE + b_off;
#pragma unroll
        for (int tap = 0; tap < KH * KW; ++tap) {
            if (KS == 2 && (tap & 1) != kh) continue;                // wave-uniform: the two k-halves take alternate taps
            const int ky = tap / KW, kx = tap - ky * KW;
            const bf16x8 ahi = *reinterpret_cast<const bf16x8*>(ab + (ky * PW + kx) * ROWB);
            const bf16x8 alo = *reinterpret_cast<const bf16x8*>(ab + (ky * PW + kx) * ROWB + 32);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 bhi = *reinterpret_cast<const bf16x8*>(bb + (tap * BN + t * 32) * ROWB);
                const bf16x8 blo = *reinterpret_cast<const bf16x8*>(bb + (tap * BN + t * 32) * ROWB + 32);
                acc[t] = split_mfma<H>(alo, bhi, acc[t]);
                acc[t] = split_mfma<H>(ahi, blo, acc[t]);
                acc[t] = split_mfma<H>(ahi, bhi, acc[t]);
            }
        }
        if (has_next) store_chunk((c + 1) & 1);
        __syncthreads();
    }

    char* slab = smem + (size_t)MW * 16 * 64 * sizeof(float) + (size_t)wave * EPI_STAGE_BYTES;
    auto epilogue = [&](const f32x16& a, int t) {
        if constexpr (H) {
            f32x16 b;
#pragma unroll
            for (int r = 0; r < 16; ++r) b[r] = a[r] * SSR_F32H_UNSCALE;
            conv_epilogue<float>(d, b, co0 + t * 32, n, gy0 + 2 * wm, gx0, lane, slab);
        } else {
            conv_epilogue<float>(d, a, co0 + t * 32, n, gy0 + 2 * wm, gx0, lane, slab);
        }
    };
    if (KS == 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t) epilogue(acc[t], t);
    } else {
        // combine the two tap halves through LDS; with NT == 2 each half finishes one of the two output tiles
        float* red = reinterpret_cast<float*>(smem);          // [MW][16][64]
        float* mine = red + (wm * 16) * 64 + lane;
        if (NT == 2) {
            if (kh == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[r * 64] = acc[1][r];
            }
            __syncthreads();
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[1][r] += mine[r * 64];
            }
            __syncthreads();
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[r * 64] = acc[0][r];
            }
            __syncthreads();
            if (kh == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += mine[r * 64];
                epilogue(acc[0], 0);
            } else {
                epilogue(acc[1], 1);
            }
        } else {
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[r * 64] = acc[0][r];
            }
            __syncthreads();
            if (kh == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += mine[r * 64];
                epilogue(acc[0], 0);
            }
        }
    }
}

template <int KH, int KW, int NT, int MW, int KS>
__global__ __launch_bounds__(64 * MW * KS) void conv_x3_kernel(const ssr_conv_desc d) {
    conv_body_x3<KH, KW, NT, MW, KS>(d);
}
template <int KH, int KW, int NT, int MW, int KS>
__global__ __launch_bounds__(64 * MW * KS) void conv_h3_kernel(const ssr_conv_desc d) {      // the fp16-split form (SSR_F32H)
    conv_body_x3<KH, KW, NT, MW, KS, true>(d);
}

template <int KH, int KW, int NT, int MW, int KS, bool H = false>
int launch_conv_x3(const ssr_conv_desc& d, hipStream_t st) {
    constexpr int BN = 32 * NT, TH = 2 * MW, TW = 16, PH = TH - 1 + KH, PW = TW - 1 + KW;
    constexpr size_t stage = (size_t)(PH * PW + KH * KW * BN) * 80;
    constexpr size_t red = (size_t)MW * 16 * 64 * sizeof(float) + (size_t)MW * KS * EPI_STAGE_BYTES;
    constexpr size_t lds = 2 * stage > red ? 2 * stage : red;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = H ? conv_h3_kernel<KH, KW, NT, MW, KS> : conv_x3_kernel<KH, KW, NT, MW, KS>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    const int tiles = ((d.Gw + TW - 1) / TW) * ((d.Gh + TH - 1) / TH) * d.N;
    hipLaunchKernelGGL(kern, dim3(tiles, d.CoutPad / BN, 1), dim3(64 * MW * KS), lds, st, d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// Round 5: the split kernel with a DEEP load pipeline.  rocprofv3 of the fp32x3 step (profiles/r05a_*): a 32-channel-output body
// convolution at B = 32 took 15.5 us for 1.3 us of MFMA work - 2.2 us per 16-channel chunk, i.e. one memory round trip per chunk:
// the loads of chunk c+1 were issued when chunk c's 14 MFMAs per wave began and waited for when they ended.  Here
//   * a pipeline STAGE is CPS 16-channel chunks (NT = 1: two, 75 KB per LDS stage; NT = 2: one), so a layer has half the stages;
//   * PD stages are in flight in REGISTERS beside the two LDS stages: the loads of stage s + PD are issued when stage s starts;
//   * every staging load is ONE unconditional buffer load (a lane that must read zeros - outside the image, past the last
//     channel, past the weight table - uses an offset beyond num_records), so hipcc can count them: the wait in front of a
//     stage's LDS store is vmcnt(loads of the PD - 1 later stages), not vmcnt(0);
//   * the two tap halves (KS = 2) take alternate (chunk, tap) pairs of a stage.
// Same packed weights ([chunk16][tap][co][16 hi | 16 lo]), LDS row format, MFMA order inside a chunk and epilogue as conv_body_x3.
// ------------------------------------------------------------------------------------------------------------------
#ifdef SSR_PROBE   // tools/x3_probe.hip: s_memtime stamps of thread 0 of every workgroup, 24 slots per workgroup
#define XPROBE(k) do { if (threadIdx.x == 0) g_probe[(blockIdx.y * gridDim.x + blockIdx.x) * 24 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XPROBE(k)
#endif
constexpr int X3_OOB = 0x7ffffff0;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x3_rsrc(const void* p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffff00L ? 0x7fffff00L : bytes), 0x00020000);
}

template <int KH, int KW, int NT, int MW, int KS, int CPS, int PD>
__device__ __forceinline__ void conv_body_x3p(const ssr_conv_desc& d) {
    constexpr int VPR = 4, ROWB = 80, BN = 32 * NT, NTAP = KH * KW;
    constexpr int TH = 2 * MW, TW = 16;
    constexpr int PH = TH - 1 + KH, PW = TW - 1 + KW;
    constexpr int NTHR = 64 * MW * KS;
    constexpr int PV1 = PH * PW * VPR, WV1 = NTAP * BN * VPR;           // vectors of one 16-channel chunk
    constexpr int PVEC = CPS * PV1, WVEC = CPS * WV1;
    constexpr int NPV = (PVEC + NTHR - 1) / NTHR, NWV = (WVEC + NTHR - 1) / NTHR;
    constexpr int SUB = (PH * PW + NTAP * BN) * ROWB;                    // bytes of one chunk in LDS
    constexpr int STAGE = CPS * SUB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    XPROBE(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % MW, kh = wave / MW;
    const int tiles_x = (d.Gw + TW - 1) / TW, tiles_y = (d.Gh + TH - 1) / TH;
    int b = blockIdx.x;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty_i * TH, gx0 = tx_i * TW;
    const int co0 = blockIdx.y * BN;
    const int upshift = d.up == 2 ? 1 : 0;
    const int LH = d.Hi << upshift, LW = d.Wi << upshift;
    const int Cin = d.Cin, Cin2 = d.Cin2, Ktot = Cin + Cin2;
    const int nchunks = (Ktot + 15) / 16, nst = (nchunks + CPS - 1) / CPS;
    const long xbytes = (long)d.N * d.Hi * d.Wi * 4;                     // x view.cs = bytes of an input-side tensor
    const void* xp = d.x.p;
    const void* x2p = d.x2.p ? d.x2.p : d.x.p;
    const __amdgpu_buffer_rsrc_t rsw = x3_rsrc(d.w, (long)nchunks * NTAP * d.CoutPad * 64);

    // ---- per-thread staging descriptors (independent of the stage) ----
    int pgo[NPV], pgo2[NPV], plo[NPV], pk[NPV];       // byte offset of the pixel in x / x2 (X3_OOB: zeros), LDS offset, channel inside the stage
    int wgo[NWV], wlo[NWV], wcc[NWV];
#pragma unroll
    for (int q = 0; q < NPV; ++q) {
        const int v = tid + q * NTHR;
        const int cc = v / PV1, r = v - cc * PV1;
        const int pix = r / VPR, part = r - pix * VPR;
        const int py = pix / PW, px = pix - py * PW;
        const int ly = gy0 + py - d.pad_y, lx = gx0 + px - d.pad_x;
        const bool ok = v < PVEC && ly >= 0 && ly < LH && lx >= 0 && lx < LW;
        const int po = (n * d.Hi + (ly >> upshift)) * d.Wi + (lx >> upshift);
        pgo[q] = ok ? (po * d.x.cs + d.x.coff) * 4 : X3_OOB;
        pgo2[q] = (ok && d.x2.p) ? (po * d.x2.cs + d.x2.coff) * 4 : X3_OOB;
        pk[q] = cc * 16 + part * 4;
        plo[q] = v < PVEC ? cc * SUB + pix * ROWB + part * 8 : -1;      // hi half of the row; lo half 32 bytes further
    }
#pragma unroll
    for (int q = 0; q < NWV; ++q) {
        const int v = tid + q * NTHR;
        const int cc = v / WV1, r = v - cc * WV1;
        const int row = r / VPR, part = r - row * VPR;
        const int tap = row / BN, co = row - tap * BN;
        wcc[q] = v < WVEC ? cc : 0x10000;                               // (beyond the table: never below nchunks)
        wgo[q] = ((cc * NTAP + tap) * d.CoutPad + co0 + co) * 64 + part * 16;
        wlo[q] = v < WVEC ? cc * SUB + (PH * PW + row) * ROWB + part * 16 : -1;   // packed rows arrive pre-split [16 hi | 16 lo]
    }
    u32x4 rp[PD][NPV], rw[PD][NWV];
    auto load_stage = [&](int s, auto jc) {            // stage s -> register set j; a stage past the end reads zeros (no memory access)
        constexpr int j = decltype(jc)::value;
        const int c0 = s * CPS * 16;
        const bool live = s < nst;
        const bool in_x = c0 < Cin;                    // a stage lies in ONE of the two views (the dispatcher checks Cin % (16 CPS) == 0 with x2)
        const int cb = in_x ? c0 : c0 - Cin, clim = live ? (in_x ? Cin : Cin2) : 0, cs_bytes = (in_x ? d.x.cs : d.x2.cs);
        const __amdgpu_buffer_rsrc_t rs = x3_rsrc(in_x ? xp : x2p, xbytes * cs_bytes);
#pragma unroll
        for (int q = 0; q < NPV; ++q) {
            const int k = cb + pk[q];
            const int po = in_x ? pgo[q] : pgo2[q];
            rp[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (k < clim && po != X3_OOB) ? po + k * 4 : X3_OOB, 0, 0);
        }
        const int wbase = s * CPS * NTAP * d.CoutPad * 64, cleft = nchunks - s * CPS;
#pragma unroll
        for (int q = 0; q < NWV; ++q)
            rw[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wcc[q] < cleft ? wgo[q] + wbase : X3_OOB, 0, 0);
    };
    // one vector of register set j -> LDS stage `buf` (u < NPV: a patch vector, split here; else a weight vector)
    auto store_unit = [&](int buf, auto jc, auto uc) {
        constexpr int j = decltype(jc)::value, u = decltype(uc)::value;
        char* base = smem + buf * STAGE;
        if constexpr (u < NPV) {
            uint2 hi, lo;
            split4(rp[j][u], hi, lo);
            if ((u + 1) * NTHR <= PVEC || plo[u] >= 0) {
                *reinterpret_cast<uint2*>(base + plo[u]) = hi;
                *reinterpret_cast<uint2*>(base + plo[u] + 32) = lo;
            }
        } else {
            constexpr int q = u - NPV;
            if ((q + 1) * NTHR <= WVEC || wlo[q] >= 0) *reinterpret_cast<u32x4*>(base + wlo[q]) = rw[j][q];
        }
    };
    auto store_stage = [&](int buf, auto jc) { static_for<0, NPV + NWV>([&](auto uc) { store_unit(buf, jc, uc); }); };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int i = lane & 31, g = lane >> 5;
    const int ty = 2 * wm + (i >> 4), tx = i & 15;
    const int a_off = (ty * PW + tx) * ROWB + g * 16;               // lane (i, g): channels g*8 .. g*8+7 of pixel i
    const int b_off = (PH * PW + i) * ROWB + g * 16;
    // A wave's share of a stage: the (chunk, tap) pairs of its parity (KS = 2) as STRAIGHT-LINE code (one instantiation per
    // k-half, picked by a scalar branch): the operand reads of pair k + 1 are issued before the MFMAs of pair k and pinned there
    // (left alone hipcc sinks every ds_read next to its MFMA: read, wait, MFMA chains).
    auto contract_h = [&](int buf, auto hc, auto&& between) {
        constexpr int H = decltype(hc)::value;
        constexpr int NITEM = KS == 2 ? (CPS * NTAP + 1 - H) / 2 : CPS * NTAP;     // pairs H, H + 2, ...
        const char* ab = smem + buf * STAGE + a_off;
        const char* bb = smem + buf * STAGE + b_off;
        bf16x8 fa[2][2], fb[2][NT][2];
        auto issue = [&](auto kc) {
            constexpr int k = decltype(kc)::value, it = KS == 2 ? 2 * k + H : k;
            constexpr int cc = it / NTAP, tap = it % NTAP, ky = tap / KW, kx = tap % KW;
            fa[k & 1][0] = *reinterpret_cast<const bf16x8*>(ab + cc * SUB + (ky * PW + kx) * ROWB);
            fa[k & 1][1] = *reinterpret_cast<const bf16x8*>(ab + cc * SUB + (ky * PW + kx) * ROWB + 32);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                fb[k & 1][t][0] = *reinterpret_cast<const bf16x8*>(bb + cc * SUB + (tap * BN + t * 32) * ROWB);
                fb[k & 1][t][1] = *reinterpret_cast<const bf16x8*>(bb + cc * SUB + (tap * BN + t * 32) * ROWB + 32);
            }
        };
        issue(std::integral_constant<int, 0>{});
        static_for<0, NITEM>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (k + 1 < NITEM) issue(std::integral_constant<int, k + 1>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k & 1][1], fb[k & 1][t][0], acc[t], 0, 0, 0);   // a_lo * b_hi
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k & 1][0], fb[k & 1][t][1], acc[t], 0, 0, 0);   // a_hi * b_lo
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k & 1][0], fb[k & 1][t][0], acc[t], 0, 0, 0);   // a_hi * b_hi
            }
            __builtin_amdgcn_sched_barrier(0);
            between(kc, std::integral_constant<int, NITEM>{});
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    const int kh_s = __builtin_amdgcn_readfirstlane(kh);
    // The next stage's LDS store (split of the staged fp32 patch vectors + the writes) rides INSIDE the MFMA stream of this stage,
    // one or two vectors behind each (chunk, tap) pair: as a phase of its own it cost 1.2 k of a stage's 4.7 k ticks with the matrix
    // pipe idle (tools/x3_probe.hip, profiles/r05*_x3_probe.txt); the other LDS buffer is the target, so the stage's one barrier stays.
    auto contract = [&](int buf, auto jn, bool store_next) {       // jn: register set that holds the next stage
        auto none = [](auto, auto) {};
        auto units = [&](auto kc, auto nc) {
            constexpr int k = decltype(kc)::value, NI = decltype(nc)::value, NU = NPV + NWV, PER = (NU + NI - 1) / NI;
            static_for<0, PER>([&](auto ec) {
                constexpr int u = k * PER + decltype(ec)::value;
                if constexpr (u < NU) store_unit(buf ^ 1, jn, std::integral_constant<int, u>{});
            });
        };
        if (store_next) {
            if (KS == 2 && kh_s == 1) contract_h(buf, std::integral_constant<int, 1>{}, units);
            else contract_h(buf, std::integral_constant<int, 0>{}, units);
        } else {
            if (KS == 2 && kh_s == 1) contract_h(buf, std::integral_constant<int, 1>{}, none);
            else contract_h(buf, std::integral_constant<int, 0>{}, none);
        }
    };

    XPROBE(1);
    static_for<0, PD>([&](auto jc) { load_stage(decltype(jc)::value, jc); });
    store_stage(0, std::integral_constant<int, 0>{});
    __syncthreads();
    XPROBE(2);
    for (int c = 0; c < nst; c += PD) {
        static_for<0, PD>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int cur = c + j;
            if (cur < nst) {
                load_stage(cur + PD, jc);                                  // set j held stage `cur`, which is in LDS already
                contract(cur & 1, std::integral_constant<int, (j + 1) % PD>{}, cur + 1 < nst);
#ifdef SSR_PROBE
                const unsigned long long tm_ = __builtin_amdgcn_s_memtime();      // stage 0: end of the MFMAs + the interleaved LDS store
#endif
#ifdef SSR_PROBE
                if (threadIdx.x == 0 && cur == 0) { g_probe[(blockIdx.y * gridDim.x + blockIdx.x) * 24 + 20] = tm_; g_probe[(blockIdx.y * gridDim.x + blockIdx.x) * 24 + 21] = __builtin_amdgcn_s_memtime(); }
#endif
                __syncthreads();
#ifdef SSR_PROBE
                if (cur < 12) XPROBE(3 + cur);
#endif
            }
        });
    }
    XPROBE(15);

    char* slab = smem + (size_t)MW * 16 * 64 * sizeof(float) + (size_t)wave * EPI_STAGE_BYTES;
    auto epilogue = [&](const f32x16& a, int t) {
        conv_epilogue<float>(d, a, co0 + t * 32, n, gy0 + 2 * wm, gx0, lane, slab);
    };
    if (KS == 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t) epilogue(acc[t], t);
    } else {
        float* red = reinterpret_cast<float*>(smem);          // [MW][16][64]
        float* mine = red + (wm * 16) * 64 + lane;
        if (NT == 2) {
            if (kh == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[r * 64] = acc[1][r];
            }
            __syncthreads();
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[1][r] += mine[r * 64];
            }
            __syncthreads();
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[r * 64] = acc[0][r];
            }
            __syncthreads();
            if (kh == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += mine[r * 64];
                epilogue(acc[0], 0);
            } else {
                epilogue(acc[1], 1);
            }
        } else {
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[r * 64] = acc[0][r];
            }
            __syncthreads();
            if (kh == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += mine[r * 64];
                epilogue(acc[0], 0);
            }
        }
    }
    XPROBE(17);
}

template <int KH, int KW, int NT, int MW, int KS, int CPS, int PD>
__global__ __launch_bounds__(64 * MW * KS) void conv_x3p_kernel(const ssr_conv_desc d) {
    conv_body_x3p<KH, KW, NT, MW, KS, CPS, PD>(d);
}

template <int KH, int KW, int NT, int MW, int KS, int CPS, int PD>
int launch_conv_x3p(const ssr_conv_desc& d, hipStream_t st) {
    constexpr int BN = 32 * NT, TH = 2 * MW, TW = 16, PH = TH - 1 + KH, PW = TW - 1 + KW;
    constexpr size_t stage = (size_t)CPS * (PH * PW + KH * KW * BN) * 80;
    constexpr size_t red = (size_t)MW * 16 * 64 * sizeof(float) + (size_t)MW * KS * EPI_STAGE_BYTES;
    constexpr size_t lds = 2 * stage > red ? 2 * stage : red;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv_x3p_kernel<KH, KW, NT, MW, KS, CPS, PD>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    const int tiles = ((d.Gw + TW - 1) / TW) * ((d.Gh + TH - 1) / TH) * d.N;
    hipLaunchKernelGGL(kern, dim3(tiles, d.CoutPad / BN, 1), dim3(64 * MW * KS), lds, st, d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

// the deep-pipeline kernel reaches its tensors through 32-bit byte offsets and takes a pipeline stage from ONE input view
bool x3p_ok(const ssr_conv_desc& d, int cps) {
    // Opt-in (SSR_X3_PIPE=1).  Measured r05b / r05d (tools/x3_probe.hip): parity-equal and NOT faster - per 32-channel stage the
    // MFMA phase takes 2.7 k ticks for 1.7 k of MFMAs, the split + LDS store 1.2 k, the barrier 0.7 k, and neither deeper load
    // prefetch nor riding the store inside the MFMA stream changes the sum (a wave's VALU / LDS-store work does not overlap its own
    // MFMAs); a launch carries ~11.7 k ticks of fixed parts (set-up 1.7 k, first loads 4.8 k, reduce + epilogue 5.2 k).
    static const bool on = [] { const char* e = getenv("SSR_X3_PIPE"); return e && e[0] == '1'; }();
    if (!on) return false;
    const long lim = 0x7fffff00L, npx = (long)d.N * d.Hi * d.Wi * 4;
    if (npx * d.x.cs > lim || (d.x2.p && npx * d.x2.cs > lim)) return false;
    if (d.x2.p && (d.Cin % (16 * cps)) != 0) return false;
    return true;
}

template <int KH, int KW, bool H = false>
int dispatch_tile_x3(const ssr_conv_desc& d, hipStream_t st) {
    bool nt2, small;
    pick_tile(d, nt2, small);
    // The split kernel always takes the 8x16-pixel workgroup tile: its weight slab (hi + lo planes, converted by every workgroup)
    // costs as much to stage as 288 pixel rows, so halving the pixels per workgroup the way the plain kernels do for small
    // grids doubles that share.  r02m, whole fp32x3 step at B = 32: 45.3 ms with the plain kernels' threshold (384 tiles),
    // 38.1 ms without small tiles, 60.0 ms with small tiles everywhere.  SSR_X3_SMALL = threshold in 8x16 tiles (tuning hook).
    static const long thr = [] { const char* e = getenv("SSR_X3_SMALL"); return e ? atol(e) : 0L; }();
    small = (long)((d.Gw + 15) / 16) * ((d.Gh + 7) / 8) * d.N * (d.CoutPad / (nt2 ? 64 : 32)) < thr;
    if (!small && !H) {      // round 5: deep load pipeline (two chunks per stage at 32 output channels, three register stages at 64)
        static const bool p2 = [] { const char* e = getenv("SSR_X3_PIPE2"); return e && e[0] == '1'; }();   // 64 output channels: measured slower than the plain pipeline (r05b: 10.4 vs 9.6 ms per step), opt-in
        if (nt2 && p2 && x3p_ok(d, 1)) return launch_conv_x3p<KH, KW, 2, 4, 2, 1, 3>(d, st);
        if (!nt2 && x3p_ok(d, 2)) return launch_conv_x3p<KH, KW, 1, 4, 2, 2, 2>(d, st);
    }
    if (nt2) return small ? launch_conv_x3<KH, KW, 2, 2, 2, H>(d, st) : launch_conv_x3<KH, KW, 2, 4, 2, H>(d, st);
    return small ? launch_conv_x3<KH, KW, 1, 2, 2, H>(d, st) : launch_conv_x3<KH, KW, 1, 4, 2, H>(d, st);
}

bool view_ok(const ssr_view& v, bool required) {
    if (!v.p) return !required;
    return (v.cs % 8) == 0 && (v.coff % 8) == 0 && ((uintptr_t)v.p % 16) == 0;
}

}  // namespace

// K-resident LDS-DMA kernel for the small generator-body layers (conv_res.hip)
bool ssr_conv_res_try(const ssr_conv_desc& d, hipStream_t st, int* rc);
bool ssr_conv_res_qualifies(const ssr_conv_desc& d);
// weight-stationary persistent kernel for the large-spatial layers (conv_ws.hip)
bool ssr_conv_ws_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force);
bool ssr_conv_ws_qualifies(const ssr_conv_desc& d);
// thin-output VALU kernel for the logit / RGB heads (conv_thin.hip)
bool ssr_conv_thin_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force);
bool ssr_conv_thin_qualifies(const ssr_conv_desc& d);
// big-tile kernel for the wide discriminator layers (conv_big.hip)
bool ssr_conv_big_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force);
bool ssr_conv_big_qualifies(const ssr_conv_desc& d);
bool ssr_conv_big_batch_try(const ssr_conv_desc* ds, int n, hipStream_t st, int* rc);
// big-tile kernel of the split-bf16 mode (conv_big_x3.hip)
bool ssr_conv_bigx3_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force);
bool ssr_conv_bigx3_qualifies(const ssr_conv_desc& d);
bool ssr_conv_bigx3_batch_try(const ssr_conv_desc* ds, int n, hipStream_t st, int* rc);
// producer / MFMA-wave ring kernel of the split-bf16 mode for small grids (conv_x3q.hip)
bool ssr_conv_x3q_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force);
bool ssr_conv_x3q_qualifies(const ssr_conv_desc& d);
// register-tiled kernel of the split-bf16 mode for small grids, round 6 (conv_x3r.hip): takes the ring kernel's layers
bool ssr_conv_x3r_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force);
bool ssr_conv_x3r_qualifies(const ssr_conv_desc& d);

extern "C" int ssr_conv2d_s2d_ok(int32_t dtype, int32_t Cin, int32_t Cout, int32_t CoutPad) {
    static const bool off = [] { const char* e = getenv("SSR_CONV_S2D"); return e && e[0] == '0'; }();
    if (dtype == SSR_F32X3 || dtype == SSR_F32H) {   // split modes: 16-channel chunks (conv_big_x3.hip)
        static const bool offx = [] { const char* e = getenv("SSR_X3_BIGTILE"); return e && e[0] == '0'; }();
        const int cpx = Cin / 16;
        return !off && !offx && Cin >= 16 && (Cin % 16) == 0 && (cpx & (cpx - 1)) == 0 && (CoutPad % 64) == 0 && (Cout % 8) == 0;
    }
    const int cpc = Cin / 32;
    return !off && dtype == SSR_BF16 && Cin >= 32 && (Cin % 32) == 0 && (cpc & (cpc - 1)) == 0 && (CoutPad % 64) == 0 && (Cout % 8) == 0;
}

extern "C" int ssr_conv2d_variant(const ssr_conv_desc* dp) {
    if (!dp) return SSR_EINVAL;
    if (dp->s2d) return dp->KH * 1000 + dp->stride * 100 + 2 * 10 + 9;
    // ReLU epilogues (the VGG19 layers) exist only in the pipelined kernel outside the split-bf16 mode: conv2d_impl forces it (impl = 3)
    const bool split = dp->dtype == SSR_F32X3 || dp->dtype == SSR_F32H;
    const bool pipelined_only = !split && (dp->act == SSR_ACT_RELU || dp->m_relu);
    if (!pipelined_only) {
        if (dp->dtype != SSR_BF16 && ssr_conv_thin_qualifies(*dp)) return dp->KH * 1000 + dp->stride * 100 + 1 * 10 + 7;        // digit 7 = thin-output VALU kernel
        if (split && ssr_conv_bigx3_qualifies(*dp)) return dp->KH * 1000 + dp->stride * 100 + 2 * 10 + 9;   // digit 9 = big tile
        if (split && ssr_conv_x3r_qualifies(*dp))
            return dp->KH * 1000 + dp->stride * 100 + ((dp->CoutPad % 64) == 0 ? 2 : 1) * 10 + 5;                              // digit 5 = register-tiled, K split over four waves (conv_x3r.hip)
        if (dp->dtype == SSR_F32X3 && ssr_conv_x3q_qualifies(*dp))
            return dp->KH * 1000 + dp->stride * 100 + ((dp->CoutPad % 64) == 0 ? 2 : 1) * 10 + 6;                              // digit 6 = twelve-wave ring (conv_x3q.hip)
        if (!split) {
            if (ssr_conv_thin_qualifies(*dp)) return dp->KH * 1000 + dp->stride * 100 + 1 * 10 + 7;  // digit 7 = thin-output VALU kernel
            if (ssr_conv_ws_qualifies(*dp)) return dp->KH * 1000 + dp->stride * 100 + 1 * 10 + 8;    // digit 8 = weight-stationary
            if (ssr_conv_big_qualifies(*dp)) return dp->KH * 1000 + dp->stride * 100 + 2 * 10 + 9;   // digit 9 = big tile
            if (dp->dtype == SSR_F32 && ssr_conv_x3r_qualifies(*dp))
                return dp->KH * 1000 + dp->stride * 100 + ((dp->CoutPad % 64) == 0 ? 2 : 1) * 10 + 5;   // digit 5 = register-tiled (exact fp32 form)
            if (ssr_conv_res_qualifies(*dp)) return dp->KH * 1000 + dp->stride * 100 + 1 * 10 + 1;   // WAVES digit 1 = resident
        }
    }
    bool nt2, small;
    pick_tile(*dp, nt2, small);
    return dp->KH * 1000 + dp->stride * 100 + (nt2 ? 2 : 1) * 10 + (small ? 2 : 4);
}

void ssr_conv_x3r_instance(const ssr_conv_desc& d, int* ntw, int* nu, int* ep);
int ssr_conv_x3r_tile_height(const ssr_conv_desc& d);

// The kernel symbol ssr_conv2d launches for this descriptor, as rocprofv3 prints it (without the "void (anonymous namespace)::"
// prefix and the argument list): what bench.py files a launch under, so that roofline.kernel can be looked up in
// profiles/*_kernel_stats*.csv, *_pmc_sq.json and traffic_*.json by the same name.  Exact for the kernels that carry the step
// (register-tiled / ring body kernels, big-tile, thin-output); for the older pipelined / resident / weight-stationary families
// the family name with the ssr_conv2d_variant code (their template lists depend on internal tile choices).
extern "C" int ssr_conv2d_symbol(const ssr_conv_desc* dp, char* buf, int32_t buflen) {
    if (!dp || !buf || buflen < 48) return SSR_EINVAL;
    const ssr_conv_desc& d = *dp;
    const int v = ssr_conv2d_variant(dp);
    if (v < 0) return v;
    const int w = v % 10, nt = (v / 10) % 10;
    const bool f32m = d.dtype != SSR_BF16;
    if (d.dtype != SSR_BF16 && w == 5) {
        int ntw = 1, nu = 1, ep = 3;
        ssr_conv_x3r_instance(d, &ntw, &nu, &ep);
        snprintf(buf, buflen, "conv_x3r_kernel<%d, %d, %d, %d, %d>", ntw, nu, ep, ssr_conv_x3r_tile_height(d), d.dtype == SSR_F32 ? 1 : d.dtype == SSR_F32H ? 2 : 0);
    } else if (d.dtype == SSR_F32X3 && w == 6) snprintf(buf, buflen, "conv_x3q_kernel<%d>", nt);
    else if (d.dtype == SSR_F32X3 && w == 9) snprintf(buf, buflen, "conv_bigx3_kernel4<%d>", (d.s2d || d.KH == 2) ? 2 : 3);
    else if (d.dtype == SSR_F32H && w == 9) snprintf(buf, buflen, "conv_bigh3_kernel4<%d>", (d.s2d || d.KH == 2) ? 2 : 3);
    else if (f32m && w == 7) snprintf(buf, buflen, "conv_thin_f32_kernel<%d, %d>", d.Cout == 1 ? 1 : d.Cout <= 3 ? 3 : d.Cout == 4 ? 4 : 8, d.dtype == SSR_F32X3 ? 1 : d.dtype == SSR_F32H ? 2 : 0);
    else snprintf(buf, buflen, "%s<%s,K%d,S%d,NT%d,W%d>", w == 7 ? "conv_thin_kernel" : w == 8 ? "conv_ws_kernel" : w == 9 ? "conv_big_kernel" : w == 1 ? "conv_res_kernel" : d.dtype == SSR_F32X3 ? "conv_x3_kernel" : d.dtype == SSR_F32H ? "conv_h3_kernel" : "conv_kernel",
                  d.dtype == SSR_BF16 ? "bf16" : d.dtype == SSR_F32 ? "fp32" : d.dtype == SSR_F32H ? "fp32h" : "fp32x3", v / 1000, (v / 100) % 10, nt, w);
    return SSR_OK;
}

extern "C" int ssr_conv2d_ck(int32_t dtype, int32_t KH) {
    const int vec = dtype == SSR_BF16 ? 8 : 4;      // SSR_F32X3 packs like SSR_F32
    return (KH == 4 ? 1 : 2) * 2 * vec;
}

static int conv2d_impl(const ssr_conv_desc* dp, void* stream, int impl) {
    if (!dp) return SSR_EINVAL;
    const ssr_conv_desc& d = *dp;
    if (d.act < SSR_ACT_NONE || d.act > SSR_ACT_RELU) return SSR_EINVAL;
    if (!view_ok(d.x, true) || !d.w || ((uintptr_t)d.w % 16) != 0) return SSR_EINVAL;
    if (!d.y.p || d.y.cs <= 0) return SSR_EINVAL;
    if (d.Cin <= 0 || (d.Cin % 8) != 0 || d.CoutPad <= 0 || (d.CoutPad % 32) != 0 || d.Cout > d.CoutPad)
        return SSR_EINVAL;
    if (!(d.up == 1 || d.up == 2) || d.N <= 0 || d.Gh <= 0 || d.Gw <= 0) return SSR_EINVAL;
    if (d.x2.p && (d.Cin2 <= 0 || (d.Cin2 % 8) != 0 || !view_ok(d.x2, true))) return SSR_EINVAL;
    if (!d.x2.p && d.Cin2 != 0) return SSR_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = 0;
    if (d.s2d) {   // 4x4 stride 2 through the space-to-depth view: a 2x2 stride-1 layer of 4*Cin channels on the big-tile kernel
        if (!(d.KH == 4 && d.KW == 4 && d.stride == 2 && d.pad_y == 1 && d.pad_x == 1 && d.up == 1 && !d.x2.p)) return SSR_EINVAL;
        if (!ssr_conv2d_s2d_ok(d.dtype, d.Cin, d.Cout, d.CoutPad) || (d.Hi % 2) || (d.Wi % 2) || d.Gh != d.Hi / 2 || d.Gw != d.Wi / 2)
            return SSR_EINVAL;
        ssr_conv_desc e = d;
        e.KH = e.KW = 2; e.stride = 1; e.pad_y = e.pad_x = 0; e.Cin = 4 * d.Cin;
        if (d.dtype == SSR_F32X3 || d.dtype == SSR_F32H) return ssr_conv_bigx3_try(e, st, &rc, true) ? rc : SSR_EUNSUP;
        return ssr_conv_big_try(e, st, &rc, true) ? rc : SSR_EUNSUP;
    }
    if (d.dtype == SSR_F32H) {    // fp16-split FORWARD arithmetic (include/ssr_hip.h): the split-bf16 mode's kernels with the H flag
        if (d.fix_list) return SSR_EUNSUP;
        if (impl == 4) return ssr_conv_bigx3_try(d, st, &rc, true) ? rc : SSR_EUNSUP;
        if (impl == 5) return ssr_conv_thin_try(d, st, &rc, true) ? rc : SSR_EUNSUP;
        if (impl == 7) return ssr_conv_x3r_try(d, st, &rc, true) ? rc : SSR_EUNSUP;
        if (impl == 0 && ssr_conv_thin_try(d, st, &rc, false)) return rc;
        if (impl == 0 && ssr_conv_bigx3_try(d, st, &rc, false)) return rc;
        if (impl == 0 && ssr_conv_x3r_try(d, st, &rc, false)) return rc;
        if (d.KH == 3 && d.KW == 3 && d.stride == 1) return dispatch_tile_x3<3, 3, true>(d, st);
        if (d.KH == 2 && d.KW == 2 && d.stride == 1) return dispatch_tile_x3<2, 2, true>(d, st);
        ssr_conv_desc e = d;      // 4x4 stride 2 without the space-to-depth view: rows of 8 are plain fp32 (misc.hip put_packed), the exact kernel
        e.dtype = SSR_F32;
        return dispatch_geom<float>(e, st);
    }
    if (d.dtype == SSR_F32X3) {   // fp32 storage, split-bf16 matrix math (stride-1 2x2 / 3x3); 4x4 stride 2 without s2d: the exact fp32 kernel
        if (impl == 4) return ssr_conv_bigx3_try(d, st, &rc, true) ? rc : SSR_EUNSUP;
        if (impl == 5) return ssr_conv_thin_try(d, st, &rc, true) ? rc : SSR_EUNSUP;
        if (impl == 6) return ssr_conv_x3q_try(d, st, &rc, true) ? rc : SSR_EUNSUP;
        if (impl == 7) return ssr_conv_x3r_try(d, st, &rc, true) ? rc : SSR_EUNSUP;
        if (impl == 0 && ssr_conv_thin_try(d, st, &rc, false)) return rc;
        if (impl == 0 && ssr_conv_bigx3_try(d, st, &rc, false)) return rc;
        if (impl == 0 && ssr_conv_x3r_try(d, st, &rc, false)) return rc;
        if (impl == 0 && ssr_conv_x3q_try(d, st, &rc, false)) return rc;
        if (d.KH == 3 && d.KW == 3 && d.stride == 1) return dispatch_tile_x3<3, 3>(d, st);
        if (d.KH == 2 && d.KW == 2 && d.stride == 1) return dispatch_tile_x3<2, 2>(d, st);
        ssr_conv_desc e = d;
        e.dtype = SSR_F32;
        return dispatch_geom<float>(e, st);
    }
    if (d.act == SSR_ACT_RELU || d.m_relu) impl = 3;   // ReLU epilogues exist only in the pipelined kernel (conv_epilogue.h)
    if (impl == 1) return ssr_conv_ws_try(d, st, &rc, true) ? rc : SSR_EUNSUP;
    if (impl == 4) return ssr_conv_big_try(d, st, &rc, true) ? rc : SSR_EUNSUP;
    if (impl == 5) return ssr_conv_thin_try(d, st, &rc, true) ? rc : SSR_EUNSUP;
    if (impl == 0 && ssr_conv_thin_try(d, st, &rc, false)) return rc;
    if (impl == 0 && ssr_conv_ws_try(d, st, &rc, false)) return rc;
    if (impl == 0 && ssr_conv_big_try(d, st, &rc, false)) return rc;
    if (d.dtype == SSR_F32 && impl == 7) return ssr_conv_x3r_try(d, st, &rc, true) ? rc : SSR_EUNSUP;
    if (d.dtype == SSR_F32 && impl == 0 && ssr_conv_x3r_try(d, st, &rc, false)) return rc;     // exact fp32 on the register-tiled body kernel (round 6)
    if (impl != 3 && ssr_conv_res_try(d, st, &rc)) return rc;
    if (d.dtype == SSR_F32) return dispatch_geom<float>(d, st);
    if (d.dtype == SSR_BF16) return dispatch_geom<__bf16>(d, st);
    return SSR_EUNSUP;
}

extern "C" int ssr_conv2d(const ssr_conv_desc* dp, void* stream) { return conv2d_impl(dp, stream, 0); }

// ---- LeakyReLU decision fix-up of the split-bf16 mode (include/ssr_hip.h, ssr_conv_desc.fix_*) ----
// One wave per listed output: the K = Cin * KH * KW products of the reference's convolution (ssr/archs/rrdbnet_arch.py:39-42,
// discriminator_arch.py:45-69: F.conv2d of fp32 tensors) from the fp32 inputs and the unpacked fp32 weights, summed in double
// (lane k takes products k, k + 64, ...; xor-shuffle tree), + bias, LeakyReLU, stored over the split-bf16 value.  The list is
// short (|v| < fix_thr: a few 1e-4 of the outputs), the launch is a handful of microseconds.
namespace {
__global__ __launch_bounds__(256) void conv_fixup_kernel(const ssr_conv_desc d) {
    int* hdr = d.fix_list;
    const int count = min(hdr[0], d.fix_cap);          // every block reads it before the last one to finish resets it
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = gridDim.x * 4;
    const int KK = d.KH * d.KW, K = d.w_ref_cin * KK;
    const float* __restrict__ xp = reinterpret_cast<const float*>(d.x.p);
    float* __restrict__ yp = reinterpret_cast<float*>(d.y.p);
    const double inv_sigma = d.w_ref_sigma ? 1.0 / (double)d.w_ref_sigma[0] : 1.0;
    const int sh = d.up == 2 ? 1 : 0, LH = d.Hi << sh, LW = d.Wi << sh;
    for (int e = blockIdx.x * 4 + wave; e < count; e += nw) {
        const int p = hdr[4 + 2 * e], co = hdr[5 + 2 * e];
        const int ox = p % d.Wo, t = p / d.Wo, oy = t % d.Ho, n = t / d.Ho;
        const float* __restrict__ wrow = d.w_ref + (size_t)co * K;
        double s = 0.0;
        for (int k = lane; k < K; k += 64) {
            const int ci = k / KK, tap = k - ci * KK, ky = tap / d.KW, kx = tap - ky * d.KW;
            const int iy = oy * d.stride + ky - d.pad_y, ix = ox * d.stride + kx - d.pad_x;
            if (iy >= 0 && iy < LH && ix >= 0 && ix < LW) {
                const size_t px = (size_t)(n * d.Hi + (iy >> sh)) * d.Wi + (ix >> sh);
                s += (double)wrow[k] * (double)xp[px * d.x.cs + d.x.coff + ci];
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) {
            const float v = (float)(s * inv_sigma + (d.bias ? (double)d.bias[co] : 0.0));
            yp[(size_t)p * d.y.cs + d.y.coff + co] = lrelu(v);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(hdr + 1, 1) == (int)gridDim.x - 1) {     // the last block: the list is consumed
            atomicMax(hdr + 2, hdr[0]);
            hdr[0] = 0;
            hdr[1] = 0;
        }
    }
}
}  // namespace

extern "C" int ssr_conv2d_fixup(const ssr_conv_desc* dp, void* stream) {
    if (!dp) return SSR_EINVAL;
    const ssr_conv_desc& d = *dp;
    if (!d.fix_list || d.fix_cap <= 0 || !d.w_ref || d.w_ref_cin <= 0 || !d.x.p || !d.y.p) return SSR_EINVAL;
    if ((d.dtype != SSR_F32X3 && d.dtype != SSR_F32) || d.act != SSR_ACT_LRELU || d.x2.p || d.y0.p || d.y1.p || d.r1.p || d.r2.p || d.m.p ||
        d.accumulate || d.alpha != 1.f || d.stride != 1 || d.s2d || d.oys != 1 || d.oxs != 1 || d.oyo != 0 || d.oxo != 0 ||
        d.w_ref_cin > d.Cin)
        return SSR_EUNSUP;
    hipLaunchKernelGGL(conv_fixup_kernel, dim3(32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

// n (<= 4) descriptors that differ only in pointers / padding / output offsets: one launch when they are 2x2 stride-1
// bf16 layers of identical geometry (the parity classes of a 4x4 stride-2 dgrad), otherwise n launches.
extern "C" int ssr_conv2d_batch(const ssr_conv_desc* ds, int32_t n, void* stream) {
    if (!ds || n <= 0 || n > 4) return SSR_EINVAL;
    bool same = ds[0].dtype == SSR_BF16 && ds[0].KH == 2 && ds[0].KW == 2 && ds[0].stride == 1;
    for (int k = 1; k < n && same; ++k)
        same = ds[k].dtype == ds[0].dtype && ds[k].KH == 2 && ds[k].KW == 2 && ds[k].stride == 1 && ds[k].N == ds[0].N &&
               ds[k].Gh == ds[0].Gh && ds[k].Gw == ds[0].Gw && ds[k].CoutPad == ds[0].CoutPad && ds[k].Cin == ds[0].Cin &&
               ds[k].Cin2 == ds[0].Cin2 && ds[k].Hi == ds[0].Hi && ds[k].Wi == ds[0].Wi && ds[k].up == ds[0].up;
    static const bool off = [] { const char* e = getenv("SSR_CONV_BATCH"); return e && e[0] == '0'; }();
    if (!off && ds[0].dtype == SSR_F32X3 && ds[0].KH == 2 && ds[0].KW == 2 && ds[0].stride == 1) {   // split-bf16 mode: the classes in one big-tile launch
        int rcx = 0;
        if (ssr_conv_bigx3_batch_try(ds, n, reinterpret_cast<hipStream_t>(stream), &rcx)) return rcx;
    }
    if (same && !off) {
        for (int k = 0; k < n; ++k) {
            const ssr_conv_desc& d = ds[k];
            if (!view_ok(d.x, true) || !d.w || ((uintptr_t)d.w % 16) != 0 || !d.y.p || d.y.cs <= 0 || d.Cin <= 0 || (d.Cin % 8) != 0 ||
                d.CoutPad <= 0 || (d.CoutPad % 32) != 0 || d.Cout > d.CoutPad || !(d.up == 1 || d.up == 2) || d.N <= 0 || d.Gh <= 0 ||
                d.Gw <= 0 || (d.x2.p && (d.Cin2 <= 0 || (d.Cin2 % 8) != 0 || !view_ok(d.x2, true))) || (!d.x2.p && d.Cin2 != 0))
                return SSR_EINVAL;
        }
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        int rcb = 0;
        if (ssr_conv_big_batch_try(ds, n, st, &rcb)) return rcb;   // 512-pixel tiles (conv_big.hip) when the grid is large enough
        bool nt2, small;
        pick_tile(ds[0], nt2, small);
        // the class launches together fill the chip: use the tile shape the 4x larger grid would get
        const long tiles4 = (long)((ds[0].Gw + 15) / 16) * ((ds[0].Gh + 7) / 8) * ds[0].N * (ds[0].CoutPad / (nt2 ? 64 : 32)) * n;
        small = tiles4 < 384;
        if (nt2) return small ? launch_conv4<__bf16, 2, 2, 1, 2, 2, 2>(ds, n, st) : launch_conv4<__bf16, 2, 2, 1, 2, 4, 2>(ds, n, st);
        return small ? launch_conv4<__bf16, 2, 2, 1, 1, 2, 2>(ds, n, st) : launch_conv4<__bf16, 2, 2, 1, 1, 4, 2>(ds, n, st);
    }
    for (int k = 0; k < n; ++k) {
        const int rc = conv2d_impl(ds + k, stream, 0);
        if (rc != SSR_OK) return rc;
    }
    return SSR_OK;
}
extern "C" int ssr_conv2d_impl(const ssr_conv_desc* dp, void* stream, int32_t impl) {
    return conv2d_impl(dp, stream, impl);
}
