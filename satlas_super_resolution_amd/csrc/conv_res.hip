// "K-resident" 3x3 stride-1 NHWC convolution for the small-spatial generator body.
//
// The RRDB body convolutions are tiny GEMMs (M = B*32*32 pixels, N = 32/64, K = 9*64..9*192): at the
// measured configuration each CU owns only 64 pixels, one wave runs 18..54 MFMAs, and the launch is
// bound by the latency of getting the halo patch and (mostly) the weight slab into the CU — every
// chunk-by-chunk pipeline exposes that latency once per chunk (rocprofv3: 10..35 us per launch).
// This kernel makes the WHOLE reduction resident in LDS (up to 160 KiB per CU) with LDS-DMA:
//   * every wave issues all its global_load_lds_dwordx4 (16 B/lane, 1 KiB per wave-instruction, no VGPR
//     staging) for all input-channel chunks back-to-back, then a single vmcnt(0) + barrier: the memory
//     latency is paid ONCE per launch;
//   * LDS-DMA writes lane-linearly, so rows are dense 64 B (32 bf16 / 16 fp32 channels) and the bank
//     swizzle is applied on the SOURCE side: physical 16-B part = logical part ^ ((row >> 2) & 3); the
//     ds_read_b128 of 16 consecutive rows then covers all 64 banks exactly once;
//   * out-of-image halo rows are zero-filled by ds_write from the lanes that skip the DMA;
//   * tile = 4x16 pixels x 32 output channels, 4 waves = 2 pixel groups x 2 k-halves (each wave takes
//     one of the two 16-byte k-substeps of every chunk), partial sums combined through LDS.
// A second input view (x2) lets one launch contract over the channel concatenation [x | x2]: that is the
// "gather" form of the dense-block backward (d x_k = sum over all later convs), the mirror image of the
// concat-free forward.  Same descriptor, epilogue and packed-weight layout as conv.hip.
//
// Replaces the same reference ops as conv.hip for /root/reference/ssr/archs/rrdbnet_arch.py:37-44,63-68
// (ResidualDenseBlock / RRDB forward) and their autograd backward.
#include "conv_epilogue.h"

#ifdef SSR_PROBE   // tools/conv_probe.hip: per-phase s_memtime stamps (lane 0 of wave 0 of every workgroup)
#define PROBE(k)                                                                                       \
    do {                                                                                               \
        if (threadIdx.x == 0) g_probe[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PROBE(k)
#endif

namespace {

constexpr int RES_TH = 4, RES_TW = 16, RES_PH = RES_TH + 2, RES_PW = RES_TW + 2;
constexpr int RES_PROWS = RES_PH * RES_PW;                 // 108 halo pixels
constexpr int RES_PROWS_PAD = (RES_PROWS + 15) / 16 * 16;  // 112
constexpr int RES_ROWB = 64;                               // bytes per LDS row

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_res_kernel(const ssr_conv_desc d) {
    constexpr int VEC = DT<T>::VEC, CK = 4 * VEC, BN = 32 * NT;
    constexpr int WROWS = 9 * BN;                           // multiple of 16
    constexpr int ROWS = RES_PROWS_PAD + WROWS;
    constexpr int NINS = ROWS / 16;                         // wave-instructions (1 KiB) per chunk
    constexpr int STAGEB = ROWS * RES_ROWB;                 // bytes per chunk image
    constexpr int NJ = (NINS + 3) / 4;                      // instructions per wave per chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, kh = wave >> 1;
    const int tiles_x = (d.Gw + RES_TW - 1) / RES_TW, tiles_y = (d.Gh + RES_TH - 1) / RES_TH;
    int b = blockIdx.x;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty_i * RES_TH, gx0 = tx_i * RES_TW;
    const int co0 = blockIdx.y * BN;

    const T* __restrict__ x1 = reinterpret_cast<const T*>(d.x.p);
    const T* __restrict__ x2 = reinterpret_cast<const T*>(d.x2.p);
    const T* __restrict__ wg = reinterpret_cast<const T*>(d.w);
    const int K = d.Cin + d.Cin2;
    const int nchunks = (K + CK - 1) / CK;
    const size_t wchunk = (size_t)9 * d.CoutPad * CK;

    PROBE(0);
    // ---- issue every load of the launch: per wave, instruction j covers LDS rows 16j..16j+15 ----
    {
        const int lrow = lane >> 2, pp = lane & 3;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = wave + 4 * jj;
            if (j >= NINS) break;
            const int row = 16 * j + lrow;
            const int lp = pp ^ ((row >> 2) & 3);           // logical 16-B part stored at physical part pp
            if (row < RES_PROWS_PAD) {
                const int py = row / RES_PW, px = row - py * RES_PW;
                const int iy = gy0 + py - 1, ix = gx0 + px - 1;
                const bool inside = row < RES_PROWS && iy >= 0 && iy < d.Hi && ix >= 0 && ix < d.Wi;
                const size_t pix = (size_t)(n * d.Hi + iy) * d.Wi + ix;
                for (int c = 0; c < nchunks; ++c) {
                    const int k0 = c * CK + lp * VEC;
                    char* dst = smem + (size_t)c * STAGEB + j * 1024;
                    const T* src = nullptr;
                    if (inside) {
                        if (k0 < d.Cin) src = x1 + pix * d.x.cs + d.x.coff + k0;
                        else if (k0 < K) src = x2 + pix * d.x2.cs + d.x2.coff + (k0 - d.Cin);
                    }
                    if (src)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                    else
                        *reinterpret_cast<u32x4*>(dst + lane * 16) = u32x4{0u, 0u, 0u, 0u};
                }
            } else {
                const int wrow = row - RES_PROWS_PAD;
                const int tap = wrow / BN, co = wrow - tap * BN;
                const T* src = wg + ((size_t)tap * d.CoutPad + co0 + co) * CK + lp * VEC;
                for (int c = 0; c < nchunks; ++c) {
                    char* dst = smem + (size_t)c * STAGEB + j * 1024;
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(src + (size_t)c * wchunk),
                        (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                }
            }
        }
    }
    PROBE(1);
    __syncthreads();   // hipcc drains vmcnt(0) in front of the barrier: all DMA has landed
    PROBE(2);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int i = lane & 31, g = lane >> 5;
    const int ty = 2 * wm + (i >> 4), tx = i & 15;
    const int lpart = 2 * kh + g;                           // logical 16-B part of this lane's k-slice
    const int ra = ty * RES_PW + tx;
    // weight rows: (RES_PROWS_PAD + tap*BN + t*32 + i) >> 2 == (i >> 2) (mod 4): swizzle is tap-invariant
    const int b_off = (RES_PROWS_PAD + i) * RES_ROWB + ((lpart ^ ((i >> 2) & 3)) << 4);
    for (int c = 0; c < nchunks; ++c) {
        const char* base = smem + (size_t)c * STAGEB;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int row = ra + ky * RES_PW + kx;
                const u32x4 a =
                    *reinterpret_cast<const u32x4*>(base + row * RES_ROWB + ((lpart ^ ((row >> 2) & 3)) << 4));
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const u32x4 bv = *reinterpret_cast<const u32x4*>(
                        base + b_off + ((ky * 3 + kx) * BN + t * 32) * RES_ROWB);
                    mma16<T>(acc[t], a, bv);
                }
            }
    }
    PROBE(3);
    __syncthreads();   // every wave is done reading: LDS becomes the reduction scratch

    // ---- combine the two k-halves, then the fused epilogue (same contract as conv.hip) ----
    char* slab = smem + 2 * 16 * 64 * sizeof(float) + (size_t)wave * EPI_STAGE_BYTES;   // behind the reduce scratch
    auto epilogue = [&](const f32x16& a, int t) {
        conv_epilogue<T>(d, a, co0 + t * 32, n, gy0 + 2 * wm, gx0, lane, slab);
    };
    float* mine = reinterpret_cast<float*>(smem) + (wm * 16) * 64 + lane;   // [2][16][64]
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
    PROBE(4);
#ifdef SSR_PROBE
    __syncthreads();
    PROBE(5);
#endif
}

template <typename T, int NT>
int launch_res(const ssr_conv_desc& d, int nchunks, hipStream_t st) {
    constexpr int BN = 32 * NT, ROWS = RES_PROWS_PAD + 9 * BN;
    const size_t lds = (size_t)nchunks * ROWS * RES_ROWB;
    auto kern = conv_res_kernel<T, NT>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    const int tiles = ((d.Gw + RES_TW - 1) / RES_TW) * ((d.Gh + RES_TH - 1) / RES_TH) * d.N;
    hipLaunchKernelGGL(kern, dim3(tiles, d.CoutPad / BN, 1), dim3(256), lds < 24576 ? 24576 : lds, st, d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

}  // namespace

// Returns true and launches if the descriptor qualifies for the K-resident kernel:
// 3x3 stride 1, no upsampling, identity output mapping, small grid, whole reduction fits in LDS.
bool ssr_conv_res_qualifies(const ssr_conv_desc& d) {
    if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.up == 1 && d.pad_y == 1 && d.pad_x == 1)) return false;
    if (d.Gh != d.Hi || d.Gw != d.Wi) return false;
    if (d.dtype != SSR_F32 && d.dtype != SSR_BF16) return false;
    const int ck = d.dtype == SSR_F32 ? 16 : 32;
    const int nchunks = (d.Cin + d.Cin2 + ck - 1) / ck;
    const long tiles = (long)((d.Gw + RES_TW - 1) / RES_TW) * ((d.Gh + RES_TH - 1) / RES_TH) * d.N;
    if (tiles * (d.CoutPad / 32) > 2048) return false;                   // large grids: pipelined kernel
    return (size_t)nchunks * (RES_PROWS_PAD + 9 * 32) * RES_ROWB <= 160 * 1024;
}

bool ssr_conv_res_try(const ssr_conv_desc& d, hipStream_t st, int* rc) {
    if (!ssr_conv_res_qualifies(d)) return false;
    const int ck = d.dtype == SSR_F32 ? 16 : 32;
    const int nchunks = (d.Cin + d.Cin2 + ck - 1) / ck;
    if (d.dtype == SSR_F32) *rc = launch_res<float, 1>(d, nchunks, st);
    else *rc = launch_res<__bf16, 1>(d, nchunks, st);
    return true;
}
