// Weight/bias gradient of the NHWC direct convolution, batched over layers.
//
//   dW[co][ci][ky][kx] += alpha * sum_{n,gy,gx} dY[n,gy,gx,co] * X[n, gy*S+ky-pad, gx*S+kx-pad, ci]
//
// The contraction runs over pixels, which are the *strided* axis of NHWC.  With the exact-fp32 matrix
// core (v_mfma_f32_32x32x2_f32) each lane supplies ONE k-value per operand, so a plain
// [pixel][channel] LDS image feeds it with conflict-free ds_read_b32 and no transpose:
//   A[i = co][k = pixel] = dY_lds[pixel][co],   B[k = pixel][j = ci] = X_lds[pixel + tap][ci].
// A workgroup (4 waves) owns one (layer, 32 co, 32 ci) tile for a range of 8x16 pixel tiles; all
// KH*KW taps are accumulated in registers (3x3: waves split the pixels, 9x16 accumulators each,
// cross-wave reduce through LDS at the end; 4x4: wave w owns tap row ky = w).  A launch processes a
// device table of work items spanning many layers, so the 345 tiny body layers of the generator
// become ONE launch with one writer per dW element (deterministic, no atomics).
// This file is the exact-fp32 path; bf16 uses the transpose-read bf16-MFMA kernel in wgrad_bf16.hip.
//
// Replaces autograd's convolution_backward (weight, bias) for every Conv2d under
// /root/reference/ssr/models/ssr_esrgan_model.py:192,221,227.
#include "wgrad_common.h"

namespace {

constexpr int WG_ROW = 36;  // LDS row stride in floats (32 + 4 pad, 16B aligned)

template <typename T> __device__ __forceinline__ void store_widened(float* dst, const u32x4& v);
template <> __device__ __forceinline__ void store_widened<float>(float* dst, const u32x4& v) {
    *reinterpret_cast<u32x4*>(dst) = v;
}
template <> __device__ __forceinline__ void store_widened<__bf16>(float* dst, const u32x4& v) {
    const bf16x8 h = __builtin_bit_cast(bf16x8, v);
    f32x4 lo, hi;
#pragma unroll
    for (int k = 0; k < 4; ++k) { lo[k] = (float)h[k]; hi[k] = (float)h[4 + k]; }
    *reinterpret_cast<f32x4*>(dst) = lo;
    *reinterpret_cast<f32x4*>(dst + 4) = hi;
}

template <typename T, int KH, int KW, int S, bool SPLIT_TAPS>
__global__ __launch_bounds__(256) void wgrad_kernel(const ssr_wgrad_layer* __restrict__ layers,
                                                    const ssr_wgrad_item* __restrict__ items) {
    constexpr int VEC = DT<T>::VEC, VPP = 32 / VEC;  // 16-byte vectors per pixel (32 channels)
    constexpr int PH = (WG_TH - 1) * S + KH, PW = (WG_TW - 1) * S + KW;
    constexpr int NTAP = SPLIT_TAPS ? KW : KH * KW;   // taps accumulated per wave
    static_assert(!SPLIT_TAPS || KH == 4, "tap-row split assumes 4 waves = KH");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ldy = reinterpret_cast<float*>(smem);       // [128][WG_ROW]
    float* lx = ldy + WG_TH * WG_TW * WG_ROW;          // [PH*PW][WG_ROW]

    const ssr_wgrad_item it = items[blockIdx.x];
    const ssr_wgrad_layer L = layers[it.layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int tiles_x = (L.Gw + WG_TW - 1) / WG_TW, tiles_y = (L.Gh + WG_TH - 1) / WG_TH;
    const int upshift = L.up == 2 ? 1 : 0;
    const int LH = L.Hi << upshift, LW = L.Wi << upshift;
    const T* __restrict__ xg = reinterpret_cast<const T*>(L.x.p);
    const T* __restrict__ dyg = reinterpret_cast<const T*>(L.dy.p);
    const bool do_bias = L.db != nullptr && it.ci0 == 0 && (!SPLIT_TAPS || wave == 0);

    f32x16 acc[NTAP], accb;
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;

    for (int tile = it.tile_begin; tile < it.tile_end; ++tile) {
        int b = tile;
        const int tx_i = b % tiles_x; b /= tiles_x;
        const int ty_i = b % tiles_y;
        const int n = b / tiles_y;
        const int gy0 = ty_i * WG_TH, gx0 = tx_i * WG_TW;
        // ---- stage dY tile [128 px][32 co] ----
        for (int v = tid; v < WG_TH * WG_TW * VPP; v += 256) {
            const int pix = v / VPP, part = v - pix * VPP;
            const int gy = gy0 + (pix >> 4), gx = gx0 + (pix & 15);
            const int c = it.co0 + part * VEC;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (gy < L.Gh && gx < L.Gw && c < L.Cout) {
                const size_t off = ((size_t)(n * L.Gh + gy) * L.Gw + gx) * L.dy.cs + L.dy.coff + c;
                val = *reinterpret_cast<const u32x4*>(dyg + off);
            }
            store_widened<T>(ldy + pix * WG_ROW + part * VEC, val);
        }
        // ---- stage X halo patch [PH*PW px][32 ci] ----
        for (int v = tid; v < PH * PW * VPP; v += 256) {
            const int pix = v / VPP, part = v - pix * VPP;
            const int py = pix / PW, px = pix - py * PW;
            const int ly = gy0 * S + py - L.pad_y, lxx = gx0 * S + px - L.pad_x;
            const int c = it.ci0 + part * VEC;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (ly >= 0 && ly < LH && lxx >= 0 && lxx < LW && c < L.Cin) {
                const int sy = ly >> upshift, sx = lxx >> upshift;
                const size_t off = ((size_t)(n * L.Hi + sy) * L.Wi + sx) * L.x.cs + L.x.coff + c;
                val = *reinterpret_cast<const u32x4*>(xg + off);
            }
            store_widened<T>(lx + pix * WG_ROW + part * VEC, val);
        }
        __syncthreads();
        constexpr int NPAIR = SPLIT_TAPS ? 64 : 16;  // pixel pairs per wave per tile
#pragma unroll 4
        for (int kp = 0; kp < NPAIR; ++kp) {
            const int pix = 2 * kp + g;
            const int ty = SPLIT_TAPS ? (pix >> 4) : (2 * wave + (pix >> 4)), tx = pix & 15;
            const float a = ldy[(ty * WG_TW + tx) * WG_ROW + i];
            if (do_bias) accb = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 1.0f, accb, 0, 0, 0);
            const float* xb = lx + ((ty * S) * PW + tx * S) * WG_ROW + i;
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int ky = SPLIT_TAPS ? wave : t / KW, kx = SPLIT_TAPS ? t : t % KW;
                const float bv = xb[(ky * PW + kx) * WG_ROW];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    wgrad_writeout<KH, KW, SPLIT_TAPS, NTAP>(acc, accb, L, it, lx, do_bias);
}

template <typename T, int KH, int KW, int S, bool SPLIT>
int launch_wgrad(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, hipStream_t st) {
    constexpr int PH = (WG_TH - 1) * S + KH, PW = (WG_TW - 1) * S + KW;
    constexpr size_t lds = (size_t)(WG_TH * WG_TW + PH * PW) * WG_ROW * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert((size_t)PH * PW * WG_ROW >= 4 * 16 * 64, "reduction scratch fits in the patch region");
    auto kern = wgrad_kernel<T, KH, KW, S, SPLIT>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_items), dim3(256), lds, st, layers, items);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

template <typename T>
int dispatch_wgrad(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, int KH, int KW, int S,
                   hipStream_t st) {
    if (KH == 3 && KW == 3 && S == 1) return launch_wgrad<T, 3, 3, 1, false>(layers, items, n_items, st);
    if (KH == 4 && KW == 4 && S == 2) return launch_wgrad<T, 4, 4, 2, true>(layers, items, n_items, st);
    return SSR_EUNSUP;
}

}  // namespace

// bf16: transpose-read MFMA kernel (wgrad_bf16.hip)
int ssr_wgrad_bf16_dispatch(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, int KH, int KW,
                            int S, hipStream_t st);
// fp32x3, 3x3 stride 1: fp32 tiles split on the way into LDS, three products per fragment pair (wgrad_x3.hip)
int ssr_wgrad_x3_dispatch(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, int KH, int KW, int S,
                          hipStream_t st);

extern "C" int32_t ssr_wgrad_tiles(int32_t N, int32_t Gh, int32_t Gw, int32_t dtype, int32_t KH) {
    const int th = dtype == SSR_BF16 ? wgrad_bf16_th(KH) : (dtype == SSR_F32X3 && KH == 4) ? 4 : WG_TH;      // (the one-pass 4x4 kernel: 4 x 16-pixel tiles, wgrad_x3.hip)
    return N * ((Gh + th - 1) / th) * ((Gw + WG_TW - 1) / WG_TW);
}

// width of the input-channel tile of one work item (host helper for building items)
extern "C" int32_t ssr_wgrad_ci_tile(int32_t dtype, int32_t KH) { return ((dtype == SSR_BF16 || dtype == SSR_F32X3) && KH == 3) ? 64 : 32; }
// 64: the kernel takes PAIRED items (two 32-channel blocks of dY against one input patch): the 3x3 kernels, and since round 6 the one-pass 4x4 stride-2 kernel of the split mode
extern "C" int32_t ssr_wgrad_co_tile(int32_t dtype, int32_t KH) { return (((dtype == SSR_BF16 || dtype == SSR_F32X3) && KH == 3) || (dtype == SSR_F32X3 && KH == 4)) ? 64 : 32; }

extern "C" int ssr_conv2d_wgrad(const ssr_wgrad_layer* layers_dev, const ssr_wgrad_item* items_dev, int32_t n_items,
                                int32_t dtype, int32_t KH, int32_t KW, int32_t stride, void* stream) {
    if (!layers_dev || !items_dev || n_items <= 0) return SSR_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SSR_F32) return dispatch_wgrad<float>(layers_dev, items_dev, n_items, KH, KW, stride, st);
    if (dtype == SSR_BF16) return ssr_wgrad_bf16_dispatch(layers_dev, items_dev, n_items, KH, KW, stride, st);
    if (dtype == SSR_F32X3) return ssr_wgrad_x3_dispatch(layers_dev, items_dev, n_items, KH, KW, stride, st);   // fp32 buffers
    return SSR_EUNSUP;
}
