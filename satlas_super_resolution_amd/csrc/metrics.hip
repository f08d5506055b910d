// Validation metrics and image quantisation on the device (byte / integer work, HBM-bound, one pass over the images).
//
//   ssr_quantize_u8         float NCHW -> uint8 NHWC: BasicSR tensor2img (clamp, *255, round-half-even; ROUND) as used by
//                           /root/reference/ssr/models/ssr_esrgan_model.py:302-305, or the truncating astype(uint8) of
//                           /root/reference/ssr/infer_grid.py:60-64 and infer.py:58-60 (TRUNC)
//   ssr_metric_shift_sums   exact integer sums  S1 = sum d, S2 = sum d^2  of d = a[.. + (ro, co)] - b[.. + (m-ro, m-co)]
//                           for all (m+1)^2 offset pairs and every channel: PSNR (m = 0) and the 81 brightness-bias-corrected
//                           shifted MSEs of cPSNR (/root/reference/ssr/metrics/cpsnr.py:36-55) come out of ONE launch
//   ssr_metric_ssim_sums    sum of the SSIM map per channel (11x11 Gaussian window sigma 1.5, 'valid' region, fp64), as
//                           basicsr.metrics.calculate_ssim computes it (esrgan_s2naip_urban.yml:159-162)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void quantize_u8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, int N, int C,
                                                          int H, int W, int mode) {
    const long total = (long)N * H * W * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long q = e / C;
        const int x = (int)(q % W); q /= W;
        const int y = (int)(q % H);
        const int n = (int)(q / H);
        float v = src[(((long)n * C + c) * H + y) * W + x];
        v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;            // clamp(0,1) then * 255 in fp32 (NaN -> 0 via fmaxf)
        dst[e] = (uint8_t)(mode == 0 ? rintf(v) : v);       // np.round = half-to-even | astype(uint8) = truncation
    }
}

__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int lo = __shfl_down((int)(v & 0xffffffffll), o), hi = __shfl_down((int)(v >> 32), o);
        v += ((long long)hi << 32) | (unsigned)lo;
    }
    return v;
}

// grid: (offset pair, row block).  a, b: uint8 [H][W][C] (C <= 4)
__global__ __launch_bounds__(256) void shift_sums_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int H, int W,
                                                         int C, int crop, int m, unsigned long long* __restrict__ out) {
    const int no = m + 1, ro = blockIdx.x / no, co = blockIdx.x % no;
    const int hc = H - 2 * crop - m, wc = W - 2 * crop - m;      // window after border crop and offset crop
    long long s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const long total = (long)hc * wc;
    for (long e = (long)blockIdx.y * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.y * blockDim.x) {
        const int y = (int)(e / wc), x = (int)(e % wc);
        const uint8_t* pa = a + ((long)(y + crop + ro) * W + (x + crop + co)) * C;
        const uint8_t* pb = b + ((long)(y + crop + m - ro) * W + (x + crop + m - co)) * C;
        for (int c = 0; c < C; ++c) {
            const int d = (int)pa[c] - (int)pb[c];
            s1[c] += d;
            s2[c] += d * d;
        }
    }
    __shared__ long long red[4][2 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = 0; c < C; ++c) {
        const long long t1 = wave_sum(s1[c]), t2 = wave_sum(s2[c]);
        if (lane == 0) { red[wave][2 * c] = t1; red[wave][2 * c + 1] = t2; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        long long t = 0;
        for (int w = 0; w < 4; ++w) t += red[w][threadIdx.x];
        // two's complement: unsigned atomic add accumulates signed sums correctly
        atomicAdd(out + (size_t)blockIdx.x * 2 * C + threadIdx.x, (unsigned long long)t);
    }
}

__constant__ double SSIM_G[11];

__global__ __launch_bounds__(256) void ssim_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int H, int W, int C,
                                                   int crop, double* __restrict__ out) {
    const int hc = H - 2 * crop - 10, wc = W - 2 * crop - 10;    // 'valid' outputs of the 11x11 window
    const int c = blockIdx.y;
    const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
    double acc = 0.0;
    const long total = (long)hc * wc;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int y = (int)(e / wc), x = (int)(e % wc);
        double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
        for (int i = 0; i < 11; ++i) {
            double r1 = 0, r2 = 0, r11 = 0, r22 = 0, r12 = 0;
            const long row = (long)(y + crop + i) * W + (x + crop);
            for (int j = 0; j < 11; ++j) {
                const double p = a[(row + j) * C + c], q = b[(row + j) * C + c], g = SSIM_G[j];
                r1 += g * p; r2 += g * q; r11 += g * p * p; r22 += g * q * q; r12 += g * p * q;
            }
            const double g = SSIM_G[i];
            m1 += g * r1; m2 += g * r2; s11 += g * r11; s22 += g * r22; s12 += g * r12;
        }
        const double v1 = s11 - m1 * m1, v2 = s22 - m2 * m2, cov = s12 - m1 * m2;
        acc += ((2 * m1 * m2 + C1) * (2 * cov + C2)) / ((m1 * m1 + m2 * m2 + C1) * (v1 + v2 + C2));
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out + c, red[0]);
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int ssr_quantize_u8(const float* src, uint8_t* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t mode,
                               void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return SSR_EINVAL;
    const long total = (long)N * C * H * W;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(quantize_u8_kernel, dim3((int)g), dim3(256), 0, ST(stream), src, dst, N, C, H, W, mode);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_metric_shift_sums(const uint8_t* a, const uint8_t* b, int32_t H, int32_t W, int32_t C, int32_t crop,
                                     int32_t max_offset, int64_t* out, void* stream) {
    if (!a || !b || !out || C <= 0 || C > 4 || crop < 0 || max_offset < 0 || max_offset > 15) return SSR_EINVAL;
    if (H - 2 * crop - max_offset <= 0 || W - 2 * crop - max_offset <= 0) return SSR_EINVAL;
    const int no = (max_offset + 1) * (max_offset + 1);
    hipError_t e = hipMemsetAsync(out, 0, sizeof(int64_t) * no * 2 * C, ST(stream));
    if (e != hipSuccess) return (int)e;
    const long total = (long)(H - 2 * crop - max_offset) * (W - 2 * crop - max_offset);
    long gy = (total + 256 * 8 - 1) / (256 * 8);
    if (gy < 1) gy = 1;
    if (gy > 256) gy = 256;
    hipLaunchKernelGGL(shift_sums_kernel, dim3(no, (int)gy), dim3(256), 0, ST(stream), a, b, H, W, C, crop, max_offset,
                       reinterpret_cast<unsigned long long*>(out));
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_metric_ssim_sums(const uint8_t* a, const uint8_t* b, int32_t H, int32_t W, int32_t C, int32_t crop,
                                    double* out, void* stream) {
    if (!a || !b || !out || C <= 0 || crop < 0) return SSR_EINVAL;
    if (H - 2 * crop - 10 <= 0 || W - 2 * crop - 10 <= 0) return SSR_EINVAL;
    static bool init = false;
    if (!init) {   // cv2.getGaussianKernel(11, 1.5): exp(-(i - 5)^2 / (2 sigma^2)), normalised to sum 1, in double
        double g[11], s = 0;
        for (int i = 0; i < 11; ++i) { g[i] = exp(-((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += g[i]; }
        for (int i = 0; i < 11; ++i) g[i] /= s;
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(SSIM_G), g, sizeof(g));
        if (e != hipSuccess) return (int)e;
        init = true;
    }
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double) * C, ST(stream));
    if (e != hipSuccess) return (int)e;
    const long total = (long)(H - 2 * crop - 10) * (W - 2 * crop - 10);
    long gx = (total + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(ssim_kernel, dim3((int)gx, C), dim3(256), 0, ST(stream), a, b, H, W, C, crop, out);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}
