// Elementwise pieces of the VGG19 perceptual loss (BasicSR PerceptualLoss as configured by
// /root/reference/ssr/options/esrgan_s2naip_urban.yml:123-137 and called at /root/reference/ssr/models/ssr_esrgan_model.py:153-160):
// the 3x3 convolutions run on the conv kernels (SSR_ACT_RELU epilogue / ReLU' mask); what is left is HBM-bound NHWC glue.
//
//   ssr_channel_affine      y[p,c] (+)= x[p,c]*scale[c] + shift[c]   (input normalisation (x-mean)/std and its adjoint)
//   ssr_relu_maxpool2_fwd   P = maxpool2x2(relu(F)): F is a feature taken BEFORE the ReLU ('conv1_2' ... 'conv4_4'), so the
//                           ReLU and the pooling that follow it in VGG19 are one pass over F
//   ssr_relu_maxpool2_bwd   gF (+)= route(gP): the gradient goes to the first maximum of each 2x2 window (torch's max_pool2d
//                           tie rule: first in row-major order) if that F > 0, else nowhere
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void channel_affine_kernel(ssr_view x, ssr_view y, long npix, int C, ssr_vec8 scale, ssr_vec8 shift,
                                                             int accumulate) {
    const T* __restrict__ xp = reinterpret_cast<const T*>(x.p);
    T* __restrict__ yp = reinterpret_cast<T*>(y.p);
    const long total = npix * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long p = e / C;
        const int c = (int)(e - p * C);
        float v = to_f32(xp[p * x.cs + x.coff + c]) * scale.v[c] + shift.v[c];
        if (accumulate) v += to_f32(yp[p * y.cs + y.coff + c]);
        yp[p * y.cs + y.coff + c] = from_f32<T>(v);
    }
}

// one thread: one pooled pixel x VEC channels (16 bytes)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void relu_maxpool2_kernel(ssr_view f, ssr_view p, ssr_view g, int N, int H, int W, int C, int accumulate) {
    constexpr int VEC = DT<T>::VEC;
    const int Hp = H / 2, Wp = W / 2, CV = C / VEC;
    const long total = (long)N * Hp * Wp * CV;
    const T* __restrict__ fp = reinterpret_cast<const T*>(f.p);
    T* __restrict__ pp = reinterpret_cast<T*>(p.p);          // fwd: pooled output; bwd: pooled gradient (read)
    T* __restrict__ gp = reinterpret_cast<T*>(g.p);          // bwd: gradient w.r.t. F
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(e % CV);
        long q = e / CV;
        const int x = (int)(q % Wp); q /= Wp;
        const int y = (int)(q % Hp);
        const int n = (int)(q / Hp);
        const long p00 = ((long)n * H + 2 * y) * W + 2 * x;
        const long pos[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
        vecT v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const vecT*>(fp + pos[k] * f.cs + f.coff + cv * VEC);
        const long po = ((long)n * Hp + y) * Wp + x;
        if (!BWD) {
            vecT o;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                float m = 0.f;                                 // relu: max with 0
#pragma unroll
                for (int k = 0; k < 4; ++k) m = fmaxf(m, to_f32(v[k][c]));
                o[c] = from_f32<T>(m);
            }
            *reinterpret_cast<vecT*>(pp + po * p.cs + p.coff + cv * VEC) = o;
        } else {
            const vecT gin = *reinterpret_cast<const vecT*>(pp + po * p.cs + p.coff + cv * VEC);
            vecT o[4];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                int arg = 0;
                float m = fmaxf(to_f32(v[0][c]), 0.f);
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const float r = fmaxf(to_f32(v[k][c]), 0.f);
                    if (r > m) { m = r; arg = k; }             // strictly greater: the first maximum wins
                }
                const bool pass = to_f32(v[arg][c]) > 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k][c] = from_f32<T>((k == arg && pass) ? to_f32(gin[c]) : 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                T* dst = gp + pos[k] * g.cs + g.coff + cv * VEC;
                if (accumulate) {
                    const vecT old = *reinterpret_cast<const vecT*>(dst);
#pragma unroll
                    for (int c = 0; c < VEC; ++c) o[k][c] = from_f32<T>(to_f32(o[k][c]) + to_f32(old[c]));
                }
                *reinterpret_cast<vecT*>(dst) = o[k];
            }
        }
    }
}

inline int grid_of(long total) {
    long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int ssr_channel_affine(ssr_view x, ssr_view y, int32_t dtype, int64_t npix, int32_t C, const float* scale,
                                  const float* shift, int32_t accumulate, void* stream) {
    if (dtype == SSR_F32X3) dtype = SSR_F32;
    if (!x.p || !y.p || !scale || !shift || npix <= 0 || C <= 0 || C > 8) return SSR_EINVAL;
    ssr_vec8 sc, sh;
    for (int c = 0; c < 8; ++c) { sc.v[c] = c < C ? scale[c] : 0.f; sh.v[c] = c < C ? shift[c] : 0.f; }
    if (dtype == SSR_F32)
        hipLaunchKernelGGL(channel_affine_kernel<float>, dim3(grid_of(npix * C)), dim3(256), 0, ST(stream), x, y, (long)npix, C, sc, sh, accumulate);
    else if (dtype == SSR_BF16)
        hipLaunchKernelGGL(channel_affine_kernel<__bf16>, dim3(grid_of(npix * C)), dim3(256), 0, ST(stream), x, y, (long)npix, C, sc, sh, accumulate);
    else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

static int relu_maxpool2(ssr_view f, ssr_view p, ssr_view g, int32_t dtype, int N, int H, int W, int C, int accumulate, bool bwd, void* stream) {
    if (dtype == SSR_F32X3) dtype = SSR_F32;
    const int vec = dtype == SSR_F32 ? 4 : 8;
    if (!f.p || !p.p || (bwd && !g.p) || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C % vec)) return SSR_EINVAL;
    if ((f.cs % vec) || (f.coff % vec) || (p.cs % vec) || (p.coff % vec) || (bwd && ((g.cs % vec) || (g.coff % vec)))) return SSR_EINVAL;
    const long total = (long)N * (H / 2) * (W / 2) * (C / vec);
    if (dtype == SSR_F32) {
        if (bwd) hipLaunchKernelGGL((relu_maxpool2_kernel<float, true>), dim3(grid_of(total)), dim3(256), 0, ST(stream), f, p, g, N, H, W, C, accumulate);
        else hipLaunchKernelGGL((relu_maxpool2_kernel<float, false>), dim3(grid_of(total)), dim3(256), 0, ST(stream), f, p, g, N, H, W, C, 0);
    } else if (dtype == SSR_BF16) {
        if (bwd) hipLaunchKernelGGL((relu_maxpool2_kernel<__bf16, true>), dim3(grid_of(total)), dim3(256), 0, ST(stream), f, p, g, N, H, W, C, accumulate);
        else hipLaunchKernelGGL((relu_maxpool2_kernel<__bf16, false>), dim3(grid_of(total)), dim3(256), 0, ST(stream), f, p, g, N, H, W, C, 0);
    } else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_relu_maxpool2_fwd(ssr_view f, ssr_view p, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    ssr_view none = {nullptr, 0, 0};
    return relu_maxpool2(f, p, none, dtype, N, H, W, C, 0, false, stream);
}

extern "C" int ssr_relu_maxpool2_bwd(ssr_view f, ssr_view gp, ssr_view gf, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C,
                                     int32_t accumulate, void* stream) {
    return relu_maxpool2(f, gp, gf, dtype, N, H, W, C, accumulate, true, stream);
}
