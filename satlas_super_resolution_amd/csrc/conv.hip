// Direct (im2col-free) NHWC convolution on the gfx950 matrix cores.
//
// One workgroup = WAVES wavefronts computes an output tile of TH x 16 grid positions (TH = 2*WAVES)
// times BN = 32*NT output channels.  Per input-channel chunk (CK = 16 fp32 / 32 bf16 channels) the
// workgroup stages the input halo patch and the [tap][co][ci] weight slab in LDS once, then every
// tap reads its shifted window straight out of the patch: no im2col buffer ever exists.
// Each wave owns 32 grid positions (2 rows x 16) and NT 32x32 MFMA accumulators; a single 16-byte
// LDS read per operand feeds 4 v_mfma_f32_32x32x2_f32 (fp32, exact) or 1 v_mfma_f32_32x32x16_bf16.
// LDS rows are 80 bytes (64 data + 16 pad): conflict-free for the 16-lane ds_read_b128 groups.
//
// Replaces: every nn.Conv2d forward on the ESRGAN path and (with flipped weights) its dgrad —
// /root/reference/ssr/archs/rrdbnet_arch.py:26-30,99-112,123-136 and discriminator_arch.py:28-40,44-69 —
// with torch.cat (rrdbnet_arch.py:39-42), LeakyReLU, the 0.2-residuals (:44,:68), the trunk add (:125),
// nearest x2 upsampling (:127-128), the U-Net skip adds (discriminator_arch.py:53-64) and the
// corresponding backward masks / gradient fan-in sums folded into the load and the epilogue.
#include "common.h"

namespace {

template <typename T, int KH, int KW, int S, int NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void conv_kernel(const ssr_conv_desc d) {
    constexpr int VEC = DT<T>::VEC, CK = 4 * VEC, CKP = CK + VEC, BN = 32 * NT;
    constexpr int TH = 2 * WAVES, TW = 16;
    constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
    constexpr int NTHR = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lp = reinterpret_cast<T*>(smem);  // patch   [PH*PW][CKP]
    T* lw = lp + PH * PW * CKP;          // weights [KH*KW*BN][CKP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    const int CinPad = (d.Cin + CK - 1) / CK * CK;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int i = lane & 31, g = lane >> 5;
    const int ty = 2 * wave + (i >> 4), tx = i & 15;
    const T* abase = lp + ((ty * S) * PW + tx * S) * CKP + g * VEC;
    const T* bbase = lw + i * CKP + g * VEC;

    for (int c0 = 0; c0 < CinPad; c0 += CK) {
        // ---- stage the halo patch (zero outside the image / beyond Cin) ----
        for (int v = tid; v < PH * PW * 4; v += NTHR) {
            const int pix = v >> 2, part = v & 3;
            const int py = pix / PW, px = pix - py * PW;
            const int ly = gy0 * S + py - d.pad_y, lx = gx0 * S + px - d.pad_x;
            const int c = c0 + part * VEC;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (ly >= 0 && ly < LH && lx >= 0 && lx < LW && c < d.Cin) {
                const int sy = ly >> upshift, sx = lx >> upshift;  // nearest: floor(o/2)
                const size_t off = ((size_t)(n * d.Hi + sy) * d.Wi + sx) * d.x.cs + d.x.coff + c;
                val = *reinterpret_cast<const u32x4*>(xg + off);
            }
            *reinterpret_cast<u32x4*>(lp + pix * CKP + part * VEC) = val;
        }
        // ---- stage the weight slab [tap][co0..co0+BN)[c0..c0+CK) ----
        for (int v = tid; v < KH * KW * BN * 4; v += NTHR) {
            const int row = v >> 2, part = v & 3;
            const int tap = row / BN, co = row - tap * BN;
            const size_t off = ((size_t)tap * d.CoutPad + co0 + co) * CinPad + c0 + part * VEC;
            *reinterpret_cast<u32x4*>(lw + row * CKP + part * VEC) = *reinterpret_cast<const u32x4*>(wg + off);
        }
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const u32x4 a = *reinterpret_cast<const u32x4*>(abase + (ky * PW + kx) * CKP + kk * 2 * VEC);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const u32x4 bb = *reinterpret_cast<const u32x4*>(
                            bbase + ((ky * KW + kx) * BN + t * 32) * CKP + kk * 2 * VEC);
                        mma16<T>(acc[t], a, bb);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- fused epilogue ----
    T* __restrict__ yp = reinterpret_cast<T*>(d.y.p);
    T* __restrict__ y0p = reinterpret_cast<T*>(d.y0.p);
    T* __restrict__ y1p = reinterpret_cast<T*>(d.y1.p);
    const T* __restrict__ r1p = reinterpret_cast<const T*>(d.r1.p);
    const T* __restrict__ r2p = reinterpret_cast<const T*>(d.r2.p);
    const T* __restrict__ mp = reinterpret_cast<const T*>(d.m.p);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = co0 + t * 32 + i;
        if (co >= d.Cout) continue;
        const float bv = d.bias ? d.bias[co] : 0.f;
        const bool has_r1 = r1p && co < d.r1_nc, has_r2 = r2p && co < d.r2_nc;
        const bool has_m = mp && co >= d.m_c0 && co < d.m_c1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pi = mfma32_row(r, g);
            const int gy = gy0 + 2 * wave + (pi >> 4), gx = gx0 + (pi & 15);
            if (gy >= d.Gh || gx >= d.Gw) continue;
            const size_t p = ((size_t)(n * d.Ho + gy * d.oys + d.oyo)) * d.Wo + gx * d.oxs + d.oxo;
            float v = acc[t][r] + bv;
            if (d.act == SSR_ACT_LRELU) v = lrelu(v);
            v *= d.alpha;
            if (y0p) y0p[p * d.y0.cs + d.y0.coff + co] = from_f32<T>(v);
            if (has_r1) v += d.beta1 * to_f32(r1p[p * d.r1.cs + d.r1.coff + co]);
            if (has_r2) v += d.beta2 * to_f32(r2p[p * d.r2.cs + d.r2.coff + co]);
            if (d.accumulate) v += to_f32(yp[p * d.y.cs + d.y.coff + co]);
            if (y1p) y1p[p * d.y1.cs + d.y1.coff + co] = from_f32<T>(v);
            if (has_m) v *= lrelu_grad_from_out(to_f32(mp[p * d.m.cs + d.m.coff + co]));
            yp[p * d.y.cs + d.y.coff + co] = from_f32<T>(v);
        }
    }
}

template <typename T, int KH, int KW, int S, int NT, int WAVES>
int launch_conv(const ssr_conv_desc& d, hipStream_t st) {
    constexpr int VEC = DT<T>::VEC, CK = 4 * VEC, CKP = CK + VEC, BN = 32 * NT;
    constexpr int TH = 2 * WAVES, TW = 16, PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
    constexpr size_t lds = (size_t)(PH * PW + KH * KW * BN) * CKP * sizeof(T);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv_kernel<T, KH, KW, S, NT, WAVES>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((d.Gw + TW - 1) / TW) * ((d.Gh + TH - 1) / TH) * d.N;
    dim3 grid(tiles, d.CoutPad / BN, 1);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES), lds, st, d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

// BN = 64 when the padded output width allows it; small problems use 2-wave (4x16) tiles so that
// more workgroups exist than CUs.
inline void pick_tile(const ssr_conv_desc& d, bool& nt2, bool& small) {
    nt2 = (d.CoutPad % 64) == 0;
    const long tiles4 = (long)((d.Gw + 15) / 16) * ((d.Gh + 7) / 8) * d.N * (d.CoutPad / (nt2 ? 64 : 32));
    small = tiles4 < 384;
}

template <typename T, int KH, int KW, int S>
int dispatch_tile(const ssr_conv_desc& d, hipStream_t st) {
    bool nt2, small;
    pick_tile(d, nt2, small);
    if (nt2) return small ? launch_conv<T, KH, KW, S, 2, 2>(d, st) : launch_conv<T, KH, KW, S, 2, 4>(d, st);
    return small ? launch_conv<T, KH, KW, S, 1, 2>(d, st) : launch_conv<T, KH, KW, S, 1, 4>(d, st);
}

template <typename T>
int dispatch_geom(const ssr_conv_desc& d, hipStream_t st) {
    if (d.KH == 3 && d.KW == 3 && d.stride == 1) return dispatch_tile<T, 3, 3, 1>(d, st);
    if (d.KH == 4 && d.KW == 4 && d.stride == 2) return dispatch_tile<T, 4, 4, 2>(d, st);
    if (d.KH == 2 && d.KW == 2 && d.stride == 1) return dispatch_tile<T, 2, 2, 1>(d, st);
    return SSR_EUNSUP;
}

bool view_ok(const ssr_view& v, bool required) {
    if (!v.p) return !required;
    return (v.cs % 8) == 0 && (v.coff % 8) == 0 && ((uintptr_t)v.p % 16) == 0;
}

}  // namespace

extern "C" int ssr_conv2d_variant(const ssr_conv_desc* dp) {
    if (!dp) return SSR_EINVAL;
    bool nt2, small;
    pick_tile(*dp, nt2, small);
    return dp->KH * 1000 + dp->stride * 100 + (nt2 ? 2 : 1) * 10 + (small ? 2 : 4);
}

extern "C" int ssr_conv2d(const ssr_conv_desc* dp, void* stream) {
    if (!dp) return SSR_EINVAL;
    const ssr_conv_desc& d = *dp;
    if (!view_ok(d.x, true) || !d.w || ((uintptr_t)d.w % 16) != 0) return SSR_EINVAL;
    if (!d.y.p || d.y.cs <= 0) return SSR_EINVAL;
    if (d.Cin <= 0 || (d.Cin % 8) != 0 || d.CoutPad <= 0 || (d.CoutPad % 32) != 0 || d.Cout > d.CoutPad)
        return SSR_EINVAL;
    if (!(d.up == 1 || d.up == 2) || d.N <= 0 || d.Gh <= 0 || d.Gw <= 0) return SSR_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d.dtype == SSR_F32) return dispatch_geom<float>(d, st);
    if (d.dtype == SSR_BF16) return dispatch_geom<__bf16>(d, st);
    return SSR_EUNSUP;
}
