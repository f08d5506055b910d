// Bandwidth-bound kernels around the convolutions: weight packing, NCHW<->NHWC boundary conversion,
// bilinear/nearest x2 (backward), spectral norm, losses, fused Adam+EMA.  All are coalesced over the
// channel (fastest NHWC) axis; reductions are wave-shuffle (64 lanes) -> LDS -> one atomic per block.
#include "common.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// sum over a workgroup of up to 1024 threads; result valid in thread 0 (and broadcast through LDS)
__device__ __forceinline__ float block_sum(float v, float* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x < 64) {
        s = (threadIdx.x < nw) ? sh[threadIdx.x] : 0.f;
        s = wave_sum(s);
        if (threadIdx.x == 0) sh[0] = s;
    }
    __syncthreads();
    return sh[0];
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
// One thread owns one (output row, k) pair of the packed layout and moves all KH*KW taps of it: the fp32 source
// taps of a pair are contiguous (36 / 64 bytes) and for a fixed tap consecutive threads write consecutive packed
// elements.  (r01 rocprofv3: the element-per-thread version with 64-bit div/mod per element and stride-KK gathers
// took 110 us per launch, 4 launches per step.)
// SSR_F32X3: packed rows of 16 input channels are stored pre-split — [16 x hi = bf16(w) | 16 x lo = bf16(w - hi)], the same 64
// bytes as 16 floats — so the split-bf16 conv kernel copies weight rows to LDS as they are (the weight slab is 60-75 % of what
// a workgroup stages per chunk, and every workgroup would otherwise repeat the same conversion).  Rows of 8 (the 4x4 stride-2
// forward layers, which run on the exact fp32 kernel) stay plain fp32.
// SSR_F32H (X3 = 2, round 6; forward tables only): the same rows with fp16 pieces of 2^SSR_F32H_WSHIFT w (include/ssr_hip.h).
template <typename T, int X3>
__device__ __forceinline__ void put_packed(T* __restrict__ dst, size_t idx, int ck, float v) {
    if (X3 == 2 && ck == 16) {
        _Float16* row = reinterpret_cast<_Float16*>(dst) + (idx >> 4) * 32;
        const float vs = v * SSR_F32H_WSCALE;
        const _Float16 h = (_Float16)vs;
        row[idx & 15] = h;
        row[16 + (idx & 15)] = (_Float16)(vs - (float)h);
    } else if (X3 == 1 && ck == 16) {
        __bf16* row = reinterpret_cast<__bf16*>(dst) + (idx >> 4) * 32;
        const __bf16 h = (__bf16)v;
        row[idx & 15] = h;
        row[16 + (idx & 15)] = (__bf16)(v - (float)h);
    } else {
        dst[idx] = from_f32<T>(v);
    }
}

template <typename T, int X3 = 0>
__global__ __launch_bounds__(256) void pack_kernel(const ssr_pack_item* __restrict__ items) {
    const ssr_pack_item it = items[blockIdx.y];
    const int KK = it.KH * it.KW;
    const float inv = it.inv_scale ? 1.0f / it.inv_scale[0] : 1.0f;
    const int stride = gridDim.x * blockDim.x;
    const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (it.dst_fwd) {   // [chunk][tap][CoutPad][ck]
        T* __restrict__ dst = reinterpret_cast<T*>(it.dst_fwd);
        const int ck = it.ck_fwd, total = it.CoutPad * it.CinPad;
        for (int t = t0; t < total; t += stride) {
            const int cc = t % ck, q = t / ck;
            const int co = q % it.CoutPad, chunk = q / it.CoutPad;
            const int ci = chunk * ck + cc;
            const bool ok = co < it.Cout && ci < it.Cin;
            const float* __restrict__ sp = it.src + ((size_t)co * it.Cin + ci) * KK;
            if (it.fwd_s2d) {
                // space-to-depth order (ssr_conv_desc.s2d): tap (ky,kx) -> parity q = (ky&1, kx&1), 2x2 tap (ky>>1, kx>>1);
                // [q * nchunks + chunk][2x2 tap][CoutPad][ck]
                const int nch = it.CinPad / ck;
                for (int tap = 0; tap < 16; ++tap) {
                    const int ky = tap >> 2, kx = tap & 3, q = (ky & 1) * 2 + (kx & 1), tt = (ky >> 1) * 2 + (kx >> 1);
                    put_packed<T, X3>(dst, ((((size_t)q * nch + chunk) * 4 + tt) * it.CoutPad + co) * ck + cc, ck, ok ? sp[tap] * inv : 0.f);
                }
                continue;
            }
            const size_t dp = ((size_t)chunk * KK * it.CoutPad + co) * ck + cc;
            for (int tap = 0; tap < KK; ++tap) put_packed<T, X3>(dst, dp + (size_t)tap * it.CoutPad * ck, ck, ok ? sp[tap] * inv : 0.f);
        }
    }
    if (it.dst_dgrad) {
        T* __restrict__ dst = reinterpret_cast<T*>(it.dst_dgrad);
        const int ck = it.ck_dgrad, total = it.CinPadO * it.CoutPadI;
        for (int t = t0; t < total; t += stride) {
            const int cc = t % ck, q = t / ck;
            const int o = q % it.CinPadO, chunk = q / it.CinPadO;
            const int k = chunk * ck + cc;
            const bool ok = k < it.Cout && o < it.Cin;
            const float* __restrict__ sp = it.src + ((size_t)k * it.Cin + o) * KK;
            if (it.stride == 1) {
                // Wd[chunk][tap'][o = ci][k = co] = W[co][ci][KK-1-tap']   (180-degree rotated, transposed)
                const size_t dp = ((size_t)chunk * KK * it.CinPadO + o) * ck + cc;
                for (int tap = 0; tap < KK; ++tap)
                    put_packed<T, X3>(dst, dp + (size_t)tap * it.CinPadO * ck, ck, ok ? sp[KK - 1 - tap] * inv : 0.f);
            } else {
                // 4x4 stride-2 transposed conv as four output-parity classes of 2x2 taps:
                // class (py,px), tap (ty,tx): ky = py ? 2-2*ty : 3-2*ty (same for x); [cls][chunk][t][o][ck]
                const size_t per_cls = (size_t)4 * it.CinPadO * it.CoutPadI;
                for (int cls = 0; cls < 4; ++cls)
                    for (int tt = 0; tt < 4; ++tt) {
                        const int py = cls >> 1, px = cls & 1, ty = tt >> 1, tx = tt & 1;
                        const int ky = py ? 2 - 2 * ty : 3 - 2 * ty, kx = px ? 2 - 2 * tx : 3 - 2 * tx;
                        put_packed<T, X3>(dst, cls * per_cls + (((size_t)chunk * 4 + tt) * it.CinPadO + o) * ck + cc, ck,
                                          ok ? sp[ky * 4 + kx] * inv : 0.f);
                    }
            }
        }
    }
}

template <typename T, int X3 = 0>
__global__ __launch_bounds__(256) void pack_seg_kernel(const ssr_pack_seg* __restrict__ items) {
    const ssr_pack_seg it = items[blockIdx.y];
    T* __restrict__ dst = reinterpret_cast<T*>(it.dst);
    const int total = it.Cout * it.rows_pad;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int kk = t % it.Cout, o = t / it.Cout;    // fastest: consecutive k -> consecutive cc in dst
        const int k = it.kbase + kk, chunk = k / it.ck, cc = k - chunk * it.ck;
        const bool ok = o < it.nci;
        const float* __restrict__ sp = it.src + ((size_t)kk * it.Cin + it.ci0 + o) * 9;
        const size_t dp = ((size_t)chunk * 9 * it.rows_pad + o) * it.ck + cc;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) put_packed<T, X3>(dst, dp + (size_t)tap * it.rows_pad * it.ck, it.ck, ok ? it.scale * sp[8 - tap] : 0.f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_views_kernel(ssr_view dst, ssr_view src, long npix, int C) {
    T* __restrict__ d = reinterpret_cast<T*>(dst.p);
    const T* __restrict__ s = reinterpret_cast<const T*>(src.p);
    const long total = npix * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long p = e / C;
        const int c = (int)(e - p * C);
        const long di = p * dst.cs + dst.coff + c;
        d[di] = from_f32<T>(to_f32(d[di]) + to_f32(s[p * src.cs + src.coff + c]));
    }
}

// ------------------------------------------------------------------------------------------------
// boundary layout conversion
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, int N, int C, int H, int W,
                                                           ssr_view dst, int s, int up, float scale) {
    // logical output: C2 = C*s*s channels, H2 = H/s*up, W2 = W/s*up
    const int C2 = C * s * s, H2 = H / s * up, W2 = W / s * up;
    const long total = (long)N * H2 * W2 * C2;
    T* __restrict__ d = reinterpret_cast<T*>(dst.p);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c2 = (int)(e % C2);
        long q = e / C2;
        const int x = (int)(q % W2); q /= W2;
        const int y = (int)(q % H2);
        const int n = (int)(q / H2);
        const int ys = y / up, xs = x / up;              // nearest: floor(o / up)
        const int c = c2 / (s * s), i = (c2 / s) % s, j = c2 % s;  // pixel_unshuffle index map
        const float v = src[(((long)n * C + c) * H + ys * s + i) * W + xs * s + j] * scale;
        d[(((long)n * H2 + y) * W2 + x) * dst.cs + dst.coff + c2] = from_f32<T>(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(ssr_view src, float* __restrict__ dst, int N, int C, int H,
                                                           int W) {
    const long total = (long)N * C * H * W;
    const T* __restrict__ s = reinterpret_cast<const T*>(src.p);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e % W);
        long q = e / W;
        const int y = (int)(q % H); q /= H;
        const int c = (int)(q % C);
        const int n = (int)(q / C);
        dst[e] = to_f32(s[(((long)n * H + y) * W + x) * src.cs + src.coff + c]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(T* __restrict__ p, long n, float v) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        p[e] = from_f32<T>(v);
}

// ------------------------------------------------------------------------------------------------
// bilinear x2, align_corners=False (ATen upsample_bilinear2d semantics: src = max(0.5*o - 0.25, 0))
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bil_src(int o, int n_in, int& i0, int& i1, float& l0, float& l1) {
    float s = 0.5f * (float)o - 0.25f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

// 16-byte channel vectors (8 bf16 / 4 fp32 per thread)
template <typename T> struct VecIO;
template <> struct VecIO<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&f)[4]) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = v[k];
    }
    static __device__ __forceinline__ void store(float* p, const float (&f)[4]) {
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = f[k];
        *reinterpret_cast<f32x4*>(p) = v;
    }
};
template <> struct VecIO<__bf16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const __bf16* p, float (&f)[8]) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = (float)v[k];
    }
    static __device__ __forceinline__ void store(__bf16* p, const float (&f)[8]) {
        bf16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (__bf16)f[k];
        *reinterpret_cast<bf16x8*>(p) = v;
    }
};

// One thread = one INPUT pixel x 16 bytes of channels -> its 2x2 block of outputs: the block needs the 3x3 input
// neighbourhood (9 loads for 4 outputs instead of 16) and the index arithmetic is 32-bit and amortised over 4 outputs
// (r01 rocprofv3: the output-pixel-per-thread version with 64-bit div/mod ran at ~2.5x its HBM time).  The arithmetic
// per output is exactly bil_src's: even outputs blend rows (i-1, i) with (.25, .75) — (i, i) with (1, 0) at the border —
// odd outputs rows (i, min(i+1, n-1)) with (.75, .25).
template <typename T>
__global__ __launch_bounds__(256) void bilinear2x_fwd_kernel(ssr_view a, ssr_view b, ssr_view y, int N, int H, int W,
                                                             int C) {
    constexpr int V = VecIO<T>::N;
    const int W2 = 2 * W, CV = C / V;
    const int total = N * H * W * CV;
    const T* __restrict__ ap = reinterpret_cast<const T*>(a.p);
    const T* __restrict__ bp = reinterpret_cast<const T*>(b.p);
    T* __restrict__ yp = reinterpret_cast<T*>(y.p);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int c = (e % CV) * V;
        int q = e / CV;
        const int ix = q % W; q /= W;
        const int iy = q % H;
        const int n = q / H;
        const int ys[3] = {iy > 0 ? iy - 1 : 0, iy, iy < H - 1 ? iy + 1 : iy};
        const int xs[3] = {ix > 0 ? ix - 1 : 0, ix, ix < W - 1 ? ix + 1 : ix};
        float t[3][3][V];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                const size_t p = (size_t)(n * H + ys[r]) * W + xs[cc];
                VecIO<T>::load(ap + p * a.cs + a.coff + c, t[r][cc]);
                if (bp) {
                    float g[V];
                    VecIO<T>::load(bp + p * b.cs + b.coff + c, g);
#pragma unroll
                    for (int k = 0; k < V; ++k) t[r][cc][k] += g[k];
                }
            }
        // even output: rows t[0], t[1] with (.25, .75) — (1, 0) at the border, where t[0] == t[1] anyway; odd: t[1], t[2] (.75, .25)
        const float wya[2] = {iy > 0 ? 0.25f : 1.f, 0.75f}, wyb[2] = {iy > 0 ? 0.75f : 0.f, 0.25f};
        const float wxa[2] = {ix > 0 ? 0.25f : 1.f, 0.75f}, wxb[2] = {ix > 0 ? 0.75f : 0.f, 0.25f};
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                float o[V];
#pragma unroll
                for (int k = 0; k < V; ++k)
                    o[k] = wya[py] * (wxa[px] * t[py][px][k] + wxb[px] * t[py][px + 1][k]) +
                           wyb[py] * (wxa[px] * t[py + 1][px][k] + wxb[px] * t[py + 1][px + 1][k]);
                VecIO<T>::store(yp + ((size_t)(n * 2 * H + 2 * iy + py) * W2 + 2 * ix + px) * y.cs + y.coff + c, o);
            }
    }
}

__device__ __forceinline__ float bil_w(int o, int n_in, int i) {  // weight of input i in output o
    int i0, i1;
    float l0, l1;
    bil_src(o, n_in, i0, i1, l0, l1);
    return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// MODE 0: bilinear x2 transpose; MODE 1: nearest x2 transpose (2x2 sum)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void up2x_bwd_kernel(ssr_view dy, ssr_view r, ssr_view y1, ssr_view y, ssr_view m,
                                                       int N, int H, int W, int C) {
    constexpr int V = VecIO<T>::N;
    const int H2 = 2 * H, W2 = 2 * W, CV = C / V;
    const long total = (long)N * H * W * CV;
    const T* __restrict__ dp = reinterpret_cast<const T*>(dy.p);
    const T* __restrict__ rp = reinterpret_cast<const T*>(r.p);
    const T* __restrict__ mp = reinterpret_cast<const T*>(m.p);
    T* __restrict__ y1p = reinterpret_cast<T*>(y1.p);
    T* __restrict__ yp = reinterpret_cast<T*>(y.p);
    // (32-bit index arithmetic: the host checks total < 2^31; three 64-bit divisions per vector made this VALU bound)
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < (unsigned)total; e += gridDim.x * blockDim.x) {
        const int c = (int)(e % (unsigned)CV) * V;
        unsigned q = e / (unsigned)CV;
        const int ix = (int)(q % (unsigned)W); q /= (unsigned)W;
        const int iy = (int)(q % (unsigned)H);
        const int n = (int)(q / (unsigned)H);
        float s[V], t[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] = 0.f;
        if (MODE == 0) {
            // transpose of bil_src: input i receives 0.25 of output 2i-1, 0.75 of 2i (all of it at i = 0: the top border
            // blends (0, 0)), 0.75 of 2i+1 (all of it at i = n-1) and 0.25 of 2i+2.  Fixed weights, clamped unconditional
            // loads (weight 0 where the output does not exist), rows combined first.
            const float wy[4] = {iy > 0 ? 0.25f : 0.f, iy > 0 ? 0.75f : 1.f, iy < H - 1 ? 0.75f : 1.f, iy < H - 1 ? 0.25f : 0.f};
            const float wx[4] = {ix > 0 ? 0.25f : 0.f, ix > 0 ? 0.75f : 1.f, ix < W - 1 ? 0.75f : 1.f, ix < W - 1 ? 0.25f : 0.f};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int oy = min(max(2 * iy - 1 + a, 0), H2 - 1);
                float rs[V];
#pragma unroll
                for (int k = 0; k < V; ++k) rs[k] = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int ox = min(max(2 * ix - 1 + b, 0), W2 - 1);
                    VecIO<T>::load(dp + ((size_t)(n * H2 + oy) * W2 + ox) * dy.cs + dy.coff + c, t);
#pragma unroll
                    for (int k = 0; k < V; ++k) rs[k] += wx[b] * t[k];
                }
#pragma unroll
                for (int k = 0; k < V; ++k) s[k] += wy[a] * rs[k];
            }
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    VecIO<T>::load(dp + (((long)n * H2 + 2 * iy + a) * W2 + 2 * ix + b) * dy.cs + dy.coff + c, t);
#pragma unroll
                    for (int k = 0; k < V; ++k) s[k] += t[k];
                }
        }
        const long p = ((long)n * H + iy) * W + ix;
        if (rp) {
            VecIO<T>::load(rp + p * r.cs + r.coff + c, t);
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] += t[k];
        }
        if (y1p) VecIO<T>::store(y1p + p * y1.cs + y1.coff + c, s);
        if (mp) {
            VecIO<T>::load(mp + p * m.cs + m.coff + c, t);
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] *= lrelu_grad_from_out(t[k]);
        }
        if (yp) VecIO<T>::store(yp + p * y.cs + y.coff + c, s);
    }
}

// ------------------------------------------------------------------------------------------------
// C a multiple of 8 vectors of 16 bytes - 64 channels in bf16, 32 in fp32 storage - (the U-Net discriminator's three x2 steps,
// discriminator_arch.py:47,52,57): the same arithmetic in the
// same order as the kernels above, but the 3 x 3 (forward) / 4 x 4 (backward) neighbourhoods come out of an LDS tile.
// The per-pixel kernels above fetch every input vector 9 (16) times, and a wave covers only 1..4 pixels of 128..512
// channels, so the reuse is across waves and mostly misses L1: 0.40 + 0.41 ms per step at ~2.2 TB/s of useful traffic,
// the L2 delivering 9x / 4x that (r03).  Tile: 4 x 16 (4 x 8) input pixels x 64 channels, halo replicated by clamping
// exactly as the index clamps above do.
// ------------------------------------------------------------------------------------------------
constexpr int BF_IH = 4, BF_IW = 16, BF_PH = BF_IH + 2, BF_PW = BF_IW + 2, BF_NPX = BF_PH * BF_PW;   // 108 patch pixels
// a workgroup covers 8 channel vectors of 16 bytes: 64 channels (bf16) / 32 channels (fp32 storage)
template <typename T>
__global__ __launch_bounds__(256) void bilinear2x_fwd_tile_kernel(ssr_view a, ssr_view b, ssr_view y, int N, int H, int W) {
    constexpr int V = VecIO<T>::N, NPL = V / 4;
    // fp32 sums a + b: [4-float part of the channel vector][pixel][vector * 4 + e] - 128-byte pixel rows: the 16 lanes the LDS
    // serves together (4 vectors of 4 consecutive pixels) read four different 64-byte bank groups
    __shared__ __attribute__((aligned(16))) float patch[NPL][BF_NPX][32];
    const int tid = threadIdx.x;
    const int tiles_x = (W + BF_IW - 1) / BF_IW, tiles_y = (H + BF_IH - 1) / BF_IH;
    int t_ = blockIdx.x;
    const int tx = t_ % tiles_x; t_ /= tiles_x;
    const int ty = t_ % tiles_y;
    const int n = t_ / tiles_y;
    const int c0 = blockIdx.y * 8 * V, iy0 = ty * BF_IH, ix0 = tx * BF_IW;
    const T* __restrict__ ap = reinterpret_cast<const T*>(a.p);
    const T* __restrict__ bp = reinterpret_cast<const T*>(b.p);
    T* __restrict__ yp = reinterpret_cast<T*>(y.p);
    for (int v = tid; v < BF_NPX * 8; v += 256) {
        const int pix = v >> 3, c8 = v & 7;
        const int py = pix / BF_PW, px = pix - py * BF_PW;
        const int sy = min(max(iy0 - 1 + py, 0), H - 1), sx = min(max(ix0 - 1 + px, 0), W - 1);
        const size_t p = (size_t)(n * H + sy) * W + sx;
        float f[V];
        VecIO<T>::load(ap + p * a.cs + a.coff + c0 + c8 * V, f);
        if (bp) {
            float g[V];
            VecIO<T>::load(bp + p * b.cs + b.coff + c0 + c8 * V, g);
#pragma unroll
            for (int k = 0; k < V; ++k) f[k] += g[k];
        }
#pragma unroll
        for (int h = 0; h < NPL; ++h)
            *reinterpret_cast<f32x4*>(&patch[h][pix][c8 * 4]) = f32x4{f[4 * h], f[4 * h + 1], f[4 * h + 2], f[4 * h + 3]};
    }
    __syncthreads();
    const int W2 = 2 * W;
    for (int it = tid; it < BF_IH * BF_IW * 8; it += 256) {
        const int c8 = it & 7, q = it >> 3;
        const int cc = q % BF_IW, r = q / BF_IW;
        const int iy = iy0 + r, ix = ix0 + cc;
        if (iy >= H || ix >= W) continue;
        float t[3][3][V];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int c2 = 0; c2 < 3; ++c2) {
                const int pix = (r + rr) * BF_PW + cc + c2;
#pragma unroll
                for (int h = 0; h < NPL; ++h) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(&patch[h][pix][c8 * 4]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[rr][c2][4 * h + k] = w4[k];
                }
            }
        const float wya[2] = {iy > 0 ? 0.25f : 1.f, 0.75f}, wyb[2] = {iy > 0 ? 0.75f : 0.f, 0.25f};
        const float wxa[2] = {ix > 0 ? 0.25f : 1.f, 0.75f}, wxb[2] = {ix > 0 ? 0.75f : 0.f, 0.25f};
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                float o[V];
#pragma unroll
                for (int k = 0; k < V; ++k)
                    o[k] = wya[py] * (wxa[px] * t[py][px][k] + wxb[px] * t[py][px + 1][k]) +
                           wyb[py] * (wxa[px] * t[py + 1][px][k] + wxb[px] * t[py + 1][px + 1][k]);
                VecIO<T>::store(yp + ((size_t)(n * 2 * H + 2 * iy + py) * W2 + 2 * ix + px) * y.cs + y.coff + c0 + c8 * V, o);
            }
    }
}

constexpr int BB_IH = 4, BB_IW = 8, BB_PH = 2 * BB_IH + 2, BB_PW = 2 * BB_IW + 2, BB_NPX = BB_PH * BB_PW;   // 180 gradient pixels
constexpr int BB_PITCH = 192;   // bytes per patch pixel (128 of data): lanes step two pixels, 384 B = half the banks further
template <typename T>
__global__ __launch_bounds__(256) void bilinear2x_bwd_tile_kernel(ssr_view dy, ssr_view r, ssr_view y1, ssr_view y, ssr_view m,
                                                                  int N, int H, int W) {
    constexpr int V = VecIO<T>::N;
    __shared__ __attribute__((aligned(16))) char dyp[BB_NPX * BB_PITCH];
    const int tid = threadIdx.x;
    const int H2 = 2 * H, W2 = 2 * W;
    const int tiles_x = (W + BB_IW - 1) / BB_IW, tiles_y = (H + BB_IH - 1) / BB_IH;
    int t_ = blockIdx.x;
    const int tx = t_ % tiles_x; t_ /= tiles_x;
    const int ty = t_ % tiles_y;
    const int n = t_ / tiles_y;
    const int c0 = blockIdx.y * 8 * V, iy0 = ty * BB_IH, ix0 = tx * BB_IW;
    const T* __restrict__ dp = reinterpret_cast<const T*>(dy.p);
    const T* __restrict__ rp = reinterpret_cast<const T*>(r.p);
    const T* __restrict__ mp = reinterpret_cast<const T*>(m.p);
    T* __restrict__ y1p = reinterpret_cast<T*>(y1.p);
    T* __restrict__ yp = reinterpret_cast<T*>(y.p);
    for (int v = tid; v < BB_NPX * 8; v += 256) {
        const int pix = v >> 3, c8 = v & 7;
        const int py = pix / BB_PW, px = pix - py * BB_PW;
        const int oy = min(max(2 * iy0 - 1 + py, 0), H2 - 1), ox = min(max(2 * ix0 - 1 + px, 0), W2 - 1);
        *reinterpret_cast<u32x4*>(dyp + pix * BB_PITCH + c8 * 16) =
            *reinterpret_cast<const u32x4*>(dp + ((size_t)(n * H2 + oy) * W2 + ox) * dy.cs + dy.coff + c0 + c8 * V);
    }
    __syncthreads();
    const int c8 = tid & 7, cc = (tid >> 3) & 7, rr = tid >> 6;
    const int iy = iy0 + rr, ix = ix0 + cc;
    if (iy >= H || ix >= W) return;
    const int c = c0 + c8 * V;
    float s[V], t[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = 0.f;
    const float wy[4] = {iy > 0 ? 0.25f : 0.f, iy > 0 ? 0.75f : 1.f, iy < H - 1 ? 0.75f : 1.f, iy < H - 1 ? 0.25f : 0.f};
    const float wx[4] = {ix > 0 ? 0.25f : 0.f, ix > 0 ? 0.75f : 1.f, ix < W - 1 ? 0.75f : 1.f, ix < W - 1 ? 0.25f : 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float rs[V];
#pragma unroll
        for (int k = 0; k < V; ++k) rs[k] = 0.f;
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            VecIO<T>::load(reinterpret_cast<const T*>(dyp + ((2 * rr + a) * BB_PW + 2 * cc + bq) * BB_PITCH + c8 * 16), t);
#pragma unroll
            for (int k = 0; k < V; ++k) rs[k] += wx[bq] * t[k];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] += wy[a] * rs[k];
    }
    const long p = ((long)n * H + iy) * W + ix;
    if (rp) {
        VecIO<T>::load(rp + p * r.cs + r.coff + c, t);
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] += t[k];
    }
    if (y1p) VecIO<T>::store(y1p + p * y1.cs + y1.coff + c, s);
    if (mp) {
        VecIO<T>::load(mp + p * m.cs + m.coff + c, t);
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] *= lrelu_grad_from_out(t[k]);
    }
    if (yp) VecIO<T>::store(yp + p * y.cs + y.coff + c, s);
}

// ------------------------------------------------------------------------------------------------
// spectral norm power iteration (torch.nn.utils.spectral_norm, eps = 1e-12)
//   tmp layout: [0..rows) = s = W v ; [rows..rows+cols) = t = W^T u ; [rows+cols] = scratch
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sn_wtu_kernel(const ssr_sn_item* __restrict__ items) {
    // t = W^T u: 64 columns per workgroup (coalesced 256-B row segments), the 4 waves stride the rows
    __shared__ float part[4][64];
    const ssr_sn_item it = items[blockIdx.y];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= it.cols) return;
    float s = 0.f;
    if (j < it.cols) {
        const float* __restrict__ wp = it.w + j;
#pragma unroll 8
        for (int i = w; i < it.rows; i += 4) s += wp[(long)i * it.cols] * it.u[i];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && j < it.cols) it.tmp[it.rows + j] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}

// one wave per row: s_i = sum_j W[i][j] * vhat[j], vhat = t / max(|t|, eps) (power_iter) or v (eval)
__global__ __launch_bounds__(256) void sn_wv_kernel(const ssr_sn_item* __restrict__ items, int power_iter) {
    __shared__ float sh[16];
    const ssr_sn_item it = items[blockIdx.y];
    const float* vsrc = power_iter ? it.tmp + it.rows : it.v;
    float inv = 1.f;
    if (power_iter) {
        float ss = 0.f;
        for (int j = threadIdx.x; j < it.cols; j += blockDim.x) ss += vsrc[j] * vsrc[j];
        ss = block_sum(ss, sh);
        inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        if (blockIdx.x == 0)
            for (int j = threadIdx.x; j < it.cols; j += blockDim.x) it.v[j] = vsrc[j] * inv;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + w;
    if (row >= it.rows) return;
    float s = 0.f;
    const float* __restrict__ wr = it.w + (long)row * it.cols;
    if ((it.cols & 3) == 0 && ((uintptr_t)wr & 15) == 0 && ((uintptr_t)vsrc & 15) == 0) {
        // 16-byte loads, four independent partial sums (one wave streams a row of up to 18 KB: the scalar loop with one
        // dependent accumulator was latency bound, 31 us for 17.6 MB)
        const f32x4* __restrict__ w4 = reinterpret_cast<const f32x4*>(wr);
        const f32x4* __restrict__ v4 = reinterpret_cast<const f32x4*>(vsrc);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int j = lane; j < (it.cols >> 2); j += 64) {
            const f32x4 x = w4[j], y = v4[j];
            a.x += x.x * y.x; a.y += x.y * y.y; a.z += x.z * y.z; a.w += x.w * y.w;
        }
        s = (a.x + a.y) + (a.z + a.w);
    } else {
        for (int j = lane; j < it.cols; j += 64) s += wr[j] * vsrc[j];
    }
    s = wave_sum(s) * inv;
    if (lane == 0) it.tmp[row] = s;
}

__global__ __launch_bounds__(256) void sn_finish_kernel(const ssr_sn_item* __restrict__ items, int power_iter) {
    __shared__ float sh[16];
    const ssr_sn_item it = items[blockIdx.x];
    if (power_iter) {
        float ss = 0.f;
        for (int i = threadIdx.x; i < it.rows; i += blockDim.x) ss += it.tmp[i] * it.tmp[i];
        ss = block_sum(ss, sh);
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        for (int i = threadIdx.x; i < it.rows; i += blockDim.x) it.u[i] = it.tmp[i] * inv;
        if (threadIdx.x == 0) it.sigma[0] = ss * inv;  // u . (W v) with u = s/|s|
    } else {
        float d = 0.f;
        for (int i = threadIdx.x; i < it.rows; i += blockDim.x) d += it.tmp[i] * it.u[i];
        d = block_sum(d, sh);
        if (threadIdx.x == 0) it.sigma[0] = d;
    }
}

__global__ __launch_bounds__(256) void sn_bwd_dot_kernel(const ssr_sn_bwd_item* __restrict__ items) {
    __shared__ float sh[16];
    const ssr_sn_bwd_item it = items[blockIdx.y];
    const long n = (long)it.rows * it.cols;
    float s = 0.f;
    if ((n & 3) == 0 && (((uintptr_t)it.dw_sn | (uintptr_t)it.w) & 15) == 0) {
        const f32x4* __restrict__ a4 = reinterpret_cast<const f32x4*>(it.dw_sn);
        const f32x4* __restrict__ b4 = reinterpret_cast<const f32x4*>(it.w);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (n >> 2); e += (long)gridDim.x * blockDim.x) {
            const f32x4 x = a4[e], y = b4[e];
            a.x += x.x * y.x; a.y += x.y * y.y; a.z += x.z * y.z; a.w += x.w * y.w;
        }
        s = (a.x + a.y) + (a.z + a.w);
    } else {
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
            s += it.dw_sn[e] * it.w[e];
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) it.tmp[blockIdx.x] = s;     // one slot per block (no atomics: sn_bwd_apply_kernel adds the slots in index order)
}

__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const ssr_sn_bwd_item* __restrict__ items) {
    const ssr_sn_bwd_item it = items[blockIdx.y];
    const long n = (long)it.rows * it.cols;
    const float sigma = it.sigma[0];
    // <dW_sn, W>: the dot kernel's per-block partial sums (one per blockIdx.x of the SAME grid), added in index order by every
    // thread - a fixed-order sum (round 4: was one fp32 atomicAdd per block, i.e. arrival order)
    float dot = 0.f;
    for (unsigned b = 0; b < gridDim.x; ++b) dot += it.tmp[b];
    const float coef = dot / (sigma * sigma);  // <dW_sn, W> / sigma^2 = <dW_sn, W_sn> / sigma
    const float inv = 1.f / sigma;
    if ((it.cols & 3) == 0 && n < (1L << 31) && (((uintptr_t)it.dw_sn | (uintptr_t)it.dw | (uintptr_t)it.v) & 15) == 0) {
        // four columns per thread: one 32-bit division per 16 bytes (the per-element 64-bit division made this VALU bound)
        const unsigned n4 = (unsigned)(n >> 2), c4 = (unsigned)it.cols >> 2;
        const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(it.dw_sn);
        const f32x4* __restrict__ v4 = reinterpret_cast<const f32x4*>(it.v);
        f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(it.dw);
        for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += gridDim.x * blockDim.x) {
            const unsigned i = e / c4, j = e - i * c4;
            const float cu = coef * it.u[i];
            const f32x4 g = s4[e], v = v4[j];
            f32x4 o = d4[e];
            o.x += g.x * inv - cu * v.x; o.y += g.y * inv - cu * v.y; o.z += g.z * inv - cu * v.z; o.w += g.w * inv - cu * v.w;
            d4[e] = o;
        }
    } else {
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
            const int i = (int)(e / it.cols), j = (int)(e - (long)i * it.cols);
            it.dw[e] += it.dw_sn[e] * inv - coef * it.u[i] * it.v[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// losses
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void l1_loss_kernel(ssr_view a, ssr_view b, ssr_view grad, long npix, int C,
                                                      float weight, float* __restrict__ loss_out, bool det) {
    __shared__ float sh[16];
    const long total = npix * C;
    const float invn = 1.f / (float)total;
    const T* __restrict__ ap = reinterpret_cast<const T*>(a.p);
    const T* __restrict__ bp = reinterpret_cast<const T*>(b.p);
    T* __restrict__ gp = reinterpret_cast<T*>(grad.p);
    float s = 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long p = e / C;
        const int c = (int)(e - p * C);
        const float d = to_f32(ap[p * a.cs + a.coff + c]) - to_f32(bp[p * b.cs + b.coff + c]);
        s += fabsf(d);
        if (gp) gp[p * grad.cs + grad.coff + c] = from_f32<T>(d > 0.f ? weight * invn : (d < 0.f ? -weight * invn : 0.f));
    }
    s = block_sum(s, sh);
    // det: one slot per block, `+=` by the slot's single writer (launches on a stream are ordered) - the reader adds the slots in
    // index order; default: one fp32 atomic per block (arrival order: last-bit differences from run to run)
    if (threadIdx.x == 0 && loss_out) { if (det) loss_out[blockIdx.x] += s * weight * invn; else atomicAdd(loss_out, s * weight * invn); }
}

template <typename T>
__global__ __launch_bounds__(256) void bce_loss_kernel(ssr_view x, ssr_view grad, long npix, float target,
                                                       float weight, float* __restrict__ loss_out,
                                                       float* __restrict__ mean_out, bool det) {
    __shared__ float sh[16];
    const float invn = 1.f / (float)npix;
    const T* __restrict__ xp = reinterpret_cast<const T*>(x.p);
    T* __restrict__ gp = reinterpret_cast<T*>(grad.p);
    float s = 0.f, sm = 0.f;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const float v = to_f32(xp[p * x.cs + x.coff]);
        // BCEWithLogits: max(x,0) - x*t + log1p(exp(-|x|))
        s += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
        sm += v;
        if (gp) {
            const float sig = 1.f / (1.f + expf(-v));
            gp[p * grad.cs + grad.coff] = from_f32<T>((sig - target) * weight * invn);
        }
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0 && loss_out) { if (det) loss_out[blockIdx.x] += s * weight * invn; else atomicAdd(loss_out, s * weight * invn); }
    if (mean_out) {
        sm = block_sum(sm, sh);
        if (threadIdx.x == 0) { if (det) mean_out[blockIdx.x] += sm * invn; else atomicAdd(mean_out, sm * invn); }
    }
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor math) + EMA, one launch per flat arena
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(const ssr_adam_args a) {
    const int t = a.step[0] + 1;
    const float lr = a.lr[0];
    const float bc1 = 1.f - powf(a.beta1, (float)t);
    const float bc2 = 1.f - powf(a.beta2, (float)t);
    const float step_size = lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < a.n; e += (long)gridDim.x * blockDim.x) {
        const float g = a.grad[e] * a.grad_scale;
        float m = a.exp_avg[e], v = a.exp_avg_sq[e];
        m = m + (g - m) * (1.f - a.beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
        v = v * a.beta2 + (1.f - a.beta2) * g * g;         // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(v) / bc2_sqrt + a.eps;
        const float p = a.param[e] - step_size * (m / denom);
        a.exp_avg[e] = m;
        a.exp_avg_sq[e] = v;
        a.param[e] = p;
        if (a.ema) a.ema[e] = a.ema[e] * a.ema_decay + p * (1.f - a.ema_decay);
    }
}
__global__ void bump_kernel(int32_t* step) { step[0] += 1; }

__global__ __launch_bounds__(256) void axpby_kernel(float a, const float* __restrict__ x, float b,
                                                    float* __restrict__ y, long n) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        y[e] = a * x[e] + (b == 0.f ? 0.f : b * y[e]);
}

inline int grid_for(long total, int per_block = 256, int cap = 4096) {
    long g = (total + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int ssr_pack_weights(const ssr_pack_item* items_dev, int32_t n_items, int32_t dtype, void* stream) {
    if (!items_dev || n_items <= 0) return SSR_EINVAL;
    // a table of a few large layers (the discriminator: conv3 alone is 131k (co, ci) pairs x 16 taps) needs more than 64
    // workgroups per layer to fill the chip (r01 rocprofv3: 42 us for 35 MB); the generator's 351 small layers do not
    dim3 grid(n_items <= 16 ? 1024 : 64, n_items);
    if (dtype == SSR_F32) hipLaunchKernelGGL(pack_kernel<float>, grid, dim3(256), 0, ST(stream), items_dev);
    else if (dtype == SSR_F32X3) hipLaunchKernelGGL((pack_kernel<float, 1>), grid, dim3(256), 0, ST(stream), items_dev);
    else if (dtype == SSR_F32H) hipLaunchKernelGGL((pack_kernel<float, 2>), grid, dim3(256), 0, ST(stream), items_dev);   // (forward tables: the mode's backward is SSR_F32X3)
    else if (dtype == SSR_BF16) hipLaunchKernelGGL(pack_kernel<__bf16>, grid, dim3(256), 0, ST(stream), items_dev);
    else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_pack_dgrad_gather(const ssr_pack_seg* items_dev, int32_t n_items, int32_t dtype, void* stream) {
    if (!items_dev || n_items <= 0) return SSR_EINVAL;
    dim3 grid(8, n_items);
    if (dtype == SSR_F32) hipLaunchKernelGGL(pack_seg_kernel<float>, grid, dim3(256), 0, ST(stream), items_dev);
    else if (dtype == SSR_F32X3) hipLaunchKernelGGL((pack_seg_kernel<float, 1>), grid, dim3(256), 0, ST(stream), items_dev);
    else if (dtype == SSR_BF16) hipLaunchKernelGGL(pack_seg_kernel<__bf16>, grid, dim3(256), 0, ST(stream), items_dev);
    else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_add_views(ssr_view dst, ssr_view src, int32_t dtype, int64_t npix, int32_t C, void* stream) {
    if (dtype == SSR_F32X3) dtype = SSR_F32;   // fp32 storage: only the matrix-core kernels differ
    if (!dst.p || !src.p || npix <= 0 || C <= 0) return SSR_EINVAL;
    const int g = grid_for(npix * C, 256 * 4, 2048);
    if (dtype == SSR_F32)
        hipLaunchKernelGGL(add_views_kernel<float>, dim3(g), dim3(256), 0, ST(stream), dst, src, (long)npix, C);
    else if (dtype == SSR_BF16)
        hipLaunchKernelGGL(add_views_kernel<__bf16>, dim3(g), dim3(256), 0, ST(stream), dst, src, (long)npix, C);
    else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_nchw_to_nhwc(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, ssr_view dst,
                                int32_t dtype, int32_t unshuffle, int32_t up, float scale, void* stream) {
    if (dtype == SSR_F32X3) dtype = SSR_F32;   // fp32 storage: only the matrix-core kernels differ
    if (!src || !dst.p || unshuffle < 1 || up < 1 || H % unshuffle || W % unshuffle) return SSR_EINVAL;
    const long total = (long)N * C * H * W * up * up;
    if (dtype == SSR_F32)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, ST(stream), src, N, C, H, W,
                           dst, unshuffle, up, scale);
    else if (dtype == SSR_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, ST(stream), src, N, C, H,
                           W, dst, unshuffle, up, scale);
    else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_nhwc_to_nchw(ssr_view src, int32_t dtype, float* dst, int32_t N, int32_t C, int32_t H, int32_t W,
                                void* stream) {
    if (dtype == SSR_F32X3) dtype = SSR_F32;   // fp32 storage: only the matrix-core kernels differ
    if (!src.p || !dst) return SSR_EINVAL;
    const long total = (long)N * C * H * W;
    if (dtype == SSR_F32)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, ST(stream), src, dst, N, C,
                           H, W);
    else if (dtype == SSR_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, ST(stream), src, dst, N, C,
                           H, W);
    else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_fill(void* p, int64_t n, int32_t dtype, float value, void* stream) {
    if (dtype == SSR_F32X3) dtype = SSR_F32;   // fp32 storage: only the matrix-core kernels differ
    if (!p || n < 0) return SSR_EINVAL;
    if (n == 0) return SSR_OK;
    if (dtype == SSR_F32)
        hipLaunchKernelGGL(fill_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST(stream), (float*)p, (long)n, value);
    else if (dtype == SSR_BF16)
        hipLaunchKernelGGL(fill_kernel<__bf16>, dim3(grid_for(n)), dim3(256), 0, ST(stream), (__bf16*)p, (long)n, value);
    else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

// SSR_BILINEAR_FLAT=1 in the environment (read once) or SSR_BILINEAR_FLAT OR-ed into a call's dtype: the per-pixel kernels for
// every shape (A/B of the LDS-tile kernels)
static const bool g_bilinear_flat_env = [] { const char* e = getenv("SSR_BILINEAR_FLAT"); return e && e[0] == '1'; }();

extern "C" int ssr_bilinear2x_fwd(ssr_view a, ssr_view b, ssr_view y, int32_t dtype, int32_t N, int32_t H, int32_t W,
                                  int32_t C, void* stream) {
    const bool g_bilinear_flat = g_bilinear_flat_env || (dtype & SSR_BILINEAR_FLAT) != 0;
    dtype &= ~SSR_BILINEAR_FLAT;
    if (dtype == SSR_F32X3) dtype = SSR_F32;   // fp32 storage: only the matrix-core kernels differ
    if (!a.p || !y.p || (C % 8) != 0 || (a.cs % 8) || (a.coff % 8) || (y.cs % 8) || (y.coff % 8) ||
        (b.p && ((b.cs % 8) || (b.coff % 8))))
        return SSR_EINVAL;
    const long total = (long)N * H * W * C / (dtype == SSR_F32 ? 4 : 8);   // one thread per input pixel x 16 bytes
    if (total >= (1L << 31)) return SSR_EINVAL;
    const dim3 tgrid(N * ((H + BF_IH - 1) / BF_IH) * ((W + BF_IW - 1) / BF_IW), dtype == SSR_F32 ? C / 32 : C / 64);
    if (dtype == SSR_F32 && (C % 32) == 0 && !g_bilinear_flat)
        hipLaunchKernelGGL(bilinear2x_fwd_tile_kernel<float>, tgrid, dim3(256), 0, ST(stream), a, b, y, N, H, W);
    else if (dtype == SSR_F32)
        hipLaunchKernelGGL(bilinear2x_fwd_kernel<float>, dim3(grid_for(total, 256, 8192)), dim3(256), 0, ST(stream), a,
                           b, y, N, H, W, C);
    else if (dtype == SSR_BF16 && (C % 64) == 0 && !g_bilinear_flat)
        hipLaunchKernelGGL(bilinear2x_fwd_tile_kernel<__bf16>, tgrid, dim3(256), 0, ST(stream), a, b, y, N, H, W);
    else if (dtype == SSR_BF16)
        hipLaunchKernelGGL(bilinear2x_fwd_kernel<__bf16>, dim3(grid_for(total, 256, 8192)), dim3(256), 0, ST(stream), a,
                           b, y, N, H, W, C);
    else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

template <int MODE>
static int up2x_bwd(ssr_view dy, ssr_view r, ssr_view y1, ssr_view y, ssr_view m, int32_t dtype, int32_t N, int32_t H,
                    int32_t W, int32_t C, void* stream) {
    const bool g_bilinear_flat = g_bilinear_flat_env || (dtype & SSR_BILINEAR_FLAT) != 0;
    dtype &= ~SSR_BILINEAR_FLAT;
    if (dtype == SSR_F32X3) dtype = SSR_F32;   // fp32 storage: only the matrix-core kernels differ
    if (!dy.p || (!y.p && !y1.p) || (C % 8) != 0 || (dy.cs % 8) || (dy.coff % 8)) return SSR_EINVAL;
    const long total = (long)N * H * W * C / 4;
    if (total >= (1L << 31)) return SSR_EINVAL;               // the kernel indexes in 32 bits
    const dim3 tgrid(N * ((H + BB_IH - 1) / BB_IH) * ((W + BB_IW - 1) / BB_IW), dtype == SSR_F32 ? C / 32 : C / 64);
    if (dtype == SSR_F32 && MODE == 0 && (C % 32) == 0 && !g_bilinear_flat)
        hipLaunchKernelGGL(bilinear2x_bwd_tile_kernel<float>, tgrid, dim3(256), 0, ST(stream), dy, r, y1, y, m, N, H, W);
    else if (dtype == SSR_F32)
        hipLaunchKernelGGL((up2x_bwd_kernel<float, MODE>), dim3(grid_for(total, 256, 8192)), dim3(256), 0, ST(stream),
                           dy, r, y1, y, m, N, H, W, C);
    else if (dtype == SSR_BF16 && MODE == 0 && (C % 64) == 0 && !g_bilinear_flat)
        hipLaunchKernelGGL(bilinear2x_bwd_tile_kernel<__bf16>, tgrid, dim3(256), 0, ST(stream), dy, r, y1, y, m, N, H, W);
    else if (dtype == SSR_BF16)
        hipLaunchKernelGGL((up2x_bwd_kernel<__bf16, MODE>), dim3(grid_for(total, 256, 8192)), dim3(256), 0, ST(stream),
                           dy, r, y1, y, m, N, H, W, C);
    else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}
extern "C" int ssr_bilinear2x_bwd(ssr_view dy, ssr_view r, ssr_view y1, ssr_view y, ssr_view m, int32_t dtype,
                                  int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    return up2x_bwd<0>(dy, r, y1, y, m, dtype, N, H, W, C, stream);
}
extern "C" int ssr_nearest2x_bwd(ssr_view dy, ssr_view r, ssr_view y1, ssr_view y, ssr_view m, int32_t dtype,
                                 int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    if (dtype == SSR_F32X3) dtype = SSR_F32;   // fp32 storage: only the matrix-core kernels differ
    return up2x_bwd<1>(dy, r, y1, y, m, dtype, N, H, W, C, stream);
}

extern "C" int ssr_spectral_norm(const ssr_sn_item* items_dev, int32_t n_items, int32_t max_rows, int32_t max_cols,
                                 int32_t power_iter, void* stream) {
    if (!items_dev || n_items <= 0 || max_rows <= 0 || max_cols <= 0) return SSR_EINVAL;
    if (power_iter) {
        hipLaunchKernelGGL(sn_wtu_kernel, dim3((max_cols + 63) / 64, n_items), dim3(256), 0, ST(stream), items_dev);
        SSR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sn_wv_kernel, dim3((max_rows + 3) / 4, n_items), dim3(256), 0, ST(stream), items_dev, power_iter);
    SSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_finish_kernel, dim3(n_items), dim3(256), 0, ST(stream), items_dev, power_iter);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_spectral_norm_bwd(const ssr_sn_bwd_item* items_dev, int32_t n_items, int32_t max_elems,
                                     void* stream) {
    if (!items_dev || n_items <= 0) return SSR_EINVAL;
    const int gx = grid_for(max_elems, 256 * 8, SSR_SN_BWD_SLOTS);
    // every item's tmp holds SSR_SN_BWD_SLOTS floats: block b of the dot kernel stores its partial sum in tmp[b] (nothing to zero)
    hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(gx, n_items), dim3(256), 0, ST(stream), items_dev);
    SSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(gx, n_items), dim3(256), 0, ST(stream), items_dev);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

// ------------------------------------------------------------------------------------------------
// USMSharp (BasicSR 1.4.2 basicsr/utils/img_process_util.py, constructed with its defaults at
// /root/reference/ssr/models/ssr_esrgan_model.py:31 and applied to the ground truth at :109):
//   blur = filter2D(img, G51)   (reflect pad 25, G51 = outer product of cv2.getGaussianKernel(51, 0): sigma = 8)
//   residual = img - blur;  mask = |residual| * 255 > threshold;  soft = filter2D(mask, G51)
//   sharp = clip(img + weight * residual, 0, 1);  out = soft * sharp + (1 - soft) * img
// One workgroup per (image, channel) plane held in LDS: the Gaussian is separable and so is reflect padding, so each
// filter2D is a row pass and a column pass over the plane (2 x 51 taps instead of 2601).
// ------------------------------------------------------------------------------------------------
constexpr int USM_R = 25, USM_K = 2 * USM_R + 1;
__device__ __forceinline__ int usm_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ __launch_bounds__(256) void usm_sharp_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                        float in_scale, float weight, float threshold) {
    extern __shared__ float usm_lds[];
    float* p0 = usm_lds;
    float* p1 = usm_lds + H * W;
    float* gk = usm_lds + 2 * H * W;
    const int tid = threadIdx.x, np = H * W;
    const float* __restrict__ sp = src + (size_t)blockIdx.x * np;
    float* __restrict__ dp = dst + (size_t)blockIdx.x * np;
    if (tid < USM_K) {   // cv2.getGaussianKernel(51, sigma <= 0): sigma = 0.3 * ((51 - 1) * 0.5 - 1) + 0.8 = 8
        const float d = (float)(tid - USM_R);
        gk[tid] = __expf(-d * d / 128.f);
    }
    for (int e = tid; e < np; e += 256) p0[e] = sp[e] * in_scale;
    __syncthreads();
    float gsum = 0.f;
    for (int k = 0; k < USM_K; ++k) gsum += gk[k];
    const float ginv = 1.f / gsum;
    auto row_pass = [&]() {   // p0 -> p1
        for (int e = tid; e < np; e += 256) {
            const int y = e / W, x = e - y * W;
            float s = 0.f;
            for (int k = 0; k < USM_K; ++k) s += gk[k] * p0[y * W + usm_reflect(x + k - USM_R, W)];
            p1[e] = s * ginv;
        }
    };
    auto col_at = [&](int e) {   // column pass of p1 at pixel e
        const int y = e / W, x = e - y * W;
        float s = 0.f;
        for (int k = 0; k < USM_K; ++k) s += gk[k] * p1[usm_reflect(y + k - USM_R, H) * W + x];
        return s * ginv;
    };
    row_pass();
    __syncthreads();
    // this thread's pixels e = tid + 256 j: the sharpened value is parked in dst (same thread re-reads it below)
    for (int e = tid; e < np; e += 256) {
        const float v = p0[e], res = v - col_at(e);
        dp[e] = fminf(fmaxf(v + weight * res, 0.f), 1.f);
        p0[e] = (fabsf(res) * 255.f > threshold) ? 1.f : 0.f;   // own pixel only: nobody else reads p0 in this phase
    }
    __syncthreads();
    row_pass();
    __syncthreads();
    for (int e = tid; e < np; e += 256) {
        const float soft = col_at(e), v = sp[e] * in_scale;
        dp[e] = soft * dp[e] + (1.f - soft) * v;
    }
}

extern "C" int ssr_usm_sharp(const float* src, float* dst, int32_t planes, int32_t H, int32_t W, float in_scale, float weight,
                             float threshold, void* stream) {
    if (!src || !dst || planes <= 0 || H <= USM_R || W <= USM_R) return SSR_EINVAL;   // reflect padding needs pad < size
    if ((long)H * W > 16384) return SSR_EUNSUP;               // plane must fit LDS twice (128 x 128 ground-truth tiles do)
    const size_t lds = ((size_t)2 * H * W + 64) * sizeof(float);
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(usm_sharp_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)((2 * 16384 + 64) * sizeof(float)));
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    hipLaunchKernelGGL(usm_sharp_kernel, dim3(planes), dim3(256), lds, ST(stream), src, dst, H, W, in_scale, weight, threshold);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_l1_loss(ssr_view a, ssr_view b, ssr_view grad, int32_t dtype, int64_t npix, int32_t C, float weight,
                           float* loss_out, void* stream) {
    const bool det = (dtype & SSR_DETERMINISTIC) != 0;      // loss_out = SSR_LOSS_SLOTS floats, one per block
    dtype &= ~SSR_DETERMINISTIC;
    if (dtype == SSR_F32X3) dtype = SSR_F32;   // fp32 storage: only the matrix-core kernels differ
    if (!a.p || !b.p || npix <= 0 || C <= 0) return SSR_EINVAL;
    const int g = grid_for(npix * C, 256 * 4, det ? SSR_LOSS_SLOTS : 1024);
    if (dtype == SSR_F32)
        hipLaunchKernelGGL(l1_loss_kernel<float>, dim3(g), dim3(256), 0, ST(stream), a, b, grad, (long)npix, C, weight,
                           loss_out, det);
    else if (dtype == SSR_BF16)
        hipLaunchKernelGGL(l1_loss_kernel<__bf16>, dim3(g), dim3(256), 0, ST(stream), a, b, grad, (long)npix, C, weight,
                           loss_out, det);
    else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_bce_logits_loss(ssr_view x, ssr_view grad, int32_t dtype, int64_t npix, float target, float weight,
                                   float* loss_out, float* mean_out, void* stream) {
    const bool det = (dtype & SSR_DETERMINISTIC) != 0;      // loss_out / mean_out = SSR_LOSS_SLOTS floats each, one per block
    dtype &= ~SSR_DETERMINISTIC;
    if (dtype == SSR_F32X3) dtype = SSR_F32;   // fp32 storage: only the matrix-core kernels differ
    if (!x.p || npix <= 0) return SSR_EINVAL;
    const int g = grid_for(npix, 256 * 4, det ? SSR_LOSS_SLOTS : 1024);
    if (dtype == SSR_F32)
        hipLaunchKernelGGL(bce_loss_kernel<float>, dim3(g), dim3(256), 0, ST(stream), x, grad, (long)npix, target,
                           weight, loss_out, mean_out, det);
    else if (dtype == SSR_BF16)
        hipLaunchKernelGGL(bce_loss_kernel<__bf16>, dim3(g), dim3(256), 0, ST(stream), x, grad, (long)npix, target,
                           weight, loss_out, mean_out, det);
    else return SSR_EUNSUP;
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

// dst[e] += sum_p src[p * stride + e], p = 0 .. parts-1 in that order: the fixed-order sum behind the deterministic weight-gradient
// mode (engine.WgradBatch: every pixel-range split of a layer writes its own partial gradient, one writer per element)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const ssr_reduce_item* __restrict__ items) {
    const ssr_reduce_item it = items[blockIdx.y];
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < it.n; e += (long)gridDim.x * blockDim.x) {
        float s = it.dst[e];
        for (int p = 0; p < it.parts; ++p) s += it.src[(long)p * it.stride + e];
        it.dst[e] = s;
    }
}
extern "C" int ssr_wgrad_reduce(const ssr_reduce_item* items_dev, int32_t n_items, int64_t max_elems, void* stream) {
    if (!items_dev || n_items <= 0 || max_elems <= 0) return SSR_EINVAL;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(max_elems, 256 * 4, 64), n_items), dim3(256), 0, ST(stream), items_dev);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_adam_step(const ssr_adam_args* a, void* stream) {
    if (!a || !a->param || !a->grad || !a->exp_avg || !a->exp_avg_sq || !a->lr || !a->step || a->n <= 0)
        return SSR_EINVAL;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(a->n, 256 * 4, 2048)), dim3(256), 0, ST(stream), *a);
    SSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, ST(stream), a->step);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_axpby_f32(float a, const float* x, float b, float* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return SSR_EINVAL;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, 256 * 4, 2048)), dim3(256), 0, ST(stream), a, x, b, y, (long)n);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_device_info(char* buf, int32_t buflen) {
    if (!buf || buflen <= 0) return SSR_EINVAL;
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        snprintf(buf, buflen, "no HIP device");
        return SSR_EINVAL;
    }
    snprintf(buf, buflen, "%s arch=%s CUs=%d lds=%zu clock=%dMHz mem=%zuGiB", p.name, p.gcnArchName,
             p.multiProcessorCount, p.sharedMemPerBlock, p.clockRate / 1000, p.totalGlobalMem >> 30);
    return SSR_OK;
}

// x -> (hi, lo) bf16 planes with x = hi + lo + O(2^-17 x): operands of the split-bf16 weight-gradient passes (SSR_F32X3)
namespace {
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ hi, __bf16* __restrict__ lo,
                                                         long n4) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[e];
        bf16x4 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = (__bf16)v[k];
            l[k] = (__bf16)(v[k] - (float)h[k]);
        }
        reinterpret_cast<bf16x4*>(hi)[e] = h;
        reinterpret_cast<bf16x4*>(lo)[e] = l;
    }
}
}  // namespace

extern "C" int ssr_split_bf16(const float* x, void* hi, void* lo, int64_t n, void* stream) {
    if (!x || !hi || !lo || n <= 0 || (n % 4) != 0) return SSR_EINVAL;
    hipLaunchKernelGGL(split_bf16_kernel, dim3(grid_for(n / 4, 256, 8192)), dim3(256), 0, ST(stream), x,
                       reinterpret_cast<__bf16*>(hi), reinterpret_cast<__bf16*>(lo), (long)(n / 4));
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

// all the buffers of a weight-gradient batch in ONE launch (round 5: the fp32x3 step issued 190 split launches of ~17 us - 3.2 ms -
// each a grid-stride loop of 16-byte loads and 8-byte stores at 3 TB/s): blockIdx.y = buffer, a thread moves 8 floats (two 16-byte
// loads -> one 16-byte store per plane)
namespace {
__global__ __launch_bounds__(256) void split_bf16_multi_kernel(const ssr_split_item* __restrict__ items) {
    const ssr_split_item it = items[blockIdx.y];
    const long n8 = it.n >> 3;
    const f32x4* __restrict__ x = reinterpret_cast<const f32x4*>(it.x);
    u32x4* __restrict__ hi = reinterpret_cast<u32x4*>(it.hi);
    u32x4* __restrict__ lo = reinterpret_cast<u32x4*>(it.lo);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += (long)gridDim.x * blockDim.x) {
        const f32x4 v0 = x[2 * e], v1 = x[2 * e + 1];
        bf16x8 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = (__bf16)v0[k];
            l[k] = (__bf16)(v0[k] - (float)h[k]);
            h[4 + k] = (__bf16)v1[k];
            l[4 + k] = (__bf16)(v1[k] - (float)h[4 + k]);
        }
        hi[e] = __builtin_bit_cast(u32x4, h);
        lo[e] = __builtin_bit_cast(u32x4, l);
    }
}
}  // namespace

extern "C" int ssr_split_bf16_multi(const ssr_split_item* items_dev, int32_t n_items, int64_t max_n, void* stream) {
    if (!items_dev || n_items <= 0 || max_n <= 0) return SSR_EINVAL;
    hipLaunchKernelGGL(split_bf16_multi_kernel, dim3(grid_for(max_n / 8, 256, 2048), n_items), dim3(256), 0, ST(stream), items_dev);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_abi_version(void) { return 3; }
