// Thin-output 3x3 convolution (Cout <= 8, Cin <= 64; bf16 below, the fp32 modes further down): the logit / RGB heads and the dgrad into a 3-channel input.
//
// conv9 (64 -> 1), conv_last (64 -> 3) and conv0's dgrad (64 -> 3) at 128x128 cost 38..67 us each on the MFMA kernels
// (r01 rocprofv3): a 32-wide MFMA tile is 97 % / 91 % padding there and the launch is all staging and epilogue.  The work
// is 576 MACs per pixel per output channel — one v_dot2c_f32_bf16 per channel pair — and the layer is bound by reading
// its input once (33.5 MB at B = 16).  So: no matrix core.  One thread = one pixel; an 8 x 32 pixel tile's halo patch sits
// in LDS (144-byte rows: consecutive lanes 36 banks apart -> conflict-free ds_read_b128), the weights arrive through
// scalar loads (wave-uniform addresses) as SGPR operands of the dot instructions, fp32 accumulation, full ssr_conv_desc epilogue on <= 8 values per lane.
//
// Replaces nn.Conv2d forward at /root/reference/ssr/archs/discriminator_arch.py:40,69 (conv9), rrdbnet_arch.py:113,136
// (conv_last) and autograd's input gradient of discriminator_arch.py:28,44 (conv0) in the generator phase.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int CT_TH = 8, CT_TW = 32, CT_PH = CT_TH + 2, CT_PW = CT_TW + 2, CT_NPIX = CT_PH * CT_PW;   // 340
typedef __bf16 bf16x2t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ct_dot8(const u32x4& a, const u32x4& b, float acc) {
    // written out: with a loop-indexed subscript of the vector references hipcc 7.2 folded all four to element 0
    const unsigned a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2t, a0), __builtin_bit_cast(bf16x2t, b0), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2t, a1), __builtin_bit_cast(bf16x2t, b1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2t, a2), __builtin_bit_cast(bf16x2t, b2), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2t, a3), __builtin_bit_cast(bf16x2t, b3), acc, false);
    return acc;
}

template <int NCO>   // accumulators per thread: 1, 4 or 8 (Cout rounded up)
__global__ __launch_bounds__(256) void conv_thin_kernel(const ssr_conv_desc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int cin = d.Cin, nvec = cin >> 3;                   // 16-byte vectors per pixel
    const int rowb = cin * 2 + 16;                            // padded LDS row
    const int tiles_x = (d.Gw + CT_TW - 1) / CT_TW, tiles_y = (d.Gh + CT_TH - 1) / CT_TH;
    int b = blockIdx.x;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty_i * CT_TH, gx0 = tx_i * CT_TW;
    const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(d.x.p);
    const __bf16* __restrict__ wg = reinterpret_cast<const __bf16*>(d.w);
    // ---- stage the halo patch: vector v = pixel * nvec + part.  All loads are issued before the first store (a load - wait - store
    //      loop is a chain of memory latencies: DESIGN.md lesson 60), through clamped addresses; lanes outside the image keep zeros ----
    {
        constexpr int NSV = (CT_NPIX * 8 + 255) / 256;        // 11 at 64 channels
        const int total = CT_NPIX * nvec;
        const float rn = 1.0f / (float)nvec;                  // v / nvec without an integer division: exact for v < 2720, nvec <= 8
        u32x4 sv[NSV];
        unsigned okm = 0;
#pragma unroll
        for (int q = 0; q < NSV; ++q) {
            const int v = tid + q * 256;
            const int pix = (int)(((float)v + 0.5f) * rn), part = v - pix * nvec;
            const int py = pix / CT_PW, px = pix - py * CT_PW;
            const int ly = gy0 + py - 1, lx = gx0 + px - 1;
            const bool ok = v < total && ly >= 0 && ly < d.Hi && lx >= 0 && lx < d.Wi;
            okm |= ok ? 1u << q : 0u;
            sv[q] = *reinterpret_cast<const u32x4*>(xg + (ok ? ((size_t)(n * d.Hi + ly) * d.Wi + lx) * d.x.cs + d.x.coff + part * 8 : (size_t)0));
        }
#pragma unroll
        for (int q = 0; q < NSV; ++q) {
            const int v = tid + q * 256;
            const int pix = (int)(((float)v + 0.5f) * rn), part = v - pix * nvec;
            if (v < total) *reinterpret_cast<u32x4*>(smem + pix * rowb + part * 16) = (okm >> q) & 1u ? sv[q] : u32x4{0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();
    const int ty = tid >> 5, tx = tid & 31;
    float acc[NCO];
#pragma unroll
    for (int c = 0; c < NCO; ++c) acc[c] = 0.f;
    const char* pb = smem + (ty * CT_PW + tx) * rowb;
    // weights come straight from the packed global array: the address is uniform across the wave, so hipcc emits scalar
    // loads and v_dot2c takes them from SGPRs — one LDS read (the pixel) per 4 * NCO dot instructions
    for (int tap = 0; tap < 9; ++tap) {
        const char* pr = pb + ((tap / 3) * CT_PW + tap % 3) * rowb;
        for (int k = 0; k < nvec; ++k) {
            const u32x4 xv = *reinterpret_cast<const u32x4*>(pr + k * 16);
            const __bf16* wk = wg + ((size_t)((k >> 2) * 9 + tap) * d.CoutPad) * 32 + (k & 3) * 8;
#pragma unroll
            for (int c = 0; c < NCO; ++c) {
                const u32x4 wv = *reinterpret_cast<const u32x4*>(wk + c * 32);
                acc[c] = ct_dot8(xv, wv, acc[c]);
            }
        }
    }
    // ---- epilogue (ssr_conv_desc contract) on this pixel's Cout values ----
    const int gy = gy0 + ty, gx = gx0 + tx;
    if (gy >= d.Gh || gx >= d.Gw) return;
    const size_t pp = (size_t)(n * d.Ho + gy * d.oys + d.oyo) * d.Wo + gx * d.oxs + d.oxo;
    __bf16* __restrict__ yp = reinterpret_cast<__bf16*>(d.y.p);
    __bf16* __restrict__ y0p = reinterpret_cast<__bf16*>(d.y0.p);
    __bf16* __restrict__ y1p = reinterpret_cast<__bf16*>(d.y1.p);
    const __bf16* __restrict__ r1p = reinterpret_cast<const __bf16*>(d.r1.p);
    const __bf16* __restrict__ r2p = reinterpret_cast<const __bf16*>(d.r2.p);
    const __bf16* __restrict__ mp = reinterpret_cast<const __bf16*>(d.m.p);
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
        if (c >= d.Cout) break;
        float v = acc[c] + (d.bias ? d.bias[c] : 0.f);
        if (d.act == SSR_ACT_LRELU) v = lrelu(v);
        v *= d.alpha;
        if (y0p) y0p[pp * d.y0.cs + d.y0.coff + c] = (__bf16)v;
        if (r1p && c < d.r1_nc) v += d.beta1 * (float)r1p[pp * d.r1.cs + d.r1.coff + c];
        if (r2p && c < d.r2_nc) v += d.beta2 * (float)r2p[pp * d.r2.cs + d.r2.coff + c];
        if (d.accumulate) v += (float)yp[pp * d.y.cs + d.y.coff + c];
        if (y1p) y1p[pp * d.y1.cs + d.y1.coff + c] = (__bf16)v;
        if (mp && c >= d.m_c0 && c < d.m_c1) v *= lrelu_grad_from_out((float)mp[pp * d.m.cs + d.m.coff + c]);
        yp[pp * d.y.cs + d.y.coff + c] = (__bf16)v;
    }
}

// ---- fp32 storage (SSR_F32 exact, SSR_F32X3 split-bf16 weights), round 5 ----
// The same layers in the fp32 modes ran on the pipelined MFMA kernels: 111-117 us each at B = 32 (conv9 x 3, conv_last, conv0's
// dgrad: 0.56 ms of the fp32x3 step) for 0.6-4.8 GFLOP - a 32-wide MFMA tile is 91-97 % padding there.  One thread = one pixel again:
//   * the input is staged 32 channels at a time (144-byte rows: consecutive lanes 36 banks apart -> conflict-free ds_read_b128);
//   * the layer's weights become ONE fp32 table in LDS, [tap][ci][NCOP] (NCOP = 1, 4 or 8), built once per workgroup from the packed
//     rows [chunk16][tap][CoutPad][16]: plain fp32 in the exact mode, w = hi + lo of the [16 hi | 16 lo] bf16 rows in the split mode
//     (activations stay exact fp32: at least as accurate as the three-product MFMA form, 2^-17 per weight);
//   * the inner loop is a pixel vector (4 channels) + the weight vectors of those channels - all lanes read the SAME address:
//     a broadcast - and one fp32 FMA per product.
// (A first form read the packed rows with 64-byte scalar loads and converted hi / lo in the scalar unit: 165 us per launch - three
//  serialized scalar-load latencies per (tap, chunk) and twice the FMAs, call r05o; with the LDS table 107 us; with all staging loads of
//  a pass in flight and the next pass requested before the current one is contracted 79 us, call r05v.)
constexpr int CTF_CB = 32, CTF_ROWB = CTF_CB * 4 + 16;        // channels staged per pass, bytes per LDS row
constexpr int CTF_PATCH = CT_NPIX * CTF_ROWB;                 // 48,960 B

// X3: 0 = plain fp32 rows (SSR_F32), 1 = [16 hi | 16 lo] bf16 rows (SSR_F32X3), 2 = fp16 rows of 2^SSR_F32H_WSHIFT w (SSR_F32H)
template <int NCO, int X3>
__global__ __launch_bounds__(256) void conv_thin_f32_kernel(const ssr_conv_desc d) {
    constexpr int NCOP = NCO == 1 ? 1 : NCO <= 4 ? 4 : 8;     // weight-table columns (a 16-byte vector holds 4)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem + CTF_PATCH);   // [tap][cin16][NCOP], cin16 = Cin rounded up to 16
    const int tid = threadIdx.x;
    const int cin = d.Cin, cin16 = (cin + 15) & ~15;
    const int tiles_x = (d.Gw + CT_TW - 1) / CT_TW, tiles_y = (d.Gh + CT_TH - 1) / CT_TH;
    int b = blockIdx.x;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty_i * CT_TH, gx0 = tx_i * CT_TW;
    const float* __restrict__ xg = reinterpret_cast<const float*>(d.x.p);
    const char* __restrict__ wg = reinterpret_cast<const char*>(d.w);
    const int ty = tid >> 5, tx = tid & 31;
    const char* pb = smem + (ty * CT_PW + tx) * CTF_ROWB;
    float acc[NCOP];
#pragma unroll
    for (int c = 0; c < NCOP; ++c) acc[c] = 0.f;
    // ---- staging of 32 channels of the halo patch: vector q of a thread = slot tid + 256 q = pixel slot / 8, part slot % 8.  ALL loads of a
    //      pass are issued before the first store (a load - wait - store loop costs one memory latency per vector: 11 in a row, twice),
    //      through clamped addresses (a lane outside the image or the channel range reads pixel 0 and keeps zeros), and the next pass
    //      is requested before the current one is contracted ----
    constexpr int NSV = (CT_NPIX * 8 + 255) / 256;             // 11
    int sofs[NSV];                                             // element offset of the vector's pixel (-1: zeros)
#pragma unroll
    for (int q = 0; q < NSV; ++q) {
        const int v = tid + q * 256, pix = v >> 3;
        const int py = pix / CT_PW, px = pix - py * CT_PW;
        const int ly = gy0 + py - 1, lx = gx0 + px - 1;
        const bool ok = v < CT_NPIX * 8 && ly >= 0 && ly < d.Hi && lx >= 0 && lx < d.Wi;
        sofs[q] = ok ? ((n * d.Hi + ly) * d.Wi + lx) * d.x.cs + d.x.coff + (v & 7) * 4 : -1;
    }
    u32x4 sv[NSV];
    auto load_pass = [&](int c0) {                              // (the zeroing waits for the data: it happens at the store)
#pragma unroll
        for (int q = 0; q < NSV; ++q) {
            const bool ok = sofs[q] >= 0 && c0 + ((tid + q * 256) & 7) * 4 < cin;
            sv[q] = *reinterpret_cast<const u32x4*>(xg + (ok ? sofs[q] + c0 : 0));
        }
    };
    auto store_pass = [&](int c0) {
#pragma unroll
        for (int q = 0; q < NSV; ++q) {
            const int v = tid + q * 256;
            const bool ok = sofs[q] >= 0 && c0 + (v & 7) * 4 < cin;
            if (v < CT_NPIX * 8) *reinterpret_cast<u32x4*>(smem + (v >> 3) * CTF_ROWB + (v & 7) * 16) = ok ? sv[q] : u32x4{0u, 0u, 0u, 0u};
        }
    };
    load_pass(0);
    // ---- the weight table (padded channels / outputs: the packed rows hold zeros there) ----
    for (int e = tid; e < 9 * cin16 * NCOP; e += 256) {
        const int c = e % NCOP, q = e / NCOP, ci = q % cin16, tap = q / cin16;
        float w = 0.f;
        if (c < NCO) {
            const char* row = wg + ((size_t)((ci >> 4) * 9 + tap) * d.CoutPad + c) * 64;
            if (X3 == 2) {
                const _Float16 h = reinterpret_cast<const _Float16*>(row)[ci & 15], l = reinterpret_cast<const _Float16*>(row)[16 + (ci & 15)];
                w = ((float)h + (float)l) * SSR_F32H_UNSCALE;
            } else if (X3 == 1) {
                const unsigned short h = reinterpret_cast<const unsigned short*>(row)[ci & 15], l = reinterpret_cast<const unsigned short*>(row)[16 + (ci & 15)];
                w = __builtin_bit_cast(float, (unsigned)h << 16) + __builtin_bit_cast(float, (unsigned)l << 16);
            } else {
                w = reinterpret_cast<const float*>(row)[ci & 15];
            }
        }
        wl[e] = w;
    }
    store_pass(0);
    __syncthreads();                                           // patch pass 0 + the weight table
    for (int c0 = 0; c0 < cin; c0 += CTF_CB) {
        const bool more = c0 + CTF_CB < cin;
        if (more) load_pass(c0 + CTF_CB);
        const int nk = (min(cin16 - c0, CTF_CB)) >> 2;         // 4-channel groups of this pass: 4 or 8
        for (int tap = 0; tap < 9; ++tap) {
            const char* pr = pb + ((tap / 3) * CT_PW + tap % 3) * CTF_ROWB;
            const float* wt = wl + (tap * cin16 + c0) * NCOP;
            for (int k = 0; k < nk; ++k) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(pr + k * 16);
                if constexpr (NCOP == 1) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wt + 4 * k);          // the 4 channels' weights: one broadcast read
                    acc[0] = fmaf(xv.x, w.x, acc[0]); acc[0] = fmaf(xv.y, w.y, acc[0]);
                    acc[0] = fmaf(xv.z, w.z, acc[0]); acc[0] = fmaf(xv.w, w.w, acc[0]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xj = j == 0 ? xv.x : j == 1 ? xv.y : j == 2 ? xv.z : xv.w;
#pragma unroll
                        for (int c4 = 0; c4 < NCOP / 4; ++c4) {
                            const f32x4 w = *reinterpret_cast<const f32x4*>(wt + (4 * k + j) * NCOP + 4 * c4);
                            acc[4 * c4] = fmaf(xj, w.x, acc[4 * c4]); acc[4 * c4 + 1] = fmaf(xj, w.y, acc[4 * c4 + 1]);
                            acc[4 * c4 + 2] = fmaf(xj, w.z, acc[4 * c4 + 2]); acc[4 * c4 + 3] = fmaf(xj, w.w, acc[4 * c4 + 3]);
                        }
                    }
                }
            }
        }
        if (more) {
            __syncthreads();                                  // everyone is finished with these 32 channels
            store_pass(c0 + CTF_CB);
            __syncthreads();
        }
    }
    // ---- epilogue (ssr_conv_desc contract) on this pixel's Cout values ----
    const int gy = gy0 + ty, gx = gx0 + tx;
    if (gy >= d.Gh || gx >= d.Gw) return;
    const size_t pp = (size_t)(n * d.Ho + gy * d.oys + d.oyo) * d.Wo + gx * d.oxs + d.oxo;
    float* __restrict__ yp = reinterpret_cast<float*>(d.y.p);
    float* __restrict__ y0p = reinterpret_cast<float*>(d.y0.p);
    float* __restrict__ y1p = reinterpret_cast<float*>(d.y1.p);
    const float* __restrict__ r1p = reinterpret_cast<const float*>(d.r1.p);
    const float* __restrict__ r2p = reinterpret_cast<const float*>(d.r2.p);
    const float* __restrict__ mp = reinterpret_cast<const float*>(d.m.p);
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
        if (c >= d.Cout) break;
        float v = acc[c] + (d.bias ? d.bias[c] : 0.f);
        if (d.act == SSR_ACT_LRELU) v = lrelu(v);
        else if (d.act == SSR_ACT_RELU) v = fmaxf(v, 0.f);
        v *= d.alpha;
        if (y0p) y0p[pp * d.y0.cs + d.y0.coff + c] = v;
        if (r1p && c < d.r1_nc) v += d.beta1 * r1p[pp * d.r1.cs + d.r1.coff + c];
        if (r2p && c < d.r2_nc) v += d.beta2 * r2p[pp * d.r2.cs + d.r2.coff + c];
        if (d.accumulate) v += yp[pp * d.y.cs + d.y.coff + c];
        if (y1p) y1p[pp * d.y1.cs + d.y1.coff + c] = v;
        if (mp && c >= d.m_c0 && c < d.m_c1) {
            const float mv = mp[pp * d.m.cs + d.m.coff + c];
            v *= d.m_relu ? (mv > 0.f ? 1.f : 0.f) : lrelu_grad_from_out(mv);
        }
        yp[pp * d.y.cs + d.y.coff + c] = v;
    }
}

template <int NCO, int X3>
int launch_thin_f32(const ssr_conv_desc& d, hipStream_t st) {
    constexpr int NCOP = NCO == 1 ? 1 : NCO <= 4 ? 4 : 8;
    const size_t lds = (size_t)CTF_PATCH + (size_t)9 * ((d.Cin + 15) & ~15) * NCOP * 4;     // <= 67,392 B
    auto kern = conv_thin_f32_kernel<NCO, X3>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CTF_PATCH + 9 * 64 * 8 * 4);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    const int tiles = d.N * ((d.Gh + CT_TH - 1) / CT_TH) * ((d.Gw + CT_TW - 1) / CT_TW);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, st, d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}
template <int X3>
int launch_thin_f32_by_cout(const ssr_conv_desc& d, hipStream_t st) {
    if (d.Cout == 1) return launch_thin_f32<1, X3>(d, st);
    if (d.Cout <= 3) return launch_thin_f32<3, X3>(d, st);
    if (d.Cout == 4) return launch_thin_f32<4, X3>(d, st);
    return launch_thin_f32<8, X3>(d, st);
}

template <int NCO>
int launch_thin(const ssr_conv_desc& d, hipStream_t st) {
    const size_t lds = (size_t)CT_NPIX * (d.Cin * 2 + 16);
    auto kern = conv_thin_kernel<NCO>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           CT_NPIX * (64 * 2 + 16) + 9 * 8 * 64 * 2);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    const int tiles = d.N * ((d.Gh + CT_TH - 1) / CT_TH) * ((d.Gw + CT_TW - 1) / CT_TW);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, st, d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

}  // namespace

bool ssr_conv_thin_shape_ok(const ssr_conv_desc& d) {
    const bool f32 = d.dtype == SSR_F32 || d.dtype == SSR_F32X3 || d.dtype == SSR_F32H;
    if (d.dtype != SSR_BF16 && !f32) return false;
    if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad_y == 1 && d.pad_x == 1) || d.x2.p || d.up != 1 || d.fix_list) return false;
    if (d.Cin > 64 || (d.Cin % 8) != 0 || d.Cout > 8 || d.Cout < 1) return false;
    if (d.Gh != d.Hi || d.Gw != d.Wi) return false;
    if (f32)   // (the fp32 form keeps 32-bit element offsets of its staging vectors)
        return (d.x.cs % 4) == 0 && (d.x.coff % 4) == 0 && ((uintptr_t)d.x.p % 16) == 0 && ((uintptr_t)d.w % 16) == 0 &&
               (long)d.N * d.Hi * d.Wi * d.x.cs < 0x7fffff00L;
    return (d.x.cs % 8) == 0 && (d.x.coff % 8) == 0 && ((uintptr_t)d.x.p % 16) == 0 && ((uintptr_t)d.w % 16) == 0;
}

bool ssr_conv_thin_qualifies(const ssr_conv_desc& d) {
    static const bool off = [] { const char* e = getenv("SSR_CONV_THIN"); return e && e[0] == '0'; }();
    if (off || !ssr_conv_thin_shape_ok(d)) return false;
    // fp32 modes: by the LAYER's grid only (64 x 64 pixels and up) - an image's bytes must not depend on how many images are launched
    // together (whole-tile inference deals chunks to ranks and batches; conv_last is such a layer).  Measured in the fp32x3 step
    // (call r05v, B = 32, 128 x 128): 79 us per launch against 117 us on the MFMA kernel; SSR_CONV_THIN_F32=0 switches it off.
    if (d.dtype != SSR_BF16) {
        static const bool off32 = [] { const char* e = getenv("SSR_CONV_THIN_F32"); return e && e[0] == '0'; }();
        return !off32 && (long)d.Gh * d.Gw >= 4096;
    }
    return (long)d.N * d.Gh * d.Gw >= 65536;                  // enough 256-pixel tiles to fill the chip
}

bool ssr_conv_thin_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force) {
    if (force ? !ssr_conv_thin_shape_ok(d) : !ssr_conv_thin_qualifies(d)) return false;
    if (d.dtype == SSR_F32H) { *rc = launch_thin_f32_by_cout<2>(d, st); return true; }
    if (d.dtype == SSR_F32X3) { *rc = launch_thin_f32_by_cout<1>(d, st); return true; }
    if (d.dtype == SSR_F32) { *rc = launch_thin_f32_by_cout<0>(d, st); return true; }
    if (d.Cout == 1) *rc = launch_thin<1>(d, st);
    else if (d.Cout <= 4) *rc = launch_thin<4>(d, st);
    else *rc = launch_thin<8>(d, st);
    return true;
}
