// Shared pieces of the register-tiled split-bf16 body kernels: csrc/conv_x3r.hip (one convolution per launch) and csrc/conv_x3c.hip (a
// dependent CHAIN of convolutions in one persistent launch).  Tile geometry, LDS ring layout, the counter poll and the straight-line
// epilogues of the dense block in the transposed domain; see conv_x3r.hip for the design.
#pragma once
#include "conv_epilogue.h"
#include <cstdlib>

namespace {

constexpr int XR_NMFMA = 4, XR_NPROD = 4, XR_NTHR = 64 * (XR_NMFMA + XR_NPROD);      // (the 32-channel form; XrT<NTW, NU>::NMF / NTHR in general)
constexpr int XR_ROWB = 80, XR_PH = 10, XR_PW = 18, XR_NPIX = XR_PH * XR_PW;       // 8 x 16 tile + halo, rows [16 hi | 16 lo | pad]
constexpr int XR_SUB = XR_NPIX * XR_ROWB;                      // one 16-channel chunk of the patch: 14,400 B
constexpr int XR_NS = 8;                                       // ring stages (one chunk each)
constexpr int XR_RING = XR_NS * XR_SUB;                        // 115,200 B
constexpr int XR_PV = XR_NPIX * 4;                             // 16-byte vectors of a chunk's patch: 720
constexpr int XR_NPV = (XR_PV + 255) / 256;                    // per producer thread: 3
#ifndef XR_PQ_DEPTH
#define XR_PQ_DEPTH 4
#endif
constexpr int XR_PQ = XR_PQ_DEPTH;                             // patch chunks a producer keeps in flight in registers
constexpr int XR_ROT = 14;                                     // rotation of a tile's second pixel row (32 - PW): conflict-free fragment reads
constexpr int XR_OOB = 0x7ffffff0;
constexpr int XR_SPIN_MAX = 1 << 22;                           // polls of a flag before a wave TRAPS (~0.3 s; a hand-over takes ~1 us)
constexpr int XR_TILEB = 2 * XR_PW * XR_ROWB;                  // LDS bytes between the pixel tiles of a wave (two patch rows): 2,880
constexpr int XR_SLOT = 32 * 32 * 4;                           // one 32 x 32 fp32 partial tile / one transpose slab

// NTW = 32-channel output tiles per MFMA wave, NU = groups of such waves: NU * 4 MFMA waves (K quarter = wave & 3, group = wave >> 2) + 4
// producer waves, NTW * NU * 32 output channels per workgroup.  Instantiated: <1, 1> the 32-channel layers; <2, 1> the 64-channel layers on
// four waves with two channel tiles each (default); <1, 2> the 64-channel layers on EIGHT MFMA waves (two per SIMD; every wave one channel
// tile: SSR_X3_REGTILE_NT2=8, faster alone, not in the step - conv_x3r.hip, xr_wide_form).
template <int NTW, int NU = 1, int TH = 8> struct XrT {
    static_assert(TH == 8 || TH == 4, "tile height");
    // tile geometry: TH x 16 pixels = MT 32-pixel MFMA tiles per wave (TH = 4: the half-height tiles of launches that would leave half
    // the chip idle - per-GPU batch 16 on the 32 x 32 body; the same products in the same order per pixel, tests/test_gpu_conv_x3.py)
    static constexpr int MT = TH / 2;
    static constexpr int PH = TH + 2, NPIX = PH * XR_PW;       // tile + halo
    static constexpr int SUB = NPIX * XR_ROWB;                 // one 16-channel chunk of the patch: 14,400 / 8,640 B
    static constexpr int RING = XR_NS * SUB;
    static constexpr int PV = NPIX * 4, NPV = (PV + 255) / 256;         // 16-byte vectors of a chunk's patch; per producer thread: 3 / 2
    static constexpr int NTT = NTW * NU;                       // channel tiles of the workgroup
    static constexpr int BN = 32 * NTT;
    static constexpr int NMF = 4 * NU;                         // MFMA waves
    static constexpr int NTHR = 64 * (NMF + 4);
    static constexpr int WR = (NTW == 1 && NU == 1) ? 6 : 3;   // (chunk, tap) steps of weight fragments in flight per wave (8 NTW registers each)
#ifndef XR_TPS1
#define XR_TPS1 4
#endif
    static constexpr int TPS = (NTW == 1 ? XR_TPS1 : 2) < MT ? (NTW == 1 ? XR_TPS1 : 2) : MT;      // pixel tiles per fragment set
    static constexpr int SPT = MT / TPS;                       // sub-steps per tap
    static constexpr int NSETS = SPT == 1 ? 2 : 3;             // fragment sets in registers; reads run NSETS - 1 sub-steps ahead
    static constexpr int SLOTS = 4 * MT * NTT;                 // [source K quarter][pixel tile][channel tile] partial tiles of the K-quarter sum
    static constexpr int RED = SLOTS * XR_SLOT;                // 64 / 128 KB (the ring is dead by then)
    static constexpr int CTL = RED > RING ? RED : RING;        // control words behind both: pdone[4] | cdone[NMF]
    static constexpr int LDS = CTL + 256;
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t xr_rsrc(const void* p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffff00L ? 0x7fffff00L : (bytes < 0 ? 0 : bytes)), 0x00020000);
}
__device__ __forceinline__ void xr_split4(const u32x4& v, uint2& hi, uint2& lo) {
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
typedef __attribute__((address_space(3))) int* xr_lds_int;

// min of the four counters at LDS address `a` (one ds_read_b128; xr_minN<8>: eight counters, two reads).  Inline asm: a compiler-visible LDS read would make hipcc drain
// the wave's global loads first (vmcnt(0)); "=&v": the output must not share registers with the address
__device__ __forceinline__ int xr_min4(int a) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    return __builtin_amdgcn_readfirstlane((int)min(min(v[0], v[1]), min(v[2], v[3])));
}

template <int N> __device__ __forceinline__ int xr_minN(int a) {
    if constexpr (N == 4) return xr_min4(a);
    else {
        u32x4 v, w;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v), "=&v"(w) : "v"(a) : "memory");
        const unsigned m = min(min(min(v[0], v[1]), min(v[2], v[3])), min(min(w[0], w[1]), min(w[2], w[3])));
        return __builtin_amdgcn_readfirstlane((int)m);
    }
}

enum { XR_EP_LRELU = 0, XR_EP_LIN = 1, XR_EP_MASK = 2, XR_EP_GENERIC = 3 };

// the dense block's epilogues in the transposed domain: lane = (pixel slot lane >> 3 (+ 8 h), channels 4 (lane & 7) .. + 3)
// AUX: cache bits of the result stores (0 plain; 16 = sc1, write-through: the chain kernel hands its results to other workgroups)
template <int EP, int AUX = 0>
__device__ __forceinline__ void xr_epilogue(const ssr_conv_desc& d, const f32x16& acc, int co_base, int n, int gy_row0, int gx0, int lane,
                                             char* slab) {
    const int i = lane & 31, g = lane >> 5;
    const int part = lane & 7, c = co_base + part * 4;
    const bool cok = c < d.Cout;
    const long npix = (long)d.N * d.Ho * d.Wo * 4;
    const __amdgpu_buffer_rsrc_t rs_y = xr_rsrc(d.y.p, npix * d.y.cs);
    int pp[4];
    bool ok[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int pix = (lane >> 3) + 8 * h;
        const int gy = gy_row0 + (pix >> 4), gx = gx0 + epi_col<XR_ROT>(pix);
        ok[h] = cok && gy < d.Gh && gx < d.Gw;
        pp[h] = (n * d.Ho + gy) * d.Wo + gx;
    }
    // ---- every load first ----
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (d.bias) {
        const __amdgpu_buffer_rsrc_t rs_b = xr_rsrc(d.bias, (long)d.Cout * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) bv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_b, (c + k) * 4, 0, 0));   // beyond Cout: zeros
    }
    u32x4 q1[4], q2[4];
    if constexpr (EP == XR_EP_LIN) {
        const __amdgpu_buffer_rsrc_t rs_r1 = xr_rsrc(d.r1.p, d.r1.p ? npix * d.r1.cs : 0), rs_r2 = xr_rsrc(d.r2.p, d.r2.p ? npix * d.r2.cs : 0);
        const bool h1 = d.r1.p != nullptr, h2 = d.r2.p != nullptr;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            q1[h] = __builtin_amdgcn_raw_buffer_load_b128(rs_r1, ok[h] && h1 ? (pp[h] * d.r1.cs + d.r1.coff + c) * 4 : XR_OOB, 0, 0);
            q2[h] = __builtin_amdgcn_raw_buffer_load_b128(rs_r2, ok[h] && h2 ? (pp[h] * d.r2.cs + d.r2.coff + c) * 4 : XR_OOB, 0, 0);
        }
    }
    if constexpr (EP == XR_EP_MASK) {
        const __amdgpu_buffer_rsrc_t rs_m = xr_rsrc(d.m.p, npix * d.m.cs);
#pragma unroll
        for (int h = 0; h < 4; ++h) q1[h] = __builtin_amdgcn_raw_buffer_load_b128(rs_m, ok[h] ? (pp[h] * d.m.cs + d.m.coff + c) * 4 : XR_OOB, 0, 0);
    }
    // ---- transpose: [32 pixel slots][32 channels] fp32 ----
    float* sl = reinterpret_cast<float*>(slab);
#pragma unroll
    for (int r = 0; r < 16; ++r) sl[mfma32_row(r, g) * 32 + i] = acc[r];
    const float alpha = d.alpha, beta1 = d.beta1, beta2 = d.beta2;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int pix = (lane >> 3) + 8 * h;
        f32x4 v = *reinterpret_cast<const f32x4*>(sl + pix * 32 + part * 4);
        if constexpr (EP == XR_EP_LRELU) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = lrelu_max(v[k] + bv[k]);
        } else if constexpr (EP == XR_EP_LIN) {
            const f32x4 a = __builtin_bit_cast(f32x4, q1[h]), b = __builtin_bit_cast(f32x4, q2[h]);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = alpha * (v[k] + bv[k]) + (beta1 * a[k] + beta2 * b[k]);
        } else {
            const f32x4 m = __builtin_bit_cast(f32x4, q1[h]);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (v[k] + bv[k]) * lrelu_grad_from_out(m[k]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_y, ok[h] ? (pp[h] * d.y.cs + d.y.coff + c) * 4 : XR_OOB, 0, AUX);
    }
}

}  // namespace
