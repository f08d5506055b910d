// Fused epilogue shared by the convolution kernels (contract: include/ssr_hip.h, ssr_conv_desc):
//   s0 = alpha * act(acc + bias) -> y0 ; s1 = s0 + beta1*r1 + beta2*r2 (+ y_old) -> y1 ; y = s1 * lrelu'(m)
// One lane owns one output channel `co` and the 16 pixel rows of its 32x32 MFMA C fragment.
//
// Two measured facts shape it (tools/conv_probe.hip, s_memtime stamps inside a generator-body launch):
//  (1) interleaving loads and stores makes hipcc put `s_waitcnt vmcnt(0)` (which on gfx950 also waits
//      for stores) in front of every load that may alias an earlier store: 16 serialized round trips.
//      -> ALL loads (residuals, old value, mask) are issued first, as unconditional straight-line code
//      on clamped addresses, and masked afterwards.
//  (2) 16 narrow stores per lane (one bf16 each, the natural MFMA C layout) cost ~8000 cycles per
//      wave — store-issue bound, ~45 % of a body launch.  -> the finished 32x32 tile is transposed
//      through a wave-private LDS slab and written as 16-byte vectors (8 bf16 channels of one pixel per
//      lane): 2 store instructions per wave instead of 16.
#pragma once
#include "common.h"

constexpr int EPI_STAGE_BYTES = 32 * 32 * 4;   // wave-private LDS slab (fp32 worst case)

// column of pixel slot pi (0..31) of a wave's 2 x 16-pixel tile.  ROT != 0 (csrc/conv_x3q.hip): the second row is rotated by ROT
// columns, so that the LDS rows a 16-lane read group touches are distinct mod 16 for every tap (conflict-free operand reads)
template <int ROT> __device__ __forceinline__ int epi_col(int pi) { return ROT ? ((pi & 15) + (pi >= 16 ? ROT : 0)) & 15 : (pi & 15); }

template <typename T, int ROT = 0>
__device__ __forceinline__ void conv_epilogue(const ssr_conv_desc& d, const f32x16& acc, int co_base, int n,
                                              int gy_row0, int gx0, int lane, char* stage) {
    constexpr int VEC = DT<T>::VEC;
    const int i = lane & 31, g = lane >> 5;
    const int co = co_base + i;
    T* __restrict__ yp = reinterpret_cast<T*>(d.y.p);
    T* __restrict__ y0p = reinterpret_cast<T*>(d.y0.p);
    T* __restrict__ y1p = reinterpret_cast<T*>(d.y1.p);
    const T* __restrict__ r1p = reinterpret_cast<const T*>(d.r1.p);
    const T* __restrict__ r2p = reinterpret_cast<const T*>(d.r2.p);
    const T* __restrict__ mp = reinterpret_cast<const T*>(d.m.p);
    const bool co_ok = co < d.Cout;
    const int cs = co_ok ? co : d.Cout - 1;                 // clamped channel for addressing
    const float bv = (d.bias && co_ok) ? d.bias[cs] : 0.f;
    const bool has_r1 = r1p && co < d.r1_nc, has_r2 = r2p && co < d.r2_nc;
    const bool has_m = mp && co >= d.m_c0 && co < d.m_c1;
    const bool has_acc = d.accumulate != 0;
    int po[16], pc[16];   // pixel index (-1 = outside the grid) and a clamped, always-addressable twin
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pi = mfma32_row(r, g);
        const int gy = gy_row0 + (pi >> 4), gx = gx0 + epi_col<ROT>(pi);
        const int cy = gy < d.Gh ? gy : d.Gh - 1, cx = gx < d.Gw ? gx : d.Gw - 1;
        pc[r] = (n * d.Ho + cy * d.oys + d.oyo) * d.Wo + cx * d.oxs + d.oxo;
        po[r] = (gy < d.Gh && gx < d.Gw) ? pc[r] : -1;
    }
    // ---- phase 1: every load ----
    float q1[16], q2[16], qa[16], qm[16];
    if (r1p) {
        const int c1 = d.r1.coff + (cs < d.r1_nc ? cs : d.r1_nc - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) q1[r] = to_f32(r1p[(size_t)pc[r] * d.r1.cs + c1]);
    }
    if (r2p) {
        const int c2 = d.r2.coff + (cs < d.r2_nc ? cs : d.r2_nc - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) q2[r] = to_f32(r2p[(size_t)pc[r] * d.r2.cs + c2]);
    }
    if (has_acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) qa[r] = to_f32(yp[(size_t)pc[r] * d.y.cs + d.y.coff + cs]);
    }
    if (mp) {
        const int cm = d.m.coff + (cs < d.m_c0 ? d.m_c0 : (cs < d.m_c1 ? cs : d.m_c1 - 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) qm[r] = to_f32(mp[(size_t)pc[r] * d.m.cs + cm]);
    }
    // ---- phase 2: arithmetic ----
    float s0[16], s1[16], s2[16];
    if (d.fix_list && d.act == SSR_ACT_LRELU && co_ok) {
        // split-bf16 mode: pre-activations at rounding level go on the list ssr_conv2d_fixup recomputes exactly (include/ssr_hip.h)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (po[r] >= 0 && fabsf(acc[r] + bv) < d.fix_thr) {
                const int k = atomicAdd(d.fix_list, 1);
                if (k < d.fix_cap) { d.fix_list[4 + 2 * k] = po[r]; d.fix_list[5 + 2 * k] = co; }
            }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r] + bv;
        if (d.act == SSR_ACT_LRELU) v = lrelu(v);
        else if (d.act == SSR_ACT_RELU) v = fmaxf(v, 0.f);
        v *= d.alpha;
        s0[r] = v;
        v += (has_r1 ? d.beta1 * q1[r] : 0.f) + (has_r2 ? d.beta2 * q2[r] : 0.f) + (has_acc ? qa[r] : 0.f);
        s1[r] = v;
        if (has_m) v *= d.m_relu ? (qm[r] > 0.f ? 1.f : 0.f) : lrelu_grad_from_out(qm[r]);
        s2[r] = v;
    }
    // ---- phase 3: stores ----
    auto store_scalar = [&](T* __restrict__ p, const ssr_view& vw, const float (&s)[16]) {
        if (!co_ok) return;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (po[r] >= 0) p[(size_t)po[r] * vw.cs + vw.coff + co] = from_f32<T>(s[r]);
    };
    auto store_wide = [&](T* __restrict__ p, const ssr_view& vw, const float (&s)[16]) {
        // transpose through the wave-private slab: [32 pixels][32 channels] of T, then 16-B vectors
        T* sl = reinterpret_cast<T*>(stage);
#pragma unroll
        for (int r = 0; r < 16; ++r) sl[mfma32_row(r, g) * 32 + i] = from_f32<T>(s[r]);
        constexpr int PARTS = 32 / VEC, NV = 32 * PARTS / 64;   // 16-B parts per pixel, vectors per lane
#pragma unroll
        for (int h = 0; h < NV; ++h) {
            const int v = h * 64 + lane;
            const int pix = v / PARTS, part = v - pix * PARTS;
            const int gy = gy_row0 + (pix >> 4), gx = gx0 + epi_col<ROT>(pix);
            const int c = co_base + part * VEC;
            const u32x4 val = *reinterpret_cast<const u32x4*>(sl + pix * 32 + part * VEC);
            if (gy < d.Gh && gx < d.Gw && c < d.Cout) {
                const size_t pp = (size_t)((n * d.Ho + gy * d.oys + d.oyo) * d.Wo + gx * d.oxs + d.oxo);
                *reinterpret_cast<u32x4*>(p + pp * vw.cs + vw.coff + c) = val;
            }
        }
    };
    auto wide_ok = [&](const ssr_view& vw) {
        return (d.Cout % VEC) == 0 && (vw.cs % VEC) == 0 && (vw.coff % VEC) == 0 && ((uintptr_t)vw.p % 16) == 0;
    };
    if (y0p) { if (wide_ok(d.y0)) store_wide(y0p, d.y0, s0); else store_scalar(y0p, d.y0, s0); }
    if (y1p) { if (wide_ok(d.y1)) store_wide(y1p, d.y1, s1); else store_scalar(y1p, d.y1, s1); }
    if (wide_ok(d.y)) store_wide(yp, d.y, s2); else store_scalar(yp, d.y, s2);
}
