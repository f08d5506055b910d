// Shared pieces of the weight-gradient kernels (fp32-MFMA and bf16-MFMA variants).
#pragma once
#include "common.h"

constexpr int WG_TH = 8, WG_TW = 16;   // pixel tile of one wgrad step (fp32 kernels)
// bf16 kernels: 16 rows for 3x3 layers (4 k-steps per MFMA wave per ring hand-over), 8 for the 4x4 stride-2 layers (LDS)
constexpr int wgrad_bf16_th(int KH) { return KH == 3 ? 16 : 8; }

// Write the per-wave accumulators D[row = co][col = ci] (32x32 MFMA C layout) of all taps to
// dW[co][ci][ky][kx] (fp32, OIHW).  3x3: the 4 waves hold partial sums over different pixels -> reduce
// through LDS scratch `red` (>= 4*16*64 floats) first; 4x4: wave w owns tap row ky = w.
template <int KH, int KW, bool SPLIT_TAPS, int NTAP>
__device__ __forceinline__ void wgrad_writeout(f32x16 (&acc)[NTAP], f32x16& accb, const ssr_wgrad_layer& L,
                                               const ssr_wgrad_item& it, float* red, bool do_bias) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    // ---- write-out: D[row = co][col = ci] ----
    float* __restrict__ dw = L.dw;
    const int KK = KH * KW;
    if (SPLIT_TAPS) {
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int tap = wave * KW + t;
            const int ci = it.ci0 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = it.co0 + mfma32_row(r, g);
                if (co < L.Cout && ci < L.Cin_w) {
                    const size_t idx = ((size_t)co * L.Cin_w + ci) * KK + tap;
                    const float v = L.alpha * acc[t][r];
                    if (it.atomic) atomicAdd(dw + idx, v); else dw[idx] += v;
                }
            }
        }
        if (do_bias && i == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = it.co0 + mfma32_row(r, g);
                if (co < L.Cout) {
                    const float v = L.alpha * accb[r];
                    if (it.atomic) atomicAdd(L.db + co, v); else L.db[co] += v;
                }
            }
        }
    } else {
        const bool bias_round = L.db != nullptr && it.ci0 == 0;
        auto round = [&](const f32x16& part, int t, bool is_bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = part[r];
            __syncthreads();
            for (int e = tid; e < 1024; e += 256) {
                const int r = e >> 6, ln = e & 63;
                const float s = red[(0 * 16 + r) * 64 + ln] + red[(1 * 16 + r) * 64 + ln] +
                                red[(2 * 16 + r) * 64 + ln] + red[(3 * 16 + r) * 64 + ln];
                const int co = it.co0 + mfma32_row(r, ln >> 5), col = ln & 31;
                const float v = L.alpha * s;
                if (!is_bias) {
                    const int ci = it.ci0 + col;
                    if (co < L.Cout && ci < L.Cin_w) {
                        const size_t idx = ((size_t)co * L.Cin_w + ci) * KK + t;
                        if (it.atomic) atomicAdd(dw + idx, v); else dw[idx] += v;
                    }
                } else if (col == 0 && co < L.Cout) {
                    if (it.atomic) atomicAdd(L.db + co, v); else L.db[co] += v;
                }
            }
            __syncthreads();
        };
        static_for<0, NTAP>([&](auto tc) { constexpr int t = decltype(tc)::value; round(acc[t], t, false); });
        if (bias_round) round(accb, 0, true);
    }
}

