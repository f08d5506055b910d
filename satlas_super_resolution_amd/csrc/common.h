// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA 32x32 tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ssr_hip.h"
#include <type_traits>

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ---- split-operand matrix math (SSR_F32X3: bf16 pieces; SSR_F32H: fp16 pieces, include/ssr_hip.h) ----
// The kernels keep the pieces in bf16x8 / uint2 containers whatever the encoding; H selects the conversion and the MFMA.
template <bool H> __device__ __forceinline__ f32x16 split_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// four fp32 values -> four hi pieces, four lo pieces (x = hi + lo + O(2^-17 x) in bf16, O(2^-22 x) in fp16 above its subnormal range)
template <bool H> __device__ __forceinline__ void split_f32x4(const u32x4& v, uint2& hi, uint2& lo) {
    const f32x4 f = __builtin_bit_cast(f32x4, v);
    if constexpr (H) {
        f16x4 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = (_Float16)f[k];
            l[k] = (_Float16)(f[k] - (float)h[k]);
        }
        hi = __builtin_bit_cast(uint2, h);
        lo = __builtin_bit_cast(uint2, l);
    } else {
        bf16x4 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = (__bf16)f[k];
            l[k] = (__bf16)(f[k] - (float)h[k]);
        }
        hi = __builtin_bit_cast(uint2, h);
        lo = __builtin_bit_cast(uint2, l);
    }
}
constexpr float SSR_F32H_WSCALE = (float)(1 << SSR_F32H_WSHIFT), SSR_F32H_UNSCALE = 1.0f / (float)(1 << SSR_F32H_WSHIFT);

#define LRELU_SLOPE 0.2f

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int VEC = 4; };
template <> struct DT<__bf16> { static constexpr int VEC = 8; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : LRELU_SLOPE * v; }
// derivative recovered from the *output* of LeakyReLU (sign is preserved for slope > 0);
// at exactly 0 torch's leaky_relu_backward uses (x > 0 ? 1 : slope).
__device__ __forceinline__ float lrelu_grad_from_out(float out) { return out > 0.f ? 1.f : LRELU_SLOPE; }
// lrelu(v) == max(v, 0.2 v) as ONE v_max_f32: fmaxf() makes hipcc canonicalise the accumulator first (a second v_max per
// value; the branch-free epilogues are VALU-bound at 4 cycles per instruction).  NaNs propagate (0.2 * NaN = NaN).
__device__ __forceinline__ float lrelu_max(float v) {
    float r;
    const float s = LRELU_SLOPE * v;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(s));
    return r;
}
// v * lrelu_grad_from_out(m) for the two bf16 masks packed in one dword: integer compares on the raw bits (bf16 m > 0
// <=> its 16 bits, read as a signed integer, are > 0), no unpacking
__device__ __forceinline__ float lrelu_mask_lo(float v, unsigned m2) { return (int)(m2 << 16) > 0 ? v : LRELU_SLOPE * v; }
__device__ __forceinline__ float lrelu_mask_hi(float v, unsigned m2) { return (int)m2 > 0xFFFF ? v : LRELU_SLOPE * v; }

// One 16-byte LDS/global operand read feeds the matrix core:
//   fp32 : 4 x v_mfma_f32_32x32x2_f32  (lane (i,g) holds channels g*4+s, s = 0..3  -> 8 channels / read)
//   bf16 : 1 x v_mfma_f32_32x32x16_bf16 (lane (i,g) holds channels g*8+t, t = 0..7 -> 16 channels / read)
template <typename T> __device__ __forceinline__ void mma16(f32x16& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma16<float>(f32x16& acc, const u32x4& a, const u32x4& b) {
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<__bf16>(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                  0, 0, 0);
}

// C/D fragment of a 32x32 MFMA: lane l holds column (l & 31), rows (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15
__device__ __forceinline__ int mfma32_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: launchers remember it per device ordinal
// (one process per GPU is the design, but a process that drives a second GPU must not launch without the opt-in there)
constexpr int SSR_MAX_DEVICES = 32;
inline int ssr_device_ordinal() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SSR_MAX_DEVICES) dev = 0;
    return dev;
}

#define SSR_LAUNCH_CHECK()                       \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)
