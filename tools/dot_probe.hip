// semantics of __builtin_amdgcn_fdot2_f32_bf16 on gfx950 (v_dot2c_f32_bf16)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* o) {
    bf16x2 a = {(__bf16)1.0f, (__bf16)2.0f}, b = {(__bf16)3.0f, (__bf16)4.0f};
    float acc = 0.5f;
    acc = __builtin_amdgcn_fdot2_f32_bf16(a, b, acc, false);
    o[0] = acc;
    unsigned ua = 0x40003f80u, ub = 0x40804040u;   // (1.0, 2.0), (3.0, 4.0) packed lo,hi
    o[1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, ua), __builtin_bit_cast(bf16x2, ub), 0.f, false);
    float c2 = 10.f;
    for (int i = 0; i < 3; ++i) c2 = __builtin_amdgcn_fdot2_f32_bf16(a, b, c2, false);
    o[2] = c2;
}
int main() { float* d; hipMalloc(&d, 16); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("%f (expect 11.5)  %f (expect 11)  %f (expect 43)\n", h[0], h[1], h[2]); return 0; }
