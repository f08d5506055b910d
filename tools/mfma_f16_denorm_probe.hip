// Does the gfx950 matrix core keep fp16 SUBNORMAL inputs, and does v_cvt_f16_f32 produce them?  (The fp16-split forward arithmetic of
// DESIGN.md section 2 needs both: the lo part of a small activation is an fp16 subnormal.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_f16_denorm_probe.hip -o tools/mfma_f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const float* in, float* out, unsigned short* bits) {
    const int lane = threadIdx.x;
    const float a = in[0], b = in[1];
    const _Float16 ah = (_Float16)a, bh = (_Float16)b;       // v_cvt_f16_f32
    h8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = ah; bv[i] = bh; }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    if (lane == 0) {
        out[0] = acc[0];
        out[1] = (float)ah;
        unsigned short u; __builtin_memcpy(&u, &ah, 2); bits[0] = u;
        const float lo = a - (float)ah;
        const _Float16 lh = (_Float16)lo;
        out[2] = (float)lh;
        __builtin_memcpy(&u, &lh, 2); bits[1] = u;
    }
}
int main() {
    float *din, *dout; unsigned short* dbits;
    hipMalloc(&din, 8); hipMalloc(&dout, 16); hipMalloc(&dbits, 4);
    const float cases[][2] = {{9.5367431640625e-07f, 1024.f}, {3.0e-6f, 1.f}, {1.0e-7f, 4096.f}, {0.1f, 1.f}, {0.001f, 1.f}};
    for (auto& c : cases) {
        hipMemcpy(din, c, 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dbits);
        float o[3]; unsigned short b[2];
        hipMemcpy(o, dout, 12, hipMemcpyDeviceToHost); hipMemcpy(b, dbits, 4, hipMemcpyDeviceToHost);
        printf("a = %.9g (fp16 bits 0x%04x -> %.9g), b = %g: mfma sum over K=16 = %.9g (kept subnormals: %.9g); lo part of a: bits 0x%04x = %.9g\n",
               c[0], b[0], o[1], c[1], o[0], 16.0 * (double)o[1] * c[1], b[1], o[2]);
    }
    return 0;
}
