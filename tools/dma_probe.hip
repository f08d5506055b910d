// Issue cost of LDS-DMA vs register staging for an 18 KB slab per 4-wave workgroup (per-wave cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, unsigned long long* out, int mode, int nslab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* base = src + (size_t)blockIdx.x * 1152 * nslab + lane;   // 18 KB = 1152 vectors per slab
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < nslab; ++s) {
        char* stage = smem + (s & 1) * 18432;
        const u32x4* sp = base + s * 1152;
        if (mode == 0) {
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {
                const int j = wave + 4 * jj;
                if (j < 18)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sp + j * 64),
                                                     (__attribute__((address_space(3))) void*)(stage + j * 1024), 16, 0, 0);
            }
        } else {
            u32x4 r[5];
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) { const int j = wave + 4 * jj; if (j < 18) r[jj] = sp[j * 64]; }
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) { const int j = wave + 4 * jj; if (j < 18) *reinterpret_cast<u32x4*>(stage + j * 1024 + lane * 16) = r[jj]; }
        }
        __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (smem[threadIdx.x] == 77) out[0] = 1;
}
int main() {
    const int nblk = 256, nslab = 26;
    u32x4* src; hipMalloc(&src, (size_t)nblk * 1152 * nslab * 16); hipMemset(src, 1, (size_t)nblk * 1152 * nslab * 16);
    unsigned long long *out, h[256]; hipMalloc(&out, 256 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40000);
    for (int mode = 0; mode < 2; ++mode) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(nblk), dim3(256), 36864, 0, src, out, mode, nslab);
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int b = 0; b < nblk; ++b) s += h[b];
        printf("%s: %.0f cycles per 18KB slab step (incl. barrier), %d slabs, L2-warm\n", mode ? "regs+ds_write" : "LDS-DMA", s / nblk / nslab, nslab);
    }
    return 0;
}
