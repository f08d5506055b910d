// Achievable HBM bandwidth of plain streaming kernels (read-only, write-only, copy) over buffers much larger than the
// 256 MB Infinity Cache: the floor the "HBM-bound" kernels of DESIGN.md are compared with.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hbm_probe.hip -o tools/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ p, size_t n, unsigned* out) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32x4 v = __builtin_nontemporal_load(p + i);
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}
__global__ __launch_bounds__(256) void wr(u32x4* __restrict__ p, size_t n) {
    const u32x4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
__global__ __launch_bounds__(256) void cp(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 2048, bytes = mb << 20, n = bytes / 16;
    u32x4 *a, *b; unsigned* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        float ms[3];
        for (int k = 0; k < 3; ++k) {
            for (int it = 0; it < 2; ++it) {
                hipEventRecord(e0);
                if (k == 0) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, a, n, o);
                if (k == 1) hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, b, n);
                if (k == 2) hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, a, b, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[k], e0, e1);
            }
        }
        printf("%zu MB, grid %5d: read %.2f TB/s  write %.2f TB/s  copy %.2f TB/s (read+write bytes)\n", mb, grid,
               bytes / ms[0] / 1e9, bytes / ms[1] / 1e9, 2.0 * bytes / ms[2] / 1e9);
    }
    return 0;
}
