// LDS throughput of ds_read_b64_tr_b16 under different lane -> address patterns (which lanes does the LDS pipe serve together?).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tr_bw_probe.hip -o tools/tr_bw_probe ;  tools/tr_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ __launch_bounds__(512) void k(int pattern, int iters, unsigned long long* out, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<int*>(smem)[i] = i;
    __syncthreads();
    const int g = lane >> 5, t16 = lane & 15, hs = (lane >> 4) & 1;
    const int px = 8 * g + (t16 >> 2), ch2 = 32 * hs + 8 * (t16 & 3);   // byte offset of the channel quad
    int addr;
    switch (pattern) {
        case 0: addr = px * 64 + ch2; break;                              // the kernels' pattern
        case 1: addr = px * 64 + (ch2 ^ (g * 32)); break;                 // halves swapped for the upper 32 lanes
        case 2: addr = px * 64 + (ch2 ^ (g * 16)); break;
        case 3: addr = lane * 8; break;                                   // linear
        case 4: addr = px * 80 + ch2; break;                              // 80-byte rows
        case 5: addr = (px & 3) * 64 + g * 1024 + 256 * 0 + ch2; break;   // upper lanes 1 KiB away (same banks)
        case 6: addr = (px & 3) * 64 + g * (1024 + 128) + ch2; break;     // upper lanes 1 KiB + 128 B away
        case 7: addr = (t16 >> 2) * 64 + ((lane >> 4) * 8) ; break;       // only 4 rows, 8-byte columns by 16-lane group
        case 8: addr = px * 64 + ch2 + ((lane >> 4) & 1) * 0 + g * 8; break;
        default: addr = 0;
    }
    addr += wave * 4096;
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    i32x2 r[16];
    int acc = 0;
    const int a = addr;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u & 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256" : "=v"(r[u]) : "v"(a) : "memory");
            else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[u]) : "v"(a) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= r[u][0] ^ r[u][1];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc == 12345) sink[0] = acc;
}
int main() {
    unsigned long long* out; int* sink; hipMalloc(&out, 8 * 8 * 304); hipMalloc(&sink, 4);
    const int iters = 2000;
    for (int nw : {1, 4, 8}) for (int p = 0; p <= 8; ++p) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * nw), 65536, 0, p, iters, out, sink);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        double mx = 0; for (int w = 0; w < nw; ++w) mx = h[w] > mx ? h[w] : mx;
        // s_memtime ticks at 100 MHz; report ticks per wave-instruction across the workgroup (relative numbers are what matter)
        printf("waves %d pattern %d: %.3f ticks per instruction per workgroup (%.1f per wave)\n", nw, p, mx / (iters * 16.0 * nw), mx / (iters * 16.0));
    }
    return 0;
}
