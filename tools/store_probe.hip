// How fast does a CU get a burst of result stores out?  (round 5: the fp32x3 big-tile kernel's epilogue - 128 KB of 16-byte stores per
// workgroup and image, 4 waves - takes 32 k ticks = 4 B/clk per CU, coalesced or not.)  Each workgroup stores `kb` KB as 16-byte
// vectors (a wave instruction = 1 KB contiguous), `rounds` times, with `waves` waves; s_memtime around the issue loop and around the
// final s_waitcnt vmcnt(0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_probe.hip -o tools/store_probe && tools/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_store(u32x4* out, unsigned long long* t, int per_wave_stores, int rounds, size_t wg_stride_vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    u32x4* base = out + (size_t)blockIdx.x * wg_stride_vec;
    const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), ti = 0, tw = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        for (int s = 0; s < per_wave_stores; ++s)
            __builtin_nontemporal_store(v, base + ((size_t)(r * per_wave_stores + s) * nw + wave) * 64 + lane), __builtin_amdgcn_sched_barrier(0);
        const unsigned long long b = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long c = __builtin_amdgcn_s_memtime();
        ti += b - a; tw += c - b;
        __syncthreads();
    }
    if (threadIdx.x == 0) { t[blockIdx.x * 4] = __builtin_amdgcn_s_memtime() - t0; t[blockIdx.x * 4 + 1] = ti; t[blockIdx.x * 4 + 2] = tw; }
}
__global__ void k_store_plain(u32x4* out, unsigned long long* t, int per_wave_stores, int rounds, size_t wg_stride_vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    u32x4* base = out + (size_t)blockIdx.x * wg_stride_vec;
    const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), ti = 0, tw = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        for (int s = 0; s < per_wave_stores; ++s)
            base[((size_t)(r * per_wave_stores + s) * nw + wave) * 64 + lane] = v, __builtin_amdgcn_sched_barrier(0);
        const unsigned long long b = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long c = __builtin_amdgcn_s_memtime();
        ti += b - a; tw += c - b;
        __syncthreads();
    }
    if (threadIdx.x == 0) { t[blockIdx.x * 4] = __builtin_amdgcn_s_memtime() - t0; t[blockIdx.x * 4 + 1] = ti; t[blockIdx.x * 4 + 2] = tw; }
}
int main() {
    const size_t total = (size_t)1 << 30;
    u32x4* out; hipMalloc(&out, total);
    unsigned long long* t; hipMalloc(&t, 4096 * 4 * 8);
    std::vector<unsigned long long> h(4096 * 4);
    for (int nt = 0; nt < 2; ++nt)
    for (int nwg : {256, 128, 32, 8})
        for (int waves : {4, 8, 16}) {
            const int kb = 128, rounds = 4, psw = kb / waves;            // 1 KB per wave store
            const size_t stride = (size_t)kb * 1024 / 16 * rounds;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int it = 0; it < 2; ++it) {
                hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL(k_store, dim3(nwg), dim3(waves * 64), 0, 0, out, t, psw, rounds, stride);
                else hipLaunchKernelGGL(k_store_plain, dim3(nwg), dim3(waves * 64), 0, 0, out, t, psw, rounds, stride);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), t, nwg * 4 * 8, hipMemcpyDeviceToHost);
            double a = 0, b = 0, c = 0; for (int i = 0; i < nwg; ++i) { a += h[i * 4]; b += h[i * 4 + 1]; c += h[i * 4 + 2]; }
            printf("%s workgroups=%3d waves=%2d: %d KB x %d rounds per workgroup: %8.0f ticks per round (issue %7.0f + drain %7.0f) = %5.2f B/tick per CU; launch %.1f us = %.2f TB/s\n",
                   nt ? "nontemporal" : "plain      ", nwg, waves, kb, rounds, a / nwg / rounds, b / nwg / rounds, c / nwg / rounds, kb * 1024.0 / (a / nwg / rounds), ms * 1e3,
                   (double)nwg * kb * 1024 * rounds / (ms * 1e-3) * 1e-12);
        }
    return 0;
}
