// Does straight-line code feed the matrix pipe as well as a loop?  810 back-to-back v_mfma_f32_32x32x16_bf16 on 3 accumulators
// (+ fillers), as ONE unrolled stream (macro-expanded) vs a loop of 18-MFMA bodies; 1 or 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ifetch_probe.hip -o tools/ifetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MF(a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[a], 0, 0, 0);
#define FILLER()                                                                                              \
    if (FILL == 1) asm volatile("s_nop 0\n\ts_nop 0");                                                        \
    if (FILL == 2) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(junk) : "v"(addr));                  \
    if (FILL == 3) asm volatile("ds_read_b128 %0, %1 offset:1024\n\ts_nop 0" : "=v"(junk) : "v"(addr));      \
    if (FILL == 4) asm volatile("ds_read_b128 %0, %2 offset:1024\n\tds_read_b128 %1, %2 offset:2048\n\ts_waitcnt lgkmcnt(8)" : "=v"(junk), "=v"(junk2) : "v"(addr));
#define B3 __builtin_amdgcn_sched_barrier(0); MF(0) FILLER() __builtin_amdgcn_sched_barrier(0); MF(1) FILLER() __builtin_amdgcn_sched_barrier(0); MF(2) FILLER()
#define R6(x) x x x x x x
#define R9(x) x x x x x x x x x
#define R5(x) x x x x x

template <int MODE, int FILL>   // MODE 0: loop of 18; 1: straight line 810
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int waves_active) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = threadIdx.x; q < 8192; q += 512) reinterpret_cast<unsigned*>(smem)[q] = 0x3c003c00u + q;
    __syncthreads();
    if (wave >= waves_active) return;
    f32x16 acc[3];
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 av = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, bv = av;
    unsigned addr = lane * 16;
    asm volatile("" : "+v"(addr));
    u32x4 junk = av, junk2 = av;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 0) {
        for (int it = 0; it < 45; ++it) { R6(B3) }
    } else {
        R5(R9(R6(B3)))
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    s += __builtin_bit_cast(float, junk.x) + __builtin_bit_cast(float, junk2.x);
    if (s == 12345.f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int FILL> void run(const char* name, int waves) {
    const int blocks = 256;
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * 8 * 8); hipMalloc(&sink, 4);
    hipMemset(out, 0, blocks * 8 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k<MODE, FILL>), dim3(blocks), dim3(512), 160 * 1024, 0, out, sink, waves);
    hipLaunchKernelGGL((k<MODE, FILL>), dim3(blocks), dim3(512), 160 * 1024, 0, out, sink, waves);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; int cnt = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) { avg += double(h[b * 8 + w]); ++cnt; }
    printf("%-52s waves/CU=%d  %.1f ticks per MFMA per wave\n", name, waves, avg / cnt / 810.0);
    hipFree(out); hipFree(sink);
}

int main() {
    for (int waves : {4, 8}) {
        run<0, 0>("loop of 18, bare", waves);
        run<1, 0>("straight line 810, bare", waves);
        run<0, 2>("loop of 18, + 1 ds_read_b128 per MFMA", waves);
        run<1, 2>("straight line 810, + 1 ds_read_b128 per MFMA", waves);
        run<0, 3>("loop of 18, + ds_read_b128 + s_nop", waves);
        run<1, 3>("straight line 810, + ds_read_b128 + s_nop", waves);
        run<0, 4>("loop of 18, + 2 ds_read_b128 + waitcnt", waves);
        run<1, 4>("straight line 810, + 2 ds_read_b128 + waitcnt", waves);
    }
    return 0;
}
