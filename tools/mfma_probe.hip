// What can one wave per SIMD sustain?  Back-to-back v_mfma_f32_32x32x16_bf16, alone and with ds_read_b128
// operand traffic at the fused-RDB ratio (1.5 reads per MFMA), stamped with s_memtime.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int q = threadIdx.x; q < 40960; q += 256) reinterpret_cast<unsigned*>(smem)[q] = 0x3c003c00u + q;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 av = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, bv = av;
    const char* pa = smem + (lane & 31) * 80 + (lane >> 5) * 16;
    const char* pb = smem + 65536 + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // bare MFMAs
#pragma unroll
            for (int n = 0; n < 18; ++n)
#pragma unroll
                for (int a = 0; a < NACC; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[a], 0, 0, 0);
        } else if (MODE == 3) {   // one full pixel-fragment read per 3 taps; the other two taps are lane shifts (DPP wave_shl:1)
            // merged with a masked fix-up read of the few lanes at row ends: the A traffic of a 3x3 row from 3 reads to ~1.1
            u32x4 bq[18], aq[18][NACC], fx[18][NACC];
            const bool edge = ((lane & 31) % 14) >= 12;          // lanes whose right neighbours are in the next region row
            auto issue = [&](int n) {
                bq[n] = *reinterpret_cast<const u32x4*>(pb + n * 1024);
#pragma unroll
                for (int a = 0; a < NACC; ++a) {
                    if (n % 3 == 0) aq[n][a] = *reinterpret_cast<const u32x4*>(pa + (n * 2 + a) * 2560 % 60000);
                    else if (edge) fx[n][a] = *reinterpret_cast<const u32x4*>(pa + (n * 2 + a) * 2560 % 60000);
                }
            };
#pragma unroll
            for (int n = 0; n < 3; ++n) issue(n);
#pragma unroll
            for (int n = 0; n < 18; ++n) {
                __builtin_amdgcn_sched_barrier(0);
                if (n + 3 < 18) issue(n + 3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < NACC; ++a) {
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[n]), __builtin_bit_cast(bf16x8, aq[n][a]), acc[a], 0, 0, 0);
                    if (n % 3 != 2 && n + 1 < 18) {               // derive the next tap's fragment while this MFMA runs
                        u32x4 nx;
                        nx.x = __builtin_amdgcn_update_dpp(0u, aq[n][a].x, 0x130, 0xf, 0xf, true);
                        nx.y = __builtin_amdgcn_update_dpp(0u, aq[n][a].y, 0x130, 0xf, 0xf, true);
                        nx.z = __builtin_amdgcn_update_dpp(0u, aq[n][a].z, 0x130, 0xf, 0xf, true);
                        nx.w = __builtin_amdgcn_update_dpp(0u, aq[n][a].w, 0x130, 0xf, 0xf, true);
                        aq[n + 1][a].x = edge ? fx[n + 1][a].x : nx.x;
                        aq[n + 1][a].y = edge ? fx[n + 1][a].y : nx.y;
                        aq[n + 1][a].z = edge ? fx[n + 1][a].z : nx.z;
                        aq[n + 1][a].w = edge ? fx[n + 1][a].w : nx.w;
                    }
                }
            }
        } else {           // software-pipelined reads: 1 B + NACC A reads per NACC MFMAs, 3 steps ahead
            u32x4 bq[18], aq[18][NACC];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                bq[n] = *reinterpret_cast<const u32x4*>(pb + n * 1024);
#pragma unroll
                for (int a = 0; a < NACC; ++a) aq[n][a] = *reinterpret_cast<const u32x4*>(pa + (n * 2 + a) * 2560 % 60000);
            }
#pragma unroll
            for (int n = 0; n < 18; ++n) {
                __builtin_amdgcn_sched_barrier(0);
                if (n + 3 < 18) {
                    bq[n + 3] = *reinterpret_cast<const u32x4*>(pb + (n + 3) * 1024);
#pragma unroll
                    for (int a = 0; a < NACC; ++a)
                        aq[n + 3][a] = *reinterpret_cast<const u32x4*>(pa + ((n + 3) * 2 + a) * 2560 % 60000);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < NACC; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[n]), __builtin_bit_cast(bf16x8, aq[n][a]), acc[a], 0, 0, 0);
            }
            if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int NACC>
void run(const char* name, int blocks) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * 4 * 8); hipMalloc(&sink, 4);
    const int iters = 200;
    auto kern = k<MODE, NACC>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 160 * 1024, 0, out, sink, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 160 * 1024, 0, out, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += double(v); avg /= h.size();
    const double nm = double(iters) * 18 * NACC;
    printf("%-34s blocks=%d  %.1f ticks/MFMA  (%.2f ns/MFMA by events, %.1f TFLOP/s chip)\n", name, blocks, avg / nm,
           ms * 1e6 / nm, blocks * 4 * nm * 32768.0 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(sink);
}

int main() {
    run<0, 2>("bare mfma, 2 acc", 256);
    run<0, 1>("bare mfma, 1 acc (dependent)", 256);
    run<1, 2>("mfma + 1.5 ds_read_b128, 2 acc", 256);
    run<1, 1>("mfma + 2 ds_read_b128, 1 acc", 256);
    run<2, 2>("same as 3 + barrier per 36 mfma", 256);
    run<3, 2>("dpp-shifted taps, 2 acc", 256);
    run<3, 1>("dpp-shifted taps, 1 acc", 256);
    run<0, 2>("bare mfma, 2 acc, 1 block", 1);
    return 0;
}
