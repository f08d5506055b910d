// The k-step pattern of the fused dense block's MFMA waves in isolation: per k-step 1 weight-fragment + NMT pixel-fragment
// ds_read_b128 into a rotating register queue (prefetch distance PF, NB buffers) and NMT v_mfma_f32_32x32x16_bf16 that consume
// the fragments read PF steps earlier.  One wave per SIMD (4 per CU) or two (8).  ticks (s_memtime) per MFMA per wave.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kstep_probe.hip -o tools/kstep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDS3 __attribute__((address_space(3)))
template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
__device__ __forceinline__ u32x4 ldsr(unsigned a) { return *(const LDS3 u32x4*)(uintptr_t)a; }

// VAR 0: as the kernel (reads bunched before the MFMAs); 1: reads interleaved one per MFMA gap; 2: no reads (operands constant)
template <int NMT, int PF, int NB, int VAR>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int waves_active, int iters) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = threadIdx.x; q < 36000; q += 512) reinterpret_cast<unsigned*>(smem)[q] = 0x3c003c00u + (q & 7);
    __syncthreads();
    if (wave >= waves_active) return;
    f32x16 acc[NMT];
    for (int a = 0; a < NMT; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int i = lane & 31, g = lane >> 5;
    unsigned wb = i * 64 + ((g ^ ((i >> 2) & 3)) << 4);                       // weight fragment, swizzled 64-B rows
    unsigned ab[NMT];
    for (int m = 0; m < NMT; ++m) {
        const int row = (i & 15) + 16 * ((i >> 4) + 2 * m) + 40 * (wave & 3);   // distinct rows mod 16 inside a 16-lane group
        ab[m] = 24576 + row * 64 + ((g ^ ((row >> 2) & 3)) << 4);
        asm volatile("" : "+v"(ab[m]));
    }
    asm volatile("" : "+v"(wb));
    u32x4 bq[NB], aq[NB][NMT];
    constexpr int NSTEP = 12;                                                  // per loop iteration (multiple of NB for NB = 4, 6; see main)
    auto issue = [&](auto n_c, auto part_c) {
        constexpr int n = decltype(n_c)::value, part = decltype(part_c)::value;
        if constexpr (VAR == 2) {
            if constexpr (part == 0) bq[n % NB] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
            else aq[n % NB][part - 1] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
        } else {
            if constexpr (part == 0) bq[n % NB] = ldsr(wb + (n % 6) * 2048);
            else aq[n % NB][part - 1] = ldsr(ab[part - 1] + (n % 9) * 1664);
        }
    };
    auto issue_all = [&](auto n_c) { sfor<0, NMT + 1>([&](auto p_c) { issue(n_c, p_c); }); };
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    sfor<0, PF>(issue_all);
    for (int it = 0; it < iters; ++it) {
        sfor<0, NSTEP>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value;
            using NX = std::integral_constant<int, n + PF>;      // (n + PF) % NB is what matters: NSTEP % NB == 0
            if constexpr (VAR == 1) {
                __builtin_amdgcn_sched_barrier(0);
                issue(NX{}, std::integral_constant<int, 0>{});
                sfor<0, NMT>([&](auto m_c) {
                    constexpr int m = decltype(m_c)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[n % NB]), __builtin_bit_cast(bf16x8, aq[n % NB][m]), acc[m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    issue(NX{}, std::integral_constant<int, m + 1>{});
                });
            } else {
                __builtin_amdgcn_sched_barrier(0);
                issue_all(NX{});
                __builtin_amdgcn_sched_barrier(0);
                sfor<0, NMT>([&](auto m_c) {
                    constexpr int m = decltype(m_c)::value;
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[n % NB]), __builtin_bit_cast(bf16x8, aq[n % NB][m]), acc[m], 0, 0, 0);
                });
            }
        });
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int m = 0; m < NMT; ++m) asm volatile("" : "+v"(acc[m][0]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
    for (int a = 0; a < NMT; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int b = 0; b < NB; ++b) s += __builtin_bit_cast(float, bq[b].x) + __builtin_bit_cast(float, aq[b][0].x);
    if (s == 12345.f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NMT, int PF, int NB, int VAR> void run(const char* name, int waves) {
    const int blocks = 256, iters = 24;
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * 8 * 8); hipMalloc(&sink, 4);
    hipMemset(out, 0, blocks * 8 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NMT, PF, NB, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NMT, PF, NB, VAR>), dim3(blocks), dim3(512), 160 * 1024, 0, out, sink, waves, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; int cnt = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) { avg += double(h[b * 8 + w]); ++cnt; }
    printf("NMT=%d PF=%d NB=%d %-28s waves/CU=%d  %.1f ticks per MFMA per wave\n", NMT, PF, NB, name, waves, avg / cnt / (iters * 12.0 * NMT));
    hipFree(out); hipFree(sink);
}

int main() {
    for (int waves : {4, 8}) {
        run<3, 3, 4, 0>("bunched reads", waves);
        run<3, 3, 6, 0>("bunched reads", waves);
        run<3, 3, 4, 1>("interleaved reads", waves);
        run<3, 3, 6, 1>("interleaved reads", waves);
        run<3, 3, 4, 2>("no reads", waves);
        run<2, 4, 6, 0>("bunched reads", waves);
        run<2, 4, 6, 1>("interleaved reads", waves);
        run<2, 3, 4, 0>("bunched reads", waves);
        run<1, 4, 6, 0>("bunched reads", waves);
    }
    return 0;
}
