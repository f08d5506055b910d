// Round 4: how fast can every CU stream the SAME L2-resident 479 KB (the dense block's weights), by path and by wave count?
//   mode 0: global_load_dwordx4 -> registers (D loads of 1 KiB in flight per wave), as the dense-block producers do
//   mode 1: global_load_lds_dwordx4 (LDS-DMA, no registers): D loads in flight per wave, each wave owns D KiB of LDS
// Sweeps waves per CU, D, a per-block rotated start (de-phasing), and the number of blocks (64 / 128 / 256: does the per-CU
// rate change with the number of CUs asking for the same lines?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_probe2.hip -o tools/l2_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;

template <int D, int MODE>
__global__ __launch_bounds__(768) void k(const char* __restrict__ w, int bytes, int rotate, int reps, unsigned long long* out, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int npiece = bytes / 1024;
    const int start = rotate ? (blockIdx.x * 37) % npiece : 0;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        u32x4 r[D];
        for (int rep = 0; rep < reps; ++rep) {
            int p = wave;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                int pp = (p + j * nw + start) % npiece;
                r[j] = *reinterpret_cast<const u32x4*>(w + (size_t)pp * 1024 + lane * 16);
            }
            for (p = wave + D * nw; p < npiece + D * nw; p += D * nw) {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    acc += r[j][0] ^ r[j][3];
                    int pp = p + j * nw;
                    if (pp < npiece) { pp = (pp + start) % npiece; r[j] = *reinterpret_cast<const u32x4*>(w + (size_t)pp * 1024 + lane * 16); }
                }
            }
        }
    } else {
        char* mine = smem + wave * D * 1024;
        for (int rep = 0; rep < reps; ++rep) {
            int cnt = 0;
            for (int p = wave; p < npiece; p += nw, ++cnt) {
                const int pp = (p + start) % npiece;
                // keep D in flight: before re-using slot cnt % D wait until at most D-1 are outstanding
                if (cnt >= D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
                __builtin_amdgcn_global_load_lds((gptr)(w + (size_t)pp * 1024 + lane * 16), (lptr)(mine + (cnt % D) * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += *reinterpret_cast<unsigned*>(mine + lane * 4);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x12345678u) sink[0] = acc;
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int D, int MODE>
void run(const char* w, int bytes, int nwaves, int rotate, int blocks) {
    unsigned long long* out; unsigned* sink;
    hipMalloc(&out, blocks * 16 * 8); hipMalloc(&sink, 4);
    hipMemset(out, 0, blocks * 16 * 8);
    const int reps = 8;
    const int lds = MODE ? nwaves * D * 1024 : 0;
    hipFuncSetAttribute((const void*)k<D, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k<D, MODE>), dim3(blocks), dim3(64 * nwaves), lds, 0, w, bytes, rotate, reps, out, sink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<D, MODE>), dim3(blocks), dim3(64 * nwaves), lds, 0, w, bytes, rotate, reps, out, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 16);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double mx = 0; for (int b = 0; b < blocks; ++b) for (int v = 0; v < nwaves; ++v) mx += double(h[b * 16 + v]);
    mx /= blocks * nwaves;
    printf("%s D=%2d waves=%2d rotate=%d blocks=%3d : %6.1f B/clk/CU  (%.2f TB/s chip, %.1f us for %d x %d KB)\n", MODE ? "lds-dma" : "regs   ", D, nwaves, rotate, blocks,
           double(bytes) * reps / mx, double(bytes) * reps * blocks / (ms * 1e-3) / 1e12, ms * 1e3, reps, bytes / 1024);
    hipFree(out); hipFree(sink);
}

int main() {
    const int bytes = 479 * 1024;
    char* w; hipMalloc(&w, bytes); hipMemset(w, 0x3c, bytes);
    for (int rot = 0; rot < 2; ++rot) {
        run<6, 0>(w, bytes, 4, rot, 256);
        run<18, 0>(w, bytes, 4, rot, 256);
        run<6, 1>(w, bytes, 4, rot, 256);
        run<8, 1>(w, bytes, 4, rot, 256);
    }
    for (int nw : {1, 2, 4, 6, 8, 12}) { run<9, 0>(w, bytes, nw, 0, 256); run<8, 1>(w, bytes, nw, 0, 256); }
    for (int blocks : {32, 64, 128, 256}) { run<18, 0>(w, bytes, 4, 0, blocks); run<8, 1>(w, bytes, 4, 0, blocks); }
    run<2, 1>(w, bytes, 4, 0, 256);
    run<4, 1>(w, bytes, 4, 0, 256);
    run<16, 1>(w, bytes, 4, 0, 256);
    run<4, 1>(w, bytes, 12, 0, 256);
    return 0;
}
