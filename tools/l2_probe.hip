// How fast can every CU stream the SAME L2-resident weights?  (the fused RDB kernels: 479 KB per block, 256 blocks)
// W waves per block each keep D 1-KiB loads (16 B / lane) in flight over a `bytes` region, optionally starting at a
// per-block rotated offset.  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_probe.hip -o tools/l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(512) void k(const char* __restrict__ w, int bytes, int rotate, int reps, unsigned long long* out,
                                         unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int npiece = bytes / 1024;                       // 1-KiB pieces; wave takes pieces wave, wave+nw, ...
    const int start = rotate ? (blockIdx.x * 37) % npiece : 0;
    u32x4 r[D];
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        int p = wave;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            int pp = p + j * nw + start; pp = pp >= npiece ? pp - npiece : pp; pp = pp >= npiece ? pp - npiece : pp;
            r[j] = *reinterpret_cast<const u32x4*>(w + (size_t)pp * 1024 + lane * 16);
        }
        for (p = wave + D * nw; p < npiece + D * nw; p += D * nw) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                acc += r[j][0] ^ r[j][3];
                int pp = p + j * nw;
                if (pp < npiece) {
                    pp += start; pp = pp >= npiece ? pp - npiece : pp;
                    r[j] = *reinterpret_cast<const u32x4*>(w + (size_t)pp * 1024 + lane * 16);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x12345678u) sink[0] = acc;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int D>
void run(const char* w, int bytes, int nwaves, int rotate, int blocks) {
    unsigned long long* out; unsigned* sink;
    hipMalloc(&out, blocks * 8 * 8); hipMalloc(&sink, 4);
    hipMemset(out, 0, blocks * 8 * 8);
    const int reps = 8;
    hipLaunchKernelGGL(k<D>, dim3(blocks), dim3(64 * nwaves), 0, 0, w, bytes, rotate, reps, out, sink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<D>, dim3(blocks), dim3(64 * nwaves), 0, 0, w, bytes, rotate, reps, out, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double mx = 0; for (int b = 0; b < blocks; ++b) for (int v = 0; v < nwaves; ++v) mx += double(h[b * 8 + v]);
    mx /= blocks * nwaves;
    printf("D=%2d waves=%d rotate=%d blocks=%3d : %6.1f B/clk/CU  (%.2f TB/s chip, %.1f us)\n", D, nwaves, rotate, blocks,
           double(bytes) * reps / mx, double(bytes) * reps * blocks / (ms * 1e-3) / 1e12, ms * 1e3);
    hipFree(out); hipFree(sink);
}

int main() {
    const int bytes = 479 * 1024;
    char* w; hipMalloc(&w, bytes); hipMemset(w, 0x3c, bytes);
    for (int rot = 0; rot < 1; ++rot) {
        run<9>(w, bytes, 2, rot, 256);
        run<18>(w, bytes, 2, rot, 256);
        run<9>(w, bytes, 4, rot, 256);
        run<18>(w, bytes, 4, rot, 256);
    }
    run<9>(w, bytes, 8, 0, 256);
    run<18>(w, bytes, 8, 0, 256);
    run<4>(w, bytes, 8, 0, 256);
    run<9>(w, bytes, 1, 0, 256);
    run<9>(w, bytes, 3, 0, 256);
    run<9>(w, bytes, 6, 0, 256);
    return 0;
}
