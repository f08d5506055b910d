// In-kernel phase timing of the producer / MFMA-wave ring kernel (conv_x3q_kernel, csrc/conv_x3q.hip): s_memtime stamps of one thread
// per role, standalone:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSR_PROBE -Iinclude -Isatlas_super_resolution_amd/csrc tools/x3q_probe.hip -o tools/x3q_probe
//   tools/x3q_probe [N=32] [Cin=64] [Cout=32] [H=32] [W=32]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_probe;
#include "../satlas_super_resolution_amd/csrc/conv_x3q.hip"
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32, Cin = argc > 2 ? atoi(argv[2]) : 64, Cout = argc > 3 ? atoi(argv[3]) : 32;
    const int H = argc > 4 ? atoi(argv[4]) : 32, W = argc > 5 ? atoi(argv[5]) : 32, CS = 192;
    const int CoutPad = (Cout + 31) / 32 * 32, nchunks = (Cin + 15) / 16;
    float *x, *y, *w;
    const size_t nb = (size_t)N * H * W * CS * 4;
    hipMalloc(&x, nb); hipMalloc(&y, nb);
    hipMalloc(&w, (size_t)nchunks * 9 * CoutPad * 64);
    hipMemset(x, 0, nb); hipMemset(w, 0, (size_t)nchunks * 9 * CoutPad * 64);
    const int nblk = N * ((H + 7) / 8) * ((W + 15) / 16) * ((CoutPad % 64) == 0 ? CoutPad / 64 : CoutPad / 32);
    unsigned long long* probe; hipMalloc(&probe, (size_t)nblk * 16 * 8); hipMemset(probe, 0, (size_t)nblk * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    ssr_conv_desc d{};
    d.dtype = SSR_F32X3; d.x = {x, CS, 0}; d.N = N; d.Hi = H; d.Wi = W; d.up = 1; d.Cin = Cin; d.w = w; d.CoutPad = CoutPad;
    d.KH = d.KW = 3; d.stride = 1; d.pad_y = d.pad_x = 1; d.Gh = H; d.Gw = W; d.Ho = H; d.Wo = W; d.oys = d.oxs = 1;
    d.Cout = Cout; d.y = {y, CS, 64}; d.alpha = 1.f; d.act = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int rc = 0;
    for (int it = 0; it < 5; ++it) ssr_conv_x3q_try(d, 0, &rc, true);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 50; ++it) ssr_conv_x3q_try(d, 0, &rc, true);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nblk * 16);
    hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    printf("N=%d Cin=%d Cout=%d %dx%d workgroups=%d chunks=%d rc=%d: avg launch (back to back, events) = %.2f us\n", N, Cin, Cout, H, W, nblk, nchunks, rc, ms * 1000 / 50);
    auto avg = [&](int a, int b_) { double s = 0; for (int b = 0; b < nblk; ++b) s += double(h[b * 16 + b_] - h[b * 16 + a]); return s / nblk; };
    printf("  MFMA wave 0: entry -> loop          %8.1f ticks\n", avg(0, 1));
    int prev = 1;
    for (int c = 0; c < nchunks && c < 6; ++c) { printf("    chunk %d (wait + MFMAs)            %8.1f\n", c, avg(prev, 2 + c)); prev = 2 + c; }
    printf("    remaining chunks + final barrier  %8.1f\n", avg(prev, 10));
    printf("    reduce + epilogue                 %8.1f\n", avg(10, 11));
    printf("    whole workgroup                   %8.1f\n", avg(0, 11));
    printf("  producer wave 0: entry -> first loads issued %8.1f, stores + refills of all chunks %8.1f\n", avg(0, 8), avg(8, 9));
    return 0;
}
