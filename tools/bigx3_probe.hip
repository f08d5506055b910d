// In-kernel phase timing of the fp32x3 big-tile kernel (conv_bigx3_kernel4, csrc/conv_big_x3.hip): s_memtime sums of wave 0 per
// workgroup, standalone:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSR_PROBE -Iinclude -Isatlas_super_resolution_amd/csrc tools/bigx3_probe.hip -o tools/bigx3_probe
//   tools/bigx3_probe [N=32] [Cin=64] [Cout=64] [H=128] [W=128] [epi=0: plain lrelu | 1: + r1 + y0 | 2: mask (dgrad)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_probe;
#include "../satlas_super_resolution_amd/csrc/conv_big_x3.hip"
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32, Cin = argc > 2 ? atoi(argv[2]) : 64, Cout = argc > 3 ? atoi(argv[3]) : 64;
    const int H = argc > 4 ? atoi(argv[4]) : 128, W = argc > 5 ? atoi(argv[5]) : 128, epi = argc > 6 ? atoi(argv[6]) : 0;
    const int CoutPad = (Cout + 63) / 64 * 64, nchunks = (Cin + 15) / 16;
    float *x, *y, *w, *r1, *y0;
    const size_t nbx = (size_t)N * H * W * Cin * 4, nby = (size_t)N * H * W * Cout * 4;
    hipMalloc(&x, nbx); hipMalloc(&y, nby); hipMalloc(&r1, nby); hipMalloc(&y0, nby);
    hipMalloc(&w, (size_t)nchunks * 9 * CoutPad * 64);
    hipMemset(x, 0, nbx); hipMemset(r1, 0, nby); hipMemset(w, 0, (size_t)nchunks * 9 * CoutPad * 64);
    const int nblk_max = 4096;
    unsigned long long* probe; hipMalloc(&probe, (size_t)nblk_max * 8 * 8); hipMemset(probe, 0, (size_t)nblk_max * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    ssr_conv_desc d{};
    d.dtype = SSR_F32X3; d.x = {x, Cin, 0}; d.N = N; d.Hi = H; d.Wi = W; d.up = 1; d.Cin = Cin; d.w = w; d.CoutPad = CoutPad;
    d.KH = d.KW = 3; d.stride = 1; d.pad_y = d.pad_x = 1; d.Gh = H; d.Gw = W; d.Ho = H; d.Wo = W; d.oys = d.oxs = 1;
    d.Cout = Cout; d.y = {y, Cout, 0}; d.alpha = 1.f; d.act = epi == 2 ? 0 : 1;
    if (epi == 1) { d.r1 = {r1, Cout, 0}; d.r1_nc = Cout; d.beta1 = 1.f; d.y0 = {y0, Cout, 0}; }
    if (epi == 2) { d.m = {r1, Cout, 0}; d.m_c0 = 0; d.m_c1 = Cout; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int rc = 0;
    for (int it = 0; it < 3; ++it) ssr_conv_bigx3_try(d, 0, &rc, true);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) ssr_conv_bigx3_try(d, 0, &rc, true);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nblk_max * 8);
    hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    int nblk = 0; while (nblk < nblk_max && h[nblk * 8] != 0) ++nblk;
    const double gf = 2.0 * N * H * W * Cout * 9.0 * Cin * 1e-9;
    printf("N=%d Cin=%d Cout=%d %dx%d epi=%d workgroups=%d rc=%d: avg launch (back to back) = %.2f us = %.1f TFLOP/s algorithmic\n", N, Cin, Cout, H, W, epi, nblk, rc,
           ms * 1000 / 20, gf / (ms / 20));
    const char* names[8] = {"whole workgroup", "prologue (first loads issued, bias)", "wait: everyone leaves the previous chunk", "store (incl. wait for staging loads)",
                            "barrier behind the store", "k-steps (MFMAs)", "epilogues", "chunks"};
    for (int k = 0; k < 8; ++k) { double s = 0; for (int b = 0; b < nblk; ++b) s += (double)h[b * 8 + k]; printf("  %-44s %10.1f\n", names[k], s / nblk); }
    return 0;
}
