// Phase timing of the big-tile conv kernel (conv6 forward shape by default: N=16, 128x128, 128 -> 64).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSR_PROBE -Iinclude -Isatlas_super_resolution_amd/csrc tools/big_probe.hip -o tools/big_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_probe;
#include "../satlas_super_resolution_amd/csrc/conv_big.hip"
int main(int argc, char** argv) {
    const int N = argc > 6 ? atoi(argv[6]) : 16, H = argc > 1 ? atoi(argv[1]) : 128, W = H, Cin = argc > 2 ? atoi(argv[2]) : 128, Cout = argc > 3 ? atoi(argv[3]) : 64;
    const int KT = argc > 4 ? atoi(argv[4]) : 3, ep21 = argc > 5 ? atoi(argv[5]) : 0;   // ep21 = 1: LeakyReLU + residual + second output (the U-Net decoder's forward convs)
    __bf16 *x, *w, *y; const size_t nx = (size_t)N * H * W * Cin * 2, ny = (size_t)N * H * W * Cout * 2, nw = (size_t)Cin * 9 * Cout * 2;
    hipMalloc(&x, nx); hipMalloc(&y, ny); hipMalloc(&w, nw);
    hipMemset(x, 0x3c, nx); hipMemset(w, 0x3c, nw);
    ssr_conv_desc d{};
    d.dtype = SSR_BF16; d.x = {x, Cin, 0}; d.N = N; d.Hi = H; d.Wi = W; d.up = 1; d.Cin = Cin; d.w = w; d.CoutPad = Cout;
    d.KH = d.KW = KT; d.stride = 1; d.pad_y = d.pad_x = KT == 3 ? 1 : 0; d.Gh = H; d.Gw = W; d.Ho = H; d.Wo = W; d.oys = d.oxs = 1;
    d.Cout = Cout; d.y = {y, Cout, 0}; d.alpha = 1.f; d.act = 1;
    if (ep21) { __bf16 *r1, *y0; hipMalloc(&r1, ny); hipMalloc(&y0, ny); hipMemset(r1, 0x3c, ny); d.r1 = {r1, Cout, 0}; d.r1_nc = Cout; d.beta1 = 1.f; d.y0 = {y0, Cout, 0}; }
    const int nb = N * ((H + 31) / 32) * ((W + 15) / 16) * (Cout / 64);
    unsigned long long* probe; hipMalloc(&probe, (size_t)nb * 16 * 8); hipMemset(probe, 0, (size_t)nb * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    int rc = 0; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) ssr_conv_big_try(d, 0, &rc, true);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) ssr_conv_big_try(d, 0, &rc, true);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nb * 16); hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    const double gf = 2.0 * N * H * W * Cin * Cout * KT * KT / 1e9;
    printf("rc=%d blocks=%d launch %.1f us  %.1f GFLOP -> %.0f TFLOP/s\n", rc, nb, ms * 100, gf, gf / (ms / 10 * 1e-3) / 1e3);
    const char* nm[] = {"", "prologue (chunk 0 loads + barrier)", "", "barrier A (chunk 1)", "store chunk", "barrier B", "MFMA phase", "", "epilogue"};
    double ph[16] = {0};
    for (int b = 0; b < nb; ++b) { for (int k : {1, 3, 4, 5, 6, 8}) ph[k] += double(h[b * 16 + k] - h[b * 16 + k - 1]); ph[9] += double(h[b * 16 + 8] - h[b * 16]); }
    for (int k : {1, 3, 4, 5, 6, 8}) printf("  %-36s %9.1f cycles\n", nm[k], ph[k] / nb);
    printf("  whole block %9.1f cycles\n", ph[9] / nb);
    return 0;
}
