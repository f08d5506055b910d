#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ unsigned long long* g_probe;
#include "../satlas_super_resolution_amd/csrc/conv_ws.hip"
int main() {
    const int N = 16, H = 128, W = 128, C = 64;
    __bf16 *x, *w, *y; size_t nb = (size_t)N * H * W * C * 2;
    hipMalloc(&x, nb); hipMalloc(&y, nb); hipMalloc(&w, 2 * 9 * 64 * 32 * 2);
    hipMemset(x, 0x3c, nb); hipMemset(w, 0x3c, 2 * 9 * 64 * 32 * 2);
    unsigned long long* probe; hipMalloc(&probe, 512 * 8 * 8); hipMemset(probe, 0, 512 * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    ssr_conv_desc d{};
    d.dtype = SSR_BF16; d.x = {x, C, 0}; d.N = N; d.Hi = H; d.Wi = W; d.up = 1; d.Cin = C; d.w = w; d.CoutPad = 64;
    d.KH = d.KW = 3; d.stride = 1; d.pad_y = d.pad_x = 1; d.Gh = H; d.Gw = W; d.Ho = H; d.Wo = W; d.oys = d.oxs = 1;
    d.Cout = 64; d.y = {y, C, 0}; d.alpha = 1.f; d.act = 1;
    int rc = 0; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) ssr_conv_ws_try(d, 0, &rc, true);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) ssr_conv_ws_try(d, 0, &rc, true);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8); hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    double ph[8] = {0}; int nb2 = 256;
    for (int b = 0; b < nb2; ++b) for (int k = 1; k < 7; ++k) ph[k] += double(h[b * 8 + k] - h[b * 8 + k - 1]);
    printf("rc=%d avg launch %.2f us (19.3 GFLOP -> %.0f TFLOP/s)\n", rc, ms * 50, 19.327 / (ms * 50e-6) / 1e3);
    const char* nm[] = {"", "issue next-patch loads", "pixel decode", "MFMA loop", "epilogue", "store_patch (waits loads)", "barrier"};
    for (int k = 1; k < 7; ++k) printf("  %-28s %9.1f cycles\n", nm[k], ph[k] / nb2);
    return 0;
}
