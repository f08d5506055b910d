// Phase timing of the bf16 weight-gradient kernel on one generator-body layer (N=16, 32x32, Cout=32, Cin given).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSR_PROBE -Iinclude -Isatlas_super_resolution_amd/csrc tools/wgrad_probe.hip -o tools/wgrad_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_probe;
#include "../satlas_super_resolution_amd/csrc/wgrad_bf16.hip"
extern "C" int32_t ssr_wgrad_tiles(int32_t N, int32_t Gh, int32_t Gw, int32_t dtype, int32_t KH) { return N * ((Gh + wgrad_bf16_th(KH) - 1) / wgrad_bf16_th(KH)) * ((Gw + 15) / 16); }
int main(int argc, char** argv) {
    const int N = 16, H = 32, W = 32, CS = 192, cin = argc > 1 ? atoi(argv[1]) : 192, cout = argc > 2 ? atoi(argv[2]) : 32;
    const int nrep = argc > 3 ? atoi(argv[3]) : 8;      // replicate the layer to fill the chip like the batched launch
    __bf16 *x, *dy; float* dw;
    hipMalloc(&x, (size_t)N * H * W * CS * 2); hipMalloc(&dy, (size_t)N * H * W * CS * 2);
    hipMemset(x, 0x3c, (size_t)N * H * W * CS * 2); hipMemset(dy, 0x3c, (size_t)N * H * W * CS * 2);
    hipMalloc(&dw, (size_t)nrep * cout * cin * 9 * 4); hipMemset(dw, 0, (size_t)nrep * cout * cin * 9 * 4);
    std::vector<ssr_wgrad_layer> L(nrep); std::vector<ssr_wgrad_item> I;
    const int tiles = N * (H / 16) * (W / 16);
    for (int r = 0; r < nrep; ++r) {
        L[r] = ssr_wgrad_layer{{x, CS, 0}, {dy, CS, 64}, N, H, W, 1, cin, cout, 1, 1, H, W, 1.f, dw + (size_t)r * cout * cin * 9, cin, nullptr};
        for (int co = 0; co < cout; co += 32) for (int ci = 0; ci < cin; ci += 64) I.push_back({r, co, ci, 0, tiles, 0, 1, 0, 0});
    }
    ssr_wgrad_layer* Ld; ssr_wgrad_item* Id;
    hipMalloc(&Ld, L.size() * sizeof(L[0])); hipMalloc(&Id, I.size() * sizeof(I[0]));
    hipMemcpy(Ld, L.data(), L.size() * sizeof(L[0]), hipMemcpyHostToDevice);
    hipMemcpy(Id, I.data(), I.size() * sizeof(I[0]), hipMemcpyHostToDevice);
    unsigned long long* probe; hipMalloc(&probe, I.size() * 16 * 8); hipMemset(probe, 0, I.size() * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) ssr_wgrad_bf16_dispatch(Ld, Id, (int)I.size(), 3, 3, 1, 0);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) ssr_wgrad_bf16_dispatch(Ld, Id, (int)I.size(), 3, 3, 1, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(I.size() * 16); hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    const double gf = 2.0 * N * H * W * cin * cout * 9 * nrep / 1e9;
    printf("items=%zu launch %.1f us  %.1f GFLOP -> %.0f TFLOP/s\n", I.size(), ms * 200, gf, gf / (ms / 5 * 1e-3) / 1e3);
    double ph[16] = {0}; const int nb = (int)I.size();
    for (int b = 0; b < nb; ++b) { ph[3] += double(h[b*16+3] - h[b*16+2]); ph[4] += double(h[b*16+4] - h[b*16+3]); ph[7] += double(h[b*16+7] - h[b*16+0]); ph[8] += double(h[b*16+8] - h[b*16+7]); }
    const int tiles2 = ssr_wgrad_tiles(N, H, W, SSR_BF16, 3);
    printf("  poll %.1f  contraction %.1f (tile 20)\n", ph[3] / nb, ph[4] / nb);
    printf("  whole loop     %8.1f cycles (%d iterations -> %.1f per iteration)\n  write-out      %8.1f cycles\n", ph[7] / nb, tiles2, ph[7] / nb / tiles2, ph[8] / nb);
    return 0;
}
