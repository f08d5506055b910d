// In-kernel phase timing of the K-resident conv kernel (s_memtime stamps), standalone:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSR_PROBE -Iinclude -Isatlas_super_resolution_amd/csrc tools/conv_probe.hip -o tools/conv_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_probe;
#include "../satlas_super_resolution_amd/csrc/conv_res.hip"

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16, Cin = argc > 2 ? atoi(argv[2]) : 64, Cout = argc > 3 ? atoi(argv[3]) : 32;
    const int H = 32, W = 32, CS = 192;
    const int CoutPad = (Cout + 31) / 32 * 32, nchunks = (Cin + 31) / 32;
    __bf16 *x, *w, *y;
    hipMalloc(&x, (size_t)N * H * W * CS * 2); hipMalloc(&y, (size_t)N * H * W * CS * 2);
    hipMalloc(&w, (size_t)nchunks * 9 * CoutPad * 32 * 2);
    hipMemset(x, 0x3c, (size_t)N * H * W * CS * 2); hipMemset(w, 0x3c, (size_t)nchunks * 9 * CoutPad * 32 * 2);
    const int tiles = N * (H / 4) * (W / 16), nblk = tiles * (CoutPad / 32);
    unsigned long long* probe; hipMalloc(&probe, (size_t)nblk * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    ssr_conv_desc d{};
    d.dtype = SSR_BF16; d.x = {x, CS, 0}; d.N = N; d.Hi = H; d.Wi = W; d.up = 1; d.Cin = Cin; d.w = w; d.CoutPad = CoutPad;
    d.KH = d.KW = 3; d.stride = 1; d.pad_y = d.pad_x = 1; d.Gh = H; d.Gw = W; d.Ho = H; d.Wo = W; d.oys = d.oxs = 1;
    d.Cout = Cout; d.y = {y, CS, 64}; d.alpha = 1.f; d.act = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int rc = 0;
    for (int it = 0; it < 5; ++it) ssr_conv_res_try(d, 0, &rc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 50; ++it) ssr_conv_res_try(d, 0, &rc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nblk * 8);
    hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0; double ph[8] = {0};
    for (int b = 0; b < nblk; ++b) {
        t0 = std::min(t0, h[b * 8]); t1 = std::max(t1, h[b * 8 + 5]);
        for (int k = 1; k < 6; ++k) ph[k] += double(h[b * 8 + k] - h[b * 8 + k - 1]);
    }
    printf("N=%d Cin=%d Cout=%d blocks=%d rc=%d  avg launch (back-to-back, event) = %.2f us\n", N, Cin, Cout, nblk, rc, ms * 1000 / 50);
    printf("kernel span (first start -> last end) = %llu ticks\n", t1 - t0);
    const char* names[] = {"", "issue DMA", "wait+barrier", "MFMA loop", "barrier+reduce", "epilogue"};
    for (int k = 1; k < 6; ++k) printf("  %-16s %10.1f ticks avg per block\n", names[k], ph[k] / nblk);
    std::vector<unsigned long long> starts; for (int b = 0; b < nblk; ++b) starts.push_back(h[b * 8] - t0);
    std::sort(starts.begin(), starts.end());
    printf("block start skew: median %llu, p90 %llu, max %llu ticks\n", starts[nblk / 2], starts[nblk * 9 / 10], starts[nblk - 1]);
    return 0;
}
