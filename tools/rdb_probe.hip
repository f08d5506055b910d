// Phase timing of the fused RDB forward kernel.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSR_PROBE -Iinclude -Isatlas_super_resolution_amd/csrc tools/rdb_probe.hip -o tools/rdb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_probe;
#include "../satlas_super_resolution_amd/csrc/rdb_fwd.hip"
// the 8x16-tile kernel (round 3) is a separate translation unit in the library; this probe times the 8x8 kernel only (desc.tile = 8 / SSR_RDB_TILE=0)
int rdbt_launch(const ssr_rdb_desc&, void*, bool, int) { return SSR_EUNSUP; }
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16, H = 32, W = 32, CS = 192;
    __bf16 *cur, *out, *w[5];
    hipMalloc(&cur, (size_t)N * H * W * CS * 2); hipMalloc(&out, (size_t)N * H * W * CS * 2);
    hipMemset(cur, 0x3c, (size_t)N * H * W * CS * 2);
    const int cin[5] = {64, 96, 128, 160, 192}, cp[5] = {32, 32, 32, 32, 64};
    ssr_rdb_desc d{}; d.tile = 8;
    for (int k = 0; k < 5; ++k) { size_t b = (size_t)cin[k] * 9 * cp[k] * 2; hipMalloc(&w[k], b); hipMemset(w[k], 0x3c, b); d.w[k] = w[k]; }
    d.dtype = SSR_BF16; d.N = N; d.H = H; d.W = W; d.in = {cur, CS, 0}; d.slices = {cur, CS, 0}; d.mask = {cur, CS, 0}; d.out = {out, CS, 0}; d.alpha5 = 0.2f; d.beta1 = 1.f;
    const int nblk = N * 16;
    unsigned long long* probe; hipMalloc(&probe, (size_t)nblk * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const bool bwd = argc > 2;
    auto run = [&]() { return bwd ? ssr_rdb_backward(&d, 0) : ssr_rdb_forward(&d, 0); };
    for (int it = 0; it < 3; ++it) run();
    hipDeviceSynchronize();
    hipMemset(probe, 0, (size_t)nblk * 16 * 8);               // slots 14 / 15 accumulate over the 20 timed launches
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) run();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nblk * 16);
    hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    double ph[8] = {0};
    for (int b = 0; b < nblk; ++b) for (int k = 1; k < 8; ++k) ph[k] += double(h[b * 16 + k] - h[b * 16 + k - 1]);
    printf("N=%d blocks=%d avg launch = %.2f us\n", N, nblk, ms * 1000 / 20);
    const char* names[] = {"", "issue x + slab0", "conv1", "conv2", "conv3", "conv4", "conv5 mfma", "conv5 epilogue"};
    for (int k = 1; k < 8; ++k) printf("  %-18s %10.1f ticks avg\n", names[k], ph[k] / nblk);
    double st1[3] = {0, 0, 0};                                // stage 1 of wave 0: setup, wait for slab 0, both chunks
    for (int b = 0; b < nblk; ++b) {
        st1[0] += double(h[b * 16 + 8] - h[b * 16 + 1]); st1[1] += double(h[b * 16 + 9] - h[b * 16 + 8]);
        st1[2] += double(h[b * 16 + 10] - h[b * 16 + 9]);
    }
    printf("    conv1: setup %.0f, wait for slab 0 %.0f, two chunks %.0f, slice store + flag %.0f\n", st1[0] / nblk, st1[1] / nblk,
           st1[2] / nblk, (ph[2] - st1[0] - st1[1] - st1[2]) / nblk);
    double pp[3] = {0, 0, 0};
    for (int b = 0; b < nblk; ++b) for (int k = 0; k < 3; ++k) pp[k] += double(h[b * 16 + 11 + k]);
    printf("  producer wave 4 per block: waits for consumers %.0f, waits for its loads + stores %.0f, issues loads %.0f ticks\n",
           pp[0] / nblk / 20, pp[1] / nblk / 20, pp[2] / nblk / 20);
    double wt = 0, wn = 0;
    for (int b = 0; b < nblk; ++b) { wt += double(h[b * 16 + 14]); wn += double(h[b * 16 + 15]); }
    printf("  slab polls (wave 0): %.1f ticks per launch-block, %.2f fallback polls (x1000) + waits\n", wt / nblk / 20, wn / nblk / 20);
    return 0;
}
