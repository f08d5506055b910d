// In-kernel phase timing of the register-tiled body kernel (conv_x3r_kernel, csrc/conv_x3r.hip) beside the ring kernel it replaces
// (conv_x3q_kernel): back-to-back launch time by events for both, s_memtime stamps of one thread per role for the new one, standalone:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSR_PROBE -Iinclude -Isatlas_super_resolution_amd/csrc tools/x3r_probe.hip -o tools/x3r_probe
//   tools/x3r_probe [N=32] [Cin=64] [Cout=32] [H=32] [W=32] [epi=0 lrelu | 1 residual | 2 mask | 3 generic (y0)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_probe;
#include "../satlas_super_resolution_amd/csrc/conv_x3r.hip"
#undef SSR_PROBE
#include "../satlas_super_resolution_amd/csrc/conv_x3q.hip"
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32, Cin = argc > 2 ? atoi(argv[2]) : 64, Cout = argc > 3 ? atoi(argv[3]) : 32;
    const int H = argc > 4 ? atoi(argv[4]) : 32, W = argc > 5 ? atoi(argv[5]) : 32, epi = argc > 6 ? atoi(argv[6]) : 0, CS = 192;
    const int CoutPad = (Cout + 31) / 32 * 32, nchunks = (Cin + 15) / 16;
    float *x, *y, *w, *r, *y0;
    const size_t nb = (size_t)N * H * W * CS * 4;
    hipMalloc(&x, nb); hipMalloc(&y, nb); hipMalloc(&r, nb); hipMalloc(&y0, nb);
    hipMalloc(&w, (size_t)nchunks * 9 * CoutPad * 64);
    hipMemset(x, 0, nb); hipMemset(r, 0, nb); hipMemset(w, 0, (size_t)nchunks * 9 * CoutPad * 64);
    const int nblk = N * ((H + 3) / 4) * ((W + 15) / 16) * (CoutPad / 32);      // (upper bound: half-height tiles, 32-channel workgroups)
    unsigned long long* probe; hipMalloc(&probe, (size_t)nblk * 16 * 8); hipMemset(probe, 0, (size_t)nblk * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    ssr_conv_desc d{};
    d.dtype = SSR_F32X3; d.x = {x, CS, 0}; d.N = N; d.Hi = H; d.Wi = W; d.up = 1; d.Cin = Cin; d.w = w; d.CoutPad = CoutPad;
    d.KH = d.KW = 3; d.stride = 1; d.pad_y = d.pad_x = 1; d.Gh = H; d.Gw = W; d.Ho = H; d.Wo = W; d.oys = d.oxs = 1;
    d.Cout = Cout; d.y = {y, CS, 64}; d.alpha = 1.f; d.act = 1;
    if (epi == 1) { d.act = 0; d.alpha = 0.2f; d.r1 = {r, CS, 0}; d.r1_nc = Cout; d.beta1 = 1.f; }
    if (epi == 2) { d.act = 0; d.m = {r, CS, 64}; d.m_c0 = 0; d.m_c1 = Cout; }
    if (epi == 3) { d.y0 = {y0, CS, 0}; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int rc = 0;
    float ms[2];
    for (int k = 0; k < 2; ++k) {
        auto run = [&]() { if (k == 0) ssr_conv_x3r_try(d, 0, &rc, true); else ssr_conv_x3q_try(d, 0, &rc, true); };
        for (int it = 0; it < 5; ++it) run();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int it = 0; it < 50; ++it) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[k], e0, e1);
    }
    hipMemset(probe, 0, (size_t)nblk * 16 * 8);
    ssr_conv_x3r_try(d, 0, &rc, true);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)nblk * 16);
    hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    printf("N=%d Cin=%d Cout=%d %dx%d epi=%d workgroups=%d chunks=%d rc=%d: avg launch (back to back, events) register-tiled %.2f us | ring %.2f us\n",
           N, Cin, Cout, H, W, epi, nblk, nchunks, rc, ms[0] * 1000 / 50, ms[1] * 1000 / 50);
    auto avg = [&](int a, int b_) { double s = 0; int cnt = 0; for (int b = 0; b < nblk; ++b) if (h[b * 16 + b_] && h[b * 16 + a]) { s += double(h[b * 16 + b_] - h[b * 16 + a]); ++cnt; } return cnt ? s / cnt : -1.0; };
    printf("  MFMA wave 0: entry -> weight loads issued %8.1f ticks, -> first chunk in LDS %8.1f\n", avg(0, 1), avg(0, 2));
    int prev = 2;
    const int nj = (3 * nchunks + 3) / 4;
    for (int j = 0; j < nj && j < 5; ++j) { printf("    item %d (3 taps x 4 tiles)            %8.1f\n", j, avg(prev, 3 + j)); prev = 3 + j; }
    printf("    remaining items + barrier           %8.1f\n", avg(prev, 10));
    printf("    K-quarter sum + epilogue            %8.1f\n", avg(10, 11));
    printf("    whole workgroup                     %8.1f\n", avg(0, 11));
    printf("  producer wave 0: entry -> first loads issued %8.1f, stores + refills of all chunks %8.1f\n", avg(0, 8), avg(8, 9));
    return 0;
}
