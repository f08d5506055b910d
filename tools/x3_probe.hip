// In-kernel phase timing of the deep-pipeline split kernel (conv_x3p_kernel, csrc/conv.hip): s_memtime stamps of thread 0 of every
// workgroup, standalone:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSR_PROBE -Iinclude -Isatlas_super_resolution_amd/csrc tools/x3_probe.hip -o tools/x3_probe
//   tools/x3_probe [N=32] [Cin=64] [Cout=32] [H=32] [W=32]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_probe;
#include "../satlas_super_resolution_amd/csrc/conv.hip"
// the other kernel families are not linked into the probe
bool ssr_conv_res_try(const ssr_conv_desc&, hipStream_t, int*) { return false; }
bool ssr_conv_res_qualifies(const ssr_conv_desc&) { return false; }
bool ssr_conv_ws_try(const ssr_conv_desc&, hipStream_t, int*, bool) { return false; }
bool ssr_conv_ws_qualifies(const ssr_conv_desc&) { return false; }
bool ssr_conv_thin_try(const ssr_conv_desc&, hipStream_t, int*, bool) { return false; }
bool ssr_conv_thin_qualifies(const ssr_conv_desc&) { return false; }
bool ssr_conv_big_try(const ssr_conv_desc&, hipStream_t, int*, bool) { return false; }
bool ssr_conv_big_qualifies(const ssr_conv_desc&) { return false; }
bool ssr_conv_big_batch_try(const ssr_conv_desc*, int, hipStream_t, int*) { return false; }
bool ssr_conv_bigx3_try(const ssr_conv_desc&, hipStream_t, int*, bool) { return false; }
bool ssr_conv_bigx3_qualifies(const ssr_conv_desc&) { return false; }
bool ssr_conv_bigx3_batch_try(const ssr_conv_desc*, int, hipStream_t, int*) { return false; }

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32, Cin = argc > 2 ? atoi(argv[2]) : 64, Cout = argc > 3 ? atoi(argv[3]) : 32;
    const int H = argc > 4 ? atoi(argv[4]) : 32, W = argc > 5 ? atoi(argv[5]) : 32, CS = 192;
    const int CoutPad = (Cout + 31) / 32 * 32, nchunks = (Cin + 15) / 16;
    float *x, *y, *w;
    const size_t nb = (size_t)N * H * W * CS * 4;
    hipMalloc(&x, nb); hipMalloc(&y, nb);
    hipMalloc(&w, (size_t)nchunks * 9 * CoutPad * 64);
    hipMemset(x, 0, nb); hipMemset(w, 0, (size_t)nchunks * 9 * CoutPad * 64);
    const int BN = (CoutPad % 64) == 0 ? 64 : 32;
    const int nblk = N * ((H + 7) / 8) * ((W + 15) / 16) * (CoutPad / BN);
    unsigned long long* probe; hipMalloc(&probe, (size_t)nblk * 24 * 8); hipMemset(probe, 0, (size_t)nblk * 24 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    ssr_conv_desc d{};
    d.dtype = SSR_F32X3; d.x = {x, CS, 0}; d.N = N; d.Hi = H; d.Wi = W; d.up = 1; d.Cin = Cin; d.w = w; d.CoutPad = CoutPad;
    d.KH = d.KW = 3; d.stride = 1; d.pad_y = d.pad_x = 1; d.Gh = H; d.Gw = W; d.Ho = H; d.Wo = W; d.oys = d.oxs = 1;
    d.Cout = Cout; d.y = {y, CS, 64}; d.alpha = 1.f; d.act = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) ssr_conv2d(&d, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 50; ++it) ssr_conv2d(&d, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nblk * 24);
    hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    const int nst = BN == 64 ? nchunks : (nchunks + 1) / 2;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nblk; ++b) { t0 = std::min(t0, h[b * 24]); t1 = std::max(t1, h[b * 24 + 17]); }
    printf("N=%d Cin=%d Cout=%d %dx%d blocks=%d stages=%d: avg launch (back to back, events) = %.2f us; kernel span = %llu ticks (100 MHz)\n", N, Cin, Cout, H, W, nblk, nst,
           ms * 1000 / 50, t1 - t0);
    auto avg = [&](int a, int b_) { double s = 0; for (int b = 0; b < nblk; ++b) s += double(h[b * 24 + b_] - h[b * 24 + a]); return s / nblk; };
    printf("  set-up (descriptors)        %8.1f ticks\n", avg(0, 1));
    printf("  first loads + store + sync  %8.1f\n", avg(1, 2));
    int prev = 2;
    for (int s = 0; s < nst && s < 12; ++s) { printf("  stage %2d                    %8.1f\n", s, avg(prev, 3 + s)); prev = 3 + s; }
    printf("    (stage 0: MFMAs done at +%.1f, LDS store done at +%.1f, barrier passed at +%.1f)\n", avg(2, 20), avg(2, 21), avg(2, 3));
    printf("  reduce + epilogue           %8.1f\n", avg(15, 17));
    printf("  whole workgroup             %8.1f\n", avg(0, 17));
    std::vector<unsigned long long> st; for (int b = 0; b < nblk; ++b) st.push_back(h[b * 24] - t0);
    std::sort(st.begin(), st.end());
    printf("  workgroup start skew: median %llu, p90 %llu, max %llu ticks\n", st[nblk / 2], st[nblk * 9 / 10], st[nblk - 1]);
    return 0;
}
