// Phase stamps of the persistent dense-block chain kernel (conv_x3c_kernel, csrc/conv_x3c.hip), MFMA wave 0 of every workgroup:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSR_PROBE -Iinclude -Isatlas_super_resolution_amd/csrc tools/x3c_probe.hip -o tools/x3c_probe
//   tools/x3c_probe [N=32]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_probe;
#include "../satlas_super_resolution_amd/csrc/conv_x3c.hip"
#undef SSR_PROBE
#include "../satlas_super_resolution_amd/csrc/conv_x3r.hip"
extern "C" int ssr_conv2d(const ssr_conv_desc*, void*) { return SSR_EUNSUP; }
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32, H = 32, W = 32, CS = 192;
    float *x, *w;
    const size_t nb = (size_t)N * H * W * CS * 4;
    hipMalloc(&x, nb); hipMemset(x, 0, nb);
    hipMalloc(&w, (size_t)12 * 9 * 32 * 64); hipMemset(w, 0, (size_t)12 * 9 * 32 * 64);
    const int nblk = N * 8;
    unsigned long long* probe; hipMalloc(&probe, (size_t)nblk * 32 * 8); hipMemset(probe, 0, (size_t)nblk * 32 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    void* state; hipMalloc(&state, ssr_conv2d_chain_state_bytes(N, H, W)); hipMemset(state, 0, ssr_conv2d_chain_state_bytes(N, H, W));
    ssr_conv_desc d[4] = {};
    for (int k = 0; k < 4; ++k) {
        d[k].dtype = SSR_F32X3; d[k].x = {x, CS, 0}; d[k].N = N; d[k].Hi = H; d[k].Wi = W; d[k].up = 1; d[k].Cin = 64 + 32 * k; d[k].w = w; d[k].CoutPad = 32;
        d[k].KH = d[k].KW = 3; d[k].stride = 1; d[k].pad_y = d[k].pad_x = 1; d[k].Gh = H; d[k].Gw = W; d[k].Ho = H; d[k].Wo = W; d[k].oys = d[k].oxs = 1;
        d[k].Cout = 32; d[k].y = {x, CS, 64 + 32 * k}; d[k].alpha = 1.f; d[k].act = 1;
    }
    printf("chain ok = %d\n", ssr_conv2d_chain_ok(d, 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) ssr_conv2d_chain(d, 4, state, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 50; ++it) ssr_conv2d_chain(d, 4, state, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemset(probe, 0, (size_t)nblk * 32 * 8);
    int rc = ssr_conv2d_chain(d, 4, state, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)nblk * 32);
    hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    printf("N=%d rc=%d: chain launch (back to back, events) %.2f us\n", N, rc, ms * 1000 / 50);
    auto avg = [&](int a, int b_) { double s = 0; int c = 0; for (int b = 0; b < nblk; ++b) if (h[b * 32 + a] && h[b * 32 + b_]) { s += double(h[b * 32 + b_] - h[b * 32 + a]); ++c; } return c ? s / c : -1.0; };
    printf("  entry -> first chunk of conv 1 in LDS        %8.1f ticks\n", avg(0, 1));
    int prev = 1;
    for (int ci = 0; ci < 4; ++ci) {
        printf("  conv %d: MFMA loop (%2d chunks)               %8.1f\n", ci + 1, 4 + 2 * ci, avg(ci == 0 ? 1 : 4 + 4 * (ci - 1), 2 + 4 * ci));
        printf("          partials written + met the others     %8.1f\n", avg(2 + 4 * ci, 3 + 4 * ci));
        printf("          sum + epilogue + drained + published  %8.1f\n", avg(3 + 4 * ci, 4 + 4 * ci));
        prev = 4 + 4 * ci;
    }
    printf("  exit                                          %8.1f\n  whole workgroup                               %8.1f\n", avg(prev, 31), avg(0, 31));
    return 0;
}
