// Fused dense block: second-generation kernel (csrc/rdb_tile.hip, 8x16 / 8x8 tiles) against the first one (csrc/rdb_fwd.hip),
// which the layer-local oracle tests pin.  Both kernels add the same products in the same order (chunk, kernel row, column,
// k-substep), so every byte of x1..x4 / dpre4..1 and of the block output must be IDENTICAL; then both are timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Isatlas_super_resolution_amd/csrc tools/rdbt_check.hip -o tools/rdbt_check
//   tools/rdbt_check [check|time|all]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#ifdef RT_TRACE
__device__ unsigned long long* g_trace;
#endif
#ifdef SSR_PROBE
__device__ unsigned long long* g_probe;
__device__ unsigned long long* g_probe2;
#endif
#include "../satlas_super_resolution_amd/csrc/rdb_fwd.hip"
#include "../satlas_super_resolution_amd/csrc/rdb_tile.hip"
static int g_rdb_tile_override = 0;   // the harness sets ssr_rdb_desc.tile from it: 0 (here) = the 8 x 8 kernel, 16 = the 8 x 16 kernel

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float urand(float a) { return a * ((rnd() & 0xffff) / 32768.f - 1.f); }

struct Case { int N, H, W; bool r2; };

struct Bufs {
    int N, H, W;
    size_t elems;
    uint16_t *cur, *out, *dout, *dcur, *r2;     // device
    uint16_t* w[2][5];                           // [fwd/bwd][k]
    float* bias[5];
};
static const int CS = 192;

static void fill_dev(uint16_t* d, size_t n, float amp) {
    std::vector<uint16_t> h(n);
    for (size_t k = 0; k < n; ++k) h[k] = bf16_of(urand(amp));
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

static Bufs make(int N, int H, int W) {
    Bufs b{};
    b.N = N; b.H = H; b.W = W; b.elems = (size_t)N * H * W * CS;
    hipMalloc(&b.cur, b.elems * 2); hipMalloc(&b.out, b.elems * 2); hipMalloc(&b.dout, b.elems * 2);
    hipMalloc(&b.dcur, b.elems * 2); hipMalloc(&b.r2, b.elems * 2);
    fill_dev(b.cur, b.elems, 1.f); fill_dev(b.dout, b.elems, 1.f); fill_dev(b.r2, b.elems, 1.f);
    hipMemset(b.out, 0, b.elems * 2); hipMemset(b.dcur, 0, b.elems * 2);
    const int cin[5] = {64, 96, 128, 160, 192}, cp[5] = {32, 32, 32, 32, 64};
    for (int f = 0; f < 2; ++f)
        for (int k = 0; k < 5; ++k) {
            const size_t n = (size_t)cin[k] * 9 * cp[k];
#ifdef RT_X_WCOPIES   // probe: RT_X_WCOPIES identical copies back to back (the kernel deals them to the workgroups of an XCD)
            hipMalloc(&b.w[f][k], n * 2 * RT_X_WCOPIES);
            fill_dev(b.w[f][k], n, 0.06f);
            for (int cpy = 1; cpy < RT_X_WCOPIES; ++cpy) hipMemcpy(b.w[f][k] + n * cpy, b.w[f][k], n * 2, hipMemcpyDeviceToDevice);
#else
            hipMalloc(&b.w[f][k], n * 2);
            fill_dev(b.w[f][k], n, 0.06f);
#endif
        }
    for (int k = 0; k < 5; ++k) {
        std::vector<float> h(64);
        for (auto& v : h) v = urand(0.1f);
        hipMalloc(&b.bias[k], 64 * 4);
        hipMemcpy(b.bias[k], h.data(), 64 * 4, hipMemcpyHostToDevice);
    }
    return b;
}

static ssr_rdb_desc desc_fwd(const Bufs& b, bool r2) {
    ssr_rdb_desc d{};
    d.tile = g_rdb_tile_override == 16 ? 16 : 8;
    d.dtype = SSR_BF16; d.N = b.N; d.H = b.H; d.W = b.W;
    d.in = {b.cur, CS, 0}; d.slices = {b.cur, CS, 0}; d.out = {b.out, CS, 0}; d.mask = {nullptr, 0, 0};
    for (int k = 0; k < 5; ++k) { d.w[k] = b.w[0][k]; d.bias[k] = b.bias[k]; }
    if (r2) { d.alpha5 = 0.04f; d.beta1 = 0.2f; d.r2 = {b.r2, CS, 0}; d.beta2 = 1.f; }
    else { d.alpha5 = 0.2f; d.beta1 = 1.f; d.r2 = {nullptr, 0, 0}; d.beta2 = 0.f; }
    return d;
}
static ssr_rdb_desc desc_bwd(const Bufs& b, bool r2) {
    ssr_rdb_desc d{};
    d.tile = g_rdb_tile_override == 16 ? 16 : 8;
    d.dtype = SSR_BF16; d.N = b.N; d.H = b.H; d.W = b.W;
    d.in = {b.dout, CS, 0}; d.slices = {b.dcur, CS, 0}; d.out = {b.dcur, CS, 0}; d.mask = {b.cur, CS, 0};
    for (int k = 0; k < 5; ++k) { d.w[k] = b.w[1][k]; d.bias[k] = nullptr; }
    d.alpha5 = 1.f; d.beta1 = r2 ? 0.2f : 1.f;
    if (r2) { d.r2 = {b.r2, CS, 0}; d.beta2 = 1.f; } else { d.r2 = {nullptr, 0, 0}; d.beta2 = 0.f; }
    return d;
}

// run one variant and fetch the two buffers it writes
static int run_variant(const Bufs& b, bool bwd, bool r2, int tile, std::vector<uint16_t>& slices, std::vector<uint16_t>& out) {
    g_rdb_tile_override = tile;
    // fresh targets
    std::vector<uint16_t> cur0;
    if (!bwd) { hipMemset(b.out, 0, b.elems * 2); }
    else hipMemset(b.dcur, 0, b.elems * 2);
    ssr_rdb_desc d = bwd ? desc_bwd(b, r2) : desc_fwd(b, r2);
    int rc = bwd ? ssr_rdb_backward(&d, 0) : ssr_rdb_forward(&d, 0);
    hipError_t e = hipDeviceSynchronize();
    if (rc != 0 || e != hipSuccess) { printf("  launch failed rc=%d hip=%d (%s)\n", rc, (int)e, hipGetErrorString(e)); return 1; }
    slices.resize(b.elems); out.resize(b.elems);
    hipMemcpy(slices.data(), bwd ? b.dcur : b.cur, b.elems * 2, hipMemcpyDeviceToHost);
    hipMemcpy(out.data(), bwd ? b.dcur : b.out, b.elems * 2, hipMemcpyDeviceToHost);
    return 0;
}

static long compare(const Bufs& b, const std::vector<uint16_t>& a, const std::vector<uint16_t>& c, int c0, int c1, const char* what, bool verbose) {
    long bad = 0;
    int shown = 0;
    for (size_t p = 0; p < (size_t)b.N * b.H * b.W; ++p)
        for (int ch = c0; ch < c1; ++ch)
            if (a[p * CS + ch] != c[p * CS + ch]) {
                ++bad;
                if (verbose && shown < 6) {
                    const int x = p % b.W, y = (p / b.W) % b.H, n = p / ((size_t)b.W * b.H);
                    printf("      %s mismatch n=%d y=%d x=%d ch=%d: ref %04x new %04x\n", what, n, y, x, ch, a[p * CS + ch], c[p * CS + ch]);
                    ++shown;
                }
            }
    return bad;
}

static int check_case(const Case& cs) {
    Bufs b = make(cs.N, cs.H, cs.W);
    int fails = 0;
    for (int bwd = 0; bwd < 2; ++bwd) {
        // the forward writes x1..x4 into `cur` (channels 64..191): keep the input part, reset the slices before each variant
        std::vector<uint16_t> ref_s, ref_o, s, o;
        std::vector<uint16_t> cur_backup(b.elems);
        hipMemcpy(cur_backup.data(), b.cur, b.elems * 2, hipMemcpyDeviceToHost);
        if (run_variant(b, bwd, cs.r2, 0, ref_s, ref_o)) return 1;
        std::vector<uint16_t> cur_after(b.elems);
        hipMemcpy(cur_after.data(), b.cur, b.elems * 2, hipMemcpyDeviceToHost);
        for (int tile : {16}) {
            if (!bwd) hipMemcpy(b.cur, cur_backup.data(), b.elems * 2, hipMemcpyHostToDevice);
            if (run_variant(b, bwd, cs.r2, tile, s, o)) return 1;
            long bad_total = 0;
            printf("  N=%d %dx%d r2=%d %s tile %2d:", cs.N, cs.H, cs.W, (int)cs.r2, bwd ? "bwd" : "fwd", tile);
            for (int k = 1; k <= 4; ++k) {
                char nm[32]; snprintf(nm, sizeof nm, "slice%d", k);
                long bad = compare(b, ref_s, s, 64 + 32 * (k - 1), 64 + 32 * k, nm, true);
                printf(" s%d:%ld", k, bad);
                bad_total += bad;
            }
            long bad = compare(b, ref_o, o, 0, 64, "out", true);
            printf(" out:%ld %s\n", bad, (bad_total + bad) ? "MISMATCH" : "identical");
            fails += (bad_total + bad) != 0;
        }
        // leave the forward activations of the reference run in `cur` for the backward's masks
        if (!bwd) hipMemcpy(b.cur, cur_after.data(), b.elems * 2, hipMemcpyHostToDevice);
    }
    hipFree(b.cur); hipFree(b.out); hipFree(b.dout); hipFree(b.dcur); hipFree(b.r2);
    for (int f = 0; f < 2; ++f) for (int k = 0; k < 5; ++k) hipFree(b.w[f][k]);
    for (int k = 0; k < 5; ++k) hipFree(b.bias[k]);
    return fails;
}

static void time_case(int N, int H, int W) {
    Bufs b = make(N, H, W);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double gflop = 2.0 * 9 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64) * (double)N * H * W * 1e-9;
    for (int bwd = 0; bwd < 2; ++bwd)
        for (int tile : {0, 16}) {
            g_rdb_tile_override = tile;
            ssr_rdb_desc d = bwd ? desc_bwd(b, false) : desc_fwd(b, false);
            auto run = [&]() { return bwd ? ssr_rdb_backward(&d, 0) : ssr_rdb_forward(&d, 0); };
            for (int it = 0; it < 5; ++it) run();
            hipDeviceSynchronize();
            float best = 1e30f, sum = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < 20; ++it) run();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best; sum += ms;
            }
            const double us = best * 1000 / 20;
            printf("  time N=%d %dx%d %s tile %2d: %.2f us per launch (mean %.2f)  %.0f TFLOP/s = %.3f of 2500\n", N, H, W,
                   bwd ? "bwd" : "fwd", tile, us, sum * 1000 / 100, gflop / us * 1e3, gflop / us / 2.5);
        }
    hipFree(b.cur); hipFree(b.out); hipFree(b.dout); hipFree(b.dcur); hipFree(b.r2);
}

// The step's launch sequence: NB dense blocks back to back, each with its OWN buffers; weights distinct per block (as in
// the network: cold in L2 when the launch starts) or shared (hot after the first launch).  Forward only, 8 x 16 kernel.
static void chain_case(int NB, int N) {
    std::vector<Bufs> bs;
    for (int k = 0; k < NB; ++k) bs.push_back(make(N, 32, 32));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    g_rdb_tile_override = 16;
    const int cin[5] = {64, 96, 128, 160, 192}, cp[5] = {32, 32, 32, 32, 64};
    for (int shared = 0; shared < 3; ++shared) {             // 2: distinct weights, each launch warms L2 for the next
        std::vector<ssr_rdb_desc> ds;
        for (int k = 0; k < NB; ++k) {
            ssr_rdb_desc d = desc_fwd(bs[k], false);
            if (shared == 1) for (int j = 0; j < 5; ++j) d.w[j] = bs[0].w[0][j];
            if (shared == 2) for (int j = 0; j < 5; ++j) { d.w_next[j] = bs[(k + 1) % NB].w[0][j]; d.w_next_bytes[j] = cin[j] * 9 * cp[j] * 2; }
            ds.push_back(d);
        }
        for (int k = 0; k < NB; ++k) ssr_rdb_forward(&ds[k], 0);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < NB; ++k) ssr_rdb_forward(&ds[k], 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("  chain of %d blocks, N=%d, weights %s: %.2f us per launch\n", NB, N, shared == 1 ? "shared (hot)" : shared == 2 ? "distinct + w_next warm-up" : "distinct (cold)", best * 1000 / NB);
    }
}

#ifdef SSR_PROBE
// phase timing (s_memtime ticks of wave 0 / producer wave 4, averaged over the blocks of one launch) of the new kernel
static void probe_case(int N, int tile, bool bwd) {
    Bufs b = make(N, 32, 32);
    g_rdb_tile_override = tile;
    const int nblk = N * 4 * (tile == 16 ? 2 : 4);
    unsigned long long* probe; hipMalloc(&probe, (size_t)nblk * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    unsigned long long* probe2; hipMalloc(&probe2, (size_t)nblk * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe2), &probe2, sizeof(probe2));
    ssr_rdb_desc d = bwd ? desc_bwd(b, false) : desc_fwd(b, false);
    for (int it = 0; it < 3; ++it) { bwd ? ssr_rdb_backward(&d, 0) : ssr_rdb_forward(&d, 0); }
    hipDeviceSynchronize();
    hipMemset(probe, 0, (size_t)nblk * 16 * 8);
    bwd ? ssr_rdb_backward(&d, 0) : ssr_rdb_forward(&d, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)nblk * 16);
    hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    double ph[8] = {0}, ex[5] = {0};
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int k = 0; k < nblk; ++k) {
        for (int j = 1; j < 8; ++j) ph[j] += double(h[k * 16 + j] - h[k * 16 + j - 1]);
        for (int j = 0; j < 5; ++j) ex[j] += double(h[k * 16 + 11 + j]);
        tmin = h[k * 16] < tmin ? h[k * 16] : tmin; tmax = h[k * 16 + 7] > tmax ? h[k * 16 + 7] : tmax;
    }
    printf("probe N=%d tile %d %s: blocks %d, first start -> last end %llu ticks (s_memtime, 100 MHz?)\n", N, tile, bwd ? "bwd" : "fwd", nblk, tmax - tmin);
    const char* names[] = {"", "prologue (x0 + queue)", "conv1", "conv2", "conv3", "conv4", "conv5 mfma", "conv5 epilogue"};
    double tot = 0;
    for (int j = 1; j < 8; ++j) { printf("    %-22s %9.1f\n", names[j], ph[j] / nblk); tot += ph[j] / nblk; }
    printf("    total %.1f | wave 0: slice waits %.1f, slab waits %.1f over %.1f slabs | producer 4: waits for consumers %.1f, loads+store %.1f\n",
           tot, ex[2] / nblk, ex[3] / nblk, ex[4] / nblk, ex[0] / nblk, ex[1] / nblk);
    std::vector<unsigned long long> h2((size_t)nblk * 16);
    hipMemcpy(h2.data(), probe2, h2.size() * 8, hipMemcpyDeviceToHost);
    for (int K = 1; K <= 4; ++K) {
        double seg[3] = {0, 0, 0};
        for (int k = 0; k < nblk; ++k) for (int u = 0; u < 3; ++u) seg[u] += double(h2[k * 16 + 4 * (K - 1) + u + 1] - h2[k * 16 + 4 * (K - 1) + u]);
        printf("    stage %d of wave 0: setup + prime %.0f, main loop %.0f, slice store %.0f\n", K, seg[0] / nblk, seg[1] / nblk, seg[2] / nblk);
    }
    hipFree(probe); hipFree(probe2);
}
#endif

#ifdef RT_TRACE
static void trace_case(int N, bool bwd) {
    Bufs b = make(N, 32, 32);
    g_rdb_tile_override = 16;
    unsigned long long* tr; hipMalloc(&tr, 12 * 512 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &tr, sizeof(tr));
    ssr_rdb_desc d = bwd ? desc_bwd(b, false) : desc_fwd(b, false);
    for (int it = 0; it < 3; ++it) { hipMemset(tr, 0, 12 * 512 * 8); bwd ? ssr_rdb_backward(&d, 0) : ssr_rdb_forward(&d, 0); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(12 * 512);
    hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < 12; ++w) for (unsigned long long k = 0; k < h[w * 512]; ++k) { unsigned long long t = h[w * 512 + 1 + k] & 0xffffffffffffull; if (t < t0) t0 = t; }
    for (int w = 0; w < 12; ++w)
        for (unsigned long long k = 0; k < h[w * 512]; ++k) {
            const unsigned long long e = h[w * 512 + 1 + k];
            printf("T %d %d %d %llu\n", w, (int)(e >> 56), (int)((e >> 48) & 0xff), (e & 0xffffffffffffull) - t0);
        }
}
#endif

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "all";
    int fails = 0;
#ifdef RT_TRACE
    if (!strcmp(mode, "trace")) { trace_case(32, argc > 2); return 0; }
#endif
#ifdef SSR_PROBE
    if (!strcmp(mode, "probe")) {
        const int pn = argc > 2 ? atoi(argv[2]) : 32;
        for (int bwd = 0; bwd < 2; ++bwd) probe_case(pn, 16, bwd);
        return 0;
    }
#endif
    if (!strcmp(mode, "check") || !strcmp(mode, "all")) {
        const Case cases[] = {{2, 32, 32, false}, {8, 32, 32, true}, {3, 24, 40, false}, {1, 8, 8, true}, {2, 20, 12, false}, {32, 32, 32, false}};
        for (const Case& c : cases) fails += check_case(c);
        printf("check: %d failing variant(s)\n", fails);
    }
    if (!strcmp(mode, "time32")) time_case(32, 32, 32);
    if (!strcmp(mode, "chain")) chain_case(argc > 2 ? atoi(argv[2]) : 69, 32);
    if (!strcmp(mode, "timen")) for (int a = 2; a < argc; ++a) time_case(atoi(argv[a]), 32, 32);
    if (!strcmp(mode, "time") || !strcmp(mode, "all")) {
        time_case(32, 32, 32);
        time_case(16, 32, 32);
        time_case(64, 32, 32);
    }
    return fails ? 1 : 0;
}
