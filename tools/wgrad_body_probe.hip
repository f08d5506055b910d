// The generator-body weight-gradient launch as the step issues it: NRDB dense blocks x 14 (conv, co, ci) items over DISTINCT
// 192-channel buffers (N = 32, 32 x 32), item order 0 = layer-major (rounds 1-2), 1 = one dense block per XCD queue.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DSSR_PROBE -Iinclude -Isatlas_super_resolution_amd/csrc tools/wgrad_body_probe.hip -o tools/wgrad_body_probe
//   tools/wgrad_body_probe [nrdb=69] [order=0|1|2 (2 = heavy first)] [shared=0|1] [pair=0|1] [cut: 0 none, N > 0 at most N tiles per item,
//                          -1 = equal shares: the cost-ordered item sequence cut at multiples of total / 256, longest piece first]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_probe;
#include "../satlas_super_resolution_amd/csrc/wgrad_bf16.hip"
int main(int argc, char** argv) {
    const int N = 32, H = 32, W = 32, CS = 192;
    const int nrdb = argc > 1 ? atoi(argv[1]) : 69, order = argc > 2 ? atoi(argv[2]) : 0, shared = argc > 3 ? atoi(argv[3]) : 0, pairing = argc > 4 ? atoi(argv[4]) : 0, cut = argc > 5 ? atoi(argv[5]) : 0;
    const size_t bufb = (size_t)N * H * W * CS * 2;
    const int nbuf = shared ? 1 : nrdb + 1;
    char *x, *dy; float* dw;
    hipMalloc(&x, bufb * nbuf); hipMalloc(&dy, bufb * nbuf);
    hipMemset(x, 0x3c, bufb * nbuf); hipMemset(dy, 0x3c, bufb * nbuf);
    const size_t wper = 9 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64);
    hipMalloc(&dw, nrdb * wper * 4); hipMemset(dw, 0, nrdb * wper * 4);
    std::vector<ssr_wgrad_layer> L; std::vector<std::vector<ssr_wgrad_item>> groups(nrdb);
    const int tiles = N * (H / 16) * (W / 16);
    double flop = 0;
    for (int r = 0; r < nrdb; ++r) {
        const int b = shared ? 0 : r;
        size_t woff = 0;
        for (int k = 0; k < 5; ++k) {
            const int cin = 64 + 32 * k, cout = k == 4 ? 64 : 32;
            ssr_view xv{x + bufb * b, CS, 0};
            ssr_view dv = k < 4 ? ssr_view{dy + bufb * b, CS, 64 + 32 * k} : ssr_view{dy + bufb * (shared ? 0 : b + 1), CS, 0};
            L.push_back(ssr_wgrad_layer{xv, dv, N, H, W, 1, cin, cout, 1, 1, H, W, 1.f, dw + r * wper + woff, cin, nullptr});
            woff += (size_t)9 * cin * cout;
            flop += 2.0 * N * H * W * cin * cout * 9;
            for (int co = 0; co < cout; co += 32) for (int ci = 0; ci < cin; ci += 64) groups[r].push_back({(int)L.size() - 1, co, ci, 0, tiles, 1, 1, 0, 0});
        }
        if (pairing) {   // same ci0 and the same number of valid input channels -> one item with two dY planes
            std::vector<ssr_wgrad_item> out; std::vector<char> used(groups[r].size(), 0);
            for (size_t a = 0; a < groups[r].size(); ++a) {
                if (used[a]) continue;
                auto ia = groups[r][a]; used[a] = 1;
                const bool fa = L[ia.layer].Cin_w - ia.ci0 > 32;
                for (size_t b = a + 1; b < groups[r].size(); ++b) {
                    auto ib = groups[r][b];
                    if (!used[b] && ib.ci0 == ia.ci0 && (L[ib.layer].Cin_w - ib.ci0 > 32) == fa) { used[b] = 1; ia.nco = 2; ia.layer_b = ib.layer; ia.co0_b = ib.co0; break; }
                }
                out.push_back(ia);
            }
            groups[r] = out;
        }
    }
    auto weight = [&](const ssr_wgrad_item& it) { return (L[it.layer].Cin_w - it.ci0 > 32 ? 2 : 1) * (it.nco == 2 ? 2 : 1); };
    std::vector<ssr_wgrad_item> I;
    if (order == 0) { for (auto& g : groups) for (auto& it : g) I.push_back(it); }
    else if (order == 2) { for (auto& g : groups) for (auto& it : g) I.push_back(it);
                           std::stable_sort(I.begin(), I.end(), [&](auto& a, auto& b) { return weight(a) > weight(b); }); }
    else {
        std::vector<std::vector<ssr_wgrad_item>> q(8);
        for (int r = 0; r < nrdb; ++r) for (auto& it : groups[r]) q[r % 8].push_back(it);
        for (;;) { auto lo = std::min_element(q.begin(), q.end(), [](auto& a, auto& b) { return a.size() < b.size(); });
                   auto hi = std::max_element(q.begin(), q.end(), [](auto& a, auto& b) { return a.size() < b.size(); });
                   if (hi->size() - lo->size() <= 1) break; lo->push_back(hi->back()); hi->pop_back(); }
        std::stable_sort(q.begin(), q.end(), [](auto& a, auto& b) { return a.size() > b.size(); });
        for (size_t j = 0; j < q[0].size(); ++j) for (auto& qq : q) if (j < qq.size()) I.push_back(qq[j]);
    }
    if (cut != 0) {
        auto cost = [&](const ssr_wgrad_item& it) { return (double)(it.tile_end - it.tile_begin) * (weight(it) == 4 ? 5900.0 : weight(it) == 2 ? 4300.0 : 3000.0); };
        std::vector<ssr_wgrad_item> out;
        if (cut <= -2) {   // whole rounds stay whole; the items of the last, partial round are cut into (-cut - 1) * 256 pieces
            const size_t whole = I.size() / 256 * 256, rest = I.size() - whole;
            for (size_t i = 0; i < whole; ++i) out.push_back(I[i]);
            if (rest) {
                const int per_item = std::max(1, (int)(((-cut - 1) * 256 + rest - 1) / rest));
                for (size_t i = whole; i < I.size(); ++i) {
                    const auto it = I[i]; const int n = it.tile_end - it.tile_begin, step = (n + per_item - 1) / per_item;
                    for (int b = it.tile_begin; b < it.tile_end; b += step) { auto p = it; p.tile_begin = b; p.tile_end = std::min(it.tile_end, b + step); p.atomic = 1; out.push_back(p); }
                }
            }
        } else if (cut > 0) {
            for (auto it : I) for (int b = it.tile_begin; b < it.tile_end; b += cut) { auto p = it; p.tile_begin = b; p.tile_end = std::min(it.tile_end, b + cut); p.atomic = 1; out.push_back(p); }
        } else {
            double total = 0; for (auto& it : I) total += cost(it);
            const double T = total / 256; double acc = 0; int k = 1;
            for (auto it : I) {
                const double per = cost(it) / (it.tile_end - it.tile_begin);
                int b = it.tile_begin;
                while (b < it.tile_end) {
                    const double room = k * T - acc;                       // work left in the current share
                    int n = std::min(it.tile_end - b, std::max(1, (int)(room / per + 0.5)));
                    if (it.tile_end - b - n < 4) n = it.tile_end - b;       // no crumbs
                    auto p = it; p.tile_begin = b; p.tile_end = b + n; p.atomic = 1; out.push_back(p);
                    acc += n * per; b += n;
                    if (acc >= k * T - 0.5 * per) ++k;
                }
            }
            std::stable_sort(out.begin(), out.end(), [&](auto& a, auto& b) { return cost(a) > cost(b); });
        }
        I = out;
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
    const int reps = 5;
    for (int it = 0; it < reps; ++it) ssr_wgrad_bf16_dispatch(Ld, Id, (int)I.size(), 3, 3, 1, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(I.size() * 16); hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost);
    printf("nrdb=%d order=%d shared=%d items=%zu launch %.1f us  %.1f GFLOP -> %.0f TFLOP/s\n", nrdb, order, shared, I.size(), ms * 1000 / reps,
           flop / 1e9, flop / (ms / reps * 1e-3) / 1e12);
    const int nb = (int)I.size();
    double loop = 0, wo = 0, loop18 = 0, loop9 = 0; int n18 = 0, n9 = 0;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nb; ++b) {
        const double l = double(h[b * 16 + 7] - h[b * 16 + 0]);
        loop += l; wo += double(h[b * 16 + 8] - h[b * 16 + 7]);
        const bool half = weight(I[b]) < 4;
        (half ? loop9 : loop18) += l; (half ? n9 : n18)++;
        t0 = std::min(t0, h[b * 16 + 0]); t1 = std::max(t1, h[b * 16 + 8]);
    }
    printf("  loop %.0f cycles per item (%.0f per tile; heaviest items %.0f, others %.0f per tile), write-out %.0f; launch span %.0f cycles\n",
           loop / nb, loop / nb / tiles, n18 ? loop18 / n18 / tiles : 0., n9 ? loop9 / n9 / tiles : 0., wo / nb, double(t1 - t0));
    {   // timeline on the device-wide 100-MHz clock: span of the launch, CU time in use, workgroups per CU
        unsigned long long r0 = ~0ull, r1 = 0; double busy = 0; std::vector<unsigned long long> ids;
        for (int b = 0; b < nb; ++b) { r0 = std::min(r0, h[b * 16 + 9]); r1 = std::max(r1, h[b * 16 + 10]); busy += double(h[b * 16 + 10] - h[b * 16 + 9]);
                                       ids.push_back(h[b * 16 + 11] & 0xf0000ff00ull | (h[b * 16 + 11] & 0xe000ull)); }
        std::sort(ids.begin(), ids.end()); const size_t ncu = std::unique(ids.begin(), ids.end()) - ids.begin();
        printf("  timeline: span %.1f us, sum of workgroup times %.1f us = %.2f of %zu CUs x span; mean workgroup %.1f us\n", (r1 - r0) / 100.0, busy / 100.0,
               busy / (double(r1 - r0) * ncu), ncu, busy / 100.0 / nb);
        // CU time in use per tenth of the span
        double use[10] = {0};
        for (int b = 0; b < nb; ++b) for (int i = 0; i < 10; ++i) {
            const double a = r0 + (r1 - r0) * i / 10.0, e = r0 + (r1 - r0) * (i + 1) / 10.0;
            use[i] += std::max(0.0, std::min(e, (double)h[b * 16 + 10]) - std::max(a, (double)h[b * 16 + 9]));
        }
        printf("  CUs in use per tenth of the span:"); for (int i = 0; i < 10; ++i) printf(" %.0f", use[i] / ((r1 - r0) / 10.0)); printf("\n");
    }
    // start-time histogram of the blocks (in 10ths of the span): how the rounds fall
    int hist[10] = {0}; for (int b = 0; b < nb; ++b) hist[std::min(9, int(10.0 * double(h[b * 16] - t0) / double(t1 - t0)))]++;
    printf("  starts per tenth of the span:"); for (int i = 0; i < 10; ++i) printf(" %d", hist[i]); printf("\n");
    return 0;
}
