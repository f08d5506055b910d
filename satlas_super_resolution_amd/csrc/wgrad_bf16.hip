// bf16 weight gradient on v_mfma_f32_32x32x16_bf16 with LDS transpose reads.
//
// wgrad contracts over PIXELS, the strided axis of NHWC, while the bf16 MFMA wants 8 consecutive
// k-values per lane.  gfx950's ds_read_b64_tr_b16 does the transpose in the LDS read path: within a
// 16-lane group, source lane 4j+q supplies 4 contiguous bf16 (channels 4q..4q+3 of pixel j) and output
// lane c receives channel c of pixels j = 0..3 (semantics measured with tools/probe_tr16.hip).  So a
// dense [pixel][32 ch] LDS image — exactly what coalesced 16-byte NHWC loads produce — feeds both MFMA
// operands directly:   A[i = co][k = pixel] <- dY tile,   B[k = pixel][j = ci] <- X halo patch + tap shift
// (the tap shift is just a different per-lane pixel address; no alignment constraint).  A 64-byte row
// (32 bf16) makes the 4 pixel rows of a 32-lane read tile the 64 banks exactly: conflict-free.
//
// Work decomposition is the same as wgrad.hip (device tables, one (layer, 32co, 32ci) tile per workgroup,
// all taps in registers); the pixel-tile loop is software pipelined: global loads of tile t+1 are in
// flight in registers while tile t is contracted, LDS is double buffered, one barrier per tile.
#include "wgrad_common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_pair(const __bf16* lo, const __bf16* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int KH, int KW, int S, bool SPLIT_TAPS>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const ssr_wgrad_layer* __restrict__ layers,
                                                         const ssr_wgrad_item* __restrict__ items) {
    constexpr int TH = wgrad_bf16_th(KH);                    // pixel-tile rows: 16 for 3x3, 8 for 4x4
    constexpr int PH = (TH - 1) * S + KH, PW = (WG_TW - 1) * S + KW;
    constexpr int NTAP = SPLIT_TAPS ? KW : KH * KW;
    constexpr int ROW = 32;                                  // bf16 per pixel row (64 B)
    constexpr int NDY = TH * WG_TW * 4 / 256;             // 16-B vectors per thread: dY tile
    constexpr int NXV = (PH * PW * 4 + 255) / 256;           //                           X patch
    constexpr int STAGE = (TH * WG_TW + PH * PW) * ROW;   // bf16 elements per LDS stage
    static_assert(!SPLIT_TAPS || KH == 4, "tap-row split assumes 4 waves = KH");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* lds = reinterpret_cast<__bf16*>(smem);

    const ssr_wgrad_item it = items[blockIdx.x];
    const ssr_wgrad_layer L = layers[it.layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5;
    const int tiles_x = (L.Gw + WG_TW - 1) / WG_TW, tiles_y = (L.Gh + TH - 1) / TH;
    const int upshift = L.up == 2 ? 1 : 0;
    const int LH = L.Hi << upshift, LW = L.Wi << upshift;
    const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(L.x.p);
    const __bf16* __restrict__ dyg = reinterpret_cast<const __bf16*>(L.dy.p);
    const bool do_bias = L.db != nullptr && it.ci0 == 0 && (!SPLIT_TAPS || wave == 0);

    // transpose-read source role of this lane: pixel j (0..3) of the 4-pixel group, channel quad
    const int t16 = lane & 15;
    const int src_px = 8 * g + (t16 >> 2);                    // + 4 for the second read of the pair
    const int src_ch = 16 * ((lane >> 4) & 1) + 4 * (t16 & 3);

    f32x16 acc[NTAP], accb;
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (__bf16)1.0f;

    u32x4 rdy[NDY], rx[NXV];
    auto load_tile = [&](int tile) {
        int b = tile;
        const int tx_i = b % tiles_x; b /= tiles_x;
        const int ty_i = b % tiles_y;
        const int n = b / tiles_y;
        const int gy0 = ty_i * TH, gx0 = tx_i * WG_TW;
#pragma unroll
        for (int q = 0; q < NDY; ++q) {
            const int v = tid + q * 256;
            const int pix = v >> 2, part = v & 3;
            const int gy = gy0 + (pix >> 4), gx = gx0 + (pix & 15);
            const int c = it.co0 + part * 8;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (gy < L.Gh && gx < L.Gw && c < L.Cout)
                val = *reinterpret_cast<const u32x4*>(dyg + ((size_t)(n * L.Gh + gy) * L.Gw + gx) * L.dy.cs +
                                                      L.dy.coff + c);
            rdy[q] = val;
        }
#pragma unroll
        for (int q = 0; q < NXV; ++q) {
            const int v = tid + q * 256;
            const int pix = v >> 2, part = v & 3;
            const int py = pix / PW, px = pix - py * PW;
            const int ly = gy0 * S + py - L.pad_y, lxx = gx0 * S + px - L.pad_x;
            const int c = it.ci0 + part * 8;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (v < PH * PW * 4 && ly >= 0 && ly < LH && lxx >= 0 && lxx < LW && c < L.Cin)
                val = *reinterpret_cast<const u32x4*>(
                    xg + ((size_t)(n * L.Hi + (ly >> upshift)) * L.Wi + (lxx >> upshift)) * L.x.cs + L.x.coff + c);
            rx[q] = val;
        }
    };
    auto store_tile = [&](int stage) {
        __bf16* ldy = lds + stage * STAGE;
        __bf16* lx = ldy + TH * WG_TW * ROW;
#pragma unroll
        for (int q = 0; q < NDY; ++q) *reinterpret_cast<u32x4*>(ldy + (tid + q * 256) * 8) = rdy[q];
#pragma unroll
        for (int q = 0; q < NXV; ++q) {
            const int v = tid + q * 256;
            if (v < PH * PW * 4) *reinterpret_cast<u32x4*>(lx + v * 8) = rx[q];
        }
    };

    int tile = it.tile_begin;
    if (tile < it.tile_end) {
        load_tile(tile);
        store_tile(0);
    }
    __syncthreads();
    int stage = 0;
    for (; tile < it.tile_end; ++tile) {
        const bool has_next = tile + 1 < it.tile_end;
        if (has_next) load_tile(tile + 1);
        const __bf16* ldy = lds + stage * STAGE;
        const __bf16* lx = ldy + TH * WG_TW * ROW;
        constexpr int NROW = SPLIT_TAPS ? TH : TH / 4;        // tile rows (16-pixel k-steps) per wave
#pragma unroll
        for (int s = 0; s < NROW; ++s) {
            const int ty = SPLIT_TAPS ? s : NROW * wave + s;
            const __bf16* ap = ldy + (ty * WG_TW + src_px) * ROW + src_ch;
            const bf16x8 a = tr_pair(ap, ap + 4 * ROW);
            if (do_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, accb, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int ky = SPLIT_TAPS ? wave : t / KW, kx = SPLIT_TAPS ? t : t % KW;
                const __bf16* bp = lx + ((ty * S + ky) * PW + src_px * S + kx) * ROW + src_ch;
                const bf16x8 b = tr_pair(bp, bp + 4 * S * ROW);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
            }
        }
        if (has_next) store_tile(stage ^ 1);
        __syncthreads();
        stage ^= 1;
    }
    // all waves are past the last barrier: stage buffers are free -> reduction scratch
    wgrad_writeout<KH, KW, SPLIT_TAPS, NTAP>(acc, accb, L, it, reinterpret_cast<float*>(smem), do_bias);
}

template <int KH, int KW, int S, bool SPLIT>
int launch(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, hipStream_t st) {
    constexpr int TH = wgrad_bf16_th(KH);
    constexpr int PH = (TH - 1) * S + KH, PW = (WG_TW - 1) * S + KW;
    constexpr size_t stage_bytes = (size_t)(TH * WG_TW + PH * PW) * 32 * 2;
    constexpr size_t lds = 2 * stage_bytes > 4 * 16 * 64 * 4 ? 2 * stage_bytes : 4 * 16 * 64 * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = wgrad_bf16_kernel<KH, KW, S, SPLIT>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_items), dim3(256), lds, st, layers, items);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

}  // namespace

int ssr_wgrad_bf16_dispatch(const ssr_wgrad_layer* layers, const ssr_wgrad_item* items, int n_items, int KH, int KW,
                            int S, hipStream_t st) {
    if (KH == 3 && KW == 3 && S == 1) return launch<3, 3, 1, false>(layers, items, n_items, st);
    if (KH == 4 && KW == 4 && S == 2) return launch<4, 4, 2, true>(layers, items, n_items, st);
    return SSR_EUNSUP;
}
