// Fused ResidualDenseBlock forward (bf16, num_feat = 64, num_grow_ch = 32): ONE launch for
// /root/reference/ssr/archs/rrdbnet_arch.py:37-44
//     x1 = lrelu(conv1(x)); x2 = lrelu(conv2(cat(x,x1))); ... x5 = conv5(cat(x..x4)); return x5*0.2 + x
// (and the RRDB tail `out*0.2 + x`, :68, when this is the third block).
//
// Why: at the measured batch (16 x 32x32 pixels) every per-conv launch is latency bound (8..12 us for
// 0.6..1.5 GFLOP: launch + load latency + epilogue), and the five convs of a block are strictly
// sequential.  Here a workgroup owns an 8x8 output tile of one image and keeps the WHOLE dense block
// resident in LDS: the 18x18 input halo region (5-pixel halo) and the shrinking 16x16 / 14x14 / 12x12 /
// 10x10 regions of x1..x4 never leave the CU between the convs; only the weights are streamed
// (LDS-DMA, double buffered 18/36 KB slabs).  Cost: halo recompute (1.76x the MFMAs of the block);
// gain: 5 launches -> 1, no re-load of activations, one latency chain per block instead of five.
// x1..x4 are still written to the dense buffer (8x8 core only): training needs them for dgrad masks and wgrad.
//
// LDS map (bytes): X0 2 x 336 rows | X1 256 | X2 208 | X3 144 | X4 112 rows of 64 B (32 bf16, source-side
// XOR swizzle: physical 16-B part = logical ^ ((row >> 2) & 3)) = 89,088; weight ring 4 x 18,432 = 73,728.
// Waves: 4, each owns whole 32-pixel M-tiles (tiles w, w+4); conv5: wave = (M-tile, 32-channel N-tile).
#include "common.h"

#ifdef SSR_PROBE   // tools/rdb_probe.hip
#define PROBE(k)                                                                           \
    do {                                                                                   \
        if (threadIdx.x == 0) g_probe[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PROBE(k)
#endif

namespace {

constexpr int RB_AROW = 80;                        // bytes per activation row in LDS: 32 bf16 + 16 B pad
constexpr int RB_WROW = 64;                        // bytes per weight row (dense: LDS-DMA writes lane-linearly)
constexpr int RB_X0 = 0;                           // x: 2 planes x 324 rows
constexpr int RB_X0P = 324 * RB_AROW;
constexpr int RB_X1 = RB_X0 + 2 * RB_X0P;          // x1: 256 rows
constexpr int RB_X2 = RB_X1 + 256 * RB_AROW;       // x2: 196 rows (+1 dummy row each: padding lanes write there)
constexpr int RB_X3 = RB_X2 + 197 * RB_AROW;       // x3: 144 rows
constexpr int RB_X4 = RB_X3 + 145 * RB_AROW;       // x4: 100 rows
constexpr int RB_RING = RB_X4 + 101 * RB_AROW;
constexpr int RB_SLAB = 288 * RB_WROW;             // 9 taps x 32 co rows of 32 ci
constexpr int RB_NSTAGE = 3;
constexpr int RB_BIAS = RB_RING + RB_NSTAGE * RB_SLAB;   // [5 convs][64] fp32 bias table
constexpr int RB_LDS = RB_BIAS + (4 * 32 + 64) * 4;   // conv k < 5 at k*32, conv5 at 128
static_assert(RB_LDS <= 160 * 1024, "LDS budget");
static_assert(RB_NSTAGE * RB_SLAB >= 4 * 2048 + 2 * 2 * 4096, "ring doubles as reduce scratch + output slabs");

__device__ __forceinline__ constexpr int rb_slice_base(int s) {   // slice 0 (plane 0), 1..4
    return s == 0 ? RB_X0 : s == 1 ? RB_X1 : s == 2 ? RB_X2 : s == 3 ? RB_X3 : RB_X4;
}

// Weight slab q of the block's schedule -> ring stage: 18 LDS-DMA wave-instructions (16 rows x 64 B each),
// all issued by the PRODUCER wave (wave 4) so that the four MFMA waves never spend issue slots on loads.
// Schedule: conv1 c0,c1 | conv2 c0..2 | conv3 c0..3 | conv4 c0..4 | conv5 (c0,n0),(c0,n1) ... (c5,n1) = 26 slabs.
constexpr int RB_NSLAB = 26;
__device__ __forceinline__ void rb_issue_slab(const ssr_rdb_desc& d, int q, char* stage, int lane) {
    int k, c, nt = 0;
    if (q < 2) { k = 0; c = q; }
    else if (q < 5) { k = 1; c = q - 2; }
    else if (q < 9) { k = 2; c = q - 5; }
    else if (q < 14) { k = 3; c = q - 9; }
    else { k = 4; c = (q - 14) >> 1; nt = (q - 14) & 1; }
    const int cp = k == 4 ? 64 : 32;
    const __bf16* base = reinterpret_cast<const __bf16*>(d.w[k]) + (size_t)(c * 9 * cp + nt * 32) * 32;
    const int lrow = lane >> 2, pp = lane & 3;
#pragma unroll
    for (int j = 0; j < 18; ++j) {
        const int row = 16 * j + lrow;                      // = tap*32 + co
        const int tap = row >> 5, co = row & 31, lp = pp ^ ((row >> 2) & 3);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(base + (tap * cp + co) * 32 + lp * 8),
            (__attribute__((address_space(3))) void*)(stage + j * 1024), 16, 0, 0);
    }
}

struct RbTile {   // per-lane coordinates of one 32-pixel M-tile of a conv's output region
    int oy, ox;   // position inside the region (clamped to a valid pixel for padding lanes)
};

template <int K>   // conv index 1..5; region side R = 18 - 2K, P = R*R pixels
__device__ __forceinline__ RbTile rb_tile(int mt, int i) {
    constexpr int R = 18 - 2 * K, P = R * R;
    const int p = 32 * mt + i;
    const int pc = p < P ? p : P - 1;
    RbTile t;
    t.oy = pc / R;
    t.ox = pc - t.oy * R;
    return t;
}

// Contraction of one weight slab (32 input channels of slice S) into NMT accumulators.
// KK0/NKK select the 16-channel k-substeps this wave handles (all: 0,2; k-split half: kh,1).
// Every LDS address is one per-lane base plus a compile-time immediate; reads of tap n+1 are
// scheduled in front of the MFMAs of tap n (one wave per SIMD: overlap must come from inside the wave).
template <int K, int S, int NMT, int NKK>
__device__ __forceinline__ void rb_contract(f32x16 (&acc)[NMT], const RbTile (&tl)[NMT], const char* smem,
                                            const char* slab, int plane, int i, int g, int kk0) {
    constexpr int RS = 18 - 2 * S;                 // side of slice S's region
    constexpr int DELTA = K - S - 1;               // offset of conv K's output region inside slice S's region
    const char* sb = smem + rb_slice_base(S) + plane * RB_X0P + (kk0 * 2 + g) * 16;
    const char* ab[NMT];
#pragma unroll
#ifdef RB_X_NOCONF
    for (int m = 0; m < NMT; ++m) ab[m] = sb + (i & 15) * RB_AROW + (tl[m].oy & 1) * 16 * RB_AROW;
#else
    for (int m = 0; m < NMT; ++m) ab[m] = sb + (tl[m].oy * RS + tl[m].ox) * RB_AROW;
#endif
    const int bsw = (i >> 2) & 3;
    const char* bb0 = slab + i * RB_WROW + ((((kk0 * 2 + g) ^ bsw)) << 4);
    const char* bb1 = slab + i * RB_WROW + ((((kk0 * 2 + g) ^ bsw) ^ 2) << 4);   // second k-substep (NKK == 2)
    // Software pipeline, written out: there is one wave per SIMD, so LDS latency (>= 128 cycles) can only be
    // hidden inside the wave.  Operand reads run RB_PF k-steps ahead of the MFMAs that consume them; the
    // sched_barrier fences pin that order (left alone, hipcc sinks every ds_read next to its MFMA and emits
    // `ds_read; s_waitcnt lgkmcnt(0); v_mfma` chains that run at ~40 % of the MFMA rate).
    constexpr int NSTEP = 9 * NKK, RB_PF = NMT == 1 ? 4 : 3;
    u32x4 bq[NSTEP], aq[NSTEP][NMT];
    auto issue = [&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        constexpr int tap = n / NKK, kk = n % NKK, ky = tap / 3, kx = tap % 3;
        bq[n] = *reinterpret_cast<const u32x4*>((kk ? bb1 : bb0) + tap * 32 * RB_WROW);
#pragma unroll
        for (int m = 0; m < NMT; ++m)
            aq[n][m] = *reinterpret_cast<const u32x4*>(ab[m] + ((DELTA + ky) * RS + DELTA + kx) * RB_AROW + kk * 32);
    };
    static_for<0, RB_PF>([&](auto n_c) { issue(n_c); });
    static_for<0, NSTEP>([&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + RB_PF < NSTEP) issue(std::integral_constant<int, n + RB_PF>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NMT; ++m) mma16<__bf16>(acc[m], bq[n], aq[n][m]);   // A = weights (rows = co), B = pixels
    });
}

// C fragment with the operands swapped (weights = A): lane l owns ONE pixel (column l & 31 of the M-tile) and
// 16 output channels co = 8*(r>>2) + 4*(l>>5) + (r&3): four runs of 4 consecutive channels.
// Epilogue of conv K < 5: x_K = lrelu(acc + bias), zero outside the image (one test per lane), packed to bf16 and
// written to LDS slice K with four 8-byte stores.
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
struct RbPix {            // this lane's pixel of one M-tile of stage K
    int p;                // linear index in the stage's region (may be >= P for padding lanes)
    bool inside;          // inside the image (and p < P)
    size_t gpix;          // clamped global pixel index (n*H + iy)*W + ix
};
template <int K>
__device__ __forceinline__ RbPix rb_pix(int mt, int i, int n, int ty0, int tx0, int H, int W) {
    constexpr int R = 18 - 2 * K, P = R * R, HK = 5 - K;
    RbPix t;
    t.p = 32 * mt + i;
    const int oy = t.p / R, ox = t.p - oy * R;
    const int iy = ty0 - HK + oy, ix = tx0 - HK + ox;
    t.inside = t.p < P && iy >= 0 && iy < H && ix >= 0 && ix < W;
    t.gpix = (size_t)(n * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1);
    return t;
}
// forward: x_K = lrelu(acc + bias); backward: dpre = acc * lrelu'(x_k) (mk = the saved forward activation);
// zero outside the image (zero padding of the NEXT stage's input); four 8-byte LDS stores per lane
template <int K, bool BWD>
__device__ __forceinline__ void rb_store_slice(const f32x16& acc, const RbPix& px, const float* bias_lds,
                                               const bf16x4v (&mk)[4], char* smem, int g) {
    constexpr int R = 18 - 2 * K, P = R * R;
    char* row = smem + rb_slice_base(K) + (px.p < P ? px.p : P) * RB_AROW + g * 8;   // P = dummy row
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        bf16x4v o;
        if (BWD) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = (__bf16)(px.inside ? acc[4 * q4 + e] * lrelu_grad_from_out((float)mk[q4][e]) : 0.f);
        } else {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + 8 * q4 + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)(px.inside ? lrelu(acc[4 * q4 + e] + bq[e]) : 0.f);
        }
        *reinterpret_cast<bf16x4v*>(row + 16 * q4) = o;
    }
}
// backward: the 16 channels of x_k (k = 5 - K) of this lane's pixel, from the saved forward buffer
template <int K>
__device__ __forceinline__ void rb_load_mask(const ssr_rdb_desc& d, const RbPix& px, int g, bf16x4v (&mk)[4]) {
    const __bf16* mp = reinterpret_cast<const __bf16*>(d.mask.p) + px.gpix * d.mask.cs + d.mask.coff + 64 +
                       32 * (5 - K - 1) + 4 * g;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) mk[q4] = *reinterpret_cast<const bf16x4v*>(mp + 8 * q4);
}

// cooperative write of the 8x8 core of LDS slice K (32 channels) to the dense buffer: one 16-B vector per thread
template <int K, bool BWD>
__device__ __forceinline__ void rb_flush_core(const ssr_rdb_desc& d, const char* smem, int n, int ty0, int tx0,
                                              int tid) {
    constexpr int R = 18 - 2 * K, HK = 5 - K;
    const int q = tid >> 2, part = tid & 3;
    const int cy = q >> 3, cx = q & 7;
    const int p = (cy + HK) * R + cx + HK;
    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + rb_slice_base(K) + p * RB_AROW + part * 16);
    const int iy = ty0 + cy, ix = tx0 + cx;
    if (iy < d.H && ix < d.W) {
        constexpr int KD = BWD ? 5 - K : K;      // backward stage K produces dpre_{5-K}
        __bf16* dst = reinterpret_cast<__bf16*>(d.slices.p) + ((size_t)(n * d.H + iy) * d.W + ix) * d.slices.cs +
                      d.slices.coff + 64 + 32 * (KD - 1) + part * 8;
        *reinterpret_cast<u32x4*>(dst) = v;
    }
}

template <bool BWD>
__global__ __launch_bounds__(320) void rdb_kernel(const ssr_rdb_desc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int tiles_x = (d.W + 7) / 8, tiles_y = (d.H + 7) / 8;
    int b = blockIdx.x;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int ty0 = ty_i * 8, tx0 = tx_i * 8;
    const int H = d.H, W = d.W;
    char* ring = smem + RB_RING;
    PROBE(0);
    const bool producer = wave == 4;
    float* bias_lds = reinterpret_cast<float*>(smem + RB_BIAS);
    {
        const int k = tid < 128 ? tid >> 5 : 4, c = tid < 128 ? tid & 31 : tid - 128;   // 4 x 32 + 64 entries
        if (tid < 192) bias_lds[tid] = d.bias[k] ? d.bias[k][c] : 0.f;
    }
    if (producer) rb_issue_slab(d, 0, ring, lane);
    // ---- 64-channel input halo region (x / d_out): 18x18 pixels -> X0 (2 planes of 32 channels, padded rows),
    //      staged through registers so that the rows can be padded (conflict-free, immediate offsets) ----
    {
        const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(d.in.p);
        u32x4 rx[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int v = tid + q * 320;                     // (plane, pixel, part): 2 x 324 x 4 = 2592 vectors
            const int plane = v / 1296, r2 = v - plane * 1296;
            const int pix = r2 >> 2, part = r2 & 3;
            const int py = pix / 18, px = pix - py * 18;
            const int iy = ty0 - 5 + py, ix = tx0 - 5 + px;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (v < 2592 && iy >= 0 && iy < H && ix >= 0 && ix < W)
                val = *reinterpret_cast<const u32x4*>(xg + ((size_t)(n * H + iy) * W + ix) * d.in.cs + d.in.coff +
                                                      plane * 32 + part * 8);
            rx[q] = val;
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int v = tid + q * 320;
            const int plane = v / 1296, r2 = v - plane * 1296;
            if (v < 2592) *reinterpret_cast<u32x4*>(smem + RB_X0 + plane * RB_X0P + (r2 >> 2) * RB_AROW + (r2 & 3) * 16) = rx[q];
        }
    }
    PROBE(1);
    if (producer) {
        // step q: the barrier publishes slab q (hipcc drains this wave's vmcnt(0) in front of it) and frees
        // stage (q+1) % 3, which was consumed in step q-2; then slab q+1 is put in flight under the MFMAs of step q
        for (int q = 0; q < RB_NSLAB; ++q) {
            __syncthreads();
#ifndef RB_X_NODMA
            if (q + 1 < RB_NSLAB) rb_issue_slab(d, q + 1, ring + ((q + 1) % RB_NSTAGE) * RB_SLAB, lane);
#endif
        }
        __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads();   // stage-5 tail barriers
        return;
    }
    // Consumer barrier: LDS writes complete (lgkmcnt) + s_barrier, but NO vmcnt drain — the MFMA waves have no
    // LDS-DMA of their own; their global loads (masks, residual) and core stores stay in flight across steps.
#define RB_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    int stage = 0;   // ring stage holding the slab that is consumed next
#define RB_STEP()                                                                 \
    RB_BAR(); /* slab in `stage` has landed (producer) and all slice writes are visible */ \
    const char* slab = ring + stage * RB_SLAB;                                    \
    stage = stage + 1 == RB_NSTAGE ? 0 : stage + 1;
    // chunk c of stage K reads LDS slice S (plane P of the 64-channel input for S = 0):
    //   forward  : x p0, x p1, x1, ..., x_{K-1}          (rrdbnet_arch.py:39-42 cat order)
    //   backward : dpre_{6-K} .. dpre_4 (= slices K-1 .. 1), d_out p0, d_out p1   (ParamStore.add_rdb_gather order)
#define RB_C(K, S, PL, NMT, ACC) { RB_STEP(); rb_contract<K, S, NMT, 2>(ACC, tl, smem, slab, PL, i, g, 0); }
#define RB_CF(K, S, PL, NMT, ACC) { RB_STEP(); rb_flush_core<K - 1, BWD>(d, smem, n, ty0, tx0, tid); \
                                    rb_contract<K, S, NMT, 2>(ACC, tl, smem, slab, PL, i, g, 0); }
    bf16x4v mk0[4], mk1[4];
    // ================= stage 1: region 16x16 (8 M-tiles: wave w owns tiles w, w+4), K = the 64-ch input =========
    {
        f32x16 acc[2];
        RbTile tl[2] = {rb_tile<1>(wave, i), rb_tile<1>(wave + 4, i)};
        const RbPix p0 = rb_pix<1>(wave, i, n, ty0, tx0, H, W), p1 = rb_pix<1>(wave + 4, i, n, ty0, tx0, H, W);
        if (BWD) { rb_load_mask<1>(d, p0, g, mk0); rb_load_mask<1>(d, p1, g, mk1); }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#ifdef SSR_PROBE
        PROBE(8);
        { RB_STEP(); PROBE(9); rb_contract<1, 0, 2, 2>(acc, tl, smem, slab, 0, i, g, 0); }
        PROBE(10);
        { RB_STEP(); PROBE(11); rb_contract<1, 0, 2, 2>(acc, tl, smem, slab, 1, i, g, 0); }
        PROBE(12);
#else
        RB_C(1, 0, 0, 2, acc) RB_C(1, 0, 1, 2, acc)
#endif
        rb_store_slice<1, BWD>(acc[0], p0, bias_lds, mk0, smem, g);
        rb_store_slice<1, BWD>(acc[1], p1, bias_lds, mk1, smem, g);
        PROBE(13);
    }
    PROBE(2);
    // ================= stage 2: region 14x14 (7 M-tiles) =====================================================
    {
        f32x16 acc[2];
        const int mt1 = wave + 4 < 7 ? wave + 4 : 6;
        RbTile tl[2] = {rb_tile<2>(wave, i), rb_tile<2>(mt1, i)};
        const RbPix p0 = rb_pix<2>(wave, i, n, ty0, tx0, H, W), p1 = rb_pix<2>(mt1, i, n, ty0, tx0, H, W);
        if (BWD) { rb_load_mask<2>(d, p0, g, mk0); rb_load_mask<2>(d, p1, g, mk1); }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        if (BWD) { RB_CF(2, 1, 0, 2, acc) RB_C(2, 0, 0, 2, acc) RB_C(2, 0, 1, 2, acc) }
        else     { RB_CF(2, 0, 0, 2, acc) RB_C(2, 0, 1, 2, acc) RB_C(2, 1, 0, 2, acc) }
        rb_store_slice<2, BWD>(acc[0], p0, bias_lds + 32, mk0, smem, g);
        if (wave + 4 < 7) rb_store_slice<2, BWD>(acc[1], p1, bias_lds + 32, mk1, smem, g);
    }
    PROBE(3);
    // ================= stage 3: region 12x12 (5 M-tiles: wave 0 owns tiles 0 and 4) ===========================
    {
        f32x16 acc[2];
        RbTile tl[2] = {rb_tile<3>(wave, i), rb_tile<3>(4, i)};
        const RbPix p0 = rb_pix<3>(wave, i, n, ty0, tx0, H, W), p1 = rb_pix<3>(4, i, n, ty0, tx0, H, W);
        if (BWD) { rb_load_mask<3>(d, p0, g, mk0); rb_load_mask<3>(d, p1, g, mk1); }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        if (BWD) { RB_CF(3, 2, 0, 2, acc) RB_C(3, 1, 0, 2, acc) RB_C(3, 0, 0, 2, acc) RB_C(3, 0, 1, 2, acc) }
        else     { RB_CF(3, 0, 0, 2, acc) RB_C(3, 0, 1, 2, acc) RB_C(3, 1, 0, 2, acc) RB_C(3, 2, 0, 2, acc) }
        rb_store_slice<3, BWD>(acc[0], p0, bias_lds + 64, mk0, smem, g);
        if (wave == 0) rb_store_slice<3, BWD>(acc[1], p1, bias_lds + 64, mk1, smem, g);
    }
    PROBE(4);
    // ================= stage 4: region 10x10 (4 M-tiles, one per wave) ========================================
    {
        f32x16 acc[1];
        RbTile tl[1] = {rb_tile<4>(wave, i)};
        const RbPix p0 = rb_pix<4>(wave, i, n, ty0, tx0, H, W);
        if (BWD) rb_load_mask<4>(d, p0, g, mk0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
        if (BWD) { RB_CF(4, 3, 0, 1, acc) RB_C(4, 2, 0, 1, acc) RB_C(4, 1, 0, 1, acc) RB_C(4, 0, 0, 1, acc) RB_C(4, 0, 1, 1, acc) }
        else     { RB_CF(4, 0, 0, 1, acc) RB_C(4, 0, 1, 1, acc) RB_C(4, 1, 0, 1, acc) RB_C(4, 2, 0, 1, acc) RB_C(4, 3, 0, 1, acc) }
        rb_store_slice<4, BWD>(acc[0], p0, bias_lds + 96, mk0, smem, g);
    }
    PROBE(5);
    // ================= stage 5: 8x8 core (2 M-tiles) x 64 channels: wave = (M-tile mt, k-substep kh); the 12 slabs
    //                   (6 chunks x 2 N-tiles) stream through the same ring; halves are combined through LDS ======
    {
        const int mt = wave & 1, kh = wave >> 1;
        f32x16 acc0[1], acc1[1];                               // N-tile 0 / 1
        RbTile tl[1] = {rb_tile<5>(mt, i)};
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[0][r] = 0.f; acc1[0][r] = 0.f; }
        // residual r2 (x_rrdb / d out_rrdb): this lane's pixel, 16 channels = four 8-byte loads, issued now and
        // consumed in the epilogue
        const int nt = kh;                                     // k-half kh finishes N-tile kh
        const int q = 32 * mt + i, cy = q >> 3, cx = q & 7;    // this lane's core pixel
        const int iy = ty0 + cy, ix = tx0 + cx;
        const __bf16* __restrict__ r2p = reinterpret_cast<const __bf16*>(d.r2.p);
        bf16x4v r2v[4];
        if (r2p) {
            const __bf16* rp = r2p + ((size_t)(n * H + min(iy, H - 1)) * W + min(ix, W - 1)) * d.r2.cs + d.r2.coff +
                               nt * 32 + 4 * g;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) r2v[q4] = *reinterpret_cast<const bf16x4v*>(rp + 8 * q4);
        }
#define RB_C5(S, PLANE, FLUSH)                                                                                  \
        { RB_STEP(); if (FLUSH) rb_flush_core<4, BWD>(d, smem, n, ty0, tx0, tid);                                \
          rb_contract<5, S, 1, 1>(acc0, tl, smem, slab, PLANE, i, g, kh); }                                      \
        { RB_STEP(); rb_contract<5, S, 1, 1>(acc1, tl, smem, slab, PLANE, i, g, kh); }
        if (BWD) { RB_C5(4, 0, 1) RB_C5(3, 0, 0) RB_C5(2, 0, 0) RB_C5(1, 0, 0) RB_C5(0, 0, 0) RB_C5(0, 1, 0) }
        else     { RB_C5(0, 0, 1) RB_C5(0, 1, 0) RB_C5(1, 0, 0) RB_C5(2, 0, 0) RB_C5(3, 0, 0) RB_C5(4, 0, 0) }
#undef RB_C5
        PROBE(6);
        RB_BAR();   // ring is free: reduce scratch [2 mt][16][64] floats + output transpose slabs
        // k-half kh = 0 finishes N-tile 0, kh = 1 finishes N-tile 1 (each needs the other's partial of its tile)
        float* mine = reinterpret_cast<float*>(ring) + (mt * 16) * 64 + lane;
        if (kh == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r * 64] = acc1[0][r];
        }
        RB_BAR();
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[0][r] += mine[r * 64];
        }
        RB_BAR();
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r * 64] = acc0[0][r];
        }
        RB_BAR();
        if (kh == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[0][r] += mine[r * 64];
        }
        // forward : out = alpha5*(conv5 + b5) + beta1*x + beta2*x_rrdb               (rrdbnet_arch.py:44, :68)
        // backward: d x = gathered dgrad (alpha5 = 1, conv5's scale is folded into the packed weights)
        //                 + beta1*d_out + beta2*d_out_rrdb
        {
            const int p0 = (cy + 5) * 18 + cx + 5;
            const char* xrow = smem + RB_X0 + nt * RB_X0P + p0 * RB_AROW + g * 8;
            const float* b5 = bias_lds + 128 + nt * 32;
            __bf16* slabw = reinterpret_cast<__bf16*>(ring + 2 * 16 * 64 * 4 + wave * 2048);   // [32 px][32 co] bf16
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(b5 + 8 * q4 + 4 * g);
                const bf16x4v xv = *reinterpret_cast<const bf16x4v*>(xrow + 16 * q4);
                bf16x4v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = kh == 0 ? acc0[0][4 * q4 + e] : acc1[0][4 * q4 + e];
                    float v = d.alpha5 * (a + bq[e]) + d.beta1 * (float)xv[e];
                    if (r2p) v += d.beta2 * (float)r2v[q4][e];
                    o[e] = (__bf16)v;
                }
                *reinterpret_cast<bf16x4v*>(slabw + i * 32 + 8 * q4 + 4 * g) = o;
            }
            // the wave's 32 px x 64 B tile -> 128 16-byte vectors: 2 per lane
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int v = h * 64 + lane;
                const int pix = v >> 2, part = v & 3;
                const int qq = 32 * mt + pix, oy = ty0 + (qq >> 3), ox = tx0 + (qq & 7);
                const u32x4 val = *reinterpret_cast<const u32x4*>(slabw + pix * 32 + part * 8);
                if (oy < H && ox < W) {
                    __bf16* dst = reinterpret_cast<__bf16*>(d.out.p) + ((size_t)(n * H + oy) * W + ox) * d.out.cs +
                                  d.out.coff + nt * 32 + part * 8;
                    *reinterpret_cast<u32x4*>(dst) = val;
                }
            }
        }
    }
    PROBE(7);
#undef RB_STEP
#undef RB_C
#undef RB_CF
#undef RB_BAR
}

}  // namespace

static int rdb_launch(const ssr_rdb_desc* dp, void* stream, bool bwd) {
    if (!dp) return SSR_EINVAL;
    const ssr_rdb_desc& d = *dp;
    if (d.dtype != SSR_BF16) return SSR_EUNSUP;
    if (!d.in.p || !d.slices.p || !d.out.p || d.N <= 0 || d.H <= 0 || d.W <= 0) return SSR_EINVAL;
    if ((d.in.cs % 8) || (d.in.coff % 8) || (d.slices.cs % 8) || (d.slices.coff % 8) || (d.out.cs % 8) || (d.out.coff % 8))
        return SSR_EINVAL;
    if (bwd && (!d.mask.p || (d.mask.cs % 4) || (d.mask.coff % 4))) return SSR_EINVAL;
    if (d.r2.p && ((d.r2.cs % 4) || (d.r2.coff % 4))) return SSR_EINVAL;
    for (int k = 0; k < 5; ++k)
        if (!d.w[k]) return SSR_EINVAL;
    static bool attr_done[2] = {false, false};
    const void* kern = bwd ? reinterpret_cast<const void*>(rdb_kernel<true>) : reinterpret_cast<const void*>(rdb_kernel<false>);
    if (!attr_done[bwd]) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, RB_LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[bwd] = true;
    }
    const int tiles = d.N * ((d.H + 7) / 8) * ((d.W + 7) / 8);
    if (bwd) hipLaunchKernelGGL(rdb_kernel<true>, dim3(tiles), dim3(320), RB_LDS, reinterpret_cast<hipStream_t>(stream), d);
    else hipLaunchKernelGGL(rdb_kernel<false>, dim3(tiles), dim3(320), RB_LDS, reinterpret_cast<hipStream_t>(stream), d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_rdb_forward(const ssr_rdb_desc* dp, void* stream) { return rdb_launch(dp, stream, false); }
extern "C" int ssr_rdb_backward(const ssr_rdb_desc* dp, void* stream) { return rdb_launch(dp, stream, true); }
