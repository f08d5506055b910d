// Fused ResidualDenseBlock forward / backward (bf16, num_feat = 64, num_grow_ch = 32): ONE launch for
// /root/reference/ssr/archs/rrdbnet_arch.py:37-44
//     x1 = lrelu(conv1(x)); x2 = lrelu(conv2(cat(x,x1))); ... x5 = conv5(cat(x..x4)); return x5*0.2 + x
// (and the RRDB tail `out*0.2 + x`, :68, when this is the third block); rdb_kernel<true> is the gather-form
// backward of the same block (ParamStore.add_rdb_gather): stage K produces dpre_{5-K} from [newer dpre | d_out].
//
// Why: at the measured batch (16 x 32x32 pixels) every per-conv launch is latency bound (8..12 us for
// 0.6..1.5 GFLOP: launch + load latency + epilogue), and the five convs of a block are strictly
// sequential.  Here a workgroup owns an 8x8 output tile of one image and keeps the WHOLE dense block
// resident in LDS: the 18x18 input halo region (5-pixel halo) and the shrinking 16x16 / 14x14 / 12x12 /
// 10x10 regions of x1..x4 never leave the CU between the convs; only the weights are streamed.
// Cost: halo recompute (1.76x the MFMAs of the block); gain: 5 launches -> 1, no re-load of activations,
// one latency chain per block instead of five.  x1..x4 are still written to the dense buffer (8x8 core
// only): training needs them for dgrad masks and wgrad.
//
// What bounds it (tools/rdb_probe.hip, tools/mfma_probe.hip): LDS bandwidth.  One v_mfma_f32_32x32x16_bf16 takes
// 32 cycles and eats two 1-KiB operands; four SIMDs fetching both from LDS need exactly the 256 B/clk the LDS has.
// Every wave therefore shares the weight fragment between two pixel tiles where it can (1.5 reads per MFMA),
// and every ds_read_b128 must be conflict-free:
//   * activation rows are 80 B (32 bf16 + 16 B pad), so row r starts at bank 20 r mod 64: the 16 lanes the
//     hardware serves together hit disjoint banks iff their rows are distinct mod 16;
//   * ALL slices use the same row pitch (18 pixels) and lane i of an M-tile owns a pixel with
//     18*oy + ox == i (mod 16) (table rb_map, built at compile time).  A tap shift or a change of slice adds the
//     same constant to every lane's row, so every operand read of every stage is conflict-free and its address
//     is one per-lane base plus an immediate.
//
// LDS map: X0 2 planes x 324 rows | X1 286 | X2 248 | X3 210 | X4 172 rows (pitch 18, region side 18 - 2S) | dummy row
// (padding lanes write there) = 125,200 B; weight ring 2 x 18,432 B (dense 64-B rows, 16-B parts XOR-swizzled by
// (row >> 2) & 3); bias table; control words.
//
// Waves: 4 MFMA waves (whole 32-pixel M-tiles; conv5: wave = (M-tile, 32-channel N-tile)) + 6 PRODUCER waves that
// stream the 26 weight slabs (479 KB per block): 16-byte global loads into registers, two slabs ahead, then
// ds_write_b128 into the ring.  Why six: tools/l2_probe.hip — a wave streaming L2-resident data gets ~6.4 B/clk
// however many loads it keeps in flight, and waves add up (2: 14, 4: 28, 6: 35, 8: 43 B/clk/CU); LDS-DMA is capped
// at ~18 B/clk per CU for any number of waves.  Two producers delivered a slab per ~1000 cycles, slower than the
// MFMAs consume them (576..1152 cycles).
// After the prologue there is NO s_barrier: ring hand-over and slice completion are LDS flags
//   ready[2]    (producers -> consumers)   per ring stage: producer parts stored so far (6 per slab)
//   done_w[4]   (consumer w -> producers)  number of slabs wave w is finished with
//   slice_cnt[] (consumers <-> consumers)  waves that have stored their part of slice K / arrived at the final sync
#include "common.h"
#include <stdlib.h>

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
constexpr int RB_PITCH = 18;                       // rows per image row, every slice
constexpr int RB_WROW = 64;                        // bytes per weight row
constexpr int rb_rows(int s) { return s == 0 ? 324 : (17 - 2 * s) * RB_PITCH + 18 - 2 * s; }
constexpr int RB_X0 = 0;                           // x: 2 planes x 324 rows
constexpr int RB_X0P = 324 * RB_AROW;
constexpr int RB_X1 = RB_X0 + 2 * RB_X0P;
constexpr int RB_X2 = RB_X1 + rb_rows(1) * RB_AROW;
constexpr int RB_X3 = RB_X2 + rb_rows(2) * RB_AROW;
constexpr int RB_X4 = RB_X3 + rb_rows(3) * RB_AROW;
constexpr int RB_DUMMY = RB_X4 + rb_rows(4) * RB_AROW;
constexpr int RB_RING = RB_DUMMY + RB_AROW;
constexpr int RB_SLAB = 288 * RB_WROW;             // 9 taps x 32 co rows of 32 ci
constexpr int RB_NSTAGE = 2;
constexpr int RB_BIAS = RB_RING + RB_NSTAGE * RB_SLAB;   // [4 x 32 + 64] fp32 bias table: conv k < 5 at k*32, conv5 at 128
constexpr int RB_CTL = RB_BIAS + (4 * 32 + 64) * 4;      // 16 control words
constexpr int RB_LDS = RB_CTL + 64;
static_assert(RB_LDS <= 160 * 1024, "LDS budget");
static_assert(RB_NSTAGE * RB_SLAB >= 4 * 2048, "ring doubles as the output transpose slabs");
constexpr int RB_NPROD = 6, RB_NTHREADS = 256 + 64 * RB_NPROD;
// control words
constexpr int CTL_READY = 0;     // [2] one counter per ring stage
constexpr int CTL_DONE = 4;      // [4]
constexpr int CTL_SLICE = 8;     // [1..4] slice K complete, [5] final sync

__device__ __forceinline__ constexpr int rb_slice_base(int s) {   // slice 0 (plane 0), 1..4
    return s == 0 ? RB_X0 : s == 1 ? RB_X1 : s == 2 ? RB_X2 : s == 3 ? RB_X3 : RB_X4;
}

// ---- lane -> pixel map: stage K (region side R = 18 - 2K), M-tile t, lane i & 31 -> oy | ox << 5 | valid << 10 with
//      (18*oy + ox) mod 16 == i mod 16.  Class c = pixels with that residue: for every oy at most one ox = (c - 2 oy)
//      mod 16 < R; the class is enumerated by rising oy and M-tile t takes its entries 2t (lanes 0..15) and 2t+1 (lanes
//      16..31).  Class sizes: R = 16: 16 (8 tiles), 14: 12..13 (7), 12: 8..10 (5), 10: 5..7 (4), 8: 4 (2) — the same
//      tile counts as a linear walk.  Unused slots point at the class's first pixel with valid = 0. ----
struct RbMap { unsigned short e[5][8][32]; };
constexpr int rb_ntiles(int K) { return K == 1 ? 8 : K == 2 ? 7 : K == 3 ? 5 : K == 4 ? 4 : 2; }
constexpr RbMap rb_make_map() {
    RbMap m{};
    for (int K = 1; K <= 5; ++K) {
        const int R = 18 - 2 * K, nt = rb_ntiles(K);
        for (int c = 0; c < 16; ++c) {
            int cnt = 0, first = 0;
            for (int oy = 0; oy < R; ++oy) {
                const int ox = ((c - 2 * oy) % 16 + 16) % 16;
                if (ox >= R) continue;
                const unsigned short v = (unsigned short)(oy | (ox << 5));
                if (cnt == 0) first = v;
                if (cnt < 2 * nt) m.e[K - 1][cnt >> 1][(cnt & 1) * 16 + c] = (unsigned short)(v | (1 << 10));
                ++cnt;
            }
            for (int sl = cnt; sl < 2 * nt; ++sl) m.e[K - 1][sl >> 1][(sl & 1) * 16 + c] = (unsigned short)first;
        }
    }
    return m;
}
constexpr bool rb_map_complete() {   // every region pixel is owned by exactly one (tile, lane)
    constexpr RbMap m = rb_make_map();
    for (int K = 1; K <= 5; ++K) {
        const int R = 18 - 2 * K;
        int n = 0;
        for (int t = 0; t < rb_ntiles(K); ++t)
            for (int i = 0; i < 32; ++i) {
                const int e = m.e[K - 1][t][i];
                if (!(e >> 10)) continue;
                const int oy = e & 31, ox = (e >> 5) & 31;
                if (oy >= R || ox >= R || ((RB_PITCH * oy + ox) & 15) != (i & 15)) return false;
                ++n;
            }
        if (n != R * R) return false;
    }
    return true;
}
static_assert(rb_map_complete(), "lane -> pixel map must cover every region exactly once");
__device__ const RbMap rb_map = rb_make_map();

// Weight slab q of the block's schedule: conv1 j0,j1 | conv2 j0..2 | conv3 j0..3 | conv4 j0..4 | conv5 (j0,h0),(j0,h1)
// ... (j5,h1) = 26 slabs of 18,432 B, where step j of conv K contracts LDS slice (j < 2 ? 0 (plane j) : j - 1): oldest
// first, so that the slice produced by the previous stage is needed last.  Weight chunk in memory: forward = j
// (rrdbnet_arch.py:39-42 cat order); backward = [dpre newest .. oldest | d_out p0 p1] (ParamStore.add_rdb_gather)
// -> j < 2 ? K-1+j : K-j.
//   conv1..4: [chunk of 32 ci][tap][32 co][32 ci]  -> 288 rows of 64 B, 16-B part XOR-swizzled by (row >> 2) & 3
//   conv5   : [chunk of 16 ci][tap][64 co][16 ci]  -> 576 rows of 32 B (16-channel half h of chunk j, ALL 64 output
//             channels), 16-B part XOR-swizzled by (row >> 3) & 1.  Every MFMA wave consumes every slab (9 MFMAs),
//             so the two-stage ring double-buffers; with per-N-tile slabs each wave pair would own one ring stage.
// A slab is 18 pieces of 1 KiB (one 16-B load per lane); producer pw moves pieces pw, pw+6, pw+12.
constexpr int RB_NSLAB = 26;
constexpr int RB_PV = 18 / RB_NPROD;
template <bool BWD>
__device__ __forceinline__ void rb_load_slab(const ssr_rdb_desc& d, int q, int lane, int pw, u32x4 (&r)[RB_PV]) {
#ifdef RB_X_NOWLOAD    // probe: only the first RB_RQ slabs are fetched (wrong results, the weight stream removed)
    if (q >= 8) return;
#endif
    int k, j, h = 0;
    if (q < 2) { k = 0; j = q; }
    else if (q < 5) { k = 1; j = q - 2; }
    else if (q < 9) { k = 2; j = q - 5; }
    else if (q < 14) { k = 3; j = q - 9; }
    else { k = 4; j = (q - 14) >> 1; h = (q - 14) & 1; }
    const int c = BWD ? (j < 2 ? k + j : k + 1 - j) : j;
    if (k == 4) {
        // source slab is contiguous; lane v = piece*64 + lane -> row v >> 1, part v & 1
        const __bf16* base = reinterpret_cast<const __bf16*>(d.w[4]) + (size_t)(2 * c + h) * (9 * 64 * 16);
#pragma unroll
        for (int jj = 0; jj < RB_PV; ++jj) {
            const int v = (jj * RB_NPROD + pw) * 64 + lane;
            const int row = v >> 1, lp = (v & 1) ^ ((row >> 3) & 1);
            r[jj] = *reinterpret_cast<const u32x4*>(base + row * 16 + lp * 8);
        }
        return;
    }
    const __bf16* base = reinterpret_cast<const __bf16*>(d.w[k]) + (size_t)(c * 9 * 32) * 32;
    const int lrow = lane >> 2, pp = lane & 3;
#pragma unroll
    for (int jj = 0; jj < RB_PV; ++jj) {
        const int row = 16 * (jj * RB_NPROD + pw) + lrow;   // = tap*32 + co
        const int lp = pp ^ ((row >> 2) & 3);
        r[jj] = *reinterpret_cast<const u32x4*>(base + row * 32 + lp * 8);
    }
}
__device__ __forceinline__ void rb_store_slab(char* stage, int lane, int pw, const u32x4 (&r)[RB_PV]) {
#ifdef RB_X_NOWSTORE   // probe: the producers publish without storing (wrong results, LDS write traffic removed)
    if (r[0].x != 0x7f7f7f7fu) return;
#endif
#pragma unroll
    for (int jj = 0; jj < RB_PV; ++jj)
        *reinterpret_cast<u32x4*>(stage + (jj * RB_NPROD + pw) * 1024 + lane * 16) = r[jj];
}

// operand prefetch distance in k-steps for stages with one / two pixel tiles per wave
#ifndef RB_PF1
#define RB_PF1 4
#endif
#ifndef RB_PF2
#define RB_PF2 3
#endif
#ifndef RB_PF5
#define RB_PF5 4   // conv5 (9 k-steps per slab)
#endif
// ---- LDS flags ----
__device__ __forceinline__ int rb_ld(const int* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
__device__ __forceinline__ void rb_wait_ge(const int* p, int target) {
    while (rb_ld(p) < target) __builtin_amdgcn_s_sleep(1);
}

// Contraction of weight slabs into NMT accumulators.  po[m] = this lane's pixel of M-tile m as a byte offset
// (18*oy + ox) * RB_AROW in the stage's own region.
// Software pipeline, written out: there is one MFMA wave per SIMD, so LDS latency (>= 128 cycles) can only be hidden
// inside the wave.  Operand reads run PF k-steps ahead of the MFMAs that consume them; the sched_barrier fences pin that
// order (left alone, hipcc sinks every ds_read next to its MFMA: `ds_read; s_waitcnt lgkmcnt(0); v_mfma`).
// The pipeline runs ACROSS slabs: during the last PF steps of a slab the first PF steps of the next one are read (its
// ring stage is normally complete long before), so the matrix pipe does not drain and refill 26 times per block —
// restarting cold cost ~300 cycles per slab, as much as the 9 MFMAs of a conv5 slab themselves (tools/rdb_probe.hip:
// 768 cycles per conv5 slab).  A slab is described by a reader (RbRdK / RbRd5: operand addresses of its k-steps).
template <int K, int S, int NMT> struct RbRdK {             // slab of conv K < 5 over 32 channels of slice S
    static constexpr int NSTEP = 18;                         // 9 taps x 2 sixteen-channel k-substeps
    const char* ab[NMT];
    const char* bb0;
    const char* bb1;
    __device__ __forceinline__ void init(const char* smem, const char* slab, int plane, const int (&po)[NMT], int i, int g) {
        constexpr int DELTA = K - S - 1;           // offset of conv K's output region inside slice S's region
        const char* sb = smem + rb_slice_base(S) + plane * RB_X0P + g * 16 + (DELTA * RB_PITCH + DELTA) * RB_AROW;
#pragma unroll
        for (int m = 0; m < NMT; ++m) ab[m] = sb + po[m];
#ifdef RB_X_LINA   // probe: pixel fragments from consecutive rows (wrong results; the address pattern of tools/mfma_probe.hip)
#pragma unroll
        for (int m = 0; m < NMT; ++m) ab[m] = smem + rb_slice_base(S) + plane * RB_X0P + g * 16 + (i + 32 * m) * RB_AROW;
#endif
        const int bsw = (i >> 2) & 3;
        bb0 = slab + i * RB_WROW + ((g ^ bsw) << 4);
        bb1 = slab + i * RB_WROW + (((g ^ bsw) ^ 2) << 4);   // second 16-channel k-substep
    }
    template <int n> __device__ __forceinline__ void issue(u32x4& b, u32x4 (&a)[NMT], int i, int g) const {
        constexpr int tap = n / 2, kk = n % 2, ky = tap / 3, kx = tap % 3;
#ifdef RB_X_NOB
        b = u32x4{0x3c003c00u + (unsigned)g, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#else
        b = *reinterpret_cast<const u32x4*>((kk ? bb1 : bb0) + tap * 32 * RB_WROW);
#endif
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#ifdef RB_X_NOA
            a[m] = u32x4{0x3c003c00u + (unsigned)i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#else
            a[m] = *reinterpret_cast<const u32x4*>(ab[m] + (ky * RB_PITCH + kx) * RB_AROW + kk * 32);
#endif
    }
};
// conv5: one (chunk j, 16-channel half h) slab = 9 taps x 64 co rows of 32 B; this wave's N-tile nt -> rows tap*64 + nt*32 + i
template <int S> struct RbRd5 {
    static constexpr int NSTEP = 9;
    const char* ab;
    const char* bb;
    __device__ __forceinline__ void init(const char* smem, const char* slab, int plane, int h, int nt, int po, int i, int g) {
        constexpr int DELTA = 5 - S - 1;
        ab = smem + rb_slice_base(S) + plane * RB_X0P + g * 16 + h * 32 + (DELTA * RB_PITCH + DELTA) * RB_AROW + po;
        bb = slab + (nt * 32 + i) * 32 + ((g ^ ((i >> 3) & 1)) << 4);
    }
    template <int n> __device__ __forceinline__ void issue(u32x4& b, u32x4 (&a)[1], int, int) const {
        b = *reinterpret_cast<const u32x4*>(bb + n * 64 * 32);
        a[0] = *reinterpret_cast<const u32x4*>(ab + ((n / 3) * RB_PITCH + n % 3) * RB_AROW);
    }
};
// first PF k-steps of a slab (cold start: beginning of a stage, or behind a slice wait)
template <int PF, int NMT, typename Rd>
__device__ __forceinline__ void rb_prime(const Rd& rd, u32x4 (&pb)[PF], u32x4 (&pa)[PF][NMT], int i, int g) {
    static_for<0, PF>([&](auto n_c) { rd.template issue<decltype(n_c)::value>(pb[decltype(n_c)::value], pa[decltype(n_c)::value], i, g); });
}
// One slab.  In: pb / pa = its first PF k-steps (already requested).  `sample` runs two steps before the hand-over (the
// caller reads the next slab's ready counter there, so that the answer is back when it is needed); `release` right after
// the LAST operand read of this slab has been issued (LDS executes a wave's operations in order, so the flag write cannot
// overtake the reads); with HAS_NEXT `next()` then returns the reader of the following slab (acquiring its ring stage)
// and its first PF k-steps are requested under the remaining MFMAs, out through pb / pa.
template <int NMT, int PF, bool HAS_NEXT, typename Cur, typename Sample, typename Release, typename Next>
__device__ __forceinline__ void rb_stream(f32x16 (&acc)[NMT], const Cur& cur, u32x4 (&pb)[PF], u32x4 (&pa)[PF][NMT], int i, int g,
                                          Sample&& sample, Release&& release, Next&& next) {
    constexpr int NSTEP = Cur::NSTEP;
    u32x4 bq[NSTEP], aq[NSTEP][NMT];
    static_for<0, PF>([&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        bq[n] = pb[n];
#pragma unroll
        for (int m = 0; m < NMT; ++m) aq[n][m] = pa[n][m];
    });
    decltype(next()) nx = nullptr;                           // reader of the next slab (a pointer; nullptr_t without one)
    static_for<0, NSTEP>([&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF < NSTEP) {
            cur.template issue<n + PF>(bq[n + PF], aq[n + PF], i, g);
            if constexpr (HAS_NEXT && n + PF == NSTEP - 3) sample();
            if constexpr (n + PF == NSTEP - 1) release();
        } else if constexpr (HAS_NEXT) {
            if constexpr (n + PF == NSTEP) nx = next();
            nx->template issue<n + PF - NSTEP>(pb[n + PF - NSTEP], pa[n + PF - NSTEP], i, g);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NMT; ++m) mma16<__bf16>(acc[m], bq[n], aq[n][m]);   // A = weights (rows = co), B = pixels
    });
}

// C fragment with the operands swapped (weights = A): lane l owns ONE pixel (column l & 31 of the M-tile) and
// 16 output channels co = 8*(r>>2) + 4*(l>>5) + (r&3): four runs of 4 consecutive channels.
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

struct RbPix {            // this lane's pixel of one M-tile of stage K
    int po;               // (18*oy + ox) * RB_AROW in the stage's region
    bool valid, inside;   // owns a region pixel / that pixel is inside the image
    size_t gpix;          // clamped global pixel index (n*H + iy)*W + ix
};
template <int K>
__device__ __forceinline__ RbPix rb_pix(int e, int n, int ty0, int tx0, int H, int W) {
    constexpr int HK = 5 - K;
    const int oy = e & 31, ox = (e >> 5) & 31;
    RbPix t;
    t.po = (oy * RB_PITCH + ox) * RB_AROW;
    t.valid = (e >> 10) != 0;
    const int iy = ty0 - HK + oy, ix = tx0 - HK + ox;
    t.inside = t.valid && iy >= 0 && iy < H && ix >= 0 && ix < W;
    t.gpix = (size_t)(n * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1);
    return t;
}
// accumulator start: the bias of this lane's 16 channels (forward) / zero (backward: the gather dgrad has no bias)
template <bool BWD>
__device__ __forceinline__ void rb_acc_init(f32x16& acc, const float* bias_lds, int g) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (!BWD) bq = *reinterpret_cast<const f32x4*>(bias_lds + 8 * q4 + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * q4 + e] = bq[e];
    }
}
// Epilogue of conv K < 5.  forward: x_K = lrelu(acc) (bias is already in acc); backward: dpre = acc * lrelu'(x_k)
// (mk = the saved forward activation); zero outside the image (zero padding of the NEXT stage's input), packed to
// bf16 and written to LDS slice K with four 8-byte stores.  The epilogue is issue bound (one wave per SIMD): max()
// instead of compare+select, the image test applied to the packed words.
template <int K, bool BWD>
__device__ __forceinline__ void rb_store_slice(const f32x16& acc, const RbPix& px, const u32x2v (&mk)[4], char* smem, int g) {
    char* row = smem + (px.valid ? rb_slice_base(K) + px.po : RB_DUMMY) + g * 8;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * q4 + e];
        if (BWD) {
            v[0] = lrelu_mask_lo(v[0], mk[q4][0]); v[1] = lrelu_mask_hi(v[1], mk[q4][0]);
            v[2] = lrelu_mask_lo(v[2], mk[q4][1]); v[3] = lrelu_mask_hi(v[3], mk[q4][1]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = lrelu_max(v[e]);   // == lrelu(v)
        }
        bf16x4v o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        u32x2v w = __builtin_bit_cast(u32x2v, o);
        w[0] = px.inside ? w[0] : 0u;
        w[1] = px.inside ? w[1] : 0u;
        *reinterpret_cast<u32x2v*>(row + 16 * q4) = w;
    }
}
// backward: the 16 channels of x_k (k = 5 - K) of this lane's pixel, from the saved forward buffer
template <int K>
__device__ __forceinline__ void rb_load_mask(const ssr_rdb_desc& d, const RbPix& px, int g, u32x2v (&mk)[4]) {
    const __bf16* mp = reinterpret_cast<const __bf16*>(d.mask.p) + px.gpix * d.mask.cs + d.mask.coff + 64 +
                       32 * (5 - K - 1) + 4 * g;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) mk[q4] = *reinterpret_cast<const u32x2v*>(mp + 8 * q4);
}

// cooperative write of the 8x8 core of LDS slice K (32 channels) to the dense buffer: one 16-B vector per thread
template <int K, bool BWD>
__device__ __forceinline__ void rb_flush_core(const ssr_rdb_desc& d, const char* smem, int n, int ty0, int tx0,
                                              int tid) {
    constexpr int HK = 5 - K;
    const int q = tid >> 2, part = tid & 3;
    const int cy = q >> 3, cx = q & 7;
    const int p = (cy + HK) * RB_PITCH + cx + HK;
    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + rb_slice_base(K) + p * RB_AROW + part * 16);
    const int iy = ty0 + cy, ix = tx0 + cx;
    if (iy < d.H && ix < d.W) {
        constexpr int KD = BWD ? 5 - K : K;      // backward stage K produces dpre_{5-K}
        __bf16* dst = reinterpret_cast<__bf16*>(d.slices.p) + ((size_t)(n * d.H + iy) * d.W + ix) * d.slices.cs +
                      d.slices.coff + 64 + 32 * (KD - 1) + part * 8;
        *reinterpret_cast<u32x4*>(dst) = v;
    }
}

struct RbCtx {            // what every stage needs
    const ssr_rdb_desc& d;
    char* smem;
    char* ring;
    int* ctl;
    const float* bias_lds;
    int n, ty0, tx0, tid, lane, wave, i, g;
    int hint_q, hint;     // ready counter of slab hint_q's ring stage, sampled during the previous slab
#ifdef SSR_PROBE
    unsigned long long wait_ticks = 0, wait_n = 0;
#endif
};
// step q: wait for slab q / hand its ring stage back
__device__ __forceinline__ const char* rb_acquire(RbCtx& c, int q) {
    // slab q is the (q/2 + 1)-th user of stage q % 2: complete when the stage's counter has all its producer parts.
    // (a producer cannot add for slab q + 2 before every consumer has released slab q, so the count is exact)
    const int target = RB_NPROD * (q / RB_NSTAGE + 1);
    if (!(c.hint_q == q && c.hint >= target)) {
#ifdef SSR_PROBE   // slots 14 / 15 of thread 0's probe row: ticks spent polling for slabs, number of polls that had to wait
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        int spins = 0;
        while (rb_ld(c.ctl + CTL_READY + (q % RB_NSTAGE)) < target) ++spins;
        c.wait_ticks += __builtin_amdgcn_s_memtime() - t0;
        c.wait_n += (spins > 0) + 1000;
#else
        while (rb_ld(c.ctl + CTL_READY + (q % RB_NSTAGE)) < target) {}
#endif
    }
    return c.ring + (q % RB_NSTAGE) * RB_SLAB;
}
__device__ __forceinline__ void rb_release(RbCtx& c, int upto) {   // this wave is finished with every slab < upto
    if (c.lane == 0) __atomic_store_n(c.ctl + CTL_DONE + c.wave, upto, __ATOMIC_RELAXED);
}
// release slab q and sample the ready counter of slab q + 1
__device__ __forceinline__ void rb_handover(RbCtx& c, int q) {
    rb_release(c, q + 1);
    c.hint_q = q + 1;
    c.hint = rb_ld(c.ctl + CTL_READY + ((q + 1) % RB_NSTAGE));
}

// sample the ready counter of slab q (consumed by the rb_acquire that follows two k-steps later)
__device__ __forceinline__ void rb_sample(RbCtx& c, int q) {
    c.hint_q = q;
    c.hint = rb_ld(c.ctl + CTL_READY + (q % RB_NSTAGE));
}

// Chunks J..K of growth conv K (slab Q0 + j): the reader of chunk J and its first PF k-steps come in.  Chunk j+1 is
// streamed behind chunk j except when it reads slice K-1 (the last chunk of K > 1), which the previous stage may still be
// writing: that one waits for the slice, sends its 8x8 core to the dense buffer and starts cold.
template <int K, int NMT, bool BWD, int J, typename Rd>
__device__ __forceinline__ void rb_stage_chunks(RbCtx& c, f32x16 (&acc)[NMT], const int (&po)[NMT], const Rd& cur,
                                                u32x4 (&pb)[NMT == 1 ? RB_PF1 : RB_PF2], u32x4 (&pa)[NMT == 1 ? RB_PF1 : RB_PF2][NMT]) {
    constexpr int Q0 = K == 1 ? 0 : K == 2 ? 2 : K == 3 ? 5 : 9, q = Q0 + J, PF = NMT == 1 ? RB_PF1 : RB_PF2;
    constexpr int J1 = J + 1, S1 = J1 < 2 ? 0 : J1 - 1, PL1 = J1 < 2 ? J1 : 0;
    if constexpr (J == K) {                                   // last chunk of the stage
        rb_stream<NMT, PF, false>(acc, cur, pb, pa, c.i, c.g, [] {}, [&] { rb_handover(c, q); }, [] { return nullptr; });
    } else if constexpr (J1 == K && K > 1) {                  // the next chunk starts cold
        rb_stream<NMT, PF, false>(acc, cur, pb, pa, c.i, c.g, [] {}, [&] { rb_handover(c, q); }, [] { return nullptr; });
        rb_wait_ge(c.ctl + CTL_SLICE + (K - 1), 4);
        rb_flush_core<K - 1, BWD>(c.d, c.smem, c.n, c.ty0, c.tx0, c.tid);
        RbRdK<K, S1, NMT> nxt;
        nxt.init(c.smem, rb_acquire(c, q + 1), PL1, po, c.i, c.g);
        rb_prime<PF, NMT>(nxt, pb, pa, c.i, c.g);
        rb_stage_chunks<K, NMT, BWD, J1>(c, acc, po, nxt, pb, pa);
    } else {
        RbRdK<K, S1, NMT> nxt;
        rb_stream<NMT, PF, true>(acc, cur, pb, pa, c.i, c.g, [&] { rb_sample(c, q + 1); }, [&] { rb_release(c, q + 1); },
                                 [&]() -> const RbRdK<K, S1, NMT>* {
                                     nxt.init(c.smem, rb_acquire(c, q + 1), PL1, po, c.i, c.g);
                                     return &nxt;
                                 });
        rb_stage_chunks<K, NMT, BWD, J1>(c, acc, po, nxt, pb, pa);
    }
}

// One growth conv (K = 1..4) for NMT M-tiles of this wave: chunks j = 0..K (slab Q0 + j), then the slice epilogue.
template <int K, int NMT, bool BWD>
__device__ __forceinline__ void rb_stage(RbCtx& c, const int (&ent)[NMT]) {
    constexpr int Q0 = K == 1 ? 0 : K == 2 ? 2 : K == 3 ? 5 : 9, PF = NMT == 1 ? RB_PF1 : RB_PF2;
    f32x16 acc[NMT];
    RbPix px[NMT];
    int po[NMT];
    u32x2v mk[NMT][4];
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
        px[m] = rb_pix<K>(ent[m], c.n, c.ty0, c.tx0, c.d.H, c.d.W);
        po[m] = px[m].po;
        if (BWD) rb_load_mask<K>(c.d, px[m], c.g, mk[m]);
    }
    rb_acc_init<BWD>(acc[0], c.bias_lds + 32 * (K - 1), c.g);
#pragma unroll
    for (int m = 1; m < NMT; ++m) acc[m] = acc[0];
    u32x4 pb[PF], pa[PF][NMT];
    RbRdK<K, 0, NMT> first;                                   // chunk 0: plane 0 of the block input
    if constexpr (K == 1) PROBE(8);
    first.init(c.smem, rb_acquire(c, Q0), 0, po, c.i, c.g);
    if constexpr (K == 1) PROBE(9);
    rb_prime<PF, NMT>(first, pb, pa, c.i, c.g);
    rb_stage_chunks<K, NMT, BWD, 0>(c, acc, po, first, pb, pa);
    if constexpr (K == 1) PROBE(10);
#pragma unroll
    for (int m = 0; m < NMT; ++m) rb_store_slice<K, BWD>(acc[m], px[m], mk[m], c.smem, c.g);
    if (c.lane == 0) __atomic_fetch_add(c.ctl + CTL_SLICE + K, 1, __ATOMIC_RELAXED);   // LDS ops of a wave execute in order
}

// conv5 slabs T..11 (slab 14 + t = chunk t/2, channel half t%2), same streaming; slab 10 is the first reader of slice 4
template <bool BWD, int T, typename Rd>
__device__ __forceinline__ void rb_stage5_slabs(RbCtx& c, f32x16 (&acc)[1], int po, int nt, const Rd& cur, u32x4 (&pb)[RB_PF5],
                                                u32x4 (&pa)[RB_PF5][1]) {
    constexpr int q = 14 + T, T1 = T + 1, J1 = T1 / 2, H1 = T1 % 2, S1 = J1 < 2 ? 0 : J1 - 1, PL1 = J1 < 2 ? J1 : 0;
    if constexpr (T == 11) {
        rb_stream<1, RB_PF5, false>(acc, cur, pb, pa, c.i, c.g, [] {}, [&] { rb_handover(c, q); }, [] { return nullptr; });
    } else if constexpr (T1 == 10) {
        rb_stream<1, RB_PF5, false>(acc, cur, pb, pa, c.i, c.g, [] {}, [&] { rb_handover(c, q); }, [] { return nullptr; });
        rb_wait_ge(c.ctl + CTL_SLICE + 4, 4);
        rb_flush_core<4, BWD>(c.d, c.smem, c.n, c.ty0, c.tx0, c.tid);
        RbRd5<S1> nxt;
        nxt.init(c.smem, rb_acquire(c, q + 1), PL1, H1, nt, po, c.i, c.g);
        rb_prime<RB_PF5, 1>(nxt, pb, pa, c.i, c.g);
        rb_stage5_slabs<BWD, T1>(c, acc, po, nt, nxt, pb, pa);
    } else {
        RbRd5<S1> nxt;
        rb_stream<1, RB_PF5, true>(acc, cur, pb, pa, c.i, c.g, [&] { rb_sample(c, q + 1); }, [&] { rb_release(c, q + 1); },
                              [&]() -> const RbRd5<S1>* {
                                  nxt.init(c.smem, rb_acquire(c, q + 1), PL1, H1, nt, po, c.i, c.g);
                                  return &nxt;
                              });
        rb_stage5_slabs<BWD, T1>(c, acc, po, nt, nxt, pb, pa);
    }
}

template <bool BWD>
__global__ __launch_bounds__(RB_NTHREADS) void rdb_kernel(const ssr_rdb_desc d) {
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
    int* ctl = reinterpret_cast<int*>(smem + RB_CTL);
    PROBE(0);
    const bool producer = wave >= 4;
    const int pw = wave - 4;
    float* bias_lds = reinterpret_cast<float*>(smem + RB_BIAS);
    {
        const int k = tid < 128 ? tid >> 5 : 4, c = tid < 128 ? tid & 31 : tid - 128;   // 4 x 32 + 64 entries
        if (tid < 192) bias_lds[tid] = d.bias[k] ? d.bias[k][c] : 0.f;
        if (tid >= 192 && tid < 208) ctl[tid - 192] = 0;
    }
    // MFMA waves: this lane's pixels (rb_map) for every stage, requested now
    const int w4 = wave & 3;
    const int e1a = rb_map.e[0][w4][i], e1b = rb_map.e[0][w4 + 4][i];
    const int e2a = rb_map.e[1][w4][i], e2b = rb_map.e[1][w4 + 4 < 7 ? w4 + 4 : 6][i];
    const int e3a = rb_map.e[2][w4][i], e3b = rb_map.e[2][4][i];
    const int e4 = rb_map.e[3][w4][i], e5 = rb_map.e[4][w4 & 1][i];
    const int e5s0 = rb_map.e[4][w4 & 1][lane >> 2], e5s1 = rb_map.e[4][w4 & 1][16 + (lane >> 2)];
    // ---- 64-channel input halo region (x / d_out): 18x18 pixels -> X0 (2 planes of 32 channels, padded rows),
    //      staged through registers so that the rows can be padded.  The halo is what the first MFMA needs, so its loads
    //      are issued BEFORE the producers fill their weight queue (a wave's loads return in order). ----
    constexpr int NQ = (2592 + RB_NTHREADS - 1) / RB_NTHREADS;
    u32x4 rx[NQ];
    {
        const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(d.in.p);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int v = tid + q * RB_NTHREADS;             // (plane, pixel, part): 2 x 324 x 4 = 2592 vectors
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
    }
    // producers: a queue of RB_RQ slabs (this wave's 3 KiB of each) in registers.  The ring has only two stages (LDS is
    // full), so run-ahead lives here: conv5 needs a slab per ~300 cycles, the six producers fetch one per ~480; the
    // queue fills during stages 1..4, which consume a slab per 576..1152 cycles.
    constexpr int RB_RQ = 8;
    u32x4 wq[RB_RQ][RB_PV];
    if (producer) static_for<0, RB_RQ>([&](auto uc) { rb_load_slab<BWD>(d, decltype(uc)::value, lane, pw, wq[decltype(uc)::value]); });
    {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int v = tid + q * RB_NTHREADS;
            const int plane = v / 1296, r2 = v - plane * 1296;
            if (v < 2592) *reinterpret_cast<u32x4*>(smem + RB_X0 + plane * RB_X0P + (r2 >> 2) * RB_AROW + (r2 & 3) * 16) = rx[q];
        }
    }
    PROBE(1);
    __syncthreads();   // the only barrier: X0, bias table, zeroed control words
    if (producer) {
        // slab s: registers (requested RB_RQ slabs ahead) -> ring
        // stage s % 2 once every MFMA wave is finished with slab s - 2 -> publish.  LDS operations of a wave execute
        // in order, so the flag write follows the data.
#ifdef SSR_PROBE   // producer wave 4: ticks waiting for the consumers (slot 11), waiting for its loads + storing (12), issuing loads (13)
#define PPROBE_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PPROBE_ADD(k, a, b) pacc[(k) - 11] += (b) - (a)
#else
#define PPROBE_T(var)
#define PPROBE_ADD(k, a, b)
#endif
#ifdef SSR_PROBE
        unsigned long long pacc[3] = {0, 0, 0};
#endif
        auto put = [&](int s_, const u32x4 (&r)[RB_PV]) {
            PPROBE_T(tp0);
            if (s_ >= RB_NSTAGE) {
                for (;;) {
                    // inline asm: for a volatile / atomic LDS read hipcc emits `s_waitcnt vmcnt(0)` first, which would
                    // wait for the refill loads issued a moment ago (one memory latency per slab)
                    u32x4 dn;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dn) : "v"((int)(RB_CTL + 4 * CTL_DONE)) : "memory");
                    if ((int)min(min(dn[0], dn[1]), min(dn[2], dn[3])) >= s_ - (RB_NSTAGE - 1)) break;
#ifdef RB_POLL_SLEEP
                    __builtin_amdgcn_s_sleep(RB_POLL_SLEEP);
#endif
                }
            }
            PPROBE_T(tp1);
            rb_store_slab(ring + (s_ % RB_NSTAGE) * RB_SLAB, lane, pw, r);
            if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"((int)(RB_CTL + 4 * (CTL_READY + (s_ % RB_NSTAGE)))), "v"(1) : "memory");
            PPROBE_T(tp2);
            PPROBE_ADD(11, tp0, tp1);
            PPROBE_ADD(12, tp1, tp2);
        };
        // L2 warm-up for the next launch: blocks are dealt to the 8 XCDs round-robin, so the gridDim.x / 8 blocks of an
        // XCD split the lines of w_next among themselves (one dword per 128-byte line, 64 lines per wave instruction).
        // Plain loads whose values stay live until the end of the wave: an inline-asm load would write its register
        // behind the compiler's back, after the register has been reused.
        unsigned pf = 0;
        {
            const int nper = max(1, (int)gridDim.x >> 3), share = ((int)blockIdx.x >> 3) % nper;
            unsigned pv[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const char* wn = reinterpret_cast<const char*>(d.w_next[k]);
                const int nlines = wn ? d.w_next_bytes[k] >> 7 : 0;
                for (int ln = (share * RB_NPROD + pw) * 64 + lane; ln < nlines; ln += nper * RB_NPROD * 64)
                    pv[k] ^= *reinterpret_cast<const unsigned*>(wn + (size_t)ln * 128);
            }
            pf = pv[0] ^ pv[1] ^ pv[2] ^ pv[3] ^ pv[4];
        }
        for (int s0 = 0; s0 < RB_NSLAB; s0 += RB_RQ)
            static_for<0, RB_RQ>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int s_ = s0 + u;
                if (s_ < RB_NSLAB) put(s_, wq[u]);
                PPROBE_T(tl0);
                if (s_ + RB_RQ < RB_NSLAB) rb_load_slab<BWD>(d, s_ + RB_RQ, lane, pw, wq[u]);
                PPROBE_T(tl1);
                PPROBE_ADD(13, tl0, tl1);
            });
#ifdef SSR_PROBE
        if (tid == 256) for (int k = 0; k < 3; ++k) g_probe[blockIdx.x * 16 + 11 + k] += pacc[k];
#endif
        if (pf == 0x9e3779b9u) ctl[15] = 1;   // keeps the warm-up loads alive; never true for packed bf16 weights in practice
        return;
    }
    // ---------------- MFMA waves ----------------
    RbCtx c{d, smem, ring, ctl, bias_lds, n, ty0, tx0, tid, lane, wave, i, g, -1, 0};
    // stage 1: 16x16 region, 8 M-tiles (wave w: tiles w, w+4); stage 2: 14x14, 7 tiles (wave 3 has one);
    // stage 3: 12x12, 5 tiles (wave 0 has two); stage 4: 10x10, 4 tiles
    { const int ent[2] = {e1a, e1b}; rb_stage<1, 2, BWD>(c, ent); }
    PROBE(2);
    if (wave < 3) { const int ent[2] = {e2a, e2b}; rb_stage<2, 2, BWD>(c, ent); }
    else          { const int ent[1] = {e2a};      rb_stage<2, 1, BWD>(c, ent); }
    PROBE(3);
    if (wave == 0) { const int ent[2] = {e3a, e3b}; rb_stage<3, 2, BWD>(c, ent); }
    else           { const int ent[1] = {e3a};      rb_stage<3, 1, BWD>(c, ent); }
    PROBE(4);
    { const int ent[1] = {e4}; rb_stage<4, 1, BWD>(c, ent); }
    PROBE(5);
    // ================= stage 5: 8x8 core (2 M-tiles) x 64 channels: wave = (M-tile mt, N-tile nt); the 12 slabs
    //                   (6 chunks x 2 channel halves, all 64 output channels each) stream through the same ring ======
    {
        const int nt = wave >> 1;
        f32x16 acc[1];
        const int cy = e5 & 31, cx = (e5 >> 5) & 31;           // this lane's core pixel (always valid: 2 x 32 = 64)
        const int po[1] = {(cy * RB_PITCH + cx) * RB_AROW};
        // forward : out = alpha5*(conv5 + b5) + beta1*x + beta2*x_rrdb               (rrdbnet_arch.py:44, :68)
        // backward: d x = gathered dgrad (alpha5 = 1, conv5's scale is folded into the packed weights)
        //                 + beta1*d_out + beta2*d_out_rrdb
        // (alpha5 multiplies the bias too: the accumulator starts at b5 and is scaled as a whole)
        rb_acc_init<BWD>(acc[0], bias_lds + 128 + nt * 32, g);
        // residual r2 (x_rrdb / d out_rrdb): this lane's pixel, 16 channels = four 8-byte loads, issued now and
        // consumed in the epilogue
        const int iy = ty0 + cy, ix = tx0 + cx;
        const __bf16* __restrict__ r2p = reinterpret_cast<const __bf16*>(d.r2.p);
        u32x2v r2v[4];
        if (r2p) {
            const __bf16* rp = r2p + ((size_t)(n * H + min(iy, H - 1)) * W + min(ix, W - 1)) * d.r2.cs + d.r2.coff +
                               nt * 32 + 4 * g;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) r2v[q4] = *reinterpret_cast<const u32x2v*>(rp + 8 * q4);
        }
        {
            u32x4 pb[RB_PF5], pa[RB_PF5][1];
            RbRd5<0> first;                                   // slab 14: chunk 0 (plane 0 of the block input), channels 0..15
            first.init(smem, rb_acquire(c, 14), 0, 0, nt, po[0], i, g);
            rb_prime<RB_PF5, 1>(first, pb, pa, i, g);
            rb_stage5_slabs<BWD, 0>(c, acc, po[0], nt, first, pb, pa);
        }
        PROBE(6);
        if (lane == 0) __atomic_fetch_add(ctl + CTL_SLICE + 5, 1, __ATOMIC_RELAXED);
        rb_wait_ge(ctl + CTL_SLICE + 5, 4);   // every wave is finished with the ring: it becomes the output transpose slabs
        {
            const char* xrow = smem + RB_X0 + nt * RB_X0P + ((cy + 5) * RB_PITCH + cx + 5) * RB_AROW + g * 8;
            __bf16* slabw = reinterpret_cast<__bf16*>(ring + wave * 2048);   // [32 px][32 co] bf16
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const u32x2v xw = *reinterpret_cast<const u32x2v*>(xrow + 16 * q4);
                const float xv[4] = {bf_lo(xw[0]), bf_hi(xw[0]), bf_lo(xw[1]), bf_hi(xw[1])};
                float rv[4] = {0.f, 0.f, 0.f, 0.f};
                if (r2p) { rv[0] = bf_lo(r2v[q4][0]); rv[1] = bf_hi(r2v[q4][0]); rv[2] = bf_lo(r2v[q4][1]); rv[3] = bf_hi(r2v[q4][1]); }
                bf16x4v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = d.alpha5 * acc[0][4 * q4 + e] + d.beta1 * xv[e];
                    if (r2p) v += d.beta2 * rv[e];
                    o[e] = (__bf16)v;
                }
                *reinterpret_cast<bf16x4v*>(slabw + i * 32 + 8 * q4 + 4 * g) = o;
            }
            // the wave's 32 px x 64 B tile -> 128 16-byte vectors: 2 per lane (row = the lane of the M-tile that owns it)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int v = h * 64 + lane;
                const int row = v >> 2, part = v & 3;
                const int es = h ? e5s1 : e5s0;
                const int oy = ty0 + (es & 31), ox = tx0 + ((es >> 5) & 31);
                const u32x4 val = *reinterpret_cast<const u32x4*>(slabw + row * 32 + part * 8);
                if (oy < H && ox < W) {
                    __bf16* dst = reinterpret_cast<__bf16*>(d.out.p) + ((size_t)(n * H + oy) * W + ox) * d.out.cs +
                                  d.out.coff + nt * 32 + part * 8;
                    *reinterpret_cast<u32x4*>(dst) = val;
                }
            }
        }
    }
    PROBE(7);
#ifdef SSR_PROBE
    if (threadIdx.x == 0) { g_probe[blockIdx.x * 16 + 14] += c.wait_ticks; g_probe[blockIdx.x * 16 + 15] += c.wait_n; }
#endif
}

}  // namespace

// second-generation kernel (csrc/rdb_tile.hip): 8 x 16 or 8 x 8 tiles, swizzled rows, 6-KB slab ring
int rdbt_launch(const ssr_rdb_desc& d, void* stream, bool bwd, int tw);
// Tile choice: ssr_rdb_desc.tile = 8: this file's kernel (8x8 tiles, 80-byte rows); 16: csrc/rdb_tile.hip (8x16 tiles); 0: automatic =
// SSR_RDB_TILE from the environment (8 / 16; read once), else 8x16 tiles when they still give (nearly) every CU a workgroup (one
// workgroup owns a CU: 160 KB of LDS).  The choice travels in the descriptor: no library-global switch.
static int rdb_pick_tile(const ssr_rdb_desc& d) {
    static const int env = [] {
        const char* e = getenv("SSR_RDB_TILE");
        const int v = (e && *e && *e != 'a') ? atoi(e) : -1;
        return v == 0 ? 8 : v;                       // "0" (rounds 2-3) = the 8 x 8 kernel
    }();
    const int choice = (d.tile == 8 || d.tile == 16) ? d.tile : env;
    if (choice == 8 || choice == 16) return choice;
    // 8 x 16 tiles want a workgroup for (nearly) every CU; below that the 8 x 8 tiles of this file fill the chip better
    const int t16 = d.N * ((d.H + 7) / 8) * ((d.W + 15) / 16);
    return t16 >= 192 ? 16 : 8;
}

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
    if (rdb_pick_tile(d) == 16) return rdbt_launch(d, stream, bwd, 16);
    static bool attr_done[2] = {false, false};
    const void* kern = bwd ? reinterpret_cast<const void*>(rdb_kernel<true>) : reinterpret_cast<const void*>(rdb_kernel<false>);
    if (!attr_done[bwd]) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, RB_LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[bwd] = true;
    }
    const int tiles = d.N * ((d.H + 7) / 8) * ((d.W + 7) / 8);
    if (bwd) hipLaunchKernelGGL(rdb_kernel<true>, dim3(tiles), dim3(RB_NTHREADS), RB_LDS, reinterpret_cast<hipStream_t>(stream), d);
    else hipLaunchKernelGGL(rdb_kernel<false>, dim3(tiles), dim3(RB_NTHREADS), RB_LDS, reinterpret_cast<hipStream_t>(stream), d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

extern "C" int ssr_rdb_tile_of(const ssr_rdb_desc* d) { return d ? rdb_pick_tile(*d) : SSR_EINVAL; }
extern "C" int ssr_rdb_forward(const ssr_rdb_desc* dp, void* stream) { return rdb_launch(dp, stream, false); }
extern "C" int ssr_rdb_backward(const ssr_rdb_desc* dp, void* stream) { return rdb_launch(dp, stream, true); }
