// Fused ResidualDenseBlock forward / backward, second generation (bf16, num_feat = 64, num_grow_ch = 32): ONE launch for
// /root/reference/ssr/archs/rrdbnet_arch.py:37-44
//     x1 = lrelu(conv1(x)); x2 = lrelu(conv2(cat(x,x1))); ... x5 = conv5(cat(x..x4)); return x5*0.2 + x
// (and the RRDB tail `out*0.2 + x`, :68, when this is the third block); rdbt_kernel<TW, true> is the gather-form backward
// of the same block (ParamStore.add_rdb_gather): stage K produces dpre_{5-K} from [newer dpre | d_out].
//
// Same contract, descriptor and packed-weight formats as csrc/rdb_fwd.hip (the 8x8-tile kernel of rounds 1-2, kept for
// small grids); what changed is everything the round-2 counters blamed:
//   * tile 8 x TW pixels, TW = 16 (or 8): the 5-pixel halo recompute drops from 1.94x to 1.58x issued MFMAs per
//     algorithmic MFMA, a weight fragment feeds 3 pixel tiles instead of 2 (1.33 instead of 1.5 LDS reads per MFMA),
//     the 479 KB weight stream of a block is amortised over twice the pixels;
//   * activation rows are dense 64-B rows (32 bf16), the 16-B parts XOR-swizzled by f = ((2Y + X) >> 2) & 3 of the
//     pixel's coordinates in the tile's 18 x (TW+10) frame, and every slice has a row pitch = 2 (mod 4): the bank
//     group of (pixel, part p) is 4*((2Y + X - 3S) & 3) + (p ^ f), a bijection of (2Y + X) mod 16 — so ANY 16 lanes
//     whose pixels have distinct (2Y + X) mod 16 read conflict-free, in every slice and under every tap shift.
//     The lane -> pixel table (rt_map) deals the pixels of a stage's region to lanes by that residue;
//     (80-B padded rows of the first kernel: 160 KB for this tile; swizzled: 131 KB)
//   * weight ring of 6-KB slabs (3 taps = one kernel row of one 32-channel chunk), 4 stages at TW = 16: the hand-over
//     latency of the old two-stage 18-KB ring (830 cycles per slab, exposed in conv4/conv5) is covered by depth; round 4: conv5's
//     slabs alternate between the ring and four extra stages in rows of x that are dead by then (RtGeo::C5X: eight slabs deep);
//   * 8 MFMA waves (two per SIMD) + 4 producer waves that each OWN a ring stage = 12 waves, three per SIMD, 168 registers each
//     (the old 4 + 6 had one MFMA wave per SIMD: 35 cycles per MFMA instead of 25.6);
//   * block -> tile map keeps the tiles of an image on ONE XCD (block b runs on XCD b % 8): halo re-reads hit that
//     XCD's L2 instead of fetching every image into all eight;
//   * round 4: the prologue costs ONE memory latency - the x halo comes in by LDS-DMA (no staging registers), the lane -> pixel map
//     is an LDS table filled with one load per thread, the bias / slab-source loads are unconditional and issued before any wait.
//
// LDS map (TW = 16): ring 4 x 6144 B | bias table | control words | X0 2 planes x 468 rows | X1 414 | X2 308 | X3 262 |
// X4 180 rows | dummy row | flag scratch | slab-source table | lane -> pixel map = 159.6 KB.  After the prologue there is NO s_barrier:
// LDS flags as in rdb_fwd.hip
//   ready[8]    (producer -> consumers)    per slab place (4 ring stages + 4 conv5 extras): 1 + the newest slab published there
//   done[8]     (consumer w -> producers)  number of slabs wave w is finished with
//   slice[1..5] (consumers <-> consumers)  waves that have stored their part of slice K / arrived at the final sync
#include "common.h"

#ifdef SSR_PROBE   // tools/rdbt_probe.hip
#define TPROBE(k)                                                                           \
    do {                                                                                    \
        if (threadIdx.x == 0) g_probe[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// fine stamps of wave 0 inside growth stage K: g_probe2[block][4*(K-1) + s], s = 0 stage entered, 1 pipeline primed,
// 2 last MFMA issued, 3 slice stored
#define TPROBE2(K, s)                                                                                        \
    do {                                                                                                     \
        if (threadIdx.x == 0) g_probe2[blockIdx.x * 16 + 4 * ((K) - 1) + (s)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define TPROBE(k)
#define TPROBE2(K, s)
#endif
#ifdef RT_TRACE   // tools/rdbt_check trace: every wave of block 0 logs (event, slab / stage, s_memtime) to g_trace[wave][512]
#define TRACE(ev, arg)                                                                                                   \
    do {                                                                                                                 \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                                                \
            unsigned long long* tr_ = g_trace + (threadIdx.x >> 6) * 512;                                                \
            const unsigned long long n_ = tr_[0];                                                                        \
            if (n_ < 510) { tr_[1 + n_] = ((unsigned long long)(ev) << 56) | ((unsigned long long)(arg) << 48) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffull); tr_[0] = n_ + 1; } \
        }                                                                                                                \
    } while (0)
#else
#define TRACE(ev, arg)
#endif

#ifndef RT_INTERLEAVE5
#define RT_INTERLEAVE5 0   // 1: conv5's chunks spread over the growth stages (r03 / r04: measured 1-3 us slower per launch); 0: conv5 after conv4
#endif
#ifndef RT_C5_DEEP
#define RT_C5_DEEP 1       // conv5's slabs alternate between the ring and four extra stages in dead rows of x (RtGeo::C5X)
#endif

namespace {

// LDS accesses by 32-bit LDS address (address space 3): the operand addresses are per-lane integers (XOR swizzle), and a
// generic `smem + offset` costs one v_add of the (zero) dynamic-LDS base per read
#define RT_LDS __attribute__((address_space(3)))
template <typename T> __device__ __forceinline__ T rt_lds_read(unsigned a) { return *(const RT_LDS T*)(uintptr_t)a; }
template <typename T> __device__ __forceinline__ void rt_lds_write(unsigned a, const T& v) { *(RT_LDS T*)(uintptr_t)a = v; }
typedef __bf16 rt_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned rt_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rt_bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float rt_bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

constexpr int RT_SLAB = 6144;                      // 3 taps x 32 co x 32 ci (conv5: 3 taps x 64 co x 16 ci) bf16
constexpr int RT_NSLAB = 78;                       // 6 + 9 + 12 + 15 + 36
constexpr int RT_NCONS = 8, RT_NPROD = 4, RT_NTHREADS = 64 * (RT_NCONS + RT_NPROD);   // MFMA waves (two per SIMD), producer waves (one per SIMD)

// ---- geometry of a tile of 8 x TW output pixels; frame = the 18 x (TW+10) halo region of the block input ----
template <int TW> struct RtGeo {
    static constexpr int RW0 = TW + 10, RH0 = 18;
    static constexpr int rw(int s) { return RW0 - 2 * s; }
    static constexpr int rh(int s) { return RH0 - 2 * s; }
    static constexpr int pitch(int s) { return (rw(s) & 3) == 2 ? rw(s) : rw(s) + 2; }   // = 2 (mod 4)
    static constexpr int rows(int s) { return (rh(s) - 1) * pitch(s) + rw(s); }
    static constexpr int PLANE = rows(0) * 64;    // x: two planes of 32 channels
    // the ring comes first: its offsets (and the flags') fit the 16-bit immediate of a DS instruction; the activation
    // slices are addressed through per-lane registers anyway
    static constexpr int NST = TW == 16 ? 4 : 8;  // ring stages
    static constexpr int RING = 0;
    static constexpr int BIAS = RING + NST * RT_SLAB;          // [4 x 32 + 64] fp32: conv k < 5 at k*32, conv5 at 128
    static constexpr int CTL = BIAS + (4 * 32 + 64) * 4;       // 32 control words
    static constexpr int ACT = CTL + 128;                      // multiple of 64
    static constexpr int base(int s) { return s == 0 ? ACT : s == 1 ? ACT + 2 * PLANE : base(s - 1) + rows(s - 1) * 64; }
    static constexpr int DUMMY = base(4) + rows(4) * 64;
    static constexpr int SCR = DUMMY + 64;                     // 256 B: where lanes 1..63 of a flag write go (never read)
    // M-tiles (32 pixels) per stage
    static constexpr int ntiles(int K) {
        return TW == 16 ? (K == 1 ? 12 : K == 2 ? 10 : K == 3 ? 8 : K == 4 ? 6 : 4) : (K == 1 ? 8 : K == 2 ? 7 : K == 3 ? 5 : K == 4 ? 4 : 2);
    }
    static constexpr int SRC = SCR + 256;                      // [80] 64-bit source address of every weight slab
    // the lane -> pixel map of the five stages ([tile][32 lanes] ushort, the tiles of a stage back to back: 12 + 10 + 8 + 6 + 4 at
    // TW = 16), copied from global memory with ONE 16-byte load per thread in the prologue.  (Round 4: the MFMA waves used to fetch
    // their eleven entries with separate global loads into registers that hipcc spilled at once - each spill behind an
    // s_waitcnt vmcnt(0), four memory latencies in a row in front of the first MFMA.)
    static constexpr int MAP = SRC + 640;
    static constexpr int map_tile0(int K) {
        int t0 = 0;
        for (int k = 1; k < K; ++k) t0 += ntiles(k);
        return t0;
    }
    static constexpr int MAP_TILES = map_tile0(5) + ntiles(5);
    static constexpr int LDS = MAP + MAP_TILES * 64;
    static_assert(ACT % 64 == 0, "activation rows are 64-byte aligned");
    static constexpr int TROW = 80;               // output transpose slab: [32 px][32 co] bf16, 80-B rows, one per wave
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(NST * RT_SLAB >= RT_NCONS * 32 * TROW, "ring doubles as the output transpose slabs");
    // ---- conv5's deeper ring (round 4).  conv5 consumes a 6-KB slab per 3 k-steps and wave (the growth convs: per 6 .. 18), and
    // what it waited for was the HAND-OVER, not the weights: with the producers' global loads removed (-DRT_X_NOWLOAD) it
    // still took 15.3 k of its 7.7 k no-hand-over ticks (a refill = poll + six ds_write_b128 + flag ~ 850 ticks against
    // 3 x 214 ticks of work in the other three stages).  Once every wave is past the two x chunks of conv4 (slab 32), frame rows
    // 0..3 and 14..17 of x are dead: conv5 reads rows 4..13 only.  Four more stages live there:
    //   X0: plane 0 rows 0..3 | X1, X2: plane 0 rows 14..17 + plane 1 rows 0..3 (contiguous) | X3: plane 1 rows 14..17
    // conv5 slab q (q >= 42) uses ring stage q % 4 when q % 8 < 4 and extra stage q % 4 otherwise: each producer still owns the
    // slabs q = p (mod 4), now with two places to put them.
    static constexpr bool C5X = TW == 16 && !RT_INTERLEAVE5 && RT_C5_DEEP;
    static constexpr int XROWS = 4 * RW0 * 64;                 // four frame rows of one plane
    static_assert(!C5X || XROWS >= RT_SLAB, "an extra stage fits four dead rows");
    static constexpr int xstage(int e) { return e == 0 ? ACT : e == 1 ? ACT + 14 * RW0 * 64 : e == 2 ? ACT + 14 * RW0 * 64 + RT_SLAB : ACT + PLANE + 14 * RW0 * 64; }
    static_assert(!C5X || (pitch(0) == RW0 && xstage(2) + RT_SLAB <= ACT + PLANE + 4 * RW0 * 64 && xstage(3) + RT_SLAB <= ACT + 2 * PLANE), "extra stages stay inside dead rows");
    static constexpr int Q5 = 42;                               // first conv5 slab
    static constexpr int QXFREE = 33;                           // every wave has released slab 32 = rt_qg(4, 1, 2): rows 3 / 14 of x are dead
    static constexpr bool is_extra(int q) { return C5X && q >= Q5 + 2 && (q & 4) != 0; }
    // physical place of slab q: LDS offset, index of its ready word (ring 0..3 / 0..7, extra 4..7), and the slab that used the place before
    static constexpr int stage_off(int q) { return is_extra(q) ? xstage(q & 3) : RING + (q % NST) * RT_SLAB; }
    static constexpr int stage_word(int q) { return is_extra(q) ? 4 + (q & 3) : q % NST; }
    static constexpr int XB = 49152;                            // second base for DS immediates beyond 64 KB
    // M-tile m of MFMA wave w (0..7; waves w and w + 4 share a SIMD) in growth stage K; -1: none.  A wave without a tile skips
    // the stage (it releases the stage's slabs at once).  TW = 16: the SIMDs carry 3,3,3,3 | 3,3,2,2 | 2,2,2,2 | 1,1,2,2 tiles.
    static constexpr int tile_of(int K, int w, int m) {
        if (TW == 16) {
            if (K == 1) return m == 0 ? w + (w >= 4 ? 4 : 0) : (w < 4 ? w + 4 : -1);     // w<4: {w, w+4}; w>=4: {w+4}
            if (K == 2) return m == 0 ? (w < 4 ? w : w < 6 ? w + 4 : -1) : (w < 4 ? w + 4 : -1);   // w<4: {w, w+4}; 4,5: {8, 9}
            if (K == 3) return m == 0 ? w : -1;
            if (K == 4) return m == 0 ? (w < 4 ? w : w >= 6 ? w - 2 : -1) : -1;           // w<4: {w}; 6,7: {4, 5}
            return -1;
        }
        if (m > 0) return -1;
        if (K == 1) return w;
        if (K == 2) return w < 7 ? w : -1;
        if (K == 3) return w < 5 ? w : -1;
        if (K == 4) return w < 4 ? w : -1;
        return -1;
    }
    static constexpr int nmt(int K) { return TW == 16 && K <= 2 ? 2 : 1; }              // most tiles a wave has in stage K
    static constexpr int ntl(int K, int w) { return (tile_of(K, w, 0) >= 0) + (nmt(K) > 1 && tile_of(K, w, 1) >= 0); }
    static constexpr int npart(int K) {                                                   // waves that take part in stage K
        int n = 0;
        for (int w = 0; w < RT_NCONS; ++w) n += K <= 4 ? ntl(K, w) > 0 : (TW == 16 || w < 4);
        return n;
    }
    static constexpr int prank(int K, int w) {                                            // rank of wave w among them
        int n = 0;
        for (int u = 0; u < w; ++u) n += K <= 4 ? ntl(K, u) > 0 : (TW == 16 || u < 4);
        return n;
    }
    // conv5: wave w owns (M-tile, N-tile of 32 output channels)
    static constexpr int m5(int w) { return TW == 16 ? (w & 3) : (w & 1); }
    static constexpr int n5(int w) { return TW == 16 ? (w >> 2) : ((w >> 1) & 1); }
    static constexpr bool in5(int w) { return TW == 16 || w < 4; }
};
constexpr int ctl_ready(int st) { return st; }   // [NST <= 8]
constexpr int RT_CTL_DONE = 8;                       // [8], 16-byte aligned
constexpr int RT_CTL_SLICE = 16;                     // [1..4] slice K complete, [5] final sync

// ---- lane -> pixel map: stage K (region [K, 18-K) x [K, TW+10-K) of the frame), M-tile t, lane i & 31 ->
//      Y | X << 5 | valid << 10 with (2Y + X) mod 16 == i mod 16.  Class c = region pixels with that residue, enumerated
//      by rising (Y, X); M-tile t takes the class's entries 2t (lanes 0..15) and 2t+1 (lanes 16..31).  The hardware
//      serves a ds_read_b128 in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): each holds 16 distinct
//      classes.  Unused slots point at the class's first pixel with valid = 0. ----
template <int TW> struct RtMap { unsigned short e[5][12][32]; };
template <int TW> constexpr RtMap<TW> rt_make_map() {
    using G = RtGeo<TW>;
    RtMap<TW> m{};
    for (int K = 1; K <= 5; ++K) {
        const int nt = G::ntiles(K);
        for (int c = 0; c < 16; ++c) {
            int cnt = 0, first = 0;
            for (int Y = K; Y < 18 - K; ++Y)
                for (int X = K; X < G::RW0 - K; ++X) {
                    if (((2 * Y + X) & 15) != c) continue;
                    const unsigned short v = (unsigned short)(Y | (X << 5));
                    if (cnt == 0) first = v;
                    if (cnt < 2 * nt) m.e[K - 1][cnt >> 1][(cnt & 1) * 16 + c] = (unsigned short)(v | (1 << 10));
                    ++cnt;
                }
            for (int sl = cnt; sl < 2 * nt; ++sl) m.e[K - 1][sl >> 1][(sl & 1) * 16 + c] = (unsigned short)first;
        }
    }
    return m;
}
template <int TW> constexpr bool rt_map_complete() {   // every region pixel is owned by exactly one (tile, lane)
    using G = RtGeo<TW>;
    constexpr RtMap<TW> m = rt_make_map<TW>();
    for (int K = 1; K <= 5; ++K) {
        int n = 0;
        for (int t = 0; t < G::ntiles(K); ++t)
            for (int i = 0; i < 32; ++i) {
                const int e = m.e[K - 1][t][i];
                const int Y = e & 31, X = (e >> 5) & 31;
                if (Y < K || Y >= 18 - K || X < K || X >= G::RW0 - K || ((2 * Y + X) & 15) != (i & 15)) return false;
                if (e >> 10) ++n;
            }
        if (n != (18 - 2 * K) * (G::RW0 - 2 * K)) return false;
    }
    return true;
}
static_assert(rt_map_complete<16>(), "lane -> pixel map must cover every region exactly once");
static_assert(rt_map_complete<8>(), "lane -> pixel map must cover every region exactly once");
// compact copy for the kernel: the tiles of the five stages back to back (RtGeo::map_tile0), 64 bytes per tile, padded to 16-byte vectors
template <int TW> struct RtMapC { unsigned short e[RtGeo<TW>::MAP_TILES][32]; };
template <int TW> constexpr RtMapC<TW> rt_make_mapc() {
    constexpr RtMap<TW> m = rt_make_map<TW>();
    RtMapC<TW> c{};
    for (int K = 1; K <= 5; ++K)
        for (int t = 0; t < RtGeo<TW>::ntiles(K); ++t)
            for (int i = 0; i < 32; ++i) c.e[RtGeo<TW>::map_tile0(K) + t][i] = m.e[K - 1][t][i];
    return c;
}
__device__ const RtMapC<16> rt_mapc16 = rt_make_mapc<16>();
__device__ const RtMapC<8> rt_mapc8 = rt_make_mapc<8>();
template <int TW> __device__ __forceinline__ const u32x4* rt_mapc_vec() {
    if constexpr (TW == 16) return reinterpret_cast<const u32x4*>(&rt_mapc16);
    else return reinterpret_cast<const u32x4*>(&rt_mapc8);
}
// this lane's entry of M-tile t of stage K, from the LDS copy (valid after the prologue's barrier)
template <int TW, int K> __device__ __forceinline__ int rt_map_entry(unsigned lds0, int t, int i) {
    constexpr int T0 = RtGeo<TW>::map_tile0(K);
    return rt_lds_read<unsigned short>(lds0 + RtGeo<TW>::MAP + 2 * ((T0 + t) * 32 + i));
}

// swizzle of a frame pixel: XOR applied to the 16-B part index of its 64-B row
__device__ __forceinline__ int rt_f(int Y, int X) { return ((2 * Y + X) >> 2) & 3; }

// ---- the block's schedule: "extended stages" E1..E5.  conv5 contracts chunk c over slice c-1 (chunks 0, 1: the two planes of
//      x) and needs nothing else, so its chunks are spread over the growth stages instead of forming a sixth of the block's
//      MFMAs with half of its weight bytes at the end (r03: 216 KB for 864 MFMAs = the producers' top rate, the ring ran dry):
//        E1: conv1 chunks 0,1 | conv5 chunk 0            E2: conv2 chunks 0,1 | conv5 chunk 1 | conv2 chunk 2
//        E3: conv3 chunks 0..2 | conv5 chunk 2 | conv3 chunk 3      E4: conv4 chunks 0..3 | conv5 chunk 3 | conv4 chunk 4
//        E5: conv5 chunks 4, 5
//      i.e. inside E_K the conv5 chunk needs only slices that are long complete, and sits in front of the one growth chunk that
//      needs the slice the previous stage has just produced: a wave that is early has work that does not wait for the others.
//      The order of conv5's contraction (chunk, half, row, column) is unchanged.  A slab = one kernel row (3 taps) of one
//      32-channel chunk (conv5: of one 16-channel half chunk); slab numbers run in this order.
#if RT_INTERLEAVE5
constexpr int rt_ebase(int E) { return E == 1 ? 0 : E == 2 ? 12 : E == 3 ? 27 : E == 4 ? 45 : 66; }
constexpr int rt_qg(int K, int j, int ky) { return K == 1 ? 3 * j + ky : rt_ebase(K) + (j < K ? 3 * j + ky : 3 * K + 6 + ky); }
constexpr int rt_q5(int c, int h, int ky) { return (c == 0 ? 6 : c <= 3 ? rt_ebase(c + 1) + 3 * (c + 1) : c == 4 ? 66 : 72) + 3 * h + ky; }
#else
constexpr int rt_qg(int K, int j, int ky) { return (K == 1 ? 0 : K == 2 ? 6 : K == 3 ? 15 : 27) + 3 * j + ky; }
constexpr int rt_q5(int c, int h, int ky) { return 42 + 6 * c + 3 * h + ky; }
#endif
#if RT_INTERLEAVE5
static_assert(rt_qg(1, 1, 2) == 5 && rt_q5(0, 0, 0) == 6 && rt_qg(2, 0, 0) == 12 && rt_q5(1, 0, 0) == 18 && rt_qg(2, 2, 0) == 24 &&
              rt_qg(3, 0, 0) == 27 && rt_q5(2, 0, 0) == 36 && rt_qg(3, 3, 2) == 44 && rt_qg(4, 0, 0) == 45 && rt_q5(3, 1, 2) == 62 &&
              rt_qg(4, 4, 0) == 63 && rt_q5(4, 0, 0) == 66 && rt_q5(5, 1, 2) == 77, "slab numbering");
#endif

// Source address of slab q (6144 contiguous bytes of one packed weight array), bit 0 = conv5 layout.  Evaluated ONCE per block,
// one slab per thread, into an LDS table (the producers' loop then costs one ds_read_b64 per slab instead of a 40-instruction
// decode; and hipcc turned every scalar formulation of "pick one of five kernel-argument pointers" into an indexed load from a
// scratch copy of the pointers, i.e. a memory round trip in front of every slab).
//   weight chunk in memory: forward = j (rrdbnet_arch.py:39-42 cat order); backward = [dpre newest .. oldest | d_out p0 p1]
//   (ParamStore.add_rdb_gather) -> j < 2 ? k + j : k + 1 - j for conv index k = K - 1
//   conv1..4: [chunk of 32 ci][tap][32 co][32 ci]  -> a slab is 96 rows of 64 B, 16-B part XOR-swizzled by (row >> 2) & 3
//   conv5   : [chunk of 16 ci][tap][64 co][16 ci]  -> a slab is 192 rows of 32 B, 16-B part XOR-swizzled by (row >> 3) & 1
template <bool BWD>
__device__ __forceinline__ unsigned long long rt_slab_src(const ssr_rdb_desc& d, int q) {
#if !RT_INTERLEAVE5
    {
        const int k = q < 6 ? 0 : q < 15 ? 1 : q < 27 ? 2 : q < 42 ? 3 : 4;
        const int r = q - (k == 0 ? 0 : k == 1 ? 6 : k == 2 ? 15 : k == 3 ? 27 : 42);
        const int j = k == 4 ? r / 6 : r / 3, h = k == 4 ? (r / 3) & 1 : 0, ky = r % 3;
        unsigned long long base = (unsigned long long)(uintptr_t)d.w[0];
        base = k == 1 ? (unsigned long long)(uintptr_t)d.w[1] : base;
        base = k == 2 ? (unsigned long long)(uintptr_t)d.w[2] : base;
        base = k == 3 ? (unsigned long long)(uintptr_t)d.w[3] : base;
        base = k == 4 ? (unsigned long long)(uintptr_t)d.w[4] : base;
        const int c = BWD ? (j < 2 ? k + j : k + 1 - j) : j;
        const int tapblk = k == 4 ? (2 * c + h) * 9 + 3 * ky : c * 9 + 3 * ky;
#ifdef RT_X_WCOPIES   // probe (tools/rdbt_check): the workgroups of an XCD read RT_X_WCOPIES different copies of the weights (harness allocates them back to back)
        base += (unsigned long long)((blockIdx.x >> 3) % RT_X_WCOPIES) * (unsigned long long)((k == 4 ? 192 * 64 : (64 + 32 * k) * 32) * 18);
#endif
        return (base + (unsigned long long)(tapblk * 2048)) | (k == 4 ? 1ull : 0ull);
    }
#endif
    int E, r;
    if (q < 12) { E = 1; r = q; }
    else if (q < 27) { E = 2; r = q - 12; }
    else if (q < 45) { E = 3; r = q - 27; }
    else if (q < 66) { E = 4; r = q - 45; }
    else { E = 5; r = q - 66; }
    const int ng = E == 1 ? 6 : 3 * E;            // growth slabs in front of the stage's conv5 chunk
    int k, j, ky, h = 0;                           // conv index (0..4), chunk, kernel row, half
    if (E == 5) { k = 4; j = 4 + r / 6; h = (r % 6) / 3; ky = r % 3; }
    else if (r < ng) { k = E - 1; j = r / 3; ky = r % 3; }
    else if (r < ng + 6) { k = 4; j = E - 1; h = (r - ng) / 3; ky = (r - ng) % 3; }
    else { k = E - 1; j = E; ky = r - ng - 6; }
    unsigned long long base = (unsigned long long)(uintptr_t)d.w[0];
    base = k == 1 ? (unsigned long long)(uintptr_t)d.w[1] : base;
    base = k == 2 ? (unsigned long long)(uintptr_t)d.w[2] : base;
    base = k == 3 ? (unsigned long long)(uintptr_t)d.w[3] : base;
    base = k == 4 ? (unsigned long long)(uintptr_t)d.w[4] : base;
    const int c = BWD ? (j < 2 ? k + j : k + 1 - j) : j;
    const int tapblk = k == 4 ? (2 * c + h) * 9 + 3 * ky : c * 9 + 3 * ky;
    return (base + (unsigned long long)(tapblk * 2048)) | (k == 4 ? 1ull : 0ull);   // tap block: 32 co x 32 ci (conv5: 64 co x 16 ci) bf16 = 2048 B
}
typedef const __attribute__((address_space(1))) char* rt_gptr;   // global memory, explicitly (never a flat access)
__device__ const float rt_zero_f32 = 0.f;                         // what a missing bias array reads (every prologue load is unconditional)
__device__ const u32x4 rt_zero16 = {0u, 0u, 0u, 0u};              // what the X0 slots of pixels outside the image read

#ifndef RT_POLL_SLEEP
#define RT_POLL_SLEEP 3
#endif
#ifndef RT_PF2
#define RT_PF2 3      // operand prefetch distance in k-steps, 2 M-tiles per wave (3 reads per k-step; at most 15 LDS operations can be counted)
#endif
#ifndef RT_PF1
#define RT_PF1 5      // ... 1 M-tile per wave (2 reads per k-step)
#endif
#ifndef RT_PF5
#define RT_PF5 5      // ... conv5 (2 reads per k-step)
#endif
// The MFMA waves are ISSUE bound (one wave per SIMD issues ~8 instructions per 32-cycle MFMA; r03 counters: the first cut
// of this kernel spent 13 instructions per MFMA, 7 of them address arithmetic hipcc rematerialised instead of keeping
// 40 registers).  rt_pin makes a value opaque: it must live in a register from here on.
__device__ __forceinline__ void rt_pin(unsigned& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void rt_pin(int& v) { asm volatile("" : "+v"(v)); }

// ---- LDS flags (workgroup scope: plain ds_read / ds_write, no cache or counter side effects) ----
__device__ __forceinline__ int rt_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void rt_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void rt_inc(int* p) { __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void rt_wait_ge(const int* p, int target) {
    while (rt_ld(p) < target) __builtin_amdgcn_s_sleep(1);
}

struct RtPix {            // this lane's pixel of one M-tile
    int Y, X;             // frame coordinates
    bool valid, inside;   // owns a region pixel / that pixel is inside the image
    size_t gpix;          // clamped global pixel index (n*H + iy)*W + ix
};
__device__ __forceinline__ RtPix rt_pix(int e, bool real, int n, int ty0, int tx0, int H, int W) {
    RtPix t;
    t.Y = e & 31;
    t.X = (e >> 5) & 31;
    t.valid = real && (e >> 10) != 0;
    const int iy = ty0 - 5 + t.Y, ix = tx0 - 5 + t.X;
    t.inside = t.valid && iy >= 0 && iy < H && ix >= 0 && ix < W;
    t.gpix = (size_t)(n * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1);
    return t;
}

struct RtCtx {            // what every stage needs
    const ssr_rdb_desc& d;
    unsigned lds0;        // LDS address of the dynamic shared block (0 unless the toolchain places something before it)
    int* ctl;
    const float* bias_lds;
    int n, ty0, tx0, tid, lane, wave, i, g;
    int prank;            // rank of this wave among the waves that take part in the current stage
    int wb0, wb1;         // this lane's weight-fragment offsets inside a conv1..4 slab (k-substep 0 / 1), relative to the ring
    int wb5;              // ... inside a conv5 slab (N-tile 0)
    int wb5x;             // wb5 + RtGeo::XB (the extra stage beyond the reach of a 16-bit DS immediate)
    int hint;             // ready counter of the next slab's ring stage, sampled two k-steps ahead
    unsigned ctlv;        // LDS address of the control words (pinned register: flag accesses are base + immediate)
    unsigned donev;       // LDS address of this wave's done word
#ifdef SSR_PROBE
    unsigned long long wait_ticks = 0, wait_n = 0, slice_ticks = 0;
#endif
};
template <int TW> __device__ __forceinline__ void rt_acquire(RtCtx& c, int q) {
    // the ready word of a slab's place (RtGeo::stage_word) holds 1 + the number of the newest slab published there; a later slab
    // cannot be published in the place before every consumer has released slab q, so "word >= q + 1" means "slab q is there"
    using G = RtGeo<TW>;
    const int target = q + 1;
#ifdef RT_X_NOSYNC   // probe: no hand-over (wrong results)
    return;
#endif
    TRACE(1, q);
    if (__builtin_expect(__builtin_amdgcn_readfirstlane(c.hint) < target, 0)) {   // the word is the same for every lane: scalar compare + branch
#ifdef SSR_PROBE   // slots 14 / 15 of thread 0's probe row: ticks spent polling for slabs, number of slabs that had to wait
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        while (rt_ld(c.ctl + ctl_ready(G::stage_word(q))) < target) __builtin_amdgcn_s_sleep(1);   // leave the issue slots to the producer wave of this SIMD
#ifdef SSR_PROBE
        c.wait_ticks += __builtin_amdgcn_s_memtime() - t0;
        c.wait_n += 1;
#endif
        TRACE(2, q);
    }
}
template <int TW> __device__ __forceinline__ void rt_sample(RtCtx& c, int q) {
    c.hint = __hip_atomic_load((const RT_LDS int*)(uintptr_t)(c.ctlv + 4 * ctl_ready(RtGeo<TW>::stage_word(q))), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void rt_release(RtCtx& c, int upto) {   // this wave is finished with every slab < upto
    // no exec games in the MFMA stream: lane 0 holds the address of the wave's done word, every other lane the address of
    // its own dword of a scratch row — one conflict-free ds_write_b32
    rt_lds_write<int>(c.donev, upto);
    asm volatile("" ::: "memory");
    TRACE(3, upto);
}

// accumulator start: the bias of this lane's 16 channels (forward) / zero (backward: the gather dgrad has no bias)
template <bool BWD>
__device__ __forceinline__ void rt_acc_init(f32x16& acc, const float* bias_lds, int g) {
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
// bf16 and written to LDS slice K with four 8-byte stores.  C fragment with the operands swapped (weights = A): lane l
// owns ONE pixel and 16 output channels co = 8*q4 + 4*(l>>5) + e.
template <int TW, int K, bool BWD>
__device__ __forceinline__ void rt_store_slice(const f32x16& acc, const RtPix& px, const rt_u32x2 (&mk)[4], unsigned lds0, int g) {
    using G = RtGeo<TW>;
    const int f0 = rt_f(px.Y, px.X);
    const unsigned row = lds0 + (px.valid ? G::base(K) + 64 * ((px.Y - K) * G::pitch(K) + (px.X - K)) : G::DUMMY) + 8 * g;
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
        rt_bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        rt_u32x2 w = __builtin_bit_cast(rt_u32x2, o);
        w[0] = px.inside ? w[0] : 0u;
        w[1] = px.inside ? w[1] : 0u;
        rt_lds_write<rt_u32x2>(row + ((q4 ^ f0) << 4), w);
    }
}
// backward: the 16 channels of x_k (k = 5 - K) of this lane's pixel, from the saved forward buffer
template <int K>
__device__ __forceinline__ void rt_load_mask(const ssr_rdb_desc& d, const RtPix& px, int g, rt_u32x2 (&mk)[4]) {
    const __bf16* mp = reinterpret_cast<const __bf16*>(d.mask.p) + px.gpix * d.mask.cs + d.mask.coff + 64 +
                       32 * (5 - K - 1) + 4 * g;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) mk[q4] = *reinterpret_cast<const rt_u32x2*>(mp + 8 * q4);
}

// cooperative write of the 8 x TW core of LDS slice K (32 channels) to the dense buffer: 16-B vectors, by the `nthr` threads
// (rank `rk`) of the waves that wait for the slice
template <int TW, int K, bool BWD>
__device__ __forceinline__ void rt_flush_core(const ssr_rdb_desc& d, unsigned lds0, int n, int ty0, int tx0, int rk, int nthr) {
    using G = RtGeo<TW>;
    for (int v = rk; v < 32 * TW; v += nthr) {
        const int q = v >> 2, part = v & 3;
        const int cy = q / TW, cx = q - cy * TW;
        const int Y = cy + 5, X = cx + 5;
        const u32x4 val = rt_lds_read<u32x4>(lds0 + G::base(K) + 64 * ((Y - K) * G::pitch(K) + (X - K)) + ((part ^ rt_f(Y, X)) << 4));
        const int iy = ty0 + cy, ix = tx0 + cx;
        if (iy < d.H && ix < d.W) {
            constexpr int KD = BWD ? 5 - K : K;      // backward stage K produces dpre_{5-K}
            __bf16* dst = reinterpret_cast<__bf16*>(d.slices.p) + ((size_t)(n * d.H + iy) * d.W + ix) * d.slices.cs +
                          d.slices.coff + 64 + 32 * (KD - 1) + part * 8;
            *reinterpret_cast<u32x4*>(dst) = val;
        }
    }
}

template <int TW, int S> __device__ __forceinline__ unsigned rt_chunk_base(unsigned lds0, int Y, int X) {
    // tap (0,0) neighbour of frame pixel (Y, X) in slice S (plane 0)
    using G = RtGeo<TW>;
    return lds0 + G::base(S) + 64 * ((Y - S - 1) * G::pitch(S) + (X - S - 1));
}

// ---- a wave's share of growth conv K (1..4): NMT M-tiles.  State lives across the segments of the stage. ----
template <int NMT> struct RtGrow {   // what survives between the segments of a stage: accumulators, the map entries (pixels)
    f32x16 acc[NMT];
    int ent[NMT];
    rt_u32x2 mk[NMT][4];             // backward: the epilogue's masks (loaded in the last segment)
};
template <int TW, int K, int NMT, bool BWD>
__device__ __forceinline__ void rt_grow_begin(RtCtx& c, RtGrow<NMT>& st, const int (&ent)[NMT]) {
    TPROBE2(K, 0);
#pragma unroll
    for (int m = 0; m < NMT; ++m) st.ent[m] = ent[m];
    rt_acc_init<BWD>(st.acc[0], c.bias_lds + 32 * (K - 1), c.g);
#pragma unroll
    for (int m = 1; m < NMT; ++m) st.acc[m] = st.acc[0];
}
// Chunks [J0, J1) of growth conv K as ONE software pipeline over their 18 k-steps each (kernel row ky, column kx,
// 16-channel k-substep kk); k-step n consumes weight slab rt_qg(K, j, ky).  LDS latency is hidden inside the wave and by
// the other MFMA wave of the SIMD: operand reads run PF k-steps ahead of the MFMAs that consume them and the sched_barrier
// fences pin that order (left alone, hipcc sinks every ds_read next to its MFMA).  Ring hand-over rides in the same
// stream: a slab is acquired right before its first operand read and released right after its last one (LDS executes a
// wave's operations in order, so the flag write cannot overtake the reads); the ready word of the next slab of the segment is
// sampled two k-steps early.  The last chunk (j = K) reads slice K-1, which the previous stage may still be writing: its first
// read waits for the slice and sends the slice's core to the dense buffer first.
template <int TW, int K, int NMT, bool BWD, int J0, int J1>
__device__ __forceinline__ void rt_grow_run(RtCtx& c, RtGrow<NMT>& st) {
    using G = RtGeo<TW>;
    constexpr int N0 = J0 * 18, N1 = J1 * 18, PF = NMT >= 2 ? RT_PF2 : RT_PF1, NB = PF + 1;
    rt_release(c, rt_qg(K, J0, 0));     // finished with every earlier slab (those of segments this wave had no part in included)
    c.hint = 0;
    // operand addresses of the current chunk, per M-tile: A[kk][t] = chunk base + ((kk*2 + g) ^ f_t) << 4 for the 7 values
    // t = 2(ky-1) + (kx-1) + 3 of a tap's swizzle; a read is A + immediate (tap row/column, plane of x).  Pinned registers.
    unsigned A0[NMT][7], A1[NMT][7];
    int PY[NMT], PX[NMT];
    constexpr int SJ0 = J0 < 2 ? 0 : J0 - 1;
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
        PY[m] = st.ent[m] & 31;
        PX[m] = (st.ent[m] >> 5) & 31;
        const int v0 = 2 * PY[m] + PX[m];
        const unsigned cb = rt_chunk_base<TW, SJ0>(c.lds0, PY[m], PX[m]);
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            A0[m][t] = cb + ((c.g ^ (((v0 + t - 3) >> 2) & 3)) << 4);
            A1[m][t] = A0[m][t] ^ 32u;
            rt_pin(A0[m][t]);
            rt_pin(A1[m][t]);
        }
    }
    u32x4 bq[NB], aq[NB][NMT];
    auto issue = [&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        constexpr int j = n / 18, r = n % 18, ky = r / 6, q = rt_qg(K, j, ky);
        constexpr int S = j < 2 ? 0 : j - 1;
        if constexpr (r == 0 && j > 0) {
            if constexpr (j == K && K > 1) {
#ifdef SSR_PROBE
                const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
                rt_wait_ge(c.ctl + RT_CTL_SLICE + (K - 1), G::npart(K - 1));
#ifdef SSR_PROBE
                c.slice_ticks += __builtin_amdgcn_s_memtime() - ts0;
#endif
                rt_flush_core<TW, K - 1, BWD>(c.d, c.lds0, c.n, c.ty0, c.tx0, c.prank * 64 + c.lane, G::npart(K) * 64);
            }
#ifndef RT_MASK_AHEAD
#define RT_MASK_AHEAD 1      // chunks between the request of the epilogue's masks and the epilogue (1: at the start of the stage's last chunk)
#endif
            if constexpr (j == (K + 1 - RT_MASK_AHEAD > 0 ? K + 1 - RT_MASK_AHEAD : 1) && BWD) {   // the epilogue's masks: requested RT_MASK_AHEAD chunks ahead
#pragma unroll
                for (int m = 0; m < NMT; ++m) rt_load_mask<K>(c.d, rt_pix(st.ent[m], true, c.n, c.ty0, c.tx0, c.d.H, c.d.W), c.g, st.mk[m]);
            }
            // chunk 1 is plane 1 of x: same registers, the plane is an immediate.  A later chunk moves every address by the
            // distance between the two slices' bases for this lane's pixel (a multiple of 64: the swizzle bits stay).
            if constexpr (j >= 2 && n > N0) {
                constexpr int SP = j == 2 ? 0 : j - 2;
#pragma unroll
                for (int m = 0; m < NMT; ++m) {
                    const unsigned delta = rt_chunk_base<TW, S>(0u, PY[m], PX[m]) - rt_chunk_base<TW, SP>(0u, PY[m], PX[m]);
#pragma unroll
                    for (int t = 0; t < 7; ++t) {
                        A0[m][t] += delta;
                        A1[m][t] += delta;
                        rt_pin(A0[m][t]);
                        rt_pin(A1[m][t]);
                    }
                }
            }
        }
        if constexpr (r % 6 == 0) rt_acquire<TW>(c, q);
    };
    auto issue_part = [&](auto n_c, auto p_c) {
        constexpr int n = decltype(n_c)::value, part = decltype(p_c)::value;
        constexpr int j = n / 18, r = n % 18, ky = r / 6, kx = (r % 6) / 2, kk = r % 2, q = rt_qg(K, j, ky);
        constexpr int S = j < 2 ? 0 : j - 1, plane = j < 2 ? j : 0;
        constexpr int t = 2 * (ky - 1) + (kx - 1) + 3;
        if constexpr (part == 0) {
            bq[n % NB] = rt_lds_read<u32x4>((kk ? c.wb1 : c.wb0) + (G::stage_off(q) + kx * 2048));
        } else {
            constexpr int m = part - 1;
            aq[n % NB][m] = rt_lds_read<u32x4>((kk ? A1[m][t] : A0[m][t]) + (plane * G::PLANE + 64 * (ky * G::pitch(S) + kx)));
            if constexpr (m == NMT - 1) {
                if constexpr (r % 6 == 3 && n + 3 < N1) rt_sample<TW>(c, rt_qg(K, (n + 3) / 18, ((n + 3) % 18) / 6));
                if constexpr (r % 6 == 5) rt_release(c, q + 1);
            }
        }
    };
    auto issue_all = [&](auto n_c) {
        issue(n_c);
        static_for<0, NMT + 1>([&](auto p_c) { issue_part(n_c, p_c); });
    };
    static_for<N0, N0 + PF>(issue_all);
    if constexpr (J0 == 0) TPROBE2(K, 1);
    static_for<N0, N1>([&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        using NX = std::integral_constant<int, n + PF>;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF < N1) issue_all(NX{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NMT; ++m) mma16<__bf16>(st.acc[m], bq[n % NB], aq[n % NB][m]);   // A = weights (rows = co), B = pixels
    });
    __builtin_amdgcn_sched_barrier(0);
}
template <int TW, int K, int NMT, bool BWD>
__device__ __forceinline__ void rt_grow_end(RtCtx& c, RtGrow<NMT>& st) {
    TPROBE2(K, 2);
    TRACE(7, K);
#pragma unroll
    for (int m = 0; m < NMT; ++m) rt_store_slice<TW, K, BWD>(st.acc[m], rt_pix(st.ent[m], true, c.n, c.ty0, c.tx0, c.d.H, c.d.W), st.mk[m], c.lds0, c.g);
    if (c.lane == 0) rt_inc(c.ctl + RT_CTL_SLICE + K);   // LDS ops of a wave execute in order
    TRACE(8, K);
    TPROBE2(K, 3);
}

// conv5 over the 8 x TW core: this wave's (M-tile, N-tile of 32 output channels); chunk c = 18 k-steps (16-channel half h,
// ky, kx), slab rt_q5(c, h, ky); chunk c >= 2 reads slice c-1.  One accumulator, alive from E1 to the end of the block.
template <int TW, bool BWD, int C0, int C1>
__device__ __forceinline__ void rt_c5_run(RtCtx& c, f32x16& acc, int e5) {
    using G = RtGeo<TW>;
    constexpr int N0 = C0 * 18, N1 = C1 * 18, PF = RT_PF5, NB = PF + 1;
    constexpr int S0 = C0 < 2 ? 0 : C0 - 1;
    rt_release(c, rt_q5(C0, 0, 0));
    c.hint = 0;
    const int Y = e5 & 31, X = (e5 >> 5) & 31;
    unsigned A0[7], A1[7];
    {
        const int v0 = 2 * Y + X;
        const unsigned cb = rt_chunk_base<TW, S0>(c.lds0, Y, X);
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            A0[t] = cb + ((c.g ^ (((v0 + t - 3) >> 2) & 3)) << 4);
            A1[t] = A0[t] ^ 32u;
            rt_pin(A0[t]);
            rt_pin(A1[t]);
        }
    }
    u32x4 bq[NB], aq[NB];
    auto issue = [&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        constexpr int j = n / 18, r = n % 18, h = r / 9, ky = (r % 9) / 3, kx = r % 3, q = rt_q5(j, h, ky);
        constexpr int S = j < 2 ? 0 : j - 1, plane = j < 2 ? j : 0;
        if constexpr (r == 0) {
            if constexpr (j >= 2) {   // slice j-1 must be complete (a wave without a tile in growth stage j has not waited for it yet)
#ifdef SSR_PROBE
                const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
                rt_wait_ge(c.ctl + RT_CTL_SLICE + (j - 1), G::npart(j - 1));
#ifdef SSR_PROBE
                c.slice_ticks += __builtin_amdgcn_s_memtime() - ts0;
#endif
                if constexpr (j == 5) rt_flush_core<TW, 4, BWD>(c.d, c.lds0, c.n, c.ty0, c.tx0, c.prank * 64 + c.lane, G::npart(5) * 64);
            }
            if constexpr (n > N0 && j >= 2) {
                constexpr int SP = j == 2 ? 0 : j - 2;
                const unsigned delta = rt_chunk_base<TW, S>(0u, Y, X) - rt_chunk_base<TW, SP>(0u, Y, X);
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    A0[t] += delta;
                    A1[t] += delta;
                    rt_pin(A0[t]);
                    rt_pin(A1[t]);
                }
            }
        }
        if constexpr (kx == 0) rt_acquire<TW>(c, q);
        constexpr int t = 2 * (ky - 1) + (kx - 1) + 3;
        if constexpr (G::stage_off(q) + kx * 2048 + 6144 > 65535) bq[n % NB] = rt_lds_read<u32x4>(c.wb5x + (G::stage_off(q) - G::XB + kx * 2048));   // DS immediates are 16 bits
        else bq[n % NB] = rt_lds_read<u32x4>(c.wb5 + (G::stage_off(q) + kx * 2048));
        aq[n % NB] = rt_lds_read<u32x4>((h ? A1[t] : A0[t]) + (plane * G::PLANE + 64 * (ky * G::pitch(S) + kx)));
        if constexpr (kx == 1 && n + 2 < N1) rt_sample<TW>(c, rt_q5((n + 2) / 18, ((n + 2) % 18) / 9, (((n + 2) % 18) % 9) / 3));
        if constexpr (kx == 2) rt_release(c, q + 1);
    };
    static_for<N0, N0 + PF>(issue);
    static_for<N0, N1>([&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF < N1) issue(std::integral_constant<int, n + PF>{});
        __builtin_amdgcn_sched_barrier(0);
        mma16<__bf16>(acc, bq[n % NB], aq[n % NB]);
    });
    __builtin_amdgcn_sched_barrier(0);
}

template <int TW, bool BWD>
__global__ __launch_bounds__(RT_NTHREADS) void rdbt_kernel(const ssr_rdb_desc d) {
    using G = RtGeo<TW>;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: SALU
    const int i = lane & 31, g = lane >> 5;
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + 7) / 8, tpi = tiles_x * tiles_y;
    // block -> tile: block b runs on XCD b % 8 (observed placement; only speed depends on it).  The work list (image-major)
    // is cut into 8 contiguous shares, one per XCD, so that the tiles of an image share an L2.
    int item = blockIdx.x;
    if ((gridDim.x & 7) == 0) item = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int n = item / tpi;
    const int trem = item - n * tpi;
    const int ty_i = trem / tiles_x, tx_i = trem - ty_i * tiles_x;
    const int ty0 = ty_i * 8, tx0 = tx_i * TW;
    const int H = d.H, W = d.W;
    const unsigned lds0 = (unsigned)(size_t)(RT_LDS char*)smem;
    int* ctl = reinterpret_cast<int*>(smem + G::CTL);
    float* bias_lds = reinterpret_cast<float*>(smem + G::BIAS);
    TPROBE(0);
    const bool producer = wave >= RT_NCONS;
    const int pw = wave - RT_NCONS;
    // ---- prologue, part 1: every global load the block needs before its first MFMA is ISSUED here, unconditionally, before any of
    //      them is waited for: the x halo first (below), then the bias entry, the slab-source entry and the lane -> pixel map vector
    //      of this thread.  (Round 4: the bias load used to be waited for in front of everything else, the map entries were eleven
    //      separate loads spilled one by one behind s_waitcnt vmcnt(0): six memory latencies in a row, 4.9 k of the block's 58 k
    //      ticks; tools/rdbt_check probe.)
    float pro_bias[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned long long pro_src;
    u32x4 pro_map;
    auto prologue_loads = [&]() {
        // bias table [4 x 32 + 64]: thread t reads entry t & 31 of conv1..4 and t & 63 of conv5 through the five SCALAR pointers of the
        // descriptor (a per-thread pick of one of them becomes an indexed load of the pointer + a dependent flat load: two latencies)
        if (!BWD) {
            typedef const __attribute__((address_space(1))) float* gfp;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float* bp = d.bias[k] ? d.bias[k] + (k < 4 ? (tid & 31) : (tid & 63)) : &rt_zero_f32;
                pro_bias[k] = *(gfp)bp;
            }
        }
        pro_src = rt_slab_src<BWD>(d, min(tid, RT_NSLAB - 1));
        pro_map = rt_mapc_vec<TW>()[min(tid, G::MAP_TILES * 4 - 1)];
    };
    auto prologue_stores = [&]() {
        if (tid < 64) {
            if (tid < 32) {
#pragma unroll
                for (int k = 0; k < 4; ++k) bias_lds[k * 32 + tid] = pro_bias[k];
            }
            bias_lds[128 + tid] = pro_bias[4];
        }
        if (tid >= 192 && tid < 224) ctl[tid - 192] = 0;
        if (tid < 80) rt_lds_write<unsigned long long>(lds0 + G::SRC + 8 * tid, pro_src);
        if (tid < G::MAP_TILES * 4) rt_lds_write<u32x4>(lds0 + G::MAP + 16 * tid, pro_map);
    };
    // ---- the 64-channel input halo region (x / d_out): 18 x (TW+10) pixels -> X0 (2 planes of 32 channels, swizzled rows),
    //      staged through registers by ALL waves (the MFMA waves have nothing else to do before the barrier).  It is what the
    //      first MFMA needs, so the producers issue its loads BEFORE their first weight slabs (a wave's loads return in order).
    // Round 4: by LDS-DMA (global_load_lds_dwordx4) instead of through registers.  Staged through registers the five vectors per
    // thread were spilled by hipcc one by one in the prologue (load, s_waitcnt vmcnt(0), scratch store, next load: a memory latency
    // each); a DMA needs no data register.  A wave-instruction fills 1 KiB of LDS, lane l the 16-byte slot l of it, from ANY global
    // address: the lane picks the (pixel, part) whose swizzled place that slot is; pixels outside the image read a zero block.
    constexpr int X0_SLOTS = 2 * G::PLANE / 16, X0_INST = (X0_SLOTS + 63) / 64, PL_SLOTS = G::PLANE / 16;
    auto x0_dma = [&]() {
        const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(d.in.p);
        typedef const __attribute__((address_space(1))) void* gvp;
        typedef __attribute__((address_space(3))) void* lvp;
        for (int u = wave; u < X0_INST; u += RT_NCONS + RT_NPROD) {      // wave-uniform
            const int sl = u * 64 + lane;                                // 16-byte slot of X0 this lane fills
            const int plane = sl >= PL_SLOTS ? 1 : 0, r2 = sl - plane * PL_SLOTS;
            const int pix = r2 >> 2;
            const int py = pix / G::RW0, pxx = pix - py * G::RW0;
            const int part = (r2 & 3) ^ rt_f(py, pxx);                   // the logical part whose swizzled place is this slot
            const int iy = ty0 - 5 + py, ix = tx0 - 5 + pxx;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const __bf16* src = xg + ((size_t)(n * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)) * d.in.cs + d.in.coff + plane * 32 + part * 8;
            const void* sp = in ? (const void*)src : (const void*)&rt_zero16;
            if (sl < X0_SLOTS)
                __builtin_amdgcn_global_load_lds((gvp)sp, (lvp)(smem + G::base(0) + u * 1024), 16, 0, 0);
        }
    };
    // ---- producer waves: the block's data movers ----
    if (producer) {
        // (the first RT_RQ weight slabs of this wave are requested FIRST, below; then its share of the x halo and the small tables:
        //  everything the block needs before its first MFMA is in flight within one memory latency)
        // wave pw owns the slabs q = pw (mod 4) and with them the ring stages q % NST (NST = 4: exactly one), so a stage's
        // ready word has ONE writer and a slab costs one decode, six loads, six stores and one flag; a queue of RT_RQ whole
        // slabs (6 KiB: six 16-byte vectors per lane) lives in registers.  (r03: dealing 1-KiB pieces round-robin made the
        // producers instruction bound — ~100 mostly scalar instructions per piece — and they paced the whole block.)
#ifndef RT_RQ_DEPTH
#define RT_RQ_DEPTH 3
#endif
        constexpr int RT_RQ = RT_RQ_DEPTH, NSP = (RT_NSLAB + RT_NPROD - 1) / RT_NPROD;    // up to 20 slabs per producer
        static_assert(G::NST % RT_NPROD == 0, "a ring stage belongs to one producer");
        u32x4 wq[RT_RQ][6];
        // lane's 16 bytes inside a 1-KiB piece: conv1..4 rows of 64 B (part swizzled by (row >> 2) & 3), conv5 rows of 32 B
        // (by (row >> 3) & 1); the row phase of a piece is a multiple of 16 rows, so the pattern is the same for every piece
        const int lo14 = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
        const int lo5 = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
#if defined(RT_SEL) && (RT_SEL == 2 || RT_SEL == 3)
        int dlo = lo5 - lo14, xlo = lo5 ^ lo14;
        rt_pin(dlo);
        rt_pin(xlo);
#endif
        auto load_from = [&](unsigned long long src64, u32x4 (&r)[6]) {
            rt_gptr src = (rt_gptr)(src64 & ~1ull);
            // bit 0 of a table entry: conv5 layout -> this lane's offset inside a 1-KiB piece.  RT_SEL picks the form of the select:
            //   1 (shipped): lo14 + bit * (lo5 - lo14) - one v_cndmask / v_mad, RT_RQ slabs in flight per producer;
            //   0: `bit ? lo5 : lo14` - hipcc turns it into a table of two stack addresses (scratch load + flat load + s_waitcnt
            //      vmcnt(0) lgkmcnt(0) in front of every slab's loads: ONE slab load in flight per producer); rounds 2-3 shipped it;
            //   2: the difference in a pinned register, 3: XOR mask, 4: inline-asm v_cndmask.
            // Round 3 saw forms 2-4 give wrong outputs in the library build (DESIGN.md lesson 36).  Root cause (round 4): the ring's
            // done-poll below was a two-instruction inline asm WITHOUT early-clobber outputs; when the poll address is dead after
            // the asm (forms 2-4 change the allocation so that it is rematerialised per poll) hipcc gave the first ds_read_b128's
            // destination the address register, and when that read returned before the second one issued (LDS queue backed up) the
            // second read fetched garbage "done" counts -> a producer refilled a ring stage the slowest MFMA waves were still
            // reading.  With "=&v" all five forms are byte-identical to the 8x8 kernel (tools/rdbt_check, tests/test_gpu_rdb_stress.py).
#ifndef RT_SEL
#define RT_SEL 1
#endif
#if RT_SEL == 0
            const int lo = (src64 & 1ull) ? lo5 : lo14;
#elif RT_SEL == 1
            const int lo = lo14 + (int)(src64 & 1ull) * (lo5 - lo14);
#elif RT_SEL == 2
            const int lo = lo14 + (int)(src64 & 1ull) * dlo;
#elif RT_SEL == 3
            const int lo = lo14 ^ (xlo & -(int)(src64 & 1ull));
#else
            int lo;
            asm volatile("v_cmp_ne_u32 vcc, 0, %3\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=&v"(lo) : "v"(lo14), "v"(lo5), "v"((int)(src64 & 1ull)) : "vcc");
#endif
#ifdef RT_X_NOWLOAD   // probe: the whole hand-over protocol without the weight loads (wrong results): what is left is the ring, not the L2
#pragma unroll
            for (int e = 0; e < 6; ++e) r[e] = u32x4{(unsigned)lo, 0u, 0u, (unsigned)(uintptr_t)src};
#else
#pragma unroll
            for (int e = 0; e < 6; ++e) r[e] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(src + lo + e * 1024);
#endif
        };
        // a slab's source comes from the LDS table (beyond the end: the last slab again, never stored)
        auto load_slab = [&](int q, u32x4 (&r)[6]) {
            const int qc = min(q, RT_NSLAB - 1);
            unsigned long long a;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a) : "v"((int)lds0 + G::SRC + 8 * qc) : "memory");
            const unsigned alo = __builtin_amdgcn_readfirstlane((unsigned)a), ahi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            load_from(((unsigned long long)ahi << 32) | alo, r);
        };
        // the first RT_RQ slabs go out before the barrier (the table is not there yet): decoded on the vector unit from an
        // opaque copy of q, so that the pointer pick stays a v_cndmask chain
        static_for<0, RT_RQ>([&](auto uc) {
            const int q = pw + RT_NPROD * decltype(uc)::value;
            static_assert(RT_NPROD * (RT_RQ - 1) + RT_NPROD - 1 < 15 || RT_INTERLEAVE5, "the first slabs of a producer belong to conv1 / conv2");
            unsigned long long a;
            if constexpr (!RT_INTERLEAVE5 && RT_NPROD * (RT_RQ - 1) + RT_NPROD - 1 < 15) {
                // slabs 0..14 = conv1 (0..5) and conv2 (6..14): a scalar pick between two kernel-argument pointers (s_cselect), no
                // indexed load of the pointer in front of the six loads (which cost the producers a memory latency per slab here)
                const bool c2 = q >= 6;
                const int r = c2 ? q - 6 : q, j = r / 3, ky = r - 3 * j, k = c2 ? 1 : 0;
                const int c = BWD ? (j < 2 ? k + j : k + 1 - j) : j;
                const unsigned long long base = c2 ? (unsigned long long)(uintptr_t)d.w[1] : (unsigned long long)(uintptr_t)d.w[0];
                a = base + (unsigned long long)((c * 9 + 3 * ky) * 2048);
            } else {
                int qv = q;
                rt_pin(qv);
                const unsigned long long av = rt_slab_src<BWD>(d, qv);
                a = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(av >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)av);
            }
            load_from(a, wq[decltype(uc)::value]);
        });
        prologue_loads();
        x0_dma();            // last: hipcc waits for every outstanding load (s_waitcnt vmcnt(0)) behind the DMA loop
        prologue_stores();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of X0 has landed (DMA writes are not covered by the barrier)
        __syncthreads();   // the only barrier: X0, bias table, source table, lane -> pixel map, zeroed control words
#ifndef RT_PRIO_PROD
#define RT_PRIO_PROD 0
#endif
        __builtin_amdgcn_s_setprio(RT_PRIO_PROD);
#ifdef RT_X_NOPROD   // probe: the MFMA waves alone (with RT_X_NOSYNC)
        return;
#endif
        // slab s of this wave (q = pw + 4 s): registers (requested RT_RQ slabs ahead) -> ring stage q % NST once every MFMA
        // wave is finished with slab q - NST -> publish.  LDS operations of a wave execute in order, so the flag follows
        // the data.
#ifdef SSR_PROBE   // producer wave 4: ticks waiting for the consumers (slot 11), waiting for its loads + storing (12)
        unsigned long long pacc[2] = {0, 0};
#define RT_PT(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define RT_PADD(k, a, b) pacc[k] += (b) - (a)
#else
#define RT_PT(var)
#define RT_PADD(k, a, b)
#endif
        unsigned stv = lds0 + G::RING + lane * 16;             // + stage * RT_SLAB + e * 1024
        unsigned flagv = lane == 0 ? lds0 + G::CTL : lds0 + G::SCR + 4 * lane;   // lane 0: the ready words; others: scratch
        rt_pin(stv);
        rt_pin(flagv);
        auto put = [&](int q, const u32x4 (&r)[6]) {
            RT_PT(tp0);
            TRACE(4, q);
#ifndef RT_X_NOSYNC
            // the place of slab q is free once every MFMA wave has released the slab that was there before: q - NST in the ring;
            // conv5 (C5X): q - 4 for its first two slabs (ring), "x rows dead" for the first four of the extra stages, then q - 8
            int need = q - (G::NST - 1);
            if (G::C5X && q >= G::Q5 + 2) need = q < G::Q5 + 6 ? G::QXFREE : q - 7;
            if (q >= G::NST) {
                for (;;) {
                    // inline asm: a compiler-visible LDS read here would make hipcc drain the refill loads first
                    u32x4 dn, dm;
                    // "=&v": the outputs must not share a register with the address - the first read may return before the second
                    // one issues (lesson 36; RT_X_POLL_NOEARLY rebuilds the round-3 form for the reproduction in tools/gpu_r4a.sh)
#ifdef RT_X_POLL_NOEARLY
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=v"(dn), "=v"(dm) : "v"((int)lds0 + (int)(G::CTL + 4 * RT_CTL_DONE)) : "memory");
#else
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dn), "=&v"(dm) : "v"((int)lds0 + (int)(G::CTL + 4 * RT_CTL_DONE)) : "memory");
#endif
                    const int dmin = (int)min(min(min(dn[0], dn[1]), min(dn[2], dn[3])), min(min(dm[0], dm[1]), min(dm[2], dm[3])));
                    if (__builtin_amdgcn_readfirstlane(dmin) >= need) break;
                    __builtin_amdgcn_s_sleep(RT_POLL_SLEEP);   // the ring is normally full: a poll per ~200 cycles is plenty and costs the MFMA wave of this SIMD nothing
                }
            }
#endif
            RT_PT(tp1);
            TRACE(5, q);
            const int st = q % G::NST;
            const bool extra = G::C5X && q >= G::Q5 + 2 && (q & 4) != 0;
            int off = st * RT_SLAB, word = st;
            if (extra) {
                off = (q & 3) == 0 ? G::xstage(0) : (q & 3) == 1 ? G::xstage(1) : (q & 3) == 2 ? G::xstage(2) : G::xstage(3);
                off -= G::RING;
                word = 4 + (q & 3);
            }
#pragma unroll
            for (int e = 0; e < 6; ++e) rt_lds_write<u32x4>(stv + off + e * 1024, r[e]);
            rt_lds_write<int>(flagv + (lane == 0 ? 4 * ctl_ready(0) + 4 * word : 0), q + 1);
            asm volatile("" ::: "memory");
            TRACE(6, q);
            RT_PT(tp2);
            RT_PADD(0, tp0, tp1);
            RT_PADD(1, tp1, tp2);
        };
        for (int s0 = 0; s0 < NSP; s0 += RT_RQ)
            static_for<0, RT_RQ>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int q = pw + RT_NPROD * (s0 + u);
                if (q < RT_NSLAB) put(q, wq[u]);
                // refill unconditionally so that the number of loads in flight is the same on every path and the compiler's
                // vmcnt bookkeeping keeps the queue RT_RQ slabs deep
                load_slab(q + RT_NPROD * RT_RQ, wq[u]);
            });
#ifdef SSR_PROBE
        if (tid == 64 * RT_NCONS) { g_probe[blockIdx.x * 16 + 11] = pacc[0]; g_probe[blockIdx.x * 16 + 12] = pacc[1]; }
#endif
        // (r03: an L2 warm-up of the next launch's weights from here — ssr_rdb_desc.w_next, one dword per line, dealt over the
        // producers of an XCD — did not shorten the next launch: 34.5 vs 34.1 us in a chain of 69 blocks with distinct
        // weights, 32.5 with shared ones; tools/rdbt_check chain.  Not kept.)
        return;
    }
    // ---------------- MFMA waves ----------------
    prologue_loads();
    x0_dma();
    // tile_of is constexpr in (K, w, m), the wave index is a run-time (scalar) value: select over the eight waves.  This lane's
    // pixels come out of the LDS copy of the map when a stage begins (rt_map_entry).
    auto pick8 = [&](auto f) {
        const int v0 = f(0), v1 = f(1), v2 = f(2), v3 = f(3), v4 = f(4), v5 = f(5), v6 = f(6), v7 = f(7);
        return wave == 0 ? v0 : wave == 1 ? v1 : wave == 2 ? v2 : wave == 3 ? v3 : wave == 4 ? v4 : wave == 5 ? v5 : wave == 6 ? v6 : v7;
    };
    int tl[4][2], ntl[4], prk[5];                   // scalar (wave-uniform) values
#pragma unroll
    for (int K = 1; K <= 4; ++K) {
        ntl[K - 1] = pick8([&](int w) { return G::ntl(K, w); });
#pragma unroll
        for (int m = 0; m < G::nmt(K); ++m) {
            const int t = pick8([&](int w) { return G::tile_of(K, w, m); });
            tl[K - 1][m] = t >= 0 ? t : 0;
        }
    }
#pragma unroll
    for (int K = 1; K <= 5; ++K) prk[K - 1] = pick8([&](int w) { return G::prank(K, w); });
    const int mt5 = pick8([&](int w) { return G::m5(w); }), nt5 = pick8([&](int w) { return G::n5(w); });
    prologue_stores();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TPROBE(1);
    __syncthreads();   // the only barrier (see the producer branch)
    const int e5 = rt_map_entry<TW, 5>(lds0, mt5, i);
    TRACE(9, 0);
#ifndef RT_PRIO_MFMA
#define RT_PRIO_MFMA 1
#endif
    __builtin_amdgcn_s_setprio(RT_PRIO_MFMA);
    const int bsw = (i >> 2) & 3;
    RtCtx c{d, lds0, ctl, bias_lds, n, ty0, tx0, tid, lane, wave, i, g, 0,
            (int)lds0 + i * 64 + ((g ^ bsw) << 4), (int)lds0 + i * 64 + (((g ^ bsw) ^ 2) << 4),
            (int)lds0 + nt5 * 1024 + i * 32 + ((g ^ ((i >> 3) & 1)) << 4), (int)lds0 + nt5 * 1024 + i * 32 + ((g ^ ((i >> 3) & 1)) << 4) + G::XB, 0, lds0 + G::CTL,
            lane == 0 ? lds0 + G::CTL + 4 * (RT_CTL_DONE + wave) : lds0 + G::SCR + 4 * lane};
    rt_pin(c.ctlv);
    rt_pin(c.donev);
    rt_pin(c.wb0);
    rt_pin(c.wb1);
    rt_pin(c.wb5);
    if constexpr (G::C5X) rt_pin(c.wb5x);
    // forward : out = alpha5*(conv5 + b5) + beta1*x + beta2*x_rrdb               (rrdbnet_arch.py:44, :68)
    // backward: d x = gathered dgrad (alpha5 = 1, conv5's scale is folded into the packed weights)
    //                 + beta1*d_out + beta2*d_out_rrdb
    // (alpha5 multiplies the bias too: the accumulator starts at b5 and is scaled as a whole)
    // What bounds conv5 (r03, tools/rdbt_check probe): 15.5 k of the block's 59 k ticks with the hand-over, 7.7 k with it switched
    // off (-DRT_X_NOSYNC) - the MFMA waves wait for slabs half of the time.  Not for want of buffering: four more ring stages in
    // the rows of x that conv5 never reads (frame rows 0..3 / 14..17, dead once conv4 is done: eight slabs in flight) gave 14.0 k;
    // not LDS bandwidth either: splitting K between the two waves of a SIMD so that each X fragment feeds both N-tiles (1.5 KB
    // of reads per MFMA instead of 2) gave 14.7 k.  All 256 workgroups reach conv5 together and each streams its 221 KB of
    // conv5 weights in ~14 k ticks: 8.5 TB/s out of the L2s.  Neither variant is kept.
    f32x16 acc5;
    rt_acc_init<BWD>(acc5, bias_lds + 128 + nt5 * 32, g);
    // extended stages E1..E4: growth conv K (the instantiation for the number of tiles this wave owns; none: it only hands the
    // slabs back) with conv5's chunk K-1 in front of the growth chunk that needs the newest slice
    static_for<1, 5>([&](auto K_c) {
        constexpr int K = decltype(K_c)::value;
        c.prank = prk[K - 1];
        const int nt = ntl[K - 1];
        RtGrow<G::nmt(K)> sa;       // this wave owns nmt(K) tiles ...
        RtGrow<1> sb;               // ... or one (only one of the two states is ever live)
        int ea[G::nmt(K)], eb[1];
#pragma unroll
        for (int m = 0; m < G::nmt(K); ++m) ea[m] = rt_map_entry<TW, K>(lds0, tl[K - 1][m], i);
        eb[0] = ea[0];
        constexpr bool TWO = G::nmt(K) == 2;
        constexpr int JA = K == 1 ? 2 : K;      // growth chunks in front of the stage's conv5 chunk (conv1: both, then its epilogue)
#if !RT_INTERLEAVE5
        if (TWO && nt == 2) { rt_grow_begin<TW, K, G::nmt(K), BWD>(c, sa, ea); rt_grow_run<TW, K, G::nmt(K), BWD, 0, K + 1>(c, sa); rt_grow_end<TW, K, G::nmt(K), BWD>(c, sa); }
        else if (nt >= 1) { rt_grow_begin<TW, K, 1, BWD>(c, sb, eb); rt_grow_run<TW, K, 1, BWD, 0, K + 1>(c, sb); rt_grow_end<TW, K, 1, BWD>(c, sb); }
        (void)JA;
        if constexpr (false) {
#else
        if (TWO && nt == 2) { rt_grow_begin<TW, K, G::nmt(K), BWD>(c, sa, ea); rt_grow_run<TW, K, G::nmt(K), BWD, 0, JA>(c, sa); if constexpr (K == 1) rt_grow_end<TW, K, G::nmt(K), BWD>(c, sa); }
        else if (nt >= 1) { rt_grow_begin<TW, K, 1, BWD>(c, sb, eb); rt_grow_run<TW, K, 1, BWD, 0, JA>(c, sb); if constexpr (K == 1) rt_grow_end<TW, K, 1, BWD>(c, sb); }
        rt_c5_run<TW, BWD, K - 1, K>(c, acc5, e5);
        if constexpr (K > 1) {
#endif
            if (TWO && nt == 2) { rt_grow_run<TW, K, G::nmt(K), BWD, K, K + 1>(c, sa); rt_grow_end<TW, K, G::nmt(K), BWD>(c, sa); }
            else if (nt >= 1) { rt_grow_run<TW, K, 1, BWD, K, K + 1>(c, sb); rt_grow_end<TW, K, 1, BWD>(c, sb); }
        }
        TPROBE(K + 1);
    });
    // ================= E5: conv5 chunks 4 (slice 3) and 5 (slice 4) ======
    c.prank = prk[4];
    {
        const RtPix px = rt_pix(e5, true, n, ty0, tx0, H, W);       // always a core pixel
        // residual r2 (x_rrdb / d out_rrdb): this lane's pixel, 16 channels = four 8-byte loads, issued now and consumed in
        // the epilogue
        const __bf16* __restrict__ r2p = reinterpret_cast<const __bf16*>(d.r2.p);
        rt_u32x2 r2v[4];
        if (r2p) {
            const __bf16* rp = r2p + px.gpix * d.r2.cs + d.r2.coff + nt5 * 32 + 4 * g;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) r2v[q4] = *reinterpret_cast<const rt_u32x2*>(rp + 8 * q4);
        }
#if RT_INTERLEAVE5
        rt_c5_run<TW, BWD, 4, 6>(c, acc5, e5);
#else
        rt_c5_run<TW, BWD, 0, 6>(c, acc5, e5);
#endif
        TPROBE(6);
        if (lane == 0) rt_inc(ctl + RT_CTL_SLICE + 5);
        rt_wait_ge(ctl + RT_CTL_SLICE + 5, G::npart(5));   // every wave is finished with the ring: it becomes the output transpose slabs
        {
            const int f0 = rt_f(px.Y, px.X);
            const int xrow = 64 * (px.Y * G::pitch(0) + px.X);
            const unsigned slabw = lds0 + G::RING + wave * (32 * G::TROW);     // [32 px][32 co] bf16, TROW-byte rows
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const rt_u32x2 xw = rt_lds_read<rt_u32x2>(lds0 + G::base(0) + nt5 * G::PLANE + xrow + ((q4 ^ f0) << 4) + 8 * g);
                const float xv[4] = {rt_bf_lo(xw[0]), rt_bf_hi(xw[0]), rt_bf_lo(xw[1]), rt_bf_hi(xw[1])};
                float rv[4] = {0.f, 0.f, 0.f, 0.f};
                if (r2p) { rv[0] = rt_bf_lo(r2v[q4][0]); rv[1] = rt_bf_hi(r2v[q4][0]); rv[2] = rt_bf_lo(r2v[q4][1]); rv[3] = rt_bf_hi(r2v[q4][1]); }
                rt_bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = d.alpha5 * acc5[4 * q4 + e] + d.beta1 * xv[e];
                    if (r2p) v += d.beta2 * rv[e];
                    o[e] = (__bf16)v;
                }
                rt_lds_write<rt_bf16x4>(slabw + i * G::TROW + 16 * q4 + 8 * g, o);
            }
            // the wave's 32 px x 64 B tile -> 16-byte vectors (row = the lane of the M-tile that owns the pixel)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int v = hh * 64 + lane;
                const int row = v >> 2, part = v & 3;
                const int es = rt_map_entry<TW, 5>(lds0, mt5, (hh * 64 + lane) >> 2);   // the pixel whose row of the slab this lane sends
                const int oy = ty0 - 5 + (es & 31), ox = tx0 - 5 + ((es >> 5) & 31);
                const u32x4 val = rt_lds_read<u32x4>(slabw + row * G::TROW + part * 16);
                if (oy < H && ox < W) {
                    __bf16* dst = reinterpret_cast<__bf16*>(d.out.p) + ((size_t)(n * H + oy) * W + ox) * d.out.cs +
                                  d.out.coff + nt5 * 32 + part * 8;
                    *reinterpret_cast<u32x4*>(dst) = val;
                }
            }
        }
    }
    TPROBE(7);
    TRACE(10, 0);
#ifdef SSR_PROBE
    if (threadIdx.x == 0) { g_probe[blockIdx.x * 16 + 13] = c.slice_ticks; g_probe[blockIdx.x * 16 + 14] = c.wait_ticks; g_probe[blockIdx.x * 16 + 15] = c.wait_n; }
#endif
}

template <int TW>
int rdbt_launch_tw(const ssr_rdb_desc& d, void* stream, bool bwd) {
    using G = RtGeo<TW>;
    static bool attr_done[2] = {false, false};
    const void* kern = bwd ? reinterpret_cast<const void*>(rdbt_kernel<TW, true>) : reinterpret_cast<const void*>(rdbt_kernel<TW, false>);
    if (!attr_done[bwd]) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[bwd] = true;
    }
    const int tiles = d.N * ((d.H + 7) / 8) * ((d.W + TW - 1) / TW);
    if (bwd) hipLaunchKernelGGL((rdbt_kernel<TW, true>), dim3(tiles), dim3(RT_NTHREADS), G::LDS, reinterpret_cast<hipStream_t>(stream), d);
    else hipLaunchKernelGGL((rdbt_kernel<TW, false>), dim3(tiles), dim3(RT_NTHREADS), G::LDS, reinterpret_cast<hipStream_t>(stream), d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

}  // namespace

// descriptor already validated by the caller (csrc/rdb_fwd.hip: rdb_launch)
int rdbt_launch(const ssr_rdb_desc& d, void* stream, bool bwd, int tw) {
    (void)tw;                                      // 8 x 16 tiles only
    return rdbt_launch_tw<16>(d, stream, bwd);
}
