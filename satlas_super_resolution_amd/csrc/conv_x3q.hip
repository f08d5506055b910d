// Split-bf16 (SSR_F32X3) 3x3 stride-1 convolution for the small-spatial generator body: twelve waves feed an LDS ring, eight of them
// run the matrix cores, LDS flag hand-over (no barrier in the loop).
//
// Why (round 5, tools/x3_probe.hip on the pipelined kernels of conv.hip, B = 32, 32 x 32 images, 32 output channels): a body
// convolution took 15.5 us for 2.9 us of MFMA time.  Per 32-channel stage the eight symmetric waves spent 2.7 k ticks in the MFMA
// phase (54 MFMAs per SIMD = 1.7 k), 1.2 k splitting the staged fp32 patch and writing the stage to LDS, 0.7 k in the barrier - and
// neither a deeper load pipeline nor riding the store inside the MFMA stream changed the sum: a wave's VALU / LDS-store work does
// not overlap its own MFMAs, and the operand reads of the second pixel row hit the banks of the first (2-way conflicts).  A first
// cut with four producer waves doing ALL the staging was no faster either (tools/x3q_probe.hip: 2.2 k ticks per chunk in the
// producers): a wave streams only ~4-6 B/clk from L2 however many loads it keeps in flight, and a chunk is 30 KB.  So:
//   * 8 MFMA waves (pixel tile w & 3 = two rows of 16 pixels, tap half w >> 2; NT = 2: both 32-channel output tiles per wave): operand
//     reads one (chunk, tap) pair ahead, MFMAs (a_lo w_hi + a_hi w_lo + a_hi w_hi into one fp32 accumulator, conv_x3_kernel's order) -
//     and, once per chunk, their share of the WEIGHT rows of a later chunk: 3 (5) buffer loads per lane, kept QT chunks deep in
//     registers, copied to the ring as they are (the packed weights arrive pre-split);
//   * 4 producer waves stage the PATCH: 3 buffer loads per lane and chunk, PQ chunks in flight, split (hi = bf16(x),
//     lo = bf16(x - hi)) and written as 80-byte rows [16 hi | 16 lo | pad];
//   * all twelve waves stream (12 x ~4 B/clk), every staging load is unconditional (a lane that must read zeros uses an offset
//     beyond num_records) and the sets are pinned in issue order, so hipcc's waits are vmcnt(loads of the later sets), not vmcnt(0);
//   * hand-over by LDS words: ready[stage] counts the 12 waves that have stored their part of the chunk, done[wave] the chunks an MFMA
//     wave has finished reading.  An MFMA wave writes the weights of chunk c + NS/2 after its MFMAs of chunk c: by then every wave has
//     finished the chunk that used that stage (they all published chunk c's weights after it), so only the producers poll `done`.
//     All polls are inline asm (a compiler-visible LDS read drains the wave's loads first: vmcnt(0));
//   * the second pixel row of a tile is rotated by 14 columns (conv_big.hip's map): every 16-lane read group touches LDS rows that
//     are distinct mod 16 under every tap shift - conflict-free ds_read_b128; conv_epilogue<float, 14> undoes it.
// A workgroup = 8 x 16 pixels x 32 NT output channels, ring of 2 stages (NT = 1: two 16-channel chunks per stage); the same tile, packed weights
// ([chunk16][tap][CoutPad][16 hi | 16 lo]), epilogue contract and - up to fp32 summation order - results as conv_x3_kernel.
//
// Replaces nn.Conv2d 3x3 forward and dgrad of the dense blocks, /root/reference/ssr/archs/rrdbnet_arch.py:26-30,37-44 (and any other
// stride-1 3x3 layer on a small grid) in the fp32x3 arithmetic mode.
#include "conv_epilogue.h"
#include <cstdlib>

#ifdef SSR_PROBE   // tools/x3q_probe.hip: s_memtime stamps of one thread per role, 16 slots per workgroup
#define QPROBE(cond, k) do { if (cond) g_probe[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QPROBE(cond, k)
#endif

namespace {

constexpr int XQ_NCONS = 8, XQ_NPROD = 4, XQ_NTHR = 64 * (XQ_NCONS + XQ_NPROD);
constexpr int XQ_PQ = 2;                                       // patch stages a producer keeps in flight in registers (3: no gain)
constexpr int XQ_ROWB = 80, XQ_PH = 10, XQ_PW = 18, XQ_NPIX = XQ_PH * XQ_PW;       // 8 x 16 tile + halo
constexpr int XQ_PV = XQ_NPIX * 4;                             // 16-byte vectors of a chunk's patch: 720
constexpr int XQ_NPV = (XQ_PV + 255) / 256;                    // per producer thread: 3
constexpr int XQ_ROT = 14;                                     // rotation of a tile's second pixel row (32 - PW)
constexpr int XQ_OOB = 0x7ffffff0;
constexpr int XQ_SPIN_MAX = 1 << 22;                           // polls of a flag before a wave TRAPS (~0.3 s; a hand-over takes ~1 us):
                                                               // a protocol bug must fail the launch loudly - neither hang the GPU nor fall through
// A ring STAGE holds CPS 16-channel chunks (NT = 1: two - with one, a wave had 13 MFMAs per hand-over and the per-stage work of an MFMA
// wave - poll, release, its weight rows - cost as much as the MFMAs: 1.9 k ticks per chunk for 0.9 k of MFMAs, tools/x3q_probe.hip).
template <int NT> struct XqT {
    static constexpr int BN = 32 * NT;
    static constexpr int CPS = NT == 1 ? 2 : 1;                // 16-channel chunks per stage
    static constexpr int NS = 2;                               // ring stages
    static constexpr int DW = NS / 2;                          // an MFMA wave stores the weights of stage s + DW behind stage s
    static constexpr int QT = DW + 1;                          // weight stages an MFMA wave holds in registers (6 / 5 vectors each; a third set: no
                                                               // gain at NT = 1, 45 spilled registers at NT = 2 - a wave of twelve may use 168)
    static constexpr int PF = NT == 1 ? 2 : 1;                 // operand reads run PF (chunk, tap) pairs ahead of their MFMAs
    static constexpr int WROWS = 9 * BN;
    static constexpr int SUB = (XQ_NPIX + WROWS) * XQ_ROWB;    // one chunk: 37,440 / 60,480 B
    static constexpr int STAGE = CPS * SUB;                    // 74,880 / 60,480 B
    static constexpr int CTL = NS * STAGE;                     // control words behind the ring: ready[4] | done[8]
    static constexpr int LDS = CTL + 256;
    static constexpr int WV = WROWS * 4;                       // 16-byte vectors of a chunk's weights: 1152 / 2304
    static constexpr int NWV = (WV + 511) / 512;               // per MFMA-wave thread and chunk: 3 / 5
    static constexpr int WTAIL = WV - (NWV - 1) * 512;         // threads that own a last weight vector: 128 / 256
    static constexpr int TAPSTEP = 128 / BN;                   // taps between a thread's consecutive weight rows (128 rows): 4 / 2
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(4 * 16 * 64 * 4 * NT + 8 * EPI_STAGE_BYTES <= CTL, "reduce scratch + epilogue slabs fit in the ring");
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t xq_rsrc(const void* p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffff00L ? 0x7fffff00L : bytes), 0x00020000);
}
__device__ __forceinline__ void xq_split4(const u32x4& v, uint2& hi, uint2& lo) {
    const f32x4 f = __builtin_bit_cast(f32x4, v);
    bf16x4 h, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h[k] = (__bf16)f[k];
        l[k] = (__bf16)(f[k] - (float)h[k]);
    }
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}
typedef __attribute__((address_space(3))) int* xq_lds_int;

template <int NT>
__global__ __launch_bounds__(XQ_NTHR) void conv_x3q_kernel(const ssr_conv_desc d) {
    using T = XqT<NT>;
    constexpr int NS = T::NS, STAGE = T::STAGE, SUB = T::SUB, CPS = T::CPS, BN = T::BN, DW = T::DW, QT = T::QT, NWV = T::NWV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* ctl = reinterpret_cast<int*>(smem + T::CTL);          // [0..3] ready, [8..15] done
    const int ctl_addr = (int)(size_t)(__attribute__((address_space(3))) char*)(smem + T::CTL);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (d.Gw + 15) / 16, tiles_y = (d.Gh + 7) / 8;
    int b = blockIdx.x;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty_i * 8, gx0 = tx_i * 16;
    const int co0 = blockIdx.y * BN;
    const int Cin = d.Cin, Cin2 = d.Cin2;
    const int nchunks = (Cin + Cin2 + 15) / 16, nst = (nchunks + CPS - 1) / CPS;      // 16-channel chunks, ring stages
    const int cout_pad = d.CoutPad;
    const int wchunk = 9 * cout_pad * 64;                      // packed bytes per 16-channel chunk
    QPROBE(tid == 0, 0);
    if (tid < 16) ctl[tid] = 0;
    __syncthreads();                                           // the only barrier before the epilogue

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int wm = wave & 3;
    int kh = 2;                                                // producers: neither k-half (they only pass the epilogue's barriers)

    if (wave >= XQ_NCONS) {
        // =============================== producer waves: the patch ===============================
        const int pt = tid - 64 * XQ_NCONS;                    // 0..255
        const int part = pt & 3, p4 = pt >> 2;
        // vector q of a thread = slot pt + 256 q: patch pixel p4 + 64 q, 16-byte part pt & 3
        int ppix[XQ_NPV];                                      // global pixel index of the patch vectors (-1: zeros)
#pragma unroll
        for (int q = 0; q < XQ_NPV; ++q) {
            const int pix = p4 + 64 * q;
            const int py = pix / XQ_PW, px = pix - py * XQ_PW;
            const int ly = gy0 + py - d.pad_y, lx = gx0 + px - d.pad_x;
            const bool ok = pix < XQ_NPIX && ly >= 0 && ly < d.Hi && lx >= 0 && lx < d.Wi;
            ppix[q] = ok ? (n * d.Hi + ly) * d.Wi + lx : -1;
        }
        const int plo0 = p4 * XQ_ROWB + part * 8;              // hi half of the row; the lo half lies 32 bytes further
        const long xbytes = (long)d.N * d.Hi * d.Wi * 4;
        const void* xp = d.x.p;
        const void* x2p = d.x2.p ? d.x2.p : d.x.p;
        const int x_cs = d.x.cs, x_coff = d.x.coff, x2_cs = d.x2.p ? d.x2.cs : d.x.cs, x2_coff = d.x2.p ? d.x2.coff : d.x.coff;
        u32x4 rq[XQ_PQ][CPS][XQ_NPV];
        auto load_stage = [&](int s_, auto jc) {               // stage s -> register set j; past the end: zeros, no memory access
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int cc = 0; cc < CPS; ++cc) {
                const int c = s_ * CPS + cc;
                const bool live = c < nchunks;
                const int c0 = c * 16;
                const bool in_x = c0 < Cin;                    // a chunk lies in ONE of the two views (dispatcher: Cin % 16 == 0 with x2)
                const int cb = in_x ? c0 : c0 - Cin, clim = live ? (in_x ? Cin : Cin2) : 0;
                const int cs = in_x ? x_cs : x2_cs, coff = in_x ? x_coff : x2_coff;
                const __amdgpu_buffer_rsrc_t rs = xq_rsrc(in_x ? xp : x2p, xbytes * cs);
                const int k = cb + part * 4;
#pragma unroll
                for (int q = 0; q < XQ_NPV; ++q) {
                    const int off = (ppix[q] * cs + coff + k) * 4;         // computed unconditionally, selected below: no branch around a load
                    const bool ok = (k < clim) & (ppix[q] >= 0);
                    rq[j][cc][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : XQ_OOB, 0, 0);
                }
            }
        };
        // (sched_barrier: hipcc otherwise REORDERS the independent loads of the stages - the last set's first - and the wait in front
        //  of set 0's LDS store becomes vmcnt(0): the queue drains at every stage)
        static_for<0, XQ_PQ>([&](auto jc) { __builtin_amdgcn_sched_barrier(0); load_stage(decltype(jc)::value, jc); __builtin_amdgcn_sched_barrier(0); });
        QPROBE(pt == 0, 8);
        for (int s0 = 0; s0 < nst; s0 += XQ_PQ) {
            static_for<0, XQ_PQ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int s_ = s0 + j;
                if (s_ < nst) {
                    const int st = s_ % NS;
                    if (s_ >= NS) {
                        // the stage is free once every MFMA wave has finished stage s - NS.  Inline asm: a compiler-visible LDS read
                        // here would make hipcc drain the refill loads first; "=&v": the outputs must not share a register with the
                        // address (the first read may return before the second one issues)
                        const int need = s_ - NS + 1;
                        int spin = 0;
                        for (; spin < XQ_SPIN_MAX; ++spin) {
                            u32x4 dn, dm;
                            asm volatile("ds_read_b128 %0, %2 offset:32\n\tds_read_b128 %1, %2 offset:48\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dn), "=&v"(dm) : "v"(ctl_addr) : "memory");
                            const int dmin = (int)min(min(min(dn[0], dn[1]), min(dn[2], dn[3])), min(min(dm[0], dm[1]), min(dm[2], dm[3])));
                            if (__builtin_amdgcn_readfirstlane(dmin) >= need) break;
                            __builtin_amdgcn_s_sleep(2);
                        }
                        if (spin == XQ_SPIN_MAX) __builtin_trap();         // a stalled hand-over must be loud (a failed launch), not wrong activations
                    }
                    char* base = smem + st * STAGE;
#pragma unroll
                    for (int cc = 0; cc < CPS; ++cc)
#pragma unroll
                        for (int q = 0; q < XQ_NPV; ++q) {
                            uint2 hi, lo;
                            xq_split4(rq[j][cc][q], hi, lo);
                            if (q < XQ_NPV - 1 || pt < XQ_PV - (XQ_NPV - 1) * 256) {
                                *reinterpret_cast<uint2*>(base + cc * SUB + plo0 + q * 64 * XQ_ROWB) = hi;
                                *reinterpret_cast<uint2*>(base + cc * SUB + plo0 + q * 64 * XQ_ROWB + 32) = lo;
                            }
                        }
                    // LDS operations of a wave execute in order: the flag follows the data
                    if (lane == 0) __hip_atomic_fetch_add(ctl + st, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                // refill unconditionally (past the end: out-of-range offsets) so that the number of loads in flight is the same on
                // every path and the compiler's vmcnt bookkeeping keeps the queue PQ stages deep
                __builtin_amdgcn_sched_barrier(0);
                load_stage(s_ + XQ_PQ, jc);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        QPROBE(pt == 0, 9);
    } else {
        // =============================== MFMA waves (+ the weight rows) ===============================
        kh = wave >> 2;
        const int i = lane & 31, g = lane >> 5;
        const int ty = 2 * wm + (i >> 4), tx = epi_col<XQ_ROT>(i);
        const int a_off = (ty * XQ_PW + tx) * XQ_ROWB + g * 16;          // lane (i, g): channels g*8 .. g*8+7 of pixel slot i
        const int b_off = (XQ_NPIX + i) * XQ_ROWB + g * 16;
        // weight vector q of a thread = slot tid + 512 q: packed row (tid >> 2) + 128 q = tap TAPSTEP q + (row0 / BN), co row0 % BN
        const __amdgpu_buffer_rsrc_t rsw = xq_rsrc(d.w, (long)nchunks * wchunk);
        const int row0 = tid >> 2, wpart = tid & 3;
        const int wgo0 = ((row0 / BN) * cout_pad + co0 + (row0 % BN)) * 64 + wpart * 16, wstep = T::TAPSTEP * cout_pad * 64;
        const int wlo0 = (XQ_NPIX + row0) * XQ_ROWB + wpart * 16;
        u32x4 wq[QT][CPS][NWV];
        auto load_w = [&](int s_, auto jc) {                   // weights of stage s -> register set j; past the end: zeros, no memory access
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int cc = 0; cc < CPS; ++cc) {
                const int c = s_ * CPS + cc;
                const bool live = c < nchunks;
#pragma unroll
                for (int q = 0; q < NWV; ++q) {
                    const int off = wgo0 + q * wstep + c * wchunk;
                    const bool ok = live & (q < NWV - 1 || tid < T::WTAIL);
                    wq[j][cc][q] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ok ? off : XQ_OOB, 0, 0);
                }
            }
        };
        auto store_w = [&](int s_, auto jc) {                  // register set j -> ring stage of stage s, then publish
            constexpr int j = decltype(jc)::value;
            char* base = smem + (s_ % NS) * STAGE;
#pragma unroll
            for (int cc = 0; cc < CPS; ++cc)
#pragma unroll
                for (int q = 0; q < NWV; ++q)
                    if (q < NWV - 1 || tid < T::WTAIL) *reinterpret_cast<u32x4*>(base + cc * SUB + wlo0 + q * 128 * XQ_ROWB) = wq[j][cc][q];
            if (lane == 0) __hip_atomic_fetch_add(ctl + (s_ % NS), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        static_for<0, QT>([&](auto jc) { __builtin_amdgcn_sched_barrier(0); load_w(decltype(jc)::value, jc); __builtin_amdgcn_sched_barrier(0); });
        static_for<0, DW>([&](auto jc) {                       // stages 0 .. DW - 1 go to the ring before the loop
            if (decltype(jc)::value < nst) store_w(decltype(jc)::value, jc);
            __builtin_amdgcn_sched_barrier(0);
            load_w(decltype(jc)::value + QT, jc);
            __builtin_amdgcn_sched_barrier(0);
        });
        // one stage: this wave's (chunk, tap) pairs - the pairs of parity P of the CPS * 9 of the stage - reads one pair ahead of the MFMAs
        auto contract = [&](const char* sb, auto pc) {
            constexpr int P = decltype(pc)::value, NITEM = (CPS * 9 + 1 - P) / 2;
            const char* ab = sb + a_off;
            const char* bb = sb + b_off;
            constexpr int PF = T::PF, NB = PF + 1;
            bf16x8 fa[NB][2], fb[NB][NT][2];
            auto issue = [&](auto kc) {
                constexpr int k = decltype(kc)::value, it = 2 * k + P, cc = it / 9, tap = it % 9, ky = tap / 3, kx = tap % 3;
                fa[k % NB][0] = *reinterpret_cast<const bf16x8*>(ab + cc * SUB + (ky * XQ_PW + kx) * XQ_ROWB);
                fa[k % NB][1] = *reinterpret_cast<const bf16x8*>(ab + cc * SUB + (ky * XQ_PW + kx) * XQ_ROWB + 32);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    fb[k % NB][t][0] = *reinterpret_cast<const bf16x8*>(bb + cc * SUB + (tap * BN + t * 32) * XQ_ROWB);
                    fb[k % NB][t][1] = *reinterpret_cast<const bf16x8*>(bb + cc * SUB + (tap * BN + t * 32) * XQ_ROWB + 32);
                }
            };
            static_for<0, (PF < NITEM ? PF : NITEM)>([&](auto kc) { issue(kc); });
            static_for<0, NITEM>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (k + PF < NITEM) issue(std::integral_constant<int, k + PF>{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k % NB][1], fb[k % NB][t][0], acc[t], 0, 0, 0);   // a_lo * w_hi
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k % NB][0], fb[k % NB][t][1], acc[t], 0, 0, 0);   // a_hi * w_lo
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k % NB][0], fb[k % NB][t][0], acc[t], 0, 0, 0);   // a_hi * w_hi
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        };
        QPROBE(tid == 0, 1);
        for (int s0 = 0; s0 < nst; s0 += QT) {
            static_for<0, QT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int s_ = s0 + j;
                if (s_ < nst) {
                    const int st = s_ % NS, target = (XQ_NCONS + XQ_NPROD) * (s_ / NS + 1);
                    int spin = 0;
                    for (; spin < XQ_SPIN_MAX; ++spin) {                      // all twelve waves have stored their part of the stage
                        int rdy;
                        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(rdy) : "v"(ctl_addr + 4 * st) : "memory");
                        if (__builtin_amdgcn_readfirstlane(rdy) >= target) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (spin == XQ_SPIN_MAX) __builtin_trap();
                    const char* sb = smem + st * STAGE;
                    // the two k-halves take alternate (chunk, tap) pairs; with an odd number of pairs per stage (CPS = 1) the halves swap
                    // parities from stage to stage
                    if ((((CPS & 1) ? s_ : 0) + kh) & 1) contract(sb, std::integral_constant<int, 1>{});
                    else contract(sb, std::integral_constant<int, 0>{});
                    // every operand read of the stage has returned (its MFMAs were issued): release it
                    asm volatile("" ::: "memory");
                    if (lane == 0) *(xq_lds_int)(uintptr_t)(ctl_addr + 32 + 4 * wave) = s_ + 1;
                    asm volatile("" ::: "memory");
#ifdef SSR_PROBE
                    if (s_ < 6) QPROBE(tid == 0, 2 + s_);
#endif
                    // the weights of stage s + DW: its ring place held stage s + DW - NS <= s - DW, which every wave has finished (they all
                    // published stage s's weights behind it) - no poll
                    if (s_ + DW < nst) store_w(s_ + DW, std::integral_constant<int, (j + DW) % QT>{});
                }
                __builtin_amdgcn_sched_barrier(0);
                load_w(s_ + DW + QT, std::integral_constant<int, (j + DW) % QT>{});    // refill unconditionally (see the producers)
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }
    __syncthreads();                                           // every wave is out of the ring: it becomes the reduce scratch + epilogue slabs
    QPROBE(tid == 0, 10);

    // ---- combine the two k-halves through LDS, then the fused epilogue (the contract of conv.hip); NT = 2: each half finishes one
    //      of the two output tiles ----
    float* red = reinterpret_cast<float*>(smem);               // [NT][4][16][64]
    char* slabs = smem + NT * 4 * 16 * 64 * sizeof(float);
    float* mine = red + (wm * 16) * 64 + lane;
    if (NT == 1) {
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r * 64] = acc[0][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] += mine[r * 64];
            conv_epilogue<float, XQ_ROT>(d, acc[0], co0, n, gy0 + 2 * wm, gx0, lane, slabs + (size_t)wm * EPI_STAGE_BYTES);
        }
    } else {
        float* other = mine + 4 * 16 * 64;
        if (kh == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) other[r * 64] = acc[NT - 1][r];
        } else if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r * 64] = acc[0][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] += mine[r * 64];
            conv_epilogue<float, XQ_ROT>(d, acc[0], co0, n, gy0 + 2 * wm, gx0, lane, slabs + (size_t)wave * EPI_STAGE_BYTES);
        } else if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[NT - 1][r] += other[r * 64];
            conv_epilogue<float, XQ_ROT>(d, acc[NT - 1], co0 + 32, n, gy0 + 2 * wm, gx0, lane, slabs + (size_t)wave * EPI_STAGE_BYTES);
        }
    }
    QPROBE(tid == 0, 11);
}

template <int NT>
int launch_x3q(const ssr_conv_desc& d, hipStream_t st) {
    auto kern = conv_x3q_kernel<NT>;
    static bool attr_done[SSR_MAX_DEVICES] = {};      // the attribute is per DEVICE
    const int attr_dev = ssr_device_ordinal();
    if (!attr_done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, XqT<NT>::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[attr_dev] = true;
    }
    const int tiles = ((d.Gw + 15) / 16) * ((d.Gh + 7) / 8) * d.N;
    hipLaunchKernelGGL(kern, dim3(tiles, d.CoutPad / (32 * NT), 1), dim3(XQ_NTHR), XqT<NT>::LDS, st, d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

}  // namespace

bool ssr_conv_x3q_shape_ok(const ssr_conv_desc& d) {
    if (d.dtype != SSR_F32X3 || d.fix_list) return false;
    if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad_y == 1 && d.pad_x == 1 && d.up == 1 && !d.s2d)) return false;
    if (d.Gh != d.Hi || d.Gw != d.Wi || (d.CoutPad % 32) != 0) return false;
    if (d.x2.p && (d.Cin % 16) != 0) return false;            // a 16-channel chunk comes from ONE view
    const long lim = 0x7fffff00L, npx = (long)d.N * d.Hi * d.Wi * 4;
    if (npx * d.x.cs > lim || (d.x2.p && npx * d.x2.cs > lim)) return false;
    if ((long)((d.Cin + d.Cin2 + 15) / 16) * 9 * d.CoutPad * 64 > lim) return false;
    return true;
}

bool ssr_conv_x3q_qualifies(const ssr_conv_desc& d) {
    static const bool off = [] { const char* e = getenv("SSR_X3_RING"); return e && e[0] == '0'; }();
    if (off || !ssr_conv_x3q_shape_ok(d)) return false;
    // small grids (the 32 x 32 body at any batch; conv_first / conv_body): the big-tile kernel takes the others
    return d.Gh <= 64 && d.Gw <= 64 && d.Cin + d.Cin2 >= 16;
}

bool ssr_conv_x3q_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force) {
    if (force ? !ssr_conv_x3q_shape_ok(d) : !ssr_conv_x3q_qualifies(d)) return false;
    static const bool nt2_off = [] { const char* e = getenv("SSR_X3_RING_NT2"); return e && e[0] == '0'; }();
    *rc = ((d.CoutPad % 64) == 0 && !nt2_off) ? launch_x3q<2>(d, st) : launch_x3q<1>(d, st);
    return true;
}
