// Split-bf16 (SSR_F32X3) 3x3 stride-1 convolution for the small-spatial generator body, round 6: 4 x 1 REGISTER TILING.  One MFMA
// wave owns ALL four 32-pixel tiles of the workgroup's 8 x 16 pixels for a quarter of the contraction (K split over four waves, one
// per SIMD); weight fragments go from L2 straight into registers - they never touch the LDS - and only the patch is staged.
//
// Why (round-5 counters of conv_x3q.hip, the kernel this replaces on the 32 x 32 body: 0.18 / 0.27 of 833 TF): with one pixel tile
// per wave a (chunk, tap) step reads 2 pixel + 2 weight fragments from LDS for 3 MFMAs, the weight rows of a chunk (23 KB) are
// copied global -> registers -> LDS by the MFMA waves first, and the ring holds 150 KB.  A stage moved 288 KB of fragment reads +
// 75 KB of stores for 216 MFMAs, and the twelve waves met at a flag every 27 MFMAs.  Here, per (chunk, tap):
//   * a wave issues 2 (NT = 2: 4) buffer loads of pre-split weight fragments ([chunk16][tap][CoutPad][16 hi | 16 lo] rows: lane
//     (i, g) takes 16 bytes at row i, byte 16 g, and 32 bytes further) - every weight byte is fetched ONCE per workgroup, as before,
//     but by the wave that multiplies with it, several steps ahead (WR steps deep in registers);
//   * 8 ds_read_b128 of pixel fragments feed 12 (24) MFMAs: 0.67 KB of LDS reads per MFMA instead of 1.33, no weight stores;
//   * the LDS holds the patch only: 14.4 KB per 16-channel chunk, an EIGHT-stage ring (the dense block's 4 .. 12 chunks: the
//     producers run a whole conv1 / conv2 / conv3 ahead and never wait for ring space there);
//   * hand-over: producer wave p counts the chunks it has stored in pdone[p], MFMA wave w the chunks it has finished reading in
//     cdone[w]; a wave polls only when the count it saw last no longer covers the chunk it needs (one ds_read_b128 = all four words).
// Work split: the 3 x nchunks (chunk, tap row) items go round-robin to the four MFMA waves (item g -> wave g & 3), three taps each;
// every wave walks the chunks at the same pace.  The four K-quarters of a tile are summed in wave order 0 .. 3 through LDS (fixed
// order: an image's bytes do not depend on the batch or on timing), wave m finishes pixel tile m; NT = 2: the four producer waves
// finish the second 32-channel tile, so eight waves run the epilogue.
// Epilogue: the dense block's three forms (bias + LeakyReLU; alpha (acc + bias) + beta1 r1 + beta2 r2; LeakyReLU-backward mask)
// are straight-line code in the TRANSPOSED domain - the finished 32 x 32 tile goes through a 4-KB LDS slab, then every lane owns
// 4 channels of 4 pixels: residual / mask loads and the result stores are whole 128-byte pixel rows by 16-byte buffer
// instructions, absent operands read zeros through an out-of-range offset (DESIGN.md lessons 2, 52).  Everything else takes
// conv_epilogue<float, 14> (conv_epilogue.h), the contract's generic form.
//
// Same tile, packed weights and per-acc product order (a_lo w_hi + a_hi w_lo + a_hi w_hi) as conv_x3q_kernel; results agree with it
// up to the fp32 summation order of the K-quarters.
//
// Replaces nn.Conv2d 3x3 forward and dgrad of the dense blocks, /root/reference/ssr/archs/rrdbnet_arch.py:26-30,37-44 (and any
// other stride-1 3x3 layer on a small grid) in the fp32x3 arithmetic mode.
#include "conv_x3r_core.h"

#ifdef SSR_PROBE   // tools/x3r_probe.hip: s_memtime stamps of one thread per role, 16 slots per workgroup
#define RPROBE(cond, k) do { if (cond) g_probe[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RPROBE(cond, k)
#endif

namespace {

// EX: the EXACT fp32 arithmetic mode (SSR_F32) on the same data path - the staged rows hold the 16 fp32 channels of a pixel as they are
// (64 bytes: the same pitch and the same two 16-byte fragment reads per lane as [16 hi | 16 lo]), the packed fp32 weight rows likewise,
// and a (tile, tap, chunk) step is eight v_mfma_f32_32x32x2_f32 (lane (i, g) multiplies channels 4 g + s and 8 + 4 g + s, s = 0 .. 3)
// instead of three bf16 MFMAs.  At 64 cycles per MFMA the step is 5.3 x longer than the split step while every byte moves as before:
// the mode that every gate of the reference holds in (outputs AND gradients) runs MFMA-bound here.
// AM: the arithmetic - 0 split-bf16 (SSR_F32X3), 1 = EX above (SSR_F32), 2 = the fp16-split FORWARD arithmetic (SSR_F32H, include/ssr_hip.h):
// the split-bf16 path with fp16 pieces, v_mfma_f32_32x32x16_f16 and the K-quarter sum multiplied by 2^-SSR_F32H_WSHIFT.
template <int NTW, int NU, int EP, int TH = 8, int AM = 0>
__global__ __launch_bounds__((XrT<NTW, NU, TH>::NTHR)) void conv_x3r_kernel(const ssr_conv_desc d) {
    constexpr bool EX = AM == 1, H = AM == 2;
    using T = XrT<NTW, NU, TH>;
    constexpr int NT = NTW, NTT = T::NTT, NMF = T::NMF, BN = T::BN, WR = T::WR, MT = T::MT;
    constexpr int G_NPIX = T::NPIX, G_SUB = T::SUB, G_PV = T::PV, G_NPV = T::NPV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* ctl = reinterpret_cast<int*>(smem + T::CTL);          // [0..3] pdone, [4 .. 4 + NMF) cdone
    const int ctl_addr = (int)(size_t)(__attribute__((address_space(3))) char*)(smem + T::CTL);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (d.Gw + 15) / 16, tiles_y = (d.Gh + TH - 1) / TH;
    int b = blockIdx.x;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty_i * TH, gx0 = tx_i * 16;
    const int co0 = blockIdx.y * BN;
    const int Cin = d.Cin, Cin2 = d.Cin2;
    const int nchunks = (Cin + Cin2 + 15) / 16;
    const int cout_pad = d.CoutPad;
    const int tapstride = cout_pad * 64, wchunk = 9 * tapstride;      // packed bytes per tap / per 16-channel chunk
    RPROBE(tid == 0, 0);
    if (tid < 4 + NMF) ctl[tid] = 0;
    __syncthreads();                                           // the only barrier before the reduce

    f32x16 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    if (wave >= NMF) {
        // =============================== producer waves: the patch ===============================
        const int pw = wave - NMF;
        const int pt = tid - 64 * NMF;                         // 0..255
#ifdef XR_PRIO        // (probe switch: the producers win the issue arbitration against the MFMA wave of their SIMD; 1 = until the first chunk is stored)
        __builtin_amdgcn_s_setprio(3);
#endif
        const int part = pt & 3, p4 = pt >> 2;
        // vector q of a thread = slot pt + 256 q: patch pixel p4 + 64 q, 16-byte part pt & 3
        int ppix[G_NPV];                                      // global pixel index of the patch vectors (-1: zeros)
#pragma unroll
        for (int q = 0; q < G_NPV; ++q) {
            const int pix = p4 + 64 * q;
            const int py = pix / XR_PW, px = pix - py * XR_PW;
            const int ly = gy0 + py - d.pad_y, lx = gx0 + px - d.pad_x;
            const bool okp = pix < G_NPIX && ly >= 0 && ly < d.Hi && lx >= 0 && lx < d.Wi;
            ppix[q] = okp ? (n * d.Hi + ly) * d.Wi + lx : -1;
        }
        const int plo0 = p4 * XR_ROWB + part * 8;              // hi half of the row; the lo half lies 32 bytes further
        const long xbytes = (long)d.N * d.Hi * d.Wi * 4;
        const void* xp = d.x.p;
        const void* x2p = d.x2.p ? d.x2.p : d.x.p;
        const int x_cs = d.x.cs, x_coff = d.x.coff, x2_cs = d.x2.p ? d.x2.cs : d.x.cs, x2_coff = d.x2.p ? d.x2.coff : d.x.coff;
        u32x4 rq[XR_PQ][G_NPV];
        auto load_chunk = [&](int c, auto jc) {                // chunk c -> register set j; past the end: zeros, no memory access
            constexpr int j = decltype(jc)::value;
            const bool live = c < nchunks;
            const int c0 = c * 16;
            const bool in_x = c0 < Cin;                        // a chunk lies in ONE of the two views (dispatcher: Cin % 16 == 0 with x2)
            const int cb = in_x ? c0 : c0 - Cin, clim = live ? (in_x ? Cin : Cin2) : 0;
            const int cs = in_x ? x_cs : x2_cs, coff = in_x ? x_coff : x2_coff;
            const __amdgpu_buffer_rsrc_t rs = xr_rsrc(in_x ? xp : x2p, xbytes * cs);
            const int k = cb + part * 4;
#pragma unroll
            for (int q = 0; q < G_NPV; ++q) {
                const int off = (ppix[q] * cs + coff + k) * 4;             // computed unconditionally, selected below: no branch around a load
                const bool okl = (k < clim) & (ppix[q] >= 0);
#ifdef XR_X_NOXLOAD   // (probe switch: the producers publish without loading)
                rq[j][q] = u32x4{(unsigned)off, (unsigned)okl, 0u, 0u};
#else
                rq[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, okl ? off : XR_OOB, 0, 0);
#endif
            }
        };
        // (sched_barrier: hipcc otherwise REORDERS the independent loads of the sets - the last set's first - and the wait in front of
        //  set 0's LDS store becomes vmcnt(0): the queue drains at every chunk; DESIGN.md lesson 51)
        static_for<0, XR_PQ>([&](auto jc) { __builtin_amdgcn_sched_barrier(0); load_chunk(decltype(jc)::value, jc); __builtin_amdgcn_sched_barrier(0); });
        RPROBE(pt == 0, 8);
        int cfree = XR_NS;                                     // chunks below this index have a free ring place
        for (int c0 = 0; c0 < nchunks; c0 += XR_PQ) {
            static_for<0, XR_PQ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int c = c0 + j;
                if (c < nchunks) {
                    if (c >= cfree) {
                        // the place is free once every MFMA wave has finished chunk c - NS
                        int spin = 0;
                        for (; spin < XR_SPIN_MAX; ++spin) {
                            cfree = xr_minN<NMF>(ctl_addr + 16) + XR_NS;
                            if (c < cfree) break;
                            __builtin_amdgcn_s_sleep(2);
                        }
                        if (spin == XR_SPIN_MAX) __builtin_trap();         // a protocol bug must be loud, not wrong activations
                    }
                    char* base = smem + (c % XR_NS) * G_SUB;
#pragma unroll
                    for (int q = 0; q < G_NPV; ++q) {
#ifdef XR_X_NOSTORE   // (tools/x3r_probe.hip switch, LDS bank-conflict attribution: the producers publish without storing)
                        asm volatile("" :: "v"(rq[j][q]));
                        continue;
#endif
                        if constexpr (EX) {                                 // exact mode: the four fp32 channels as they are
                            if (q < G_NPV - 1 || pt < G_PV - (G_NPV - 1) * 256)
                                *reinterpret_cast<u32x4*>(base + p4 * XR_ROWB + part * 16 + q * 64 * XR_ROWB) = rq[j][q];
                        } else {
                            uint2 hi, lo;
                            split_f32x4<H>(rq[j][q], hi, lo);
                            if (q < G_NPV - 1 || pt < G_PV - (G_NPV - 1) * 256) {
                                *reinterpret_cast<uint2*>(base + plo0 + q * 64 * XR_ROWB) = hi;
                                *reinterpret_cast<uint2*>(base + plo0 + q * 64 * XR_ROWB + 32) = lo;
                            }
                        }
                    }
                    // LDS operations of a wave execute in order: the count follows the data
                    asm volatile("" ::: "memory");
                    if (lane == 0) *(xr_lds_int)(uintptr_t)(ctl_addr + 4 * pw) = c + 1;
                    asm volatile("" ::: "memory");
#if defined(XR_PRIO) && XR_PRIO == 1
                    if (c == 0) __builtin_amdgcn_s_setprio(0);
#endif
                }
                // refill unconditionally (past the end: out-of-range offsets) so that the number of loads in flight is the same on every
                // path and the compiler's vmcnt bookkeeping keeps the queue PQ chunks deep
                __builtin_amdgcn_sched_barrier(0);
                load_chunk(c + XR_PQ, jc);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        RPROBE(pt == 0, 9);
    } else {
        // =============================== MFMA waves: all four pixel tiles, a quarter of K ===============================
        const int w = wave & 3, ug = wave >> 2;                             // K quarter, channel-tile group
        const int i = lane & 31, gq = lane >> 5;
        const int a_lane = ((i >> 4) * XR_PW + epi_col<XR_ROT>(i)) * XR_ROWB + gq * 16;     // lane (i, g): channels 8 g .. 8 g + 7 of pixel slot i of tile 0
        const __amdgpu_buffer_rsrc_t rsw = xr_rsrc(d.w, (long)nchunks * wchunk);
        const int w_lane = (co0 + 32 * NTW * ug + i) * 64 + gq * 16;
        const int nitems = 3 * nchunks;
        const int nj = nitems > w ? (nitems - w + 3) / 4 : 0;              // this wave's items g = w, w + 4, ...
        const int glast = w + 4 * (nj - 1);
        if (nj == 0) {
            if (lane == 0) *(xr_lds_int)(uintptr_t)(ctl_addr + 16 + 4 * wave) = 0x3fffffff;
        } else {
            bf16x8 wf[WR][NT][2];                                          // weight fragments of WR (chunk, tap) steps: [hi | lo]
            constexpr int TPS = T::TPS, SPT = T::SPT, NSUB = 3 * SPT, NSETS = T::NSETS, PD = NSETS - 1;
            bf16x8 af[NSETS][TPS][2];                                      // pixel fragments of NSETS sub-steps (TPS tiles each): [hi | lo]
            auto load_w = [&](int g_, auto kxc, auto sc) {                 // step (item g, tap kx) -> register set s
                constexpr int kx = decltype(kxc)::value, s = decltype(sc)::value;
                const int gg = g_ < glast ? g_ : glast;                    // past the end: the last item again (never used)
                const int c = gg / 3, ky = gg - 3 * c;
                const int off = c * wchunk + (ky * 3 + kx) * tapstride + w_lane;
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    wf[s][u][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, off + u * 2048, 0, 0));
                    wf[s][u][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, off + u * 2048 + 32, 0, 0));
                }
            };
            // fragment reads of sub-step q (tap q / SPT, tiles TPS (q % SPT) ..) -> register set s: the lo halves (H = 1: what the first MFMA
            // group of the sub-step multiplies) or the hi halves (H = 0) of all its tiles
            auto issue_a = [&](int base, auto qc, auto sc, auto hc) {
                constexpr int q = decltype(qc)::value, kx = q / SPT, part = q % SPT, s = decltype(sc)::value, H = decltype(hc)::value;
#pragma unroll
                for (int t2 = 0; t2 < TPS; ++t2)
                    af[s][t2][H] = *reinterpret_cast<const bf16x8*>(smem + base + kx * XR_ROWB + (part * TPS + t2) * XR_TILEB + 32 * H);
            };
            int avail = 0;                                                 // chunks known to be in the ring
            auto ensure = [&](int c) {
                if (avail > c) return;
                int spin = 0;
                for (; spin < XR_SPIN_MAX; ++spin) {
                    avail = xr_min4(ctl_addr);
                    if (avail > c) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (spin == XR_SPIN_MAX) __builtin_trap();
            };
            auto base_of = [&](int g_) { const int c = g_ / 3, ky = g_ - 3 * c; return (c % XR_NS) * G_SUB + ky * XR_PW * XR_ROWB + a_lane; };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            // weights of the first WR steps (they depend on nobody), then the first chunk
            // (XR_WPRE: only the first WPRE steps before the wait for the first chunk, the others behind it - the producers' first patch
            //  loads then queue behind WPRE x 2 KB of weight requests per MFMA wave instead of WR x 2 KB; tools/x3r_probe)
#ifndef XR_WPRE
#define XR_WPRE 99
#endif
            constexpr int WPRE = XR_WPRE < WR ? XR_WPRE : WR;
            static_for<0, WPRE>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                __builtin_amdgcn_sched_barrier(0);
                load_w(w + 4 * (s / 3), std::integral_constant<int, s % 3>{}, sc);
                __builtin_amdgcn_sched_barrier(0);
            });
            RPROBE(tid == 0, 1);
            int g = w, c_cur = g / 3;
            int base_cur = base_of(g);
            ensure(c_cur);
            static_for<WPRE, WR>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                __builtin_amdgcn_sched_barrier(0);
                load_w(w + 4 * (s / 3), std::integral_constant<int, s % 3>{}, sc);
                __builtin_amdgcn_sched_barrier(0);
            });
            RPROBE(tid == 0, 2);
            static_for<0, PD>([&](auto qc) {
                issue_a(base_cur, qc, std::integral_constant<int, decltype(qc)::value % NSETS>{}, I1{});
                issue_a(base_cur, qc, std::integral_constant<int, decltype(qc)::value % NSETS>{}, I0{});
            });
            int gn = g + 4 < glast ? g + 4 : glast;                        // the next item (at the end: this one again, never used)
            int c_next = gn / 3, base_next = base_of(gn);
            int gnn = gn, c_nn = c_next, base_nn = base_next;
            // one item = three taps = NSUB sub-steps of three MFMA groups (a_lo w_hi | a_hi w_lo | a_hi w_hi over the sub-step's TPS NT = 4
            // accumulators: conv_x3q's order per accumulator).  One wave per SIMD hides about five other instructions per MFMA
            // (MI355X_MICROARCH.md), and nothing hides what sits between the last MFMA of one item and the first of the next: the
            // weight refill, the fragment reads and the next item's bookkeeping are spread BETWEEN the groups (tools/x3r_probe: an item
            // without any memory instruction took 1.5 k ticks for 36 MFMAs with the bookkeeping at its head).
            // JP = parity of the item (NT = 1: the weight ring is two items deep and the fragment sets alternate from item to item)
            auto body = [&](int j, auto jpc) {
                constexpr int JP = decltype(jpc)::value;
                static_for<0, NSUB>([&](auto qc) {
                    constexpr int q = decltype(qc)::value, kx = q / SPT, part = q % SPT, qn = q + PD;
                    constexpr int ws = (JP * 3 + kx) % WR, as = (JP * NSUB + q) % NSETS, asn = (JP * NSUB + qn) % NSETS;
                    auto group = [&](auto pc) {
                        constexpr int P = decltype(pc)::value;             // 0: a_lo w_hi, 1: a_hi w_lo, 2: a_hi w_hi
                        if constexpr (EX) {
                            // exact mode: group 0 = channels 8 .. 15 (the fragments read first), groups 1 / 2 = channels 0 .. 7 in two halves
                            constexpr int H = P == 0 ? 1 : 0, S0 = P == 2 ? 2 : 0, S1 = P == 1 ? 2 : 4;
#pragma unroll
                            for (int sidx = S0; sidx < S1; ++sidx)
#pragma unroll
                                for (int t2 = 0; t2 < TPS; ++t2)
#pragma unroll
                                    for (int u = 0; u < NT; ++u) {
                                        const f32x4 av = __builtin_bit_cast(f32x4, af[as][t2][H]), bv = __builtin_bit_cast(f32x4, wf[ws][u][H]);
                                        acc[part * TPS + t2][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sidx], bv[sidx], acc[part * TPS + t2][u], 0, 0, 0);
                                    }
                        } else {
#pragma unroll
                            for (int t2 = 0; t2 < TPS; ++t2)
#pragma unroll
                                for (int u = 0; u < NT; ++u)
                                    acc[part * TPS + t2][u] = split_mfma<H>(af[as][t2][P == 0 ? 1 : 0], wf[ws][u][P == 1 ? 1 : 0], acc[part * TPS + t2][u]);
                        }
                    };
                    auto reads = [&](auto hc) {
#ifndef XR_X_NOA      // (tools/x3r_probe.hip switch: the fragment reads of the prologue are reused)
                        if constexpr (qn < NSUB) issue_a(base_cur, std::integral_constant<int, qn>{}, std::integral_constant<int, asn>{}, hc);
                        else issue_a(base_next, std::integral_constant<int, qn - NSUB>{}, std::integral_constant<int, asn>{}, hc);
#endif
                    };
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (qn == NSUB) {
                        // every fragment read of this item has been issued: the chunks below the next item's are finished
                        if (c_next > c_cur) {
                            asm volatile("" ::: "memory");
                            if (lane == 0) *(xr_lds_int)(uintptr_t)(ctl_addr + 16 + 4 * wave) = c_next;
                            asm volatile("" ::: "memory");
                        }
                        ensure(c_next);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    group(std::integral_constant<int, 0>{});
                    __builtin_amdgcn_sched_barrier(0);
                    reads(I1{});
                    if constexpr (part == 0) {
                        // the registers of the PREVIOUS step are free (all its MFMAs are issued): the step WR - 1 further on
#ifndef XR_X_NOW      // (probe switch: the weight fragments of the prologue are reused)
                        constexpr int wsp = (JP * 3 + kx + WR - 1) % WR, ioff = (kx + WR - 1) / 3, tap = (kx + WR - 1) % 3;
                        // (the very first step reloads step WR - 1 into its own registers: no branch around a load, lesson 32)
                        load_w(g + 4 * ioff, std::integral_constant<int, tap>{}, std::integral_constant<int, wsp>{});
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    group(std::integral_constant<int, 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    reads(I0{});
                    if constexpr (q == 0) {                                 // the item after the next one, under this sub-step's MFMAs
                        gnn = g + 8 < glast ? g + 8 : glast;
                        c_nn = gnn / 3;
                        base_nn = base_of(gnn);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    group(std::integral_constant<int, 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
                g += 4;
                c_cur = c_next; base_cur = base_next;
                gn = gnn; c_next = c_nn; base_next = base_nn;
#ifdef SSR_PROBE
                if (j < 5) RPROBE(tid == 0, 3 + j);
#endif
            };
            if constexpr (WR == 6 || (NSUB % NSETS) != 0) {                // the register sets of an item depend on its parity
                for (int j = 0; j < nj; j += 2) {
                    body(j, std::integral_constant<int, 0>{});
                    if (j + 1 < nj) body(j + 1, std::integral_constant<int, 1>{});
                }
            } else {
                for (int j = 0; j < nj; ++j) body(j, std::integral_constant<int, 0>{});
            }
            asm volatile("" ::: "memory");
            if (lane == 0) *(xr_lds_int)(uintptr_t)(ctl_addr + 16 + 4 * wave) = 0x3fffffff;
        }
    }
    __syncthreads();                                           // every wave is out of the ring: it becomes the partial-sum slots
    RPROBE(tid == 0, 10);

    // ---- sum the four K-quarters in wave order through LDS: slot [source wave][pixel tile][channel tile] = 16 registers x 64 lanes,
    //      as four 16-byte vectors per lane ([q][lane][4]: contiguous, conflict-free); wave m keeps tile (m, 0) in registers ----
    const bool is_mfma = wave < NMF;
    const int me = is_mfma ? (wave & 3) : wave - NMF;          // the pixel tile this wave finishes (if it is one of the MT tiles) = its K quarter
    const int ut = is_mfma ? NTW * (wave >> 2) : 1;            // ... and the channel tile (producers: the second tile of the one group, NTW = 2)
#ifndef XR_X_NORED     // (probe switch: no partial tiles through LDS - the sums below read whatever the ring left there)
    if (is_mfma) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
                if (!(t == me && u == 0)) {
                    char* sp = smem + ((me * MT + t) * NTT + ut + u) * XR_SLOT + lane * 16;      // source K quarter = wave & 3 = me
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {acc[t][u][4 * q], acc[t][u][4 * q + 1], acc[t][u][4 * q + 2], acc[t][u][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(sp + q * 1024) = v;
                    }
                }
    }
#endif
    __syncthreads();
    if ((is_mfma || NTW == 2) && me < MT) {
        f32x16 own;
#pragma unroll
        for (int r = 0; r < 16; ++r) own[r] = 0.f;
        if (is_mfma) {                                         // runtime tile index -> static register selection
#pragma unroll
            for (int t = 0; t < MT; ++t)
                if (t == me) own = acc[t][0];
        }
        f32x16 sum;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x16 part;
            const char* sp = smem + ((s * MT + me) * NTT + ut) * XR_SLOT + lane * 16;
#ifdef XR_X_NORED
            part = own;
            asm volatile("" : "+v"(part));
#else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sp + q * 1024);
                part[4 * q] = v[0]; part[4 * q + 1] = v[1]; part[4 * q + 2] = v[2]; part[4 * q + 3] = v[3];
            }
#endif
            const bool mine = is_mfma && s == me;              // (that slot was never written: its bytes are not used)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = mine ? own[r] : part[r];
                sum[r] = s == 0 ? p : sum[r] + p;
            }
        }
        if constexpr (H) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sum[r] *= SSR_F32H_UNSCALE;       // the packed weights carry 2^SSR_F32H_WSHIFT
        }
        // the transpose slab: a slot no other wave reads (an MFMA wave's own, never written, slot of its tile; a producer's: source quarter 0's)
        char* slab = smem + (((is_mfma ? me : 0) * MT + me) * NTT + ut) * XR_SLOT;
        const int cb = co0 + 32 * ut;
#ifdef XR_X_NOEPI      // (probe switch: the sums leave through one plain store per lane instead of the transposing epilogue)
        float keep = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) keep += sum[r];
        if (keep == 123.456f) reinterpret_cast<float*>(d.y.p)[tid] = keep;
#else
        if constexpr (EP == XR_EP_GENERIC) conv_epilogue<float, XR_ROT>(d, sum, cb, n, gy0 + 2 * me, gx0, lane, slab);
        else xr_epilogue<EP>(d, sum, cb, n, gy0 + 2 * me, gx0, lane, slab);
#endif
    }
    RPROBE(tid == 0, 11);
}

template <int NTW, int NU, int EP, int TH = 8, int AM = 0>
int launch_x3r(const ssr_conv_desc& d, hipStream_t st) {
    using T = XrT<NTW, NU, TH>;
    auto kern = conv_x3r_kernel<NTW, NU, EP, TH, AM>;
    static bool attr_done[SSR_MAX_DEVICES] = {};               // the attribute is per DEVICE
    const int dev = ssr_device_ordinal();
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[dev] = true;
    }
    const int tiles = ((d.Gw + 15) / 16) * ((d.Gh + TH - 1) / TH) * d.N;
    hipLaunchKernelGGL(kern, dim3(tiles, d.CoutPad / T::BN, 1), dim3(T::NTHR), T::LDS, st, d);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

// which of the straight-line epilogues covers the descriptor (else the generic one)
int xr_pick_epilogue(const ssr_conv_desc& d) {
    static const bool generic_only = [] { const char* e = getenv("SSR_X3R_EPI"); return e && e[0] == '0'; }();
    if (generic_only) return XR_EP_GENERIC;
    const long lim = 0x7fffff00L, npx = (long)d.N * d.Ho * d.Wo * 4;
    auto vok = [&](const ssr_view& v) { return (v.cs % 4) == 0 && (v.coff % 4) == 0 && ((uintptr_t)v.p % 16) == 0 && npx * v.cs <= lim; };
    if (d.fix_list || d.y0.p || d.y1.p || d.accumulate || d.m_relu || (d.Cout % 4) != 0 || !vok(d.y)) return XR_EP_GENERIC;
    if (d.oys != 1 || d.oxs != 1 || d.oyo != 0 || d.oxo != 0 || d.Ho < d.Gh || d.Wo < d.Gw) return XR_EP_GENERIC;
    const bool r1 = d.r1.p != nullptr, r2 = d.r2.p != nullptr, m = d.m.p != nullptr;
    if (d.act == SSR_ACT_LRELU && d.alpha == 1.f && !r1 && !r2 && !m) return XR_EP_LRELU;
    if (d.act == SSR_ACT_NONE && !m && (!r1 || (d.r1_nc >= d.Cout && vok(d.r1))) && (!r2 || (d.r2_nc >= d.Cout && vok(d.r2)))) return XR_EP_LIN;
    if (d.act == SSR_ACT_NONE && d.alpha == 1.f && m && !r1 && !r2 && d.m_c0 == 0 && d.m_c1 >= d.Cout && vok(d.m)) return XR_EP_MASK;
    return XR_EP_GENERIC;
}

template <int NTW, int NU, int TH = 8, int AM = 0>
int launch_x3r_ep(const ssr_conv_desc& d, hipStream_t st) {
    switch (xr_pick_epilogue(d)) {
        case XR_EP_LRELU: return launch_x3r<NTW, NU, XR_EP_LRELU, TH, AM>(d, st);
        case XR_EP_LIN: return launch_x3r<NTW, NU, XR_EP_LIN, TH, AM>(d, st);
        case XR_EP_MASK: return launch_x3r<NTW, NU, XR_EP_MASK, TH, AM>(d, st);
        default: return launch_x3r<NTW, NU, XR_EP_GENERIC, TH, AM>(d, st);
    }
}

// half-height tiles (4 x 16 pixels, <1, 1> form only): when the 8 x 16 tiling of a 32-channel layer would leave more than 3/8 of the
// CUs without a workgroup (per-GPU batch 16 on the 32 x 32 body: 128 tiles).  SSR_X3_REGTILE_TH=8 | 4 forces one
int xr_tile_height(const ssr_conv_desc& d, int form) {
    static const int forced = [] { const char* e = getenv("SSR_X3_REGTILE_TH"); return e ? atoi(e) : 0; }();
    if (form != 0) return 8;
    if (forced == 4 || forced == 8) return forced;
    const long tiles8 = (long)((d.Gw + 15) / 16) * ((d.Gh + 7) / 8) * d.N * (d.CoutPad / 32);
    return (tiles8 <= 160 && d.Gh > 4) ? 4 : 8;
}

// the form of the 64-channel layers: 1 = four MFMA waves with two channel tiles each (default); 2 = eight MFMA waves, one channel tile each
// (SSR_X3_REGTILE_NT2=8: 21.6 against 22.4 us per launch alone - the K-quarter sum + epilogue falls from 5.7 k to 3.7 k ticks - but 27.38
// against 27.24 ms per step, call r06j: twelve-wave workgroups share the chip worse with the second chain and the discriminator stream);
// 0 = as two 32-channel workgroups (SSR_X3_REGTILE_NT2=0)
int xr_wide_form(const ssr_conv_desc& d) {
    static const int f = [] { const char* e = getenv("SSR_X3_REGTILE_NT2"); return !e ? 1 : e[0] == '0' ? 0 : e[0] == '8' ? 2 : 1; }();
    if ((d.CoutPad % 64) != 0) return 0;
    // a launch that leaves half the chip idle (per-GPU batch 16 on the 32 x 32 body: 128 tiles) runs its 64-channel layers as two
    // 32-channel workgroups per tile: twice the workgroups, the same products in the same order (an image's bytes do not depend on
    // the form: every form sums the four K quarters of a (pixel tile, channel tile) in wave order)
    static const bool split_off = [] { const char* e = getenv("SSR_X3_REGTILE_SPLIT"); return e && e[0] == '0'; }();
    const long tiles = (long)((d.Gw + 15) / 16) * ((d.Gh + 7) / 8) * d.N;
    if (!split_off && tiles * (d.CoutPad / 64) <= 160) return 0;
    // the generic epilogue does not fit the 168 registers of a twelve-wave workgroup (77 spilled): those (rare) layers keep the four-wave form
    return (f == 2 && xr_pick_epilogue(d) == XR_EP_GENERIC) ? 1 : f;
}

}  // namespace

// channel tiles per workgroup, MFMA waves and straight-line-epilogue index of the instantiation ssr_conv_x3r_try launches (the rocprofv3
// symbol is conv_x3r_kernel<NTW, NU, EP>)
void ssr_conv_x3r_instance(const ssr_conv_desc& d, int* ntw, int* nu, int* ep) {
    int f = xr_wide_form(d);
    if ((d.dtype == SSR_F32 || d.dtype == SSR_F32H) && f == 2) f = 1;
    *ntw = f == 1 ? 2 : 1;
    *nu = f == 2 ? 2 : 1;
    *ep = xr_pick_epilogue(d);
}
int ssr_conv_x3r_tile_height(const ssr_conv_desc& d) { return xr_tile_height(d, xr_wide_form(d)); }

bool ssr_conv_x3r_shape_ok(const ssr_conv_desc& d) {
    static const bool ex_off = [] { const char* e = getenv("SSR_F32_REGTILE"); return e && e[0] == '0'; }();
    if (!(d.dtype == SSR_F32X3 || d.dtype == SSR_F32H || (d.dtype == SSR_F32 && !ex_off)) || d.fix_list) return false;
    if (d.dtype == SSR_F32 && (d.act == SSR_ACT_RELU || d.m_relu)) return false;      // (the VGG19 layers keep the pipelined kernel: conv2d_impl)
    if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad_y == 1 && d.pad_x == 1 && d.up == 1 && !d.s2d)) return false;
    if (d.Gh != d.Hi || d.Gw != d.Wi || (d.CoutPad % 32) != 0) return false;
    if (d.x2.p && (d.Cin % 16) != 0) return false;            // a 16-channel chunk comes from ONE view
    const long lim = 0x7fffff00L, npx = (long)d.N * d.Hi * d.Wi * 4;
    if (npx * d.x.cs > lim || (d.x2.p && npx * d.x2.cs > lim)) return false;
    if ((long)((d.Cin + d.Cin2 + 15) / 16) * 9 * d.CoutPad * 64 > lim) return false;
    return true;
}

bool ssr_conv_x3r_qualifies(const ssr_conv_desc& d) {
    static const bool off = [] { const char* e = getenv("SSR_X3_REGTILE"); return e && e[0] == '0'; }();
    if (off || !ssr_conv_x3r_shape_ok(d)) return false;
    // small grids (the 32 x 32 body at any batch; conv_first / conv_body): the big-tile kernel takes the others
    return d.Gh <= 64 && d.Gw <= 64 && d.Cin + d.Cin2 >= 16;
}

bool ssr_conv_x3r_try(const ssr_conv_desc& d, hipStream_t st, int* rc, bool force) {
    if (force ? !ssr_conv_x3r_shape_ok(d) : !ssr_conv_x3r_qualifies(d)) return false;
    const int f = xr_wide_form(d);
    if (d.dtype == SSR_F32) {                                  // exact fp32 arithmetic: four-wave forms only
        if (f != 0) *rc = launch_x3r_ep<2, 1, 8, 1>(d, st);
        else *rc = xr_tile_height(d, f) == 4 ? launch_x3r_ep<1, 1, 4, 1>(d, st) : launch_x3r_ep<1, 1, 8, 1>(d, st);
        return true;
    }
    if (d.dtype == SSR_F32H) {                                 // fp16-split forward arithmetic: four-wave forms only
        if (f != 0) *rc = launch_x3r_ep<2, 1, 8, 2>(d, st);
        else *rc = xr_tile_height(d, f) == 4 ? launch_x3r_ep<1, 1, 4, 2>(d, st) : launch_x3r_ep<1, 1, 8, 2>(d, st);
        return true;
    }
    if (f == 2) *rc = launch_x3r_ep<1, 2>(d, st);
    else if (f == 1) *rc = launch_x3r_ep<2, 1>(d, st);
    else *rc = xr_tile_height(d, f) == 4 ? launch_x3r_ep<1, 1, 4>(d, st) : launch_x3r_ep<1, 1, 8>(d, st);
    return true;
}
