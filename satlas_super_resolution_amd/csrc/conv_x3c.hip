// Split-bf16 (SSR_F32X3) dense-block CHAIN: up to four dependent 3x3 convolutions of one ResidualDenseBlock - conv1..conv4 of the forward
// (/root/reference/ssr/archs/rrdbnet_arch.py:37-41) or slices 4..1 of its gather-form backward - in ONE persistent launch, the
// workgroups of an image handing their results to each other through memory with per-tile flags (round 6).
//
// Why: with one launch per convolution (csrc/conv_x3r.hip) a 32-channel body layer spends 11.7 us for 3.5 us of MFMA time:
// ~1.3 us of launch gap, 4.9 k ticks until the first chunk is staged (wave start, descriptor, first loads), 3.2 k of K-sum + epilogue,
// and the chip idles at every boundary.  Here a workgroup (image n, tile t: ticketed, see below) walks the chain:
//   * conv k+1 reads the channel prefix [x | x1 .. xk] of the block buffer: only its LAST two chunks (xk, written by conv k of THIS
//     launch) depend on the neighbours; the producer waves stage the older chunks while the MFMA waves still sum and store conv k, and
//     the MFMA waves start conv k+1 on them at once.  (Backward: the newest gradient slice comes FIRST in the gathered K, so a conv's
//     chunks are walked last to first - `rev`.)
//   * results leave with write-through (sc1) stores; each MFMA wave drains its stores and sets its byte of the tile's flag word; a
//     producer wave polls the flag words of the 3 x 3 tile neighbourhood (one relaxed agent-scope load per lane) before it requests
//     a chunk another workgroup wrote, and reads patch data with sc1 loads (MI355X guide, Guideline 16 form R1: sc1 both sides).
//   * no state to zero per launch: flags carry the launch's EPOCH byte (state[2] mod 255 + 1), every flag byte is rewritten by
//     every launch, and the last workgroup to finish resets the ticket counter and advances the epoch.
//   * deadlock freedom: workgroups draw tickets at start (ticket -> image ticket / tiles, tile ticket % tiles), so the tiles a
//     workgroup waits for are running or next in line for the first free CU; every poll is bounded and TRAPS on time-out.
// The register-tiled MFMA stream, the patch ring and the straight-line epilogues are conv_x3r's (conv_x3r_core.h): ring of six
// stages + 64 KB of K-quarter sum slots beside it (the producers fill the ring for conv k+1 while conv k is summed).
// Forward chains add the same products in the same order as four conv_x3r launches (bit-identical results); backward chains walk
// K in the opposite direction (same products, other fp32 summation order).
#include "conv_x3r_core.h"
#include <cstdio>

#ifdef SSR_PROBE   // tools/x3c_probe.hip: s_memtime stamps of MFMA wave 0, 32 slots per workgroup
#define CPROBE(cond, k) do { if (cond) g_probe[blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CPROBE(cond, k)
#endif

namespace {

constexpr int XC_MAXN = 4;                                     // convolutions per chain
constexpr int XC_MAXCH = 16;                                   // 16-channel chunks per convolution
constexpr int XC_NS = 6;                                       // ring stages
constexpr int XC_RING = XC_NS * XR_SUB;                        // 86,400 B
constexpr int XC_RED = 16 * XR_SLOT;                           // [source wave][pixel tile] partial tiles: 65,536 B
constexpr int XC_CTL = XC_RING + XC_RED;                       // control words: pdone[4] | cdone[4] | rsync | ticket | epoch byte
constexpr int XC_LDS = XC_CTL + 256;
static_assert(XC_LDS <= 160 * 1024, "LDS budget");
constexpr int XC_STATE_HDR = 64;                               // bytes of [ticket, done, epoch, ...] in front of the flag words
constexpr int XC_POLL_MAX = 1 << 20;                           // polls of a flag word before a wave traps (~1 s)

struct ssr_chain_args {
    ssr_conv_desc d[XC_MAXN];
    int32_t n;                                                 // convolutions of the chain
    int32_t tiles_x, tiles_y;
    int32_t nch[XC_MAXN];                                      // 16-channel chunks of each
    int8_t dep[XC_MAXN][XC_MAXCH];                             // per conv and chunk (memory order): the conv of this chain that writes it, or -1
    int8_t rev[XC_MAXN];                                       // 1: chunks are walked last to first (the dependent ones come first in memory)
    uint32_t* state;                                           // [0] ticket [1] done [2] epoch; + XC_STATE_HDR: flags[N * tiles][4 convs][4 wave bytes]
};

typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned char gu8;

template <int EP>
__global__ __launch_bounds__(XR_NTHR) void conv_x3c_kernel(const ssr_chain_args a) {
    using T = XrT<1, 1>;
    constexpr int WR = T::WR, TPS = T::TPS, SPT = 4 / TPS, NSUB = 3 * SPT, NSETS = T::NSETS, PD = NSETS - 1;
    static_assert(WR == 6 && NSUB * 2 % NSETS == 0, "item-parity unrolling");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* ctl = reinterpret_cast<int*>(smem + XC_CTL);          // [0..3] pdone, [4..7] cdone, [8] rsync, [9] ticket, [10] epoch byte
    const int ctl_addr = (int)(size_t)(__attribute__((address_space(3))) char*)(smem + XC_CTL);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 9) ctl[tid] = 0;
    if (tid == 64) {                                           // tiles of an image start together, in ticket order
        ctl[9] = (int)atomicAdd(a.state, 1u);
        ctl[10] = (int)(__hip_atomic_load(a.state + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % 255u) + 1;
    }
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(ctl[9]);
    const unsigned eb = (unsigned)__builtin_amdgcn_readfirstlane(ctl[10]);
    const int tiles_x = a.tiles_x, tiles = a.tiles_x * a.tiles_y;
    const int n = ticket / tiles, tix = ticket - n * tiles;
    const int ty_i = tix / tiles_x, tx_i = tix - ty_i * tiles_x;
    const int gy0 = ty_i * 8, gx0 = tx_i * 16;
    const int nconv = a.n;
    const int flag_off = XC_STATE_HDR + (n * tiles + tix) * 16;        // this tile's flag line
    CPROBE(tid == 0, 0);

    if (wave >= XR_NMFMA) {
        // =============================== producer waves: the patch chunks of every conv of the chain, one stream ===============================
        const int pw = wave - XR_NMFMA;
        const int pt = tid - 64 * XR_NMFMA;
        const int part = pt & 3, p4 = pt >> 2;
        const ssr_conv_desc& d0 = a.d[0];                      // the chain's convs share the grid
        int ppix[XR_NPV];
#pragma unroll
        for (int q = 0; q < XR_NPV; ++q) {
            const int pix = p4 + 64 * q;
            const int py = pix / XR_PW, px = pix - py * XR_PW;
            const int ly = gy0 + py - 1, lx = gx0 + px - 1;
            const bool okp = pix < XR_NPIX && ly >= 0 && ly < d0.Hi && lx >= 0 && lx < d0.Wi;
            ppix[q] = okp ? (n * d0.Hi + ly) * d0.Wi + lx : -1;
        }
        const int plo0 = p4 * XR_ROWB + part * 8;
        const long xbytes = (long)d0.N * d0.Hi * d0.Wi * 4;
        const __amdgpu_buffer_rsrc_t rs_state = xr_rsrc(a.state, (long)XC_STATE_HDR + (long)d0.N * tiles * 16);
        int total = 0;
        for (int ci = 0; ci < nconv; ++ci) total += a.nch[ci];
        int seen = -1;                                         // convs of this chain every neighbour has published
        // the flag words of the 3 x 3 tile neighbourhood, one per lane (absent neighbours: this tile)
        int nb_off;
        {
            const int l = lane < 9 ? lane : 4;
            int nty = ty_i + l / 3 - 1, ntx = tx_i + l % 3 - 1;
            if (nty < 0 || nty >= a.tiles_y || ntx < 0 || ntx >= tiles_x) { nty = ty_i; ntx = tx_i; }
            nb_off = XC_STATE_HDR + (n * tiles + nty * tiles_x + ntx) * 16;
        }
        auto wait_published = [&](int dep) {
            const unsigned want = eb * 0x01010101u;
            int spin = 0;
            for (; spin < XC_POLL_MAX; ++spin) {
                const unsigned v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_state, nb_off + dep * 4, 0, 16);     // sc1: served past the L1
                if (__builtin_amdgcn_ballot_w64(v != want) == 0) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (spin == XC_POLL_MAX) __builtin_trap();
            seen = dep;
        };
        u32x4 rq[XR_PQ][XR_NPV];
        // job k of the stream = (conv ci, position p in its walk); past the end: zeros, no memory access
        auto load_job = [&](int k, auto jc) {
            constexpr int j = decltype(jc)::value;
            int ci = 0, p = k;
            bool live = k < total;
            if (live) {
                while (p >= a.nch[ci]) { p -= a.nch[ci]; ++ci; }
            } else { ci = 0; p = 0; }
            const ssr_conv_desc& d = a.d[ci];
            const int nch = a.nch[ci];
            const int c = a.rev[ci] ? nch - 1 - p : p;         // chunk in memory order
            if (live) {
                const int dep = a.dep[ci][c];
                if (dep > seen) wait_published(dep);           // (uniform) another workgroup's result: wait for the neighbourhood
            }
            const int Cin = d.Cin, Cin2 = d.Cin2;
            const int c0 = c * 16;
            const bool in_x = c0 < Cin;
            const int cb = in_x ? c0 : c0 - Cin, clim = live ? (in_x ? Cin : Cin2) : 0;
            const ssr_view& vw = (in_x || !d.x2.p) ? d.x : d.x2;
            const int cs = vw.cs, coff = vw.coff;
            const __amdgpu_buffer_rsrc_t rs = xr_rsrc(vw.p, xbytes * cs);
            const int kk = cb + part * 4;
#pragma unroll
            for (int q = 0; q < XR_NPV; ++q) {
                const int off = (ppix[q] * cs + coff + kk) * 4;
                const bool okl = (kk < clim) & (ppix[q] >= 0);
                rq[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, okl ? off : XR_OOB, 0, 16);      // sc1 (patch data has no L1 reuse anyway)
            }
        };
        static_for<0, XR_PQ>([&](auto jc) { __builtin_amdgcn_sched_barrier(0); load_job(decltype(jc)::value, jc); __builtin_amdgcn_sched_barrier(0); });
        int cfree = XC_NS;
        for (int k0 = 0; k0 < total; k0 += XR_PQ) {
            static_for<0, XR_PQ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int k = k0 + j;
                if (k < total) {
                    if (k >= cfree) {
                        int spin = 0;
                        for (; spin < XR_SPIN_MAX; ++spin) {
                            cfree = xr_min4(ctl_addr + 16) + XC_NS;
                            if (k < cfree) break;
                            __builtin_amdgcn_s_sleep(2);
                        }
                        if (spin == XR_SPIN_MAX) __builtin_trap();
                    }
                    char* base = smem + (k % XC_NS) * XR_SUB;
#pragma unroll
                    for (int q = 0; q < XR_NPV; ++q) {
                        uint2 hi, lo;
                        xr_split4(rq[j][q], hi, lo);
                        if (q < XR_NPV - 1 || pt < XR_PV - (XR_NPV - 1) * 256) {
                            *reinterpret_cast<uint2*>(base + plo0 + q * 64 * XR_ROWB) = hi;
                            *reinterpret_cast<uint2*>(base + plo0 + q * 64 * XR_ROWB + 32) = lo;
                        }
                    }
                    asm volatile("" ::: "memory");
                    if (lane == 0) *(xr_lds_int)(uintptr_t)(ctl_addr + 4 * pw) = k + 1;
                    asm volatile("" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                load_job(k + XR_PQ, jc);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    } else {
        // =============================== MFMA waves: all four pixel tiles, a quarter of K, conv after conv ===============================
        const int w = wave;
        const int i = lane & 31, gq = lane >> 5;
        const int a_lane = ((i >> 4) * XR_PW + epi_col<XR_ROT>(i)) * XR_ROWB + gq * 16;
        f32x16 acc[4];
        bf16x8 wf[WR][2];
        bf16x8 af[NSETS][TPS][2];
        int avail = 0;
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
        // the four MFMA waves meet at ctl[8] (the producers run on): arrive = one more wave has passed, wait = until `target` have
        auto lds_arrive = [&]() {
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(ctl + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
        };
        auto lds_wait = [&](int target) {
            int spin = 0;
            for (; spin < XR_SPIN_MAX; ++spin) {
                int v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(ctl_addr + 32) : "memory");
                if (__builtin_amdgcn_readfirstlane(v) >= target) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (spin == XR_SPIN_MAX) __builtin_trap();
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        int Gbase = 0;                                         // ring position of this conv's first chunk
        for (int ci = 0; ci < nconv; ++ci) {
            const ssr_conv_desc& d = a.d[ci];
            const int nch = a.nch[ci], rev = a.rev[ci];
            const int tapstride = d.CoutPad * 64, wchunk = 9 * tapstride;
            const __amdgpu_buffer_rsrc_t rsw = xr_rsrc(d.w, (long)nch * wchunk);
            const int w_lane = i * 64 + gq * 16;               // 32 output channels: one channel tile
            const int nitems = 3 * nch;
            const int nj = nitems > w ? (nitems - w + 3) / 4 : 0;
            const int glast = w + 4 * (nj - 1);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            if (nj > 0) {
                auto load_w = [&](int g_, auto kxc, auto sc) {
                    constexpr int kx = decltype(kxc)::value, s = decltype(sc)::value;
                    const int gg = g_ < glast ? g_ : glast;
                    const int p = gg / 3, ky = gg - 3 * p;
                    const int c = rev ? nch - 1 - p : p;
                    const int off = c * wchunk + (ky * 3 + kx) * tapstride + w_lane;
                    wf[s][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0));
                    wf[s][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, off + 32, 0, 0));
                };
                auto issue_a = [&](int base, auto qc, auto sc, auto hc) {
                    constexpr int q = decltype(qc)::value, kx = q / SPT, part = q % SPT, s = decltype(sc)::value, H = decltype(hc)::value;
#pragma unroll
                    for (int t2 = 0; t2 < TPS; ++t2)
                        af[s][t2][H] = *reinterpret_cast<const bf16x8*>(smem + base + kx * XR_ROWB + (part * TPS + t2) * XR_TILEB + 32 * H);
                };
                auto base_of = [&](int g_) { const int p = g_ / 3, ky = g_ - 3 * p; return ((Gbase + p) % XC_NS) * XR_SUB + ky * XR_PW * XR_ROWB + a_lane; };
                static_for<0, WR>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    load_w(w + 4 * (s / 3), std::integral_constant<int, s % 3>{}, sc);
                    __builtin_amdgcn_sched_barrier(0);
                });
                int g = w, c_cur = g / 3;
                int base_cur = base_of(g);
                ensure(Gbase + c_cur);
                if (ci == 0) CPROBE(tid == 0, 1);
                static_for<0, PD>([&](auto qc) {
                    issue_a(base_cur, qc, std::integral_constant<int, decltype(qc)::value % NSETS>{}, I1{});
                    issue_a(base_cur, qc, std::integral_constant<int, decltype(qc)::value % NSETS>{}, I0{});
                });
                int gn = g + 4 < glast ? g + 4 : glast;
                int c_next = gn / 3, base_next = base_of(gn);
                int gnn = gn, c_nn = c_next, base_nn = base_next;
                auto body = [&](int j, auto jpc) {
                    constexpr int JP = decltype(jpc)::value;
                    static_for<0, NSUB>([&](auto qc) {
                        constexpr int q = decltype(qc)::value, kx = q / SPT, part = q % SPT, qn = q + PD;
                        constexpr int ws = (JP * 3 + kx) % WR, as = (JP * NSUB + q) % NSETS, asn = (JP * NSUB + qn) % NSETS;
                        auto group = [&](auto pc) {
                            constexpr int P = decltype(pc)::value;
#pragma unroll
                            for (int t2 = 0; t2 < TPS; ++t2)
                                acc[part * TPS + t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[as][t2][P == 0 ? 1 : 0], wf[ws][P == 1 ? 1 : 0], acc[part * TPS + t2], 0, 0, 0);
                        };
                        auto reads = [&](auto hc) {
                            if constexpr (qn < NSUB) issue_a(base_cur, std::integral_constant<int, qn>{}, std::integral_constant<int, asn>{}, hc);
                            else issue_a(base_next, std::integral_constant<int, qn - NSUB>{}, std::integral_constant<int, asn>{}, hc);
                        };
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (qn == NSUB) {
                            if (c_next > c_cur) {
                                asm volatile("" ::: "memory");
                                if (lane == 0) *(xr_lds_int)(uintptr_t)(ctl_addr + 16 + 4 * w) = Gbase + c_next;
                                asm volatile("" ::: "memory");
                            }
                            ensure(Gbase + c_next);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        group(std::integral_constant<int, 0>{});
                        __builtin_amdgcn_sched_barrier(0);
                        reads(I1{});
                        if constexpr (part == 0) {
                            constexpr int wsp = (JP * 3 + kx + WR - 1) % WR, ioff = (kx + WR - 1) / 3, tap = (kx + WR - 1) % 3;
                            load_w(g + 4 * ioff, std::integral_constant<int, tap>{}, std::integral_constant<int, wsp>{});
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        group(std::integral_constant<int, 1>{});
                        __builtin_amdgcn_sched_barrier(0);
                        reads(I0{});
                        if constexpr (q == 0) {
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
                };
                for (int j = 0; j < nj; j += 2) {
                    body(j, std::integral_constant<int, 0>{});
                    if (j + 1 < nj) body(j + 1, std::integral_constant<int, 1>{});
                }
            }
            // this wave has finished every chunk of the conv: the producers may reuse their ring places
            asm volatile("" ::: "memory");
            if (lane == 0) *(xr_lds_int)(uintptr_t)(ctl_addr + 16 + 4 * w) = Gbase + nch;
            asm volatile("" ::: "memory");
            Gbase += nch;
            CPROBE(tid == 0, 2 + 4 * ci);
            // ---- K-quarter sum in wave order through the slots BESIDE the ring; every wave has read the previous conv's slots
            //      (second meeting of that conv) before anybody overwrites them ----
            if (ci > 0) lds_wait(8 * ci);
            char* red = smem + XC_RING;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t != w) {
                    char* sp = red + (w * 4 + t) * XR_SLOT + lane * 16;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(sp + q * 1024) = v;
                    }
                }
            lds_arrive();
            lds_wait(8 * ci + 4);
            CPROBE(tid == 0, 3 + 4 * ci);
            f32x16 own;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t == w) own = acc[t];
            f32x16 sum;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x16 part;
                const char* sp = red + (s * 4 + w) * XR_SLOT + lane * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(sp + q * 1024);
                    part[4 * q] = v[0]; part[4 * q + 1] = v[1]; part[4 * q + 2] = v[2]; part[4 * q + 3] = v[3];
                }
                const bool mine = s == w;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = mine ? own[r] : part[r];
                    sum[r] = s == 0 ? p : sum[r] + p;
                }
            }
            lds_arrive();          // second meeting: arrive only (the slot reads above are in this wave's LDS queue ahead of it)
            xr_epilogue<EP, 16>(d, sum, 0, n, gy0 + 2 * w, gx0, lane, red + (w * 4 + w) * XR_SLOT);
            // publish: this wave's stores are complete (write-through), then its byte of the tile's flag word of conv ci
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store((gu8*)((char*)a.state + flag_off + ci * 4 + w), (unsigned char)eb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            CPROBE(tid == 0, 4 + 4 * ci);
        }
    }
    __syncthreads();
    if (tid == 0) {                                            // the last workgroup to finish: tickets from 0, next epoch
        const unsigned done = atomicAdd(a.state + 1, 1u);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(a.state + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.state + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    CPROBE(tid == 0, 31);
}

template <int EP>
int launch_x3c(const ssr_chain_args& a, int blocks, hipStream_t st) {
    auto kern = conv_x3c_kernel<EP>;
    static bool attr_done[SSR_MAX_DEVICES] = {};
    const int dev = ssr_device_ordinal();
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, XC_LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(XR_NTHR), XC_LDS, st, a);
    SSR_LAUNCH_CHECK();
    return SSR_OK;
}

bool overlap(const ssr_view& a, int na, const ssr_view& b, int nb) {
    return a.p && b.p && a.p == b.p && a.coff < b.coff + nb && b.coff < a.coff + na;
}

}  // namespace

bool ssr_conv_x3r_shape_ok(const ssr_conv_desc& d);
void ssr_conv_x3r_instance(const ssr_conv_desc& d, int* ntw, int* nu, int* ep);

extern "C" int64_t ssr_conv2d_chain_state_bytes(int32_t N, int32_t Gh, int32_t Gw) {
    return XC_STATE_HDR + (int64_t)N * ((Gw + 15) / 16) * ((Gh + 7) / 8) * 16;
}

// 1 if the n descriptors can run as one chain launch (else ssr_conv2d_chain launches them one by one)
extern "C" int ssr_conv2d_chain_ok(const ssr_conv_desc* ds, int32_t n) {
    static const bool off = [] { const char* e = getenv("SSR_X3_CHAIN"); return e && e[0] == '0'; }();
    if (off || !ds || n < 2 || n > XC_MAXN) return 0;
    int ep0 = -1;
    for (int k = 0; k < n; ++k) {
        const ssr_conv_desc& d = ds[k];
        if (d.dtype != SSR_F32X3 || !ssr_conv_x3r_shape_ok(d) || d.CoutPad != 32 || d.Cout > 32) return 0;
        if (d.N != ds[0].N || d.Hi != ds[0].Hi || d.Wi != ds[0].Wi || d.Gh != ds[0].Gh || d.Gw != ds[0].Gw) return 0;
        if ((d.Cin % 16) != 0 || (d.x2.p && (d.Cin2 % 16) != 0) || (d.Cin + d.Cin2) / 16 > XC_MAXCH || d.Cin + d.Cin2 < 16) return 0;
        int nt = 0, nu = 0, ep = 3;
        ssr_conv_x3r_instance(d, &nt, &nu, &ep);
        if (ep == XR_EP_GENERIC || (ep0 >= 0 && ep != ep0)) return 0;
        ep0 = ep;
        // no later conv of the chain may overwrite what an earlier one reads or writes (a workgroup runs ahead of its neighbours)
        for (int j = 0; j < k; ++j) {
            if (overlap(d.y, d.Cout, ds[j].y, ds[j].Cout)) return 0;
            if (overlap(d.y, d.Cout, ds[j].x, ds[j].Cin) || (ds[j].x2.p && overlap(d.y, d.Cout, ds[j].x2, ds[j].Cin2))) return 0;
        }
        if (overlap(d.y, d.Cout, d.x, d.Cin) || (d.x2.p && overlap(d.y, d.Cout, d.x2, d.Cin2))) return 0;
        if (d.r1.p || d.r2.p) return 0;                        // (residual operands written inside the chain are not tracked)
        if (d.m.p) for (int j = 0; j < n; ++j) if (overlap(ds[j].y, ds[j].Cout, d.m, d.Cout)) return 0;
    }
    const int tiles = ((ds[0].Gw + 15) / 16) * ((ds[0].Gh + 7) / 8);
    if (tiles > 64) return 0;                                  // an image's tiles must be co-resident (256 CUs)
    // every chunk's producer, and a walk direction that meets the producers in order (dependent chunks last)
    for (int k = 0; k < n; ++k) {
        const ssr_conv_desc& d = ds[k];
        const int nch = (d.Cin + d.Cin2) / 16;
        int first_dep = -1, last_free = -1, prev = -2;
        bool asc = true, desc = true;
        for (int c = 0; c < nch; ++c) {
            const bool in_x = c * 16 < d.Cin;
            ssr_view v = in_x ? d.x : d.x2;
            v.coff += in_x ? c * 16 : c * 16 - d.Cin;
            int dep = -1;
            for (int j = 0; j < k; ++j)
                if (overlap(v, 16, ds[j].y, ds[j].Cout)) {
                    if (dep >= 0) return 0;                    // a chunk written by two convs
                    // the chunk must lie inside the writer's VALID channels (padding rows are never written)
                    if (v.coff < ds[j].y.coff || v.coff + 16 > ds[j].y.coff + ds[j].Cout) return 0;
                    dep = j;
                }
            if (c > 0) {
                if (dep < prev) asc = false;
                if (dep > prev) desc = false;
            }
            prev = dep;
            (void)first_dep; (void)last_free;
        }
        if (!asc && !desc) return 0;
    }
    return 1;
}

// n (2..4) stride-1 3x3 split-bf16 convolutions over one grid, each possibly reading what earlier ones of the list write (the
// dense block's conv1..conv4, or the slices 4..1 of its gather-form backward) in ONE persistent launch; `state` =
// ssr_conv2d_chain_state_bytes(N, Gh, Gw) bytes of device memory, zeroed ONCE by the host and owned by the launches of one stream
// (consecutive launches reuse it; concurrent chains need one each).  Falls back to n ssr_conv2d launches when the list does not
// qualify (ssr_conv2d_chain_ok) or state is NULL.
extern "C" int ssr_conv2d_chain(const ssr_conv_desc* ds, int32_t n, void* state, void* stream) {
    if (!ds || n <= 0) return SSR_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!state || !ssr_conv2d_chain_ok(ds, n)) {
        for (int k = 0; k < n; ++k) {
            const int rc = ssr_conv2d(ds + k, stream);
            if (rc != SSR_OK) return rc;
        }
        return SSR_OK;
    }
    ssr_chain_args a{};
    a.n = n;
    a.tiles_x = (ds[0].Gw + 15) / 16;
    a.tiles_y = (ds[0].Gh + 7) / 8;
    a.state = reinterpret_cast<uint32_t*>(state);
    int ep = 3, nt = 0, nu = 0;
    for (int k = 0; k < n; ++k) {
        a.d[k] = ds[k];
        const ssr_conv_desc& d = ds[k];
        const int nch = (d.Cin + d.Cin2) / 16;
        a.nch[k] = nch;
        int firstd = -1;
        for (int c = 0; c < nch; ++c) {
            const bool in_x = c * 16 < d.Cin;
            ssr_view v = in_x ? d.x : d.x2;
            v.coff += in_x ? c * 16 : c * 16 - d.Cin;
            int dep = -1;
            for (int j = 0; j < k; ++j)
                if (overlap(v, 16, ds[j].y, ds[j].Cout)) dep = j;
            a.dep[k][c] = (int8_t)dep;
            if (c == 0) firstd = dep;
        }
        // dependent chunks must be walked LAST: ascending when the producers grow with the chunk index, else descending
        a.rev[k] = (int8_t)((nch > 1 && firstd > a.dep[k][nch - 1]) ? 1 : 0);
        ssr_conv_x3r_instance(d, &nt, &nu, &ep);
    }
    const int blocks = a.tiles_x * a.tiles_y * ds[0].N;
    switch (ep) {
        case XR_EP_LRELU: return launch_x3c<XR_EP_LRELU>(a, blocks, st);
        case XR_EP_LIN: return launch_x3c<XR_EP_LIN>(a, blocks, st);
        case XR_EP_MASK: return launch_x3c<XR_EP_MASK>(a, blocks, st);
        default: return SSR_EUNSUP;
    }
}
