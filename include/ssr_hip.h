/*
 * ssr_hip.h — C ABI of libssr_hip.so: the MI355X (gfx950) kernels under the ESRGAN hot path of
 * allenai/satlas-super-resolution.
 *
 * The reference has no native layer: its "kernels" are the ATen/cuDNN ops that its Python issues
 * (SURVEY.md §2.3).  Each entry point below names the reference call site(s) (file:line, relative
 * to /root/reference) whose device work it replaces.  All entry points
 *   - take raw device pointers, sizes and a hipStream_t (as void*); no torch types;
 *   - never allocate, never synchronise, are hipGraph-capturable;
 *   - return 0 on success, a negative SSR_E* code on a bad descriptor, or the positive hipError_t
 *     of a failed launch.
 *
 * Data layout: activations are NHWC ("pixel-major"), element type fp32 (SSR_F32) or bf16
 * (SSR_BF16); a tensor is described by (base pointer, channel stride `cs` = channels of the
 * underlying buffer, channel offset `coff`), so a conv can read/write a channel slice of a wider
 * buffer — this is how the dense blocks run without torch.cat (rrdbnet_arch.py:39-42).
 * Channel strides/offsets are multiples of 8 elements.  Master weights are fp32 OIHW exactly as in
 * the reference's state_dict; kernels consume packed copies produced by ssr_pack_weights.
 */
#ifndef SSR_HIP_H
#define SSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSR_F32 0
#define SSR_BF16 1
/* fp32 tensors in HBM, bf16 matrix cores on split operands (hi + lo, three MFMAs per product, fp32 accumulation): the
 * arithmetic mode that meets the 1e-3 parity gate at matrix-core speed.  Storage, packing and every non-MFMA entry point
 * treat it exactly as SSR_F32. */
#define SSR_F32X3 2
/* FORWARD convolutions only (ssr_conv2d, ssr_conv2d_batch; the forward table of ssr_pack_weights), round 6: fp32 tensors in HBM,
 * fp16 matrix cores on split operands - an activation is staged as hi = f16(x), lo = f16(x - hi); the packed weights hold
 * hi = f16(2^SSR_F32H_WSHIFT w), lo = f16(2^SSR_F32H_WSHIFT w - hi) in the SSR_F32X3 row layout ([16 hi | 16 lo], 64 bytes); a product
 * is a_lo w_hi + a_hi w_lo + a_hi w_hi (three v_mfma_f32_32x32x16_f16, fp32 accumulation) and the accumulators are multiplied by
 * 2^-SSR_F32H_WSHIFT before the epilogue.  Two 11-bit pieces carry 22 bits of an operand (bf16 pieces: 16), so a pre-activation is
 * as close to the fp32 value as another fp32 summation order is, and its LeakyReLU decision is an fp32 evaluation's - at the split-bf16
 * mode's speed.  The weight shift keeps the lo pieces of ordinary convolution weights (|w| down to 2^-17 x 2^-SHIFT) in fp16's
 * normal range; activations are staged unscaled: an element below 2^-3 keeps an ABSOLUTE error of 2^-25 (its lo piece is an fp16
 * subnormal, which gfx950's matrix cores and converters keep), and |x| must stay below 65504 (beyond: inf -> NaN outputs, loudly).
 * Gradients (1e-7 .. 1e-4 in this model) do not fit fp16's range without a per-tensor scale: backward convolutions and weight
 * gradients of the mode that uses this code (hip.F32H in the Python layer) stay SSR_F32X3. */
#define SSR_F32H 3
#define SSR_F32H_WSHIFT 10

#define SSR_OK 0
#define SSR_EINVAL (-1)   /* bad descriptor (unsupported geometry / alignment) */
#define SSR_EUNSUP (-2)   /* combination not instantiated */

#define SSR_ACT_NONE 0
#define SSR_ACT_LRELU 1   /* LeakyReLU(0.2): rrdbnet_arch.py:32, discriminator_arch.py:44-68 */
#define SSR_ACT_RELU 2    /* ReLU: the VGG19 feature extractor of the perceptual loss (ssr_esrgan_model.py:153-160) */

/* A channel-sliced NHWC view. */
typedef struct ssr_view {
    void* p;          /* base pointer of the buffer (element type = desc dtype) */
    int32_t cs;       /* channels of the underlying buffer (pixel stride, in elements) */
    int32_t coff;     /* first channel of the slice */
} ssr_view;

/*
 * Direct (im2col-free) convolution on MFMA.  One descriptor covers
 *   forward:  nn.Conv2d 3x3 s1 p1 (rrdbnet_arch.py:26-30,99-112; discriminator_arch.py:28,34-40),
 *             4x4 s2 p1 (discriminator_arch.py:30-32), with F.interpolate(x2,'nearest') folded into
 *             the input read (`up`=2; rrdbnet_arch.py:127-128: source index = floor(o/2)),
 *   dgrad:    the same kernel with flipped/transposed packed weights; the stride-2 transposed conv is
 *             run as four 2x2 output-parity classes (oys/oyo/oxs/oxo),
 * plus the fused epilogue that replaces the reference's elementwise ops:
 *   s0 = alpha * act(acc + bias)                     -> y0 (optional)
 *   s1 = s0 + beta1*r1[c<r1_nc] + beta2*r2[c<r2_nc] (+ y_old if accumulate) -> y1 (optional)
 *   s2 = s1 * lrelu'(m) for channels in [m_c0, m_c1) -> y
 * (bias+LeakyReLU: rrdbnet_arch.py:38-41; x5*0.2+x and out*0.2+x: :44,:68; feat+body_feat: :125;
 *  lrelu(...)+skip: discriminator_arch.py:53-64; LeakyReLU backward masks and grad fan-in sums for
 *  autograd's backward of all of the above.)
 */
typedef struct ssr_conv_desc {
    int32_t dtype;            /* SSR_F32 | SSR_BF16 (x, w, y*, r*, m element type) */
    /* input side */
    ssr_view x;
    int32_t N, Hi, Wi;        /* stored input dims */
    int32_t up;               /* 1, or 2 = nearest x2 upsample on read (logical dims Hi*up x Wi*up) */
    int32_t Cin;              /* channels contracted from x (multiple of 8) */
    ssr_view x2;              /* optional second input (p == NULL: none): the contraction runs over the channel */
    int32_t Cin2;             /*   concatenation [x(Cin) | x2(Cin2)] — "gather" form of the dense-block backward */
    /* packed weights, chunk-major [Cin chunk][KH*KW][CoutPad][CK]; CK = ssr_conv2d_ck(dtype, KH) input
     * channels per chunk (32 bf16 / 16 fp32; half of that for 4x4 kernels), Cin zero-padded to a multiple */
    const void* w;
    int32_t CoutPad;          /* multiple of 32 */
    const float* bias;        /* Cout floats or NULL */
    /* geometry: grid position (gy,gx), tap (ty,tx) reads logical input (gy*stride+ty-pad_y, gx*stride+tx-pad_x) */
    int32_t KH, KW, stride, pad_y, pad_x;
    int32_t Gh, Gw;           /* output grid */
    /* output side: grid (gy,gx) -> stored pixel (gy*oys+oyo, gx*oxs+oxo) of Ho x Wo buffers */
    int32_t Ho, Wo, oys, oyo, oxs, oxo;
    int32_t Cout;             /* valid output channels (<= CoutPad) */
    ssr_view y, y0, y1;       /* y required; y0/y1 optional (p == NULL) */
    float alpha;
    int32_t act;
    ssr_view r1; int32_t r1_nc; float beta1;
    ssr_view r2; int32_t r2_nc; float beta2;
    int32_t accumulate;
    ssr_view m; int32_t m_c0, m_c1;   /* mask source: m[p, m.coff + c] for output channel c */
    /* 0, or 1 = space-to-depth evaluation of a 4x4 stride-2 pad-1 layer (KH = KW = 4, stride = 2, pad = 1, up = 1, even
     * Hi / Wi, Cin a power-of-two multiple of 32; see ssr_conv2d_s2d_ok): the layer is computed as a 2x2 stride-1
     * convolution over the view  x'[Y, X, q*Cin + c] = x[2Y-1 + (q>>1), 2X-1 + (q&1), c]  (zero outside), which the
     * big-tile kernel gathers while staging — nothing is materialised.  `w` must then be packed with
     * ssr_pack_item.fwd_s2d = 1: [q*Cin/32 + chunk][2x2 taps (dy,dx)][CoutPad][32], tap (ky,kx) = (2dy + (q>>1), 2dx + (q&1)). */
    int32_t s2d;
    /* 0: the mask `m` is LeakyReLU' (1 | 0.2 by the sign of m); 1: ReLU' (1 | 0) — backward through the VGG19 layers */
    int32_t m_relu;
    /* ---- LeakyReLU decision fix-up of the split-bf16 mode (SSR_F32X3; round 4).  The three-product split rounds a product at
     * 2^-16 instead of 2^-24: ~10x more pre-activations land on the other side of zero than in an fp32 evaluation, and every such
     * decision moves all upstream gradients by 0.8 x its gradient (measured: conv_first.weight 25 % of elements outside the 1e-3
     * gate with 48 of 22.8 M decisions flipped).  With fix_list set on a LeakyReLU layer the epilogue appends every output whose
     * pre-activation satisfies |v| < fix_thr to the list and ssr_conv2d_fixup recomputes exactly those from the fp32 inputs and
     * the UNPACKED fp32 weights in double precision and rewrites y: the decisions are then those of an exact evaluation (the
     * backward reads them from the sign of the stored output), all other outputs keep their 1e-5 accuracy.
     *   fix_list: int32 [4 + 2 * fix_cap]: [0] entries appended (may exceed fix_cap: the surplus is dropped), [1] internal ticket,
     *             [2] high-water mark of [0] (diagnostics), [3] unused; then (stored output pixel index, channel) pairs.  Zeroed
     *             once by the host; ssr_conv2d_fixup resets [0] and [1].
     *   w_ref: the layer's weights as the reference stores them, fp32 [Cout][w_ref_cin][KH][KW]; w_ref_sigma: NULL or the
     *          spectral-norm sigma (the effective weight is w_ref / sigma[0]).
     * Supported: stride 1, one input view (x2 = NULL), up 1 | 2, y only (no y0 / y1 / r1 / r2 / m / accumulate), alpha 1. */
    int32_t* fix_list;
    int32_t fix_cap;
    float fix_thr;
    const float* w_ref;
    const float* w_ref_sigma;
    int32_t w_ref_cin;
} ssr_conv_desc;

int ssr_conv2d(const ssr_conv_desc* d, void* stream);
/* recompute the outputs the launch of `d` listed in d->fix_list (see the fix_* fields); a no-op launch when the list is empty */
int ssr_conv2d_fixup(const ssr_conv_desc* d, void* stream);
/* n <= 4 descriptors (host array) of identical geometry in ONE launch where the kernel family supports it (the four
 * output-parity classes of a 4x4 stride-2 dgrad, built by the host as 2x2 stride-1 convolutions); otherwise n launches. */
int ssr_conv2d_batch(const ssr_conv_desc* ds, int32_t n, void* stream);
/* test / tuning hook: force a kernel family. 0 = automatic (ssr_conv2d), 1 = weight-stationary persistent
 * kernel (SSR_EUNSUP if the descriptor does not fit it), 2 = skip it (K-resident or pipelined kernel),
 * 3 = pipelined kernel only, 4 = big-tile kernel (32x16 pixels x 64 channels per workgroup; SSR_EUNSUP if unfit),
 * 5 = thin-output VALU kernel (Cout <= 8, Cin <= 64; SSR_EUNSUP if unfit).  SSR_F32X3 descriptors: 4 = the split-mode big-tile
 * kernel (csrc/conv_big_x3.hip), 6 = the producer / MFMA-wave ring kernel for small grids (csrc/conv_x3q.hip, round 5), 7 = the register-tiled
 * kernel that took those layers over in round 6 (csrc/conv_x3r.hip: four pixel tiles per MFMA wave, K split over four waves). */
int ssr_conv2d_impl(const ssr_conv_desc* d, void* stream, int32_t impl);
/* Which kernel instantiation ssr_conv2d dispatches this descriptor to, encoded as
 * KH*1000 + stride*100 + NT*10 + WAVES (NT = 32-channel output tiles per wave, WAVES per workgroup);
 * used by bench.py to attribute launch durations to kernel symbols.  Negative on error. */
int ssr_conv2d_variant(const ssr_conv_desc* d);
/* A dependent CHAIN of n (2..4) stride-1 3x3 SSR_F32X3 convolutions over one grid in ONE persistent launch (csrc/conv_x3c.hip): the dense
 * block's conv1..conv4 (rrdbnet_arch.py:37-41: each reads the channel prefix the earlier ones extend) or the slices 4..1 of its gather-form
 * backward.  The workgroups of an image hand their results to each other through memory (write-through stores, per-tile flag words,
 * bounded polls); a later descriptor may read what an earlier one writes, nothing else may alias.
 *   state: ssr_conv2d_chain_state_bytes(N, Gh, Gw) bytes of device memory, zeroed ONCE by the host, owned by the launches of ONE stream
 *          (consecutive launches reuse it: tickets, epoch and flags re-arm themselves; concurrent chains need one each).
 * When the list does not qualify (ssr_conv2d_chain_ok: 32 output channels each, one of the straight-line epilogues - bias + LeakyReLU,
 * plain, or LeakyReLU-backward mask - shared by all, <= 64 tiles per image) or state is NULL: n ssr_conv2d launches, same results
 * (forward chains bit for bit; backward chains sum K in the opposite direction). */
int ssr_conv2d_chain(const ssr_conv_desc* ds, int32_t n, void* state, void* stream);
int ssr_conv2d_chain_ok(const ssr_conv_desc* ds, int32_t n);
int64_t ssr_conv2d_chain_state_bytes(int32_t N, int32_t Gh, int32_t Gw);
/* The kernel symbol ssr_conv2d launches for this descriptor as rocprofv3 prints it (without the "void (anonymous namespace)::" prefix and
 * the argument list), e.g. "conv_x3r_kernel<1, 0>": the key under which bench.py, tools/pmc_traffic.py, tools/pmc_sq.py and
 * tools/roofline_check.py file a launch, so that roofline.kernel can be found in profiles/ by name.  buflen >= 48. */
int ssr_conv2d_symbol(const ssr_conv_desc* d, char* buf, int32_t buflen);
/* input channels per packed weight chunk for a KHxKH kernel in `dtype` */
int ssr_conv2d_ck(int32_t dtype, int32_t KH);
/* 1 if a 4x4 stride-2 layer of this shape can run through the space-to-depth path (ssr_conv_desc.s2d), else 0 */
int ssr_conv2d_s2d_ok(int32_t dtype, int32_t Cin, int32_t Cout, int32_t CoutPad);

/*
 * Fused ResidualDenseBlock (rrdbnet_arch.py:37-44, and :68 for the third block of an RRDB), bf16,
 * num_feat 64 / num_grow_ch 32.  One launch keeps the whole dense block of an 8x8 or 8x16 tile (with its 5-pixel halo)
 * resident in LDS; only the weights are streamed.
 *   forward : in = x (64 ch) -> x1..x4 written to channels [64,192) of `slices`,
 *             out[0,64) = alpha5*(conv5(cat(x..x4)) + b5) + beta1*x + beta2*r2;
 *             w[k] = forward-packed weights of conv1..5 (ssr_pack_weights, 32-channel chunks), bias[k] their biases.
 *   backward: in = d_out (64 ch, gradient w.r.t. the block output) -> dpre4..dpre1 (pre-activation gradients of
 *             conv4..conv1, masked with lrelu'(x_k) read from `mask` = the forward `slices` buffer) written to
 *             channels [64,192) of `slices`; out[0,64) = d x = gathered dgrad + beta1*d_out + beta2*r2;
 *             w[j] = gather-packed weights of slice 4-j (ssr_pack_dgrad_gather; conv5's 0.2/0.04 folded in),
 *             bias = NULL, alpha5 = 1.
 */
typedef struct ssr_rdb_desc {
    int32_t dtype, N, H, W;
    ssr_view in, slices, out, mask;
    const void* w[5];
    const float* bias[5];
    float alpha5, beta1;
    ssr_view r2;
    float beta2;
    /* optional L2 warm-up: the five weight arrays of the NEXT launch on this stream (NULL / 0 = none).  Every block
     * touches its share of the lines of its own XCD's L2 while it computes, so that the next dense block's weight
     * stream (479 KB, re-read by all 256 blocks) hits L2 instead of the Infinity Cache / HBM. */
    const void* w_next[5];
    int32_t w_next_bytes[5];
    /* which kernel runs THIS launch: 0 = automatic (SSR_RDB_TILE from the environment, else the 8 x 16 tiles of csrc/rdb_tile.hip
     * when the launch gives (nearly) every CU a workgroup, else the 8 x 8 tiles of csrc/rdb_fwd.hip), 8 = 8 x 8 tiles, 16 = 8 x 16
     * tiles.  Both kernels add the same products in the same order: their results are bit-identical (tests/test_gpu_rdb_tile.py,
     * tests/test_gpu_rdb_stress.py).  A per-descriptor field, not library state: two plans may share the library. */
    int32_t tile;
} ssr_rdb_desc;
int ssr_rdb_forward(const ssr_rdb_desc* d, void* stream);
int ssr_rdb_backward(const ssr_rdb_desc* d, void* stream);
/* the kernel ssr_rdb_forward / ssr_rdb_backward run for this descriptor: 8 = 8 x 8 tiles, 16 = 8 x 16 tiles */
int ssr_rdb_tile_of(const ssr_rdb_desc* d);

/*
 * Weight gradient (autograd's convolution_backward weight/bias part, triggered at
 * ssr_esrgan_model.py:192,221,227):
 *   dW[co][ci][ky][kx] += alpha * sum_{n,gy,gx} dY[n, gy, gx, co] * X[n, gy*stride+ky-pad_y, gx*stride+kx-pad_x, ci]
 *   db[co]             += alpha * sum dY
 * dW/db are fp32 in the reference's OIHW layout.  Many layers can be processed by one launch: the
 * host passes device tables of layer descriptors and of work items (layer, co/ci tile, pixel-tile range).
 */
typedef struct ssr_wgrad_layer {
    ssr_view x;               /* forward input (same meaning as ssr_conv_desc.x) */
    ssr_view dy;              /* gradient w.r.t. the conv output (pre-activation), Gh x Gw pixels */
    int32_t N, Hi, Wi, up, Cin, Cout;
    int32_t pad_y, pad_x, Gh, Gw;
    float alpha;
    float* dw;                /* [Cout][Cin_w][KH][KW] fp32 */
    int32_t Cin_w;            /* Cin of the weight tensor (== real input channels; <= Cin) */
    float* db;                /* [Cout] or NULL */
} ssr_wgrad_layer;

typedef struct ssr_wgrad_item {
    int32_t layer, co0, ci0, tile_begin, tile_end, atomic;
    /* second 32-channel block of output gradients contracted with the SAME input patch (kernels with
     * ssr_wgrad_co_tile() == 64 only): nco == 2 -> channels co0_b.. of layer_b, which must read the same x view, ci0 and
     * geometry as `layer` (it may be `layer` itself: the other half of a 64-output conv).  nco == 0 or 1: single. */
    int32_t nco, layer_b, co0_b;
} ssr_wgrad_item;

int ssr_conv2d_wgrad(const ssr_wgrad_layer* layers_dev, const ssr_wgrad_item* items_dev, int32_t n_items,
                     int32_t dtype, int32_t KH, int32_t KW, int32_t stride, void* stream);
/* number of pixel tiles of a layer in the wgrad kernel selected by (dtype, KH): host helper for building items */
int32_t ssr_wgrad_tiles(int32_t N, int32_t Gh, int32_t Gw, int32_t dtype, int32_t KH);
/* input-channel width of a work item (ci0 must be a multiple of it; co0 a multiple of 32) for that kernel */
int32_t ssr_wgrad_ci_tile(int32_t dtype, int32_t KH);
/* 64 when that kernel takes paired items (nco == 2), else 32 */
int32_t ssr_wgrad_co_tile(int32_t dtype, int32_t KH);

/*
 * Weight packing (once per optimizer step; replaces cuDNN's internal filter transforms).
 * Table entry: src fp32 OIHW -> fwd [tap][CoutPad][CinPad] and dgrad layouts, optionally scaled by
 * 1/sigma (spectral norm, W_orig/sigma: torch.nn.utils.spectral_norm at discriminator_arch.py:30-39).
 */
typedef struct ssr_pack_item {
    const float* src;         /* [Cout][Cin][KH][KW] */
    const float* inv_scale;   /* device scalar sigma (weights are divided by it) or NULL */
    void* dst_fwd;            /* [CinPad/ck_fwd][KH*KW][CoutPad][ck_fwd] or NULL */
    void* dst_dgrad;          /* stride 1: [CoutPadI/ck_dgrad][KH*KW (taps flipped)][CinPadO][ck_dgrad];
                                 stride 2 (4x4): [4 parity classes][CoutPadI/ck_dgrad][4 taps][CinPadO][ck_dgrad];
                                 or NULL */
    int32_t Cout, Cin, KH, KW, stride;
    int32_t CoutPad, CinPad;      /* fwd padding */
    int32_t CinPadO, CoutPadI;    /* dgrad: "output" channels (=Cin) padded to 32, "input" (=Cout) padded to ck_dgrad */
    int32_t ck_fwd, ck_dgrad;     /* channels per chunk of the consuming kernels (ssr_conv2d_ck) */
    int32_t fwd_s2d;              /* 1: dst_fwd in the space-to-depth order of ssr_conv_desc.s2d (4x4 stride 2, ck_fwd = 32) */
} ssr_pack_item;

int ssr_pack_weights(const ssr_pack_item* items_dev, int32_t n_items, int32_t dtype, void* stream);

/*
 * Packed weights for the "gather" form of the dense-block backward (rrdbnet_arch.py:37-44 under autograd):
 *   d x_k = sum_{j > k} conv3x3( dpre_j , rot180(W_j[:, slice_k])^T )
 * i.e. ONE convolution per channel slice over the channel concatenation of all later pre-activation
 * gradients — the mirror image of the concat-free forward.  One table entry copies the slice of one later
 * conv into its K range of the gathered weight: dst[chunk][tap'][o][cc] with k = chunk*ck + cc,
 * value = scale * src[k - kbase][ci0 + o][8 - tap'] for kbase <= k < kbase + Cout (rows o >= nci are zero).
 */
typedef struct ssr_pack_seg {
    const float* src;         /* later conv, fp32 OIHW [Cout][Cin][3][3] */
    void* dst;                /* [Kpad/ck][9][rows_pad][ck] */
    float scale;              /* residual scaling folded in (0.2 / 0.04 for conv5: rrdbnet_arch.py:44,68) */
    int32_t Cout, Cin;        /* of src */
    int32_t ci0, nci;         /* slice of src's input channels = output channels of the gathered conv */
    int32_t kbase;            /* offset of src's Cout channels inside the concatenated K */
    int32_t rows_pad, ck;
} ssr_pack_seg;
int ssr_pack_dgrad_gather(const ssr_pack_seg* items_dev, int32_t n_items, int32_t dtype, void* stream);

/* dst[p, c] += src[p, c] over npix pixels and C channels of two NHWC views */
int ssr_add_views(ssr_view dst, ssr_view src, int32_t dtype, int64_t npix, int32_t C, void* stream);

/* ---- layout conversion at the plugin boundary (NCHW fp32 tensors of the reference API) ---- */
/* dst NHWC[n,h,w,coff+c] = src NCHW fp32 [n,c,h,w] * scale, optionally through pixel_unshuffle
 * (arch_util.py:769-785; `unshuffle` = 1 none | 2 | 4) and nearest upsampling by `up`
 * (lr_resized, ssr_esrgan_model.py:133). */
int ssr_nchw_to_nhwc(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, ssr_view dst, int32_t dtype,
                     int32_t unshuffle, int32_t up, float scale, void* stream);
int ssr_nhwc_to_nchw(ssr_view src, int32_t dtype, float* dst, int32_t N, int32_t C, int32_t H, int32_t W,
                     void* stream);
/* inverse for gradients: dst NCHW fp32 [n,c,h,w] (+)= src NHWC slice */
int ssr_fill(void* p, int64_t n_elems, int32_t dtype, float value, void* stream);

/* ---- bilinear x2 (align_corners=False): discriminator_arch.py:50,55,60 ---- */
/* y[n,2H,2W,C] = bilinear(a (+ b)) ; b optional (skip-add folded into the read) */
int ssr_bilinear2x_fwd(ssr_view a, ssr_view b, ssr_view y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C,
                       void* stream);
/* s1 = bilinear2x^T(dy) (+ r if r.p) -> y1 (optional); y = s1 * lrelu'(m) (m optional) */
int ssr_bilinear2x_bwd(ssr_view dy, ssr_view r, ssr_view y1, ssr_view y, ssr_view m, int32_t dtype, int32_t N,
                       int32_t H, int32_t W, int32_t C, void* stream);
/* OR-ed into `dtype` of the two calls above: the per-pixel kernels for this call whatever the shape.  Default: layers whose
 * channels are whole groups of eight 16-byte vectors (64 in bf16, 32 in fp32 storage) use the LDS-tile kernels (same arithmetic,
 * same order: identical bytes in bf16).  A per-call flag, not library state.  Environment (read once): SSR_BILINEAR_FLAT=1. */
#define SSR_BILINEAR_FLAT 0x100
/* nearest x2 backward (2x2 sum) with the same epilogue: rrdbnet_arch.py:127-128 backward */
int ssr_nearest2x_bwd(ssr_view dy, ssr_view r, ssr_view y1, ssr_view y, ssr_view m, int32_t dtype, int32_t N,
                      int32_t H, int32_t W, int32_t C, void* stream);

/* ---- spectral norm (torch.nn.utils.spectral_norm, discriminator_arch.py:7,26,30-39) ---- */
typedef struct ssr_sn_item {
    const float* w;           /* weight_orig as [rows = Cout][cols = Cin*KH*KW] */
    float* u;                 /* [rows] in/out */
    float* v;                 /* [cols] in/out */
    float* sigma;             /* out: scalar */
    float* tmp;               /* [rows + cols + 4] scratch */
    int32_t rows, cols;
} ssr_sn_item;
/* one power iteration (train mode) for every layer in the table, then sigma = u.(W v);
 * with power_iter = 0 only sigma is recomputed (eval mode) */
int ssr_spectral_norm(const ssr_sn_item* items_dev, int32_t n_items, int32_t max_rows, int32_t max_cols,
                      int32_t power_iter, void* stream);
/* backward through W_sn = W/sigma: dW_orig += (dW_sn - <dW_sn, W_sn> u v^T) / sigma */
/* tmp: SSR_SN_BWD_SLOTS floats of scratch per item (per-block partial sums of <dW_sn, W>, added in index order: the result
 * does not depend on the order in which the blocks finish; nothing to zero) */
#define SSR_SN_BWD_SLOTS 64
typedef struct ssr_sn_bwd_item {
    const float* dw_sn; const float* w; const float* u; const float* v; const float* sigma;
    float* dw; float* tmp; int32_t rows, cols;
} ssr_sn_bwd_item;
int ssr_spectral_norm_bwd(const ssr_sn_bwd_item* items_dev, int32_t n_items, int32_t max_elems, void* stream);

/* ---- losses (BasicSR L1Loss / GANLoss('vanilla'); call sites ssr_esrgan_model.py:148,182,218,224) ---- */
/* USMSharp of BasicSR (img_process_util.USMSharp(radius=50, sigma=0), built at ssr_esrgan_model.py:31, applied to the
 * ground truth at :109): 51x51 Gaussian (sigma 8) blur with reflect padding, thresholded residual mask, soft blend.
 * src/dst: `planes` contiguous fp32 H x W planes (NCHW tensors: planes = N*C); src is multiplied by in_scale first
 * (1/255 for uint8-valued input), dst is in [0,1].  H*W <= 16384 (SSR_EUNSUP beyond), H, W > 25. */
int ssr_usm_sharp(const float* src, float* dst, int32_t planes, int32_t H, int32_t W, float in_scale, float weight,
                  float threshold, void* stream);

/* Deterministic reductions (run-to-run bit-identical results; the reference's fp32 CPU path is).
 * SSR_DETERMINISTIC OR-ed into the dtype of the two loss calls: loss_out / mean_out are arrays of SSR_LOSS_SLOTS floats, block b
 * adds its partial sum to slot b (single writer per slot and launch) and the reader adds the slots in index order; without the
 * flag: one fp32 atomicAdd per block into loss_out[0] (arrival order). */
#define SSR_DETERMINISTIC 0x200
#define SSR_LOSS_SLOTS 256
/* loss_out[0] += weight * mean|a-b| over (N,H,W,C valid); grad (optional) = weight*sign(a-b)/numel */
int ssr_l1_loss(ssr_view a, ssr_view b, ssr_view grad, int32_t dtype, int64_t npix, int32_t C, float weight,
                float* loss_out, void* stream);
/* BCE-with-logits against constant target t: loss_out[0] += weight*mean(softplus(x) - x*t);
 * mean_out[0] += mean(x) (optional); grad = weight*(sigmoid(x)-t)/numel (optional) */
int ssr_bce_logits_loss(ssr_view x, ssr_view grad, int32_t dtype, int64_t npix, float target, float weight,
                        float* loss_out, float* mean_out, void* stream);

/* dst[e] += sum over p = 0 .. parts-1 (in that order) of src[p * stride + e], e < n: the fixed-order sum of per-split partial
 * weight gradients (deterministic mode: each pixel-range split of a layer is given its own partial dW / db as `dw` / `db` of a
 * layer-table entry of its own, so that every gradient element has ONE writer per launch) */
typedef struct ssr_reduce_item { float* dst; const float* src; int64_t n, stride; int32_t parts, pad_; } ssr_reduce_item;
int ssr_wgrad_reduce(const ssr_reduce_item* items_dev, int32_t n_items, int64_t max_elems, void* stream);

/* ---- optimizer: torch.optim.Adam (weight_decay 0, eps 1e-8) over a flat fp32 arena, with the
 *      BasicSR model_ema update fused (ssr_esrgan_model.py:193,228,230-231) ---- */
typedef struct ssr_adam_args {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    float* ema;               /* NULL or EMA arena: ema = ema*decay + p*(1-decay) */
    int64_t n;
    const float* lr;          /* device scalar */
    int32_t* step;            /* device counter, incremented by the kernel launch */
    float beta1, beta2, eps, ema_decay, grad_scale;
} ssr_adam_args;
int ssr_adam_step(const ssr_adam_args* a, void* stream);

/* y = a*x + b*y over n fp32 elements (grad averaging / accumulation helpers) */
int ssr_axpby_f32(float a, const float* x, float b, float* y, int64_t n, void* stream);

/* ---- image quantisation and validation metrics (csrc/metrics.hip) ----
 * ssr_quantize_u8: fp32 NCHW -> uint8 NHWC, clamp(0,1) * 255 then mode 0: round half to even (basicsr tensor2img as called at
 *   /root/reference/ssr/models/ssr_esrgan_model.py:302-305), mode 1: truncate (astype(uint8) at /root/reference/ssr/infer_grid.py:60-64,
 *   infer.py:58-60).
 * ssr_metric_shift_sums: a, b uint8 [H][W][C], C <= 4.  For every offset pair (ro, co) in [0, max_offset]^2 and channel c writes
 *   out[((ro*(max_offset+1) + co)*C + c)*2 + {0,1}] = sum d, sum d^2 (exact, int64) over the window of
 *   (H-2*crop-max_offset) x (W-2*crop-max_offset) pixels, d = a[y+crop+ro][x+crop+co][c] - b[y+crop+max_offset-ro][x+crop+max_offset-co][c]:
 *   max_offset 0 -> PSNR (basicsr calculate_psnr), 8 -> cPSNR (/root/reference/ssr/metrics/cpsnr.py:36-55).
 * ssr_metric_ssim_sums: out[c] = sum over the valid region of the SSIM map of channel c (11x11 Gaussian window, sigma 1.5, fp64;
 *   basicsr calculate_ssim); the caller divides by (H-2*crop-10)*(W-2*crop-10). */
int ssr_quantize_u8(const float* src_nchw, uint8_t* dst_nhwc, int32_t N, int32_t C, int32_t H, int32_t W, int32_t mode, void* stream);
int ssr_metric_shift_sums(const uint8_t* a, const uint8_t* b, int32_t H, int32_t W, int32_t C, int32_t crop, int32_t max_offset,
                          int64_t* out, void* stream);
int ssr_metric_ssim_sums(const uint8_t* a, const uint8_t* b, int32_t H, int32_t W, int32_t C, int32_t crop, double* out, void* stream);

/* ---- VGG19 perceptual loss glue (csrc/vgg.hip; the convolutions run through ssr_conv2d with SSR_ACT_RELU / m_relu) ----
 * ssr_channel_affine: y[p, c] (+)= x[p, c] * scale[c] + shift[c] for c < C <= 8 (host float arrays, copied into the launch):
 *   the input normalisation (x - mean) / std of the feature extractor and, with accumulate = 1, its adjoint.
 * ssr_relu_maxpool2_fwd: P[N, H/2, W/2, C] = maxpool2x2(relu(F[N, H, W, C])), H and W even.
 * ssr_relu_maxpool2_bwd: gF (+)= the adjoint: each window's gradient goes to its first maximum (row-major) if F there is > 0. */
typedef struct ssr_vec8 { float v[8]; } ssr_vec8;
int ssr_channel_affine(ssr_view x, ssr_view y, int32_t dtype, int64_t npix, int32_t C, const float* scale, const float* shift,
                       int32_t accumulate, void* stream);
int ssr_relu_maxpool2_fwd(ssr_view f, ssr_view p, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int ssr_relu_maxpool2_bwd(ssr_view f, ssr_view gp, ssr_view gf, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C,
                          int32_t accumulate, void* stream);

/* SSR_F32X3 weight gradients: split an fp32 buffer (n % 4 == 0 elements) into bf16 planes hi = bf16(x), lo = bf16(x - hi); the
 * bf16 wgrad kernel then accumulates (x_hi, dy_hi) + (x_hi, dy_lo) + (x_lo, dy_hi) into the fp32 gradient. */
int ssr_split_bf16(const float* x, void* hi, void* lo, int64_t n, void* stream);
/* the same for a device table of buffers in ONE launch (every n % 8 == 0, 16-byte aligned pointers; max_n = the largest n): what
 * engine.WgradBatch issues in front of the three split passes of a batch */
typedef struct ssr_split_item {
    const float* x;
    void* hi;
    void* lo;
    int64_t n;
} ssr_split_item;
int ssr_split_bf16_multi(const ssr_split_item* items_dev, int32_t n_items, int64_t max_n, void* stream);

/* library / device info: writes "gfx950 CUs=256 ..." style text */
int ssr_device_info(char* buf, int32_t buflen);
int ssr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SSR_HIP_H */
