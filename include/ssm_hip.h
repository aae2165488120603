/* ssm_hip.h - C ABI of libssm_hip.so: the MI355X (gfx950) replacement for the
 * stock-PyTorch operators on the Super SloMo frame-pair -> intermediate-frame
 * path.  Plain pointers and sizes only; no torch types.  Every entry point
 * cites the reference interface it replaces (paths relative to the reference
 * repository root).
 *
 * Conventions
 *  - all tensors are fp32 device memory owned by the caller; the library never
 *    allocates or frees device memory and keeps no pointer past return - with
 *    ONE exception the caller opts into: a launch program (ssm_program_*)
 *    keeps a copy of the argument values, pointers included, of every launch
 *    it recorded until ssm_program_destroy; the buffers they point to must
 *    outlive the program's last ssm_program_run;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*) and
 *    re-entrant.  It launches on the CALLING THREAD's current device
 *    (hipGetDevice), which must own `stream` and every pointer passed; the
 *    library never calls hipSetDevice and keeps no "current device" of its
 *    own.  One process may drive several GPUs from several host threads (the
 *    reference's torch.nn.DataParallel replica threads, scripts/main.py:74-76:
 *    each thread has set its device): per-device state - the opt-in of a
 *    kernel to more than 64 KiB of dynamic LDS - is keyed on (kernel, device)
 *    and taken on a thread's first launch there (csrc/ssm_common.h
 *    reserve_lds; tests/test_build_fences_cpu.py fences process-wide guards);
 *  - return value 0 = ok, negative = SSM_E_* ; ssm_last_error_string() holds a
 *    message for the calling thread.  Nothing throws or exits.  The Python side
 *    turns non-zero codes into RuntimeError, mirroring the reference's
 *    assert/exception convention (scripts/utils/validators.py).
 *
 * Tensor views
 *  A `ssm_view` addresses logical element (b,c,y,x) at
 *      ptr[b*sb + c*sc + y*sh + x]           (strides in floats, x-stride 1)
 *  so the same entry points take plain contiguous NCHW (sh=W, sc=H*W,
 *  sb=C*H*W), channel slices of it, batch-broadcast tensors (sb=0) and the
 *  library's "padded plane" layout.
 *
 * Padded-plane layout (what the convolutions read)
 *  [B][C][Hp][Wp] with Hp = H + 2*SSM_PADY, Wp = roundup4(W + 2*SSM_PADX),
 *  the image at rows SSM_PADY.., cols SSM_PADX.. and an all-zero frame around
 *  it.  `ptr` of such a view points at the first INTERIOR element, is 16-byte
 *  aligned, and sh/sc/sb are multiples of 4.  The zero frame implements the
 *  convolution's zero padding (scripts/models/layers.py:22-31) with no bounds
 *  test in the kernel.  Kernels only ever write interiors, so a buffer zeroed
 *  once stays valid.  The DIRECT-form kernels (ssm_conv2d_*, ssm_conv2d_hl8_*)
 *  read a tile's overshoot past the map unpredicated: their inputs must be
 *  readable for SSM_TAIL_SLACK_FLOATS past the last element; what is read
 *  there only feeds outputs that are not stored.  The Winograd-form kernels
 *  (ssm_wino_*, ssm_wino1d_*, ssm_wino4_*, ssm_wino5_*, ssm_wino7_*) never read
 *  outside the padded plane: a tile's transform mixes its whole input patch
 *  into every output, so overshoot rows / 16-byte pieces are fetched from the
 *  zero frame instead (tests/test_hip_overshoot.py poisons the memory behind
 *  the last plane with NaN).
 */
#ifndef SSM_HIP_H
#define SSM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSM_PADX 4
#define SSM_PADY 3
#define SSM_TAIL_SLACK_FLOATS (1 << 16)

#define SSM_OK 0
#define SSM_E_ARG (-1)      /* bad shape / alignment / unsupported size      */
#define SSM_E_LAUNCH (-2)   /* HIP reported a launch error                   */
#define SSM_E_UNSUPPORTED (-3)

#define SSM_FLAG_LRELU 1    /* apply LeakyReLU(slope) after bias             */
#define SSM_FLAG_FP16_FAST 2 /* HL8 conv: hi*hi product only (plain fp16 inputs) */
#define SSM_FLAG_Q8 4        /* HL8 conv on Q8 operands: 1 fp16 MFMA + 2 block-scaled fp8 MFMAs per product */
#define SSM_FLAG_MASK 8      /* the `add` view of an *_add_fwd entry point (ssm_wino_conv2d_add_fwd, ssm_wino4_conv2d_add_fwd, ssm_wino5_conv2d_add_fwd,
                              * ssm_splitk_finish_fwd) is a MASK source m instead of a pre-activation addend: out = conv(x) * (m > 0 ? 1 :
                              * slope), no activation (SSM_FLAG_LRELU must be clear).  The training step's data gradients use it: the
                              * gradient wrt a layer's input leaves the convolution as dZ of the layer that produced that input -
                              * dX * LeakyReLU'(its output) - and the separate ssm_lrelu_bwd pass over it disappears (autograd of
                              * layers.conv's LeakyReLU(0.1), scripts/models/layers.py:21-33)                                              */

typedef struct ssm_view {
    float *ptr;
    long long sb, sc;   /* batch / channel stride, floats */
    int sh;             /* row stride, floats             */
} ssm_view;

/* "HL8" activation: every fp32 value x is carried as hi = fp16(x), lo = fp16(x - hi).
 * [B][C/8][2 = hi|lo][Hp][Wp][8 x fp16]; a pixel's 8 channels of one part are one 16-byte
 * unit (= one fp16 MFMA operand fragment).  Same zero frame / slack rules as the fp32 padded
 * planes.  Strides in 16-byte pixels; ptr = first INTERIOR pixel of group 0, hi part.      */
typedef struct ssm_hview {
    void *ptr;
    long long sb, sg, sp;   /* batch stride, channel-group stride, hi->lo stride */
    int sh;                 /* row stride */
} ssm_hview;

int ssm_abi_version(void);
const char *ssm_last_error_string(void);

/* Padded-plane geometry for an HxW image. */
void ssm_plane_dims(int H, int W, int *Hp, int *Wp);

/* Strided copy of a [B,C,H,W] block between two views (NCHW <-> padded plane,
 * channel slices, torch.cat by writing at a channel offset).                */
int ssm_copy_view(ssm_view src, ssm_view dst, int B, int C, int H, int W, void *stream);

/* Which tile configuration ssm_conv2d_fwd will use for this problem (kernel size,
 * output channels, batch, map size, fused pool or not):
 * BN = output-channel block the weights must be packed for, CK = input-channel
 * chunk (Cin and the first cat source must be multiples of it).             */
int ssm_conv_config(int k, int Cout, int B, int H, int W, int pool, int *BN, int *CK);
/* The same with the input channel count and the fused-upsample form taken into account (what ssm_conv2d_fwd /
 * ssm_conv2d_ups_fwd evaluate at launch: an estimate of the launch duration per tile configuration from the
 * workgroup count against the 512 resident slots of the chip, the MFMAs per workgroup and the tile overshoot).
 * kind = index of the configuration (diagnostics).  Pack filters for the BN / CK returned HERE.               */
int ssm_conv_plan(int k, int Cin, int Cout, int B, int H, int W, int pool, int ups, int *kind, int *BN, int *CK);
/* Tests / tuning only: force tile configuration `kind` for its kernel size (-1 = automatic).  Process-wide.
 * Returns the number of configurations.                                                                     */
int ssm_conv_force_kind(int kind);

/* Number of floats of the packed filter / packed bias for that configuration. */
size_t ssm_packed_weight_floats(int Cout, int Cin_padded, int k, int BN);
size_t ssm_packed_bias_floats(int Cout, int BN);

/* OIHW fp32 filter (the reference's state-dict layout) ->
 * [Cout/BN][Cin_padded][k*k][BN], zero-filled where padded.  Device to device. */
int ssm_pack_weights(const float *w_oihw, const float *bias, float *w_packed, float *bias_packed,
                     int Cout, int Cin, int Cin_padded, int k, int BN, void *stream);

/* Replaces layers.conv = Conv2d(stride 1, pad (k-1)/2, bias) [+ LeakyReLU(0.1)]
 * (scripts/models/layers.py:21-33) and the bare final_conv
 * (scripts/models/flow_computation.py:145-153).  fp32 implicit GEMM on
 * v_mfma_f32_32x32x2_f32.  k in {3,5,7}.
 *   x1,C1 / x2,C2 : input = torch.cat([x1, x2], dim=1) without materialising it
 *                   (fuse_conv, flow_computation.py:271-272); C2 may be 0.
 *                   Both padded-plane views with identical sh/sc.
 *   y             : any view, [B,Cout,H,W]
 *   pool          : optional (ptr NULL = off) view [B,Cout,H/2,W/2] that
 *                   receives avg_pool2(y) (scripts/models/layers.py:60-63)
 *                   fused into the epilogue.                                */
int ssm_conv2d_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed,
                   const float *bias_packed, ssm_view y, ssm_view pool, int B, int H, int W,
                   int Cout, int k, float slope, int flags, void *stream);

/* Fused  conv3x3( F.upsample(torch.cat([a, b], 1), size=(2h,2w), mode="bilinear") )  - the decoder step of
 * scripts/models/flow_computation.py:244-247 (lambdas :92-137; flow_interpolation.py:92-141,224-231) - in exact
 * fp32 (v_mfma_f32_32x32x2_f32).  a [B,C1,H/2,W/2], b [B,C2,H/2,W/2] LOW-res padded-plane views (C2 may be 0, b may
 * be batch-broadcast), H, W = OUTPUT size (even).  The concatenated, upsampled tensor is never materialised: each
 * workgroup expands the low-res chunk it staged into the hi-res patch in LDS (align_corners=False half-pixel rule,
 * edge-clamped sources, zeros outside the image = the convolution's padding).  Filters: ssm_pack_weights for the
 * BN / CK of ssm_conv_plan(3, C1+C2, Cout, B, H, W, 0, 1, ...).                                                  */
int ssm_conv2d_ups_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed,
                       ssm_view y, int B, int H, int W, int Cout, float slope, int flags, void *stream);

/* The two convolutions above with a pre-activation addend: y = act(conv(x) + bias + add[b / add_div]), add = a view
 * [B / add_div, Cout, H, W].  For the parts of a convolution's input that do not depend on the batch index - the reference
 * evaluates stage 2 once per interpolation time t (evaluate_interpolation_results.py:234-242), but the image channels of its
 * first convolution's 16-channel input (flow_interpolation.py:364-367: channels 0:3 and 13:16) and the stage-1 half of the
 * cross-skip concat in front of conv7a (flow_interpolation.py:98-101,224-231) are the same for every t of a pair: their
 * partial sums are computed once per pair by a plain call and enter the per-t launches here.  add.ptr NULL = the plain form.
 * The direct kernels read the addend unpredicated: Cout must be a multiple of the plan's BN and add a padded-plane view
 * (readable frame / tail slack for the tile overshoot); the Winograd kernels predicate their reads.                       */
int ssm_conv2d_add_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                       ssm_view pool, ssm_view add, int add_div, int B, int H, int W, int Cout, int k, float slope, int flags,
                       void *stream);
int ssm_conv2d_ups_add_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                           ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags, void *stream);

/* ---- the same 3x3 convolution as Winograd F(2x2,3x3), all arithmetic fp32 (v_mfma_f32_32x32x2_f32) ---------------
 * Same operator and operand layout as ssm_conv2d_fwd / ssm_conv2d_ups_fwd for k = 3 (layers.conv,
 * scripts/models/layers.py:21-33; decoder step scripts/models/flow_computation.py:244-247), evaluated as
 * Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A per 2x2 output tile: 16 instead of 36 multiplies per (cin, cout, 4 outputs),
 * i.e. 2.25x fewer matrix-core cycles; in fp32 the result differs from the direct form by rounding only (about as much as
 * a different summation order; profiles/DESIGN_history_r1-r3.md 3.2d).  Any H, W (r6: odd widths too - the last tile of a row stores its lone column by itself, the zero frame stays
 * untouched; the fused pool and the fused upsample need even sizes); Cin and the first cat source multiples of CK.
 * ssm_wino_plan: tile configuration for the problem (ups: the fused-upsample entry point; BN = cout block to pack for, CK =
 * channel chunk); two kernel forms - one workgroup per CU with 16 frequency accumulators per wave, or two per CU with 8.
 * ssm_wino_pack_weights: OIHW fp32 3x3 filter -> U = G g G^T as [Cout/BN][Cin][4][BN][4] (+ bias padded to BN).          */
int ssm_wino_plan(int Cin, int Cout, int B, int H, int W, int ups, int *kind, int *BN, int *CK);
int ssm_wino_force_kind(int kind);       /* tests / tuning only (-1 = automatic); returns the number of configurations */
size_t ssm_wino_packed_weight_floats(int Cout, int Cin, int BN);
int ssm_wino_pack_weights(const float *w_oihw, const float *bias, float *w_packed, float *bias_packed, int Cout, int Cin,
                          int BN, void *stream);
int ssm_wino_conv2d_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed,
                        ssm_view y, ssm_view pool, int B, int H, int W, int Cout, float slope, int flags, void *stream);
int ssm_wino_conv2d_ups_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed,
                            ssm_view y, int B, int H, int W, int Cout, float slope, int flags, void *stream);

/* Split-K form of the two convolutions above for launches that leave most of the chip idle (config 3's bottleneck layers: 512 -> 512
 * channels on 22x22 / 11x11 maps at batch 2 are 96-192 workgroups, each walking every input channel).  ssm_wino_splitk_plan proposes
 * KS in {1, 2, 4, 8} (1: do not split; $SSM_WINO_SPLITK=0: always 1) for a filter packed with BN couts per block;
 * ssm_wino_conv2d_splitk_fwd (ups = 1: conv3x3(upsample2x(cat[a, b])), H x W the output) runs KS workgroups per output tile, workgroup
 * ks summing input channels [ks, ks + 1) * Cin / KS, and stores the raw sums (the bias with ks = 0, no activation) as batch entry
 * ks * B + b of `part` ([KS * B, Cout, H, W] padded planes owned by the caller); ssm_splitk_finish_fwd (csrc/ssm_elem.hip) adds the KS
 * partial maps in the order ks = 0, 1, ... (deterministic), then the optional addend [B / add_div, C, H, W], LeakyReLU (flags) and the
 * optional fused 2x2 mean - layers.conv / avg_pool of scripts/models/layers.py:21-33,60-63 as before, in two launches.              */
int ssm_wino_splitk_plan(int Cin, int Cout, int B, int H, int W, int ups, int BN, int *KS);
int ssm_wino_conv2d_splitk_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view part,
                               int KS, int ups, int B, int H, int W, int Cout, int BN, void *stream);
int ssm_splitk_finish_fwd(ssm_view part, int KS, ssm_view y, ssm_view pool, ssm_view add, int add_div, int B, int C, int H, int W,
                          float slope, int flags, void *stream);
/* The direct-form twin (csrc/ssm_conv.hip) for the maps no Winograd form takes (odd widths: config 3's 11x11 bottleneck): the same
 * convolution as ssm_conv2d_fwd - ONE source when KS > 1 (C2 must be 0: the kernel offsets its first source by the split's channel
 * range; two sources are accepted with KS = 1 only), no fused pool - with KS workgroups per output tile; `part` and the finishing
 * launch as above (ssm_splitk_finish_fwd takes odd widths too).  ssm_conv_splitk_plan: KS for the tile configuration the filter was
 * packed for (1: do not split; $SSM_CONV_SPLITK=0: always 1).                                                                    */
int ssm_conv_splitk_plan(int k, int Cin, int Cout, int B, int H, int W, int *KS);
/* Run-time twin of $SSM_WINO_SPLITK / $SSM_CONV_SPLITK for the two plan functions: 1 = on, 0 = off, -1 = leave.  Returns the previous
 * state as (wino | conv << 1).  A split reorders a layer's fp32 sum over the input channels: tests A/B it in one process.            */
int ssm_splitk_enable(int wino, int conv);
int ssm_conv2d_splitk_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view part,
                          int KS, int B, int H, int W, int Cout, int k, void *stream);
/* ... with the pre-activation addend of ssm_conv2d_add_fwd (8-byte aligned view). */
int ssm_wino_conv2d_add_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                            ssm_view pool, ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags,
                            void *stream);
int ssm_wino_conv2d_ups_add_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags, void *stream);

/* ---- the 3x3 convolution as Winograd F(4x4,3x3), all arithmetic fp32 (v_mfma_f32_16x16x4_f32) ------------------------------
 * Same operator and operand layout as ssm_wino_conv2d_add_fwd / ssm_wino_conv2d_ups_add_fwd (layers.conv, scripts/models/layers.py:21-33;
 * decoder step scripts/models/flow_computation.py:244-247), evaluated per 4x4 output tile from its 6x6 input patch over the points
 * {0, +-1, +-2, inf}: 36 instead of 144 multiplies per (cin, cout, 16 outputs) - 1.78x fewer matrix-core cycles than F(2x2,3x3); in
 * fp32 a single layer sits ~1e-5 from a float64 evaluation at unit output scale (per-layer bar 5e-5), the pair -> frame path at
 * 736x1280 is unchanged within its fp32 noise (tests/emulate_winograd_f44_precision.py; profiles/DESIGN_history_r1-r3.md 3.2f).
 * Cin and the first cat source multiples of 4, Cout a multiple of 32; any H, W (fused upsample: even).
 * ssm_wino4_pack_weights: OIHW fp32 3x3 filter -> U = G g G^T as [Cout/32][Cin][9][32][4] (+ bias).
 * Inputs must be padded planes; nothing outside them is read (tile overshoot is fetched from the zero frame), any row stride.          */
int ssm_wino4_plan(int Cin, int Cout, int B, int H, int W, int ups, int *kind, int *BN, int *CK);
int ssm_wino4_preferred(int Cin, int Cout, int B, int H, int W, int ups);   /* 1: modelled faster than F(2x2,3x3) for this problem */
double ssm_wino_estimate(int Cin, int Cout, int B, int H, int W, int ups);     /* modelled cycles of ssm_wino_conv2d_*_fwd (-1: unsupported) */
int ssm_wino4_force_kind(int kind);      /* tests / tuning only (-1 = automatic); returns the number of configurations */
size_t ssm_wino4_packed_weight_floats(int Cout, int Cin);
int ssm_wino4_pack_weights(const float *w_oihw, const float *bias, float *w_packed, float *bias_packed, int Cout, int Cin, void *stream);
int ssm_wino4_conv2d_add_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                             ssm_view pool, ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags,
                             void *stream);
int ssm_wino4_conv2d_ups_add_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                 ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags, void *stream);
/* conv3x3(upsample2x(cat[a, b])) (scripts/models/flow_computation.py:244-247; layers.conv, scripts/models/layers.py:21-33) in its SUB-PIXEL form
 * away from the map's border (r5).  The operator is linear in the low-res input: output parity (pa, pb) of real channel c is a plain 3x3
 * convolution of the LOW-res map with the effective filter M_pa W_c M_pb^T (bilinear weights 1/4, 3/4 folded into the taps), so the layer is
 *   ssm_wino4_conv2d_shuffle_fwd: a 3x3 convolution with 4 Cout effective output channels, channel 4 c + 2 pa + pb, whose 4x4 low-res tiles are
 *     stored as 8x8 blocks of the 2H x 2W output (pixel-shuffle store).  x1 / x2: low-res padded-plane views AT THE REGION'S ORIGIN (the region's
 *     neighbours must be the real neighbouring pixels: it is the map's interior), y: the full-resolution output view at twice that origin,
 *     H x W: the low-res region, w_packed / bias_packed: ssm_wino4_pack_weights of the effective filter (bias repeated per parity).
 * where the bilinear rule does not clamp and the convolution does not zero-pad, i.e. everywhere but on the border ring, which
 *   ssm_wino4_conv2d_ups_border_fwd: the fused-upsample kernel (arguments of ssm_wino4_conv2d_ups_add_fwd) restricted to the workgroup tiles
 *     of SSM_WINO4_BORDER_TH x SSM_WINO4_BORDER_TW output pixels on the map's border ring
 * computes.  Together they equal ssm_wino4_conv2d_ups_add_fwd (no addend) when the shuffle launch covers exactly the interior tiles. */
#define SSM_WINO4_BORDER_TH 16
#define SSM_WINO4_BORDER_TW 32
int ssm_wino4_conv2d_shuffle_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view y, int B,
                                 int H, int W, int Cout4, float slope, int flags, void *stream);
int ssm_wino4_conv2d_ups_border_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed, ssm_view y, int B,
                                    int H, int W, int Cout, float slope, int flags, void *stream);

/* ---- the 7x7 / 5x5 convolutions as one-dimensional Winograd along x, all arithmetic fp32 (v_mfma_f32_32x32x2_f32) ----------
 * Same operator and operand layout as ssm_conv2d_add_fwd for k = 7 / 5 (layers.conv, scripts/models/layers.py:21-33; the layers
 * conv1a/conv1b (k = 7) and conv2a/conv2b (k = 5) of both U-Nets, scripts/models/flow_computation.py:36-45 and
 * flow_interpolation.py:36-45; fused 2x2 mean, layers.py:60-63), evaluated as F(2,7) / F(4,5) along x - eight frequencies over the
 * points {0, +-1, +-2, +-1/2, inf} - and in the direct form along y: 8 multiplies per 2 (k = 7) / 4 (k = 5) outputs and filter
 * row instead of 14 / 20, i.e. 1.75x / 2.5x fewer matrix-core cycles; in fp32 the result differs from the direct form by
 * rounding only (a single layer: 5-8e-6 at unit output scale, tests/emulate_winograd_1d_precision.py; profiles/DESIGN_history_r1-r3.md 3.2e).
 * One input source (these layers have no concat); Cin a multiple of CK (= 2; pad the view), Cout a multiple of BN.
 * ssm_wino1d_plan: tile configuration (BN = cout block to pack for, CK = channel chunk).
 * ssm_wino1d_pack_weights: OIHW fp32 filter -> U[ky] = G g[ky] as [Cout/BN][CinP][k][2][BN][4] (+ bias padded to BN).
 * Outputs, addend and pooled outputs move as aligned 8- / 16-byte pieces when the views allow (W a multiple of 2 / 4 and
 * aligned strides: padded planes always do), element by element otherwise.                                                    */
int ssm_wino1d_plan(int k, int Cin, int Cout, int B, int H, int W, int *kind, int *BN, int *CK);
int ssm_wino1d_force_kind(int kind);     /* tests / tuning only (-1 = automatic); returns the number of configurations */
size_t ssm_wino1d_packed_weight_floats(int Cout, int CinP, int k, int BN);
int ssm_wino1d_pack_weights(const float *w_oihw, const float *bias, float *w_packed, float *bias_packed, int Cout, int Cin,
                            int CinP, int k, int BN, void *stream);
int ssm_wino1d_conv2d_add_fwd(ssm_view x, int Cin, const float *w_packed, const float *bias_packed, ssm_view y, ssm_view pool,
                              ssm_view add, int add_div, int B, int H, int W, int Cout, int k, float slope, int flags, void *stream);

/* ---- the 7x7 convolutions as a blocked two-dimensional Winograd form, all arithmetic fp32 (v_mfma_f32_16x16x4_f32) ----------
 * Same operator and operand layout as ssm_wino1d_conv2d_add_fwd for k = 7 (layers.conv, scripts/models/layers.py:21-33; conv1a /
 * conv1b of both U-Nets, scripts/models/flow_computation.py:36-45 and flow_interpolation.py:36-45; fused 2x2 mean, layers.py:60-63;
 * pre-activation addend for the hoisted part of stage 2's conv1a).  The 7x7 filter (zero-padded to 8x8) is 2x2 blocks of 4x4 taps;
 * each block is a F(4x4,4x4) Winograd filter over the seven points {0, +-1, +-2, 1/2, inf}, and because the blocks of a 4x4 output tile
 * read input windows that are whole tiles apart, one input transform per window position serves all four: 4 x 49 multiplies per 16
 * outputs and (cin, cout) = 12.25 per output against 28 for F(2,7) and 49 for the direct form (csrc/ssm_wino7.hip).
 * In fp32 the result differs from the direct form by rounding only (a 32-channel layer: 2e-6 rms / 3e-5 max at unit output scale).
 * One input source, any Cin >= 1 (a k-step is the four blocks of one channel: no channel padding); Cout a multiple of 32.
 * Inputs are padded planes; nothing outside them is read (tile overshoot and the bottom window's row H + 3 come from the zero frame).
 * ssm_wino7_pack_weights: OIHW fp32 7x7 filter -> U_b = G g_b G^T as [ceil(Cout/32)][Cin][14 quads][4 blocks][32][4] (+ bias, whole
 * 32-channel blocks).  Cout may be any positive count there: the channels up to the next multiple of 32 are packed as zeros, and the
 * convolution is then launched with that padded count on an output view that holds them (the data gradient of stage 2's conv1a).     */
int ssm_wino7_plan(int Cin, int Cout, int B, int H, int W, int *kind);
int ssm_wino7_force_kind(int kind);      /* tests / tuning only (-1 = automatic); returns the number of configurations */
size_t ssm_wino7_packed_weight_floats(int Cout, int Cin);
int ssm_wino7_pack_weights(const float *w_oihw, const float *bias, float *w_packed, float *bias_packed, int Cout, int Cin, void *stream);
int ssm_wino7_conv2d_add_fwd(ssm_view x, int Cin, const float *w_packed, const float *bias_packed, ssm_view y, ssm_view pool,
                             ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags, void *stream);

/* ---- the 5x5 convolutions as two-dimensional Winograd F(4x4,5x5), all arithmetic fp32 (v_mfma_f32_16x16x4_f32) ------------------
 * Same operator and operand layout as ssm_wino1d_conv2d_add_fwd for k = 5 (layers.conv, scripts/models/layers.py:21-33; conv2a /
 * conv2b of both U-Nets, scripts/models/flow_computation.py:43-45; fused 2x2 mean, layers.py:60-63): the eight points
 * {0, +-1, +-2, +-1/2, inf} of the 1-D F(4,5) form on BOTH axes - 64 multiplies per 16 outputs and (cin, cout) = 4 per output
 * against 10 for F(4,5) along x and 25 for the direct form (csrc/ssm_wino5.hip).  In fp32 the result differs from the
 * direct form by rounding only (a 64-channel layer: 3e-6 rms / 3e-5 max at unit output scale).
 * One input source; Cin a multiple of 4 (pad the view; pack with CinP), Cout a multiple of 32.  Inputs are padded planes; nothing
 * outside them is read (tile overshoot comes from the zero frame).
 * ssm_wino5_pack_weights: OIHW fp32 5x5 filter -> U = G g G^T as [Cout/32][CinP/4][16 quads][4 channels][32][4] (+ bias).          */
int ssm_wino5_plan(int Cin, int Cout, int B, int H, int W, int *kind);
int ssm_wino5_force_kind(int kind);      /* tests / tuning only (-1 = automatic); returns the number of configurations */
size_t ssm_wino5_packed_weight_floats(int Cout, int CinP);
int ssm_wino5_pack_weights(const float *w_oihw, const float *bias, float *w_packed, float *bias_packed, int Cout, int Cin, int CinP,
                           void *stream);
int ssm_wino5_conv2d_add_fwd(ssm_view x, int Cin, const float *w_packed, const float *bias_packed, ssm_view y, ssm_view pool,
                             ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags, void *stream);

/* ---- every fp32 filter of a U-Net repacked by one launch (training: the parameters change each optimizer step) ------------
 * A job = one convolution's OIHW parameter -> its packed form (algo: the per-layer pack entry point it replaces, same arithmetic,
 * same layout).  transposed: pack the DATA-GRADIENT filter of the forward parameter, W'[ci][co][ky][kx] = W[co][ci][k-1-ky][k-1-kx]
 * (the adjoint of layers.conv, scripts/models/layers.py:21-33, is the same convolution on it), read straight from the forward
 * tensor; Cout / Cin are then the data-gradient convolution's (Cout = the forward layer's input channels), bias NULL = zeros.
 * first = sum of max(total, nbias) over the preceding jobs.                                                                   */
enum { SSM_PACK_DIRECT = 0, SSM_PACK_WINO = 1, SSM_PACK_WINO1D = 2, SSM_PACK_WINO4 = 3, SSM_PACK_WINO7 = 4, SSM_PACK_WINO5 = 5 };
typedef struct {
    const float *w;      /* OIHW parameter */
    const float *bias;   /* or NULL */
    float *wp, *bp;      /* packed filter, packed bias */
    int Cout, Cin, CinP, k, BN, algo, transposed, nbias;
    long long first, total;
} ssm_pack32_job;
/* (a thread stores four packed elements: every job's `first` and `total` are multiples of 4 - BN is - and `wp` is 16-byte aligned) */
int ssm_pack32_weights_batch(const ssm_pack32_job *jobs_device, int n_jobs, long long total_elements, void *stream);
/* The F(2x2,3x3) jobs (algo 1) with k = 3, Cout % BN == 0, Cin % 16 == 0, BN <= 64 - most of a U-Net's bytes - by TILES of BN couts x 16
 * input channels: a workgroup reads its tile as contiguous rows and writes one contiguous run of the packed filter (the element-wise
 * kernel above fetches nine weights per thread from addresses Cin x 36 bytes apart).  Same job struct: `first` = the job's first tile,
 * `total` = its tiles = (Cout / BN) x (Cin / 16); max_bn = the largest BN of the table (LDS).  Bit-identical to the element-wise form. */
int ssm_pack32_wino_tiles_batch(const ssm_pack32_job *jobs_device, int n_jobs, long long total_tiles, int max_bn, void *stream);

/* ---- fp16-MFMA convolution on HL8 activations (v_mfma_f32_32x32x16_f16) ---------------
 * Same operator as ssm_conv2d_fwd.  Default mode evaluates a*b as a_hi*b_hi + a_hi*b_lo +
 * a_lo*b_hi with fp32 accumulation (fp32-grade products at 16/3 x the fp32-MFMA rate);
 * SSM_FLAG_FP16_FAST keeps only a_hi*b_hi (plain fp16 inputs; BASELINE config 5).
 * Filters are packed once with ssm_pack16_weights (scaled by a power of two `scale`;
 * pass wscale = 1/scale to the convolution).  Cin is padded to a multiple of 16.
 * Outputs: y_hl8 (ptr NULL = off) and/or y_f32 (ptr NULL = off), optional fused 2x2 mean
 * pool_hl8.                                                                               */
int ssm_conv16_config(int k, int Cout, int W, int *BN, int *KYS);
size_t ssm_packed16_weight_halves(int Cout, int Cin_padded, int k, int BN);
int ssm_pack16_weights(const float *w_oihw, const float *bias, void *w_packed, float *bias_packed, int Cout,
                       int Cin, int Cin_padded, int k, int BN, int KYS, float scale, void *stream);
int ssm_conv2d_hl8_fwd(ssm_hview x1, int C1, ssm_hview x2, int C2, const void *w_packed,
                       const float *bias_packed, float wscale, ssm_hview y_hl8, ssm_view y_f32,
                       ssm_hview pool_hl8, int B, int H, int W, int Cout, int k, float slope, int flags,
                       void *stream);
/* Fused  conv3x3( F.upsample(torch.cat([a, b], 1), size=(2h,2w), mode="bilinear") )  - the decoder step of
 * scripts/models/flow_computation.py:244-247 (and :103-137) - on LOW-res HL8 inputs a [B,C1,H/2,W/2] and
 * b [B,C2,H/2,W/2] (C2 may be 0; b may be batch-broadcast).  H, W = OUTPUT size.  The concatenated, upsampled
 * tensor is never materialised: expander waves rebuild it tile by tile in LDS under the matrix waves' MFMAs.
 * Filters: the ssm_pack16_weights packing for (k=3, Cout) (independent of KYS).                              */
int ssm_conv16_ups_config(int Cout, int W, int *BN, int *KYS);
int ssm_conv2d_ups_hl8_fwd(ssm_hview a, int C1, ssm_hview b, int C2, const void *w_packed,
                           const float *bias_packed, float wscale, ssm_hview y_hl8, ssm_view y_f32, int B, int H,
                           int W, int Cout, float slope, int flags, void *stream);
/* Q8 operand form (SSM_FLAG_Q8 on the two convolution entry points): plane 1 of every pixel group holds
 * [8 x fp8 e4m3 (x) | 8 x fp8 ((x - fp16(x)) * 2^11)] instead of 8 x fp16(lo); a product is a_hi*b_hi on the fp16 matrix path
 * plus fp8(a)*fp8(b_lo) + fp8(a_lo)*fp8(b) on the block-scaled fp8 path (v_mfma_scale_f32_32x32x64_f8f6f4, K = 4 taps x 16
 * channels, E8M0 scale 2^-11 on the lo operand) - 1 + 2*(1/4) fp16-MFMA units per product instead of 3.  Same geometry,
 * strides and entry points as HL8; filters are packed by ssm_pack16q_weights for the tile ssm_conv16q_config reports.      */
int ssm_conv16q_config(int k, int Cout, int W, int *BN, int *KYS);
int ssm_conv16q_ups_config(int Cout, int W, int *BN, int *KYS);      /* tile of ssm_conv2d_ups_hl8_fwd with SSM_FLAG_Q8 */
size_t ssm_packed16q_weight_bytes(int Cout, int Cin_padded, int k, int BN, int KYS);
int ssm_pack16q_weights(const float *w_oihw, const float *bias, void *w_packed, float *bias_packed, int Cout, int Cin,
                        int Cin_padded, int k, int BN, int KYS, float scale, void *stream);
int ssm_flowinterp_inputs_hq8_fwd(ssm_view img6, ssm_view flow4, const float *t, ssm_hview out16, ssm_view flows4, int B, int H,
                                  int W, void *stream);      /* ssm_flowinterp_inputs_hl8_fwd with a Q8 output */
/* Sub-pixel form of the decoder's `F.upsample(cat([x, skip]), scale 2, bilinear)` + 3x3 conv (scripts/models/flow_computation.py:
 * 215-289, flow_interpolation.py:220-281): conv3x3(upsample2x(X)) is a plain 3x3 convolution of the LOW-res tensors with 4*Cr
 * outputs - one effective filter M_a * W * M_b^T per output parity (a, b), packed by the caller as channel (2a + b)*Cr + c - and a
 * pixel-shuffle store: channel (2a + b)*Cr + c of low-res pixel (y, x) lands at pixel (2y + a, 2x + b) of channel c of y_hl8
 * (transposed: (2x + b, 2y + a)).  y_hl8 points at the destination pixel of this launch's low-res pixel (0, 0); H, W = low-res
 * extent of the launch.  The rows / columns 0 and last need their own effective filters (bilinear clamping vs the convolution's
 * zero padding): the caller overwrites them from strip launches (H = 1 views; columns through ssm_hl8_gather_cols copies).
 * Q8 operands only (flags must carry SSM_FLAG_Q8).                                                                          */
int ssm_conv2d_hl8_subpixel_fwd(ssm_hview x1, int C1, ssm_hview x2, int C2, const void *w_packed, const float *bias_packed,
                                float inv_wscale, ssm_hview y_hl8, int B, int H, int W, int Cr, int transposed, float slope, int flags,
                                void *stream);
/* The main problem, the four border strips and the four corners of a sub-pixel decoder level as ONE launch: every problem is one
 * ssm_conv2d_hl8_subpixel_fwd call (all with the filter packing of ssm_conv16q_config(3, 4*Cr, wcfg)); skip_y / skip_x make a
 * problem leave the first and last row / column of its extent to the problem that owns them, so the outputs are disjoint and the
 * problems need no ordering.  ssm_conv16_subpixel_plan fills `table_host` (ssm_conv16_subpixel_table_bytes(n) bytes; copy it to
 * device memory once - it holds the raw pointers of the views) and block_start[n+1]; ssm_conv16_subpixel_run launches it.     */
typedef struct ssm_subpixel_problem {
    ssm_hview x1;
    int C1;
    ssm_hview x2;
    int C2;
    const void *w_packed;
    const float *bias_packed;
    float inv_wscale;
    ssm_hview y_hl8;
    int H, W;
    int transposed;
    int skip_y, skip_x;
} ssm_subpixel_problem;
size_t ssm_conv16_subpixel_table_bytes(int n);
int ssm_conv16_subpixel_plan(const ssm_subpixel_problem *problems, int n, int B, int Cr, int wcfg, float slope, int flags,
                             void *table_host, size_t table_bytes, int *block_start);
int ssm_conv16_subpixel_run(const void *table_device, int n, const int *block_start, int Cr, int wcfg, int Cin, void *stream);
/* Image columns of the Ga + Gb groups of a two-source HL8 / Q8 tensor as the rows of a zero-framed tensor (record copies, both
 * planes): dst rows 0..ncols-1 = columns col0.., and if col1 >= 0 dst rows ncols+1..2*ncols = columns col1.. (row ncols is left
 * alone: the zero gap between the two column strips' neighbourhoods).                                                          */
int ssm_hl8_gather_cols(ssm_hview a, int Ga, ssm_hview b, int Gb, ssm_hview dst, int B, int H, int col0, int ncols, int col1,
                        void *stream);
/* All filters of a U-Net packed (Q8 form) by ONE launch: the training step repacks every filter after each optimizer step
 * (scripts/main.py:188-197 updates the nn.Conv2d weights of scripts/models/layers.py:21-33 in place).  A job describes one
 * ssm_pack16q_weights call; transposed = 1 packs the data-gradient filter W'[co][ci][ky][kx] = w[ci][co][k-1-ky][k-1-kx] straight
 * from the forward OIHW tensor `w` (Cout/Cin are those of W'), bias = NULL packs zeros.  block_start = running sum of
 * row_blocks + bias_blocks (ssm_pack16q_job_blocks) over the preceding jobs; the job array lives in device memory.        */
typedef struct ssm_pack16q_job {
    const float *w;
    const float *bias;
    void *wp;
    float *bp;
    int Cout, Cin, CinP, k, BN, KYS;
    float scale;
    int transposed;
    int block_start, row_blocks;
} ssm_pack16q_job;
int ssm_pack16q_job_blocks(int Cout, int CinP, int k, int BN, int *row_blocks, int *bias_blocks);
int ssm_pack16q_weights_batch(const ssm_pack16q_job *jobs_device, int n_jobs, int total_blocks, void *stream);
int ssm_hq8_from_f32(ssm_view src, ssm_hview dst, int B, int C, int G, int H, int W, void *stream);
int ssm_hq8_to_f32(ssm_hview src, ssm_view dst, int B, int C, int G, int H, int W, void *stream);
/* fp32 view [B,C,H,W] <-> HL8 with G >= ceil(C/8) channel groups (extra channels are zeros). */
int ssm_hl8_from_f32(ssm_view src, ssm_hview dst, int B, int C, int G, int H, int W, void *stream);
int ssm_hl8_to_f32(ssm_hview src, ssm_view dst, int B, int C, int G, int H, int W, void *stream);

/* HL8 forms of the concat+bilinear-x2 and of compute_inputs (same semantics as the fp32 entry
 * points below; Ga/Gb = channel groups of 8).  ssm_flowinterp_inputs_hl8_fwd also writes the four
 * approximated flow channels (Ft1^ u,v | Ft0^ u,v) as fp32 `flows4` for ssm_synthesize_fwd.    */
int ssm_upsample2x_cat_hl8_fwd(ssm_hview a, int Ga, ssm_hview b, int Gb, ssm_hview y, int B, int h, int w,
                               void *stream);
int ssm_flowinterp_inputs_hl8_fwd(ssm_view img6, ssm_view flow4, const float *t, ssm_hview out16,
                                  ssm_view flows4, int B, int H, int W, void *stream);

/* layers.avg_pool(2) (scripts/models/layers.py:60-63), unfused form. */
int ssm_avgpool2_fwd(ssm_view x, ssm_view y, int B, int C, int H, int W, void *stream);

/* y = F.upsample(torch.cat([a, b], 1), size=(2h,2w), mode="bilinear")
 * (align_corners=False; scripts/models/flow_computation.py:244-245 and the
 * upsample lambdas :92-137).  Cb may be 0.  y is [B,Ca+Cb,2h,2w].            */
int ssm_upsample2x_cat_fwd(ssm_view a, int Ca, ssm_view b, int Cb, ssm_view y, int B, int h, int w,
                           void *stream);

/* layers.warp (scripts/models/layers.py:73-120): backward bilinear warp,
 * zeros outside, flow = (u,v).  img [B,C,H,W], flow [B,2,H,W], out [B,C,H,W]. */
int ssm_warp_bilinear_fwd(ssm_view img, ssm_view flow, ssm_view out, int B, int C, int H, int W,
                          void *stream);

/* FlowInterpolationModel.compute_inputs (scripts/models/flow_interpolation.py:
 * 338-372), fused: flow approximation + 2 warps + 16-channel concat.
 * img6 [B,6,H,W] (I0|I1), flow4 [B,4,H,W] (F01|F10), t[B] in (0,1) on device,
 * out16 [B,16,H,W] in the reference's channel order.                        */
int ssm_flowinterp_inputs_fwd(ssm_view img6, ssm_view flow4, const float *t, ssm_view out16, int B,
                              int H, int W, void *stream);
/* The same, writing only the ten t-dependent channels 3:13 of out16 (warped frames + approximated flows): for plans that convolve
 * the frame channels 0:3 / 13:16 (flow_interpolation.py:364-367) once per pair from the pair itself and never read them from out16
 * (ssm_amd.engine.UNetPlan.hoist) - 80 instead of 104 B/px of algorithmic traffic.                                              */
int ssm_flowinterp_inputs_t_fwd(ssm_view img6, ssm_view flow4, const float *t, ssm_view out16, int B, int H, int W, void *stream);

/* extract_outputs + compute_output_image (scripts/models/flow_interpolation.py:
 * 374-429), fused: sigmoid visibility, refined flows, 2 warps, blend.
 * y3 [B,3,H,W].  aux (ptr NULL = off) [B,5,H,W] receives Ft1(2) | Ft0(2) | V0(1),
 * the intermediates FullModel returns (scripts/models/superslomo_r.py:142-150). */
int ssm_synthesize_fwd(ssm_view img6, ssm_view in16, ssm_view out5, const float *t, ssm_view y3,
                       ssm_view aux, int B, int H, int W, void *stream);

/* final_conv [+ synthesis]: Conv2d(32 -> NC, k3, pad 1, bias), no activation (scripts/models/flow_computation.py:145-153,
 * flow_interpolation.py:149-157), exact fp32: NC = 4 / 5 (the model's two filters) as v_fma_f32 chains in (cin, ky, kx) order,
 * every other NC (and $SSM_FINAL_VALU=0) on v_mfma_f32_4x4x1_16B_f32 (4 couts x 64 pixels per instruction).  x [B,32,H,W]
 * padded-plane view; w_oihw / bias the reference's state-dict tensors as they are (device fp32), NC <= 8.
 *   out (ptr NULL = off): [B,NC,H,W].
 *   y3 (ptr NULL = plain convolution): with NC = 5, extract_outputs + compute_output_image (flow_interpolation.py:374-429)
 *   run on the five sums in registers - arguments as ssm_synthesize_fwd, the 5-channel map is never written unless `out`
 *   is given too.                                                                                                       */
int ssm_final_conv_fwd(ssm_view x, const float *w_oihw, const float *bias, int NC, ssm_view out, ssm_view img6, ssm_view in16,
                       const float *t, ssm_view y3, ssm_view aux, int B, int H, int W, void *stream);

/* ---- launch programs (csrc/ssm_program.cpp) ------------------------------------------------------------------------
 * The reference's training loop (scripts/main.py:116-145,188-197) issues its step op by op through the framework; here a step is ~900
 * launches of this library with the SAME pointers, grids and arguments every time (the plans own every activation, gradient and packed
 * filter), so the host side of a step can be recorded once and replayed with a handful of calls (HIP graphs were measured slower than
 * eager issue on ROCm 7.2).
 *   ssm_program_create / destroy   a program handle
 *   ssm_program_begin   start recording with the pass's streams as slots 0..n-1 (n <= 8): from now on every launch of the library BY
 *                       THE CALLING THREAD is executed as usual AND appended to the program; launches of other host threads (the
 *                       reference's DataParallel replicas, scripts/main.py:74-76) run eagerly and stay out of it.  One recording at a
 *                       time per process: a second begin is refused (SSM_E_ARG, "another program is recording") - record later.  A
 *                       launch of the recording thread on a stream that is no slot makes ssm_program_end fail.
 *   ssm_program_mark    number of nodes recorded so far: a cut point for host-side work (framework kernels, collectives) that must run
 *                       between two ranges of nodes at replay
 *   ssm_program_end     stop recording; *n_nodes = nodes recorded
 *   ssm_program_run     issue nodes [first, last) again on `streams` (same count as at begin; normally the same handles); buffers
 *                       the recorded arguments point to must still be alive - they are the plans' own
 *   ssm_stream_wait     dst_stream waits for everything queued on src_stream so far (event record + wait).  The cross-stream ordering
 *                       of a pass (side stream of the weight gradients, VGG stream) must go through this call to be part of a program;
 *                       outside a recording it is a plain pair of HIP calls.                                                       */
int ssm_program_create(void **handle);
int ssm_program_destroy(void *handle);
int ssm_program_begin(void *handle, void *const *streams, int n_streams);
int ssm_program_mark(void *handle, int *n_nodes);
int ssm_program_end(void *handle, int *n_nodes);
int ssm_program_run(void *handle, int first, int last, void *const *streams, int n_streams);
int ssm_stream_wait(void *src_stream, void *dst_stream);

/* ---- frame formats either side of the path (uint8 HWC RGB on the device) ----------------
 * ssm_frames_from_u8_fwd: [N,H,W,3] uint8 -> normalised fp32 [N,3,Hp,Wp], image at (top,left),
 *   fusing ToTensor + Normalize + EvalPad (scripts/utils/dataloaders/augmentations.py:141-200;
 *   pad_before_norm=0: pad value 0 in normalised space) or the visualiser's load_batch +
 *   normalize_tensor (scripts/visualize_interpolation.py:61-88,257-262; pad_before_norm=1).
 *   mean3/std3 are HOST pointers to 3 floats (MODEL.PIXEL_MEAN / PIXEL_STD).
 * ssm_frames_to_u8_fwd: normalised fp32 [N,3,*,*] -> cropped [N,H,W,3] uint8: get_crop +
 *   denormalize + *255 + astype(uint8) (scripts/evaluate_interpolation_results.py:143-163,
 *   192-202).  mode 0 = the reference's cast (truncate, wrap mod 256); 1 = round + saturate. */
int ssm_frames_from_u8_fwd(const unsigned char *frames_hwc, ssm_view out, int N, int H, int W, int Hp, int Wp,
                           int top, int left, const float *mean3, const float *std3, int pad_before_norm,
                           void *stream);
int ssm_frames_to_u8_fwd(ssm_view in, unsigned char *frames_hwc, int N, int H, int W, int top, int left,
                         const float *mean3, const float *std3, int mode, void *stream);

/* ---- backward kernels of the training step (fp32 planes; BASELINE config 3) ------------------------------
 * The data gradient of a convolution is ssm_conv2d_fwd on the transposed, spatially flipped filter.
 * ssm_lrelu_bwd           dz = (dy + 1/4 dpool[y/2][x/2]) * (y > 0 ? 1 : slope)   (dy or dpool may be NULL views;
 *                         has_act = 0 for the bare final_conv)  - LeakyReLU' + adjoint of the fused 2x2 mean
 * ssm_bias_grad           db[c] = sum dz
 * ssm_conv2d_wgrad        dw[:, ci_offset:ci_offset+Cin] of an OIHW fp32 filter with cin_total inputs (zeroed first if
 *                         zero_first; two-source convs call it once per source) += sum_{b,y,x} dz * x(shifted); x padded planes
 * ssm_upsample2x_cat_bwd  adjoint of ssm_upsample2x_cat_fwd; acc_a/acc_b: add into da/db instead of overwriting
 * ssm_synthesize_bwd      adjoint of ssm_synthesize_fwd fused with d(L1 reconstruction) and, if stage2_terms, the two
 *                         refined-flow warp-loss terms (scripts/models/losses.py:160-161,217); c_rec[b], c_warp[b] =
 *                         lambda * upstream gradient / (3*H*W) per sample (device arrays); est4 = Ft1^ | Ft0^;
 *                         dy_extra (NULL view = none): gradient of further loss terms wrt the frame (the perceptual term)
 * ssm_flowinterp_inputs_bwd  adjoint of ssm_flowinterp_inputs_fwd wrt the stage-1 flows, fused with the two stage-1
 *                         warp-loss terms (losses.py:152-154) if stage1_terms                                   */
/* Adjoint of ssm_warp_bilinear_fwd (autograd of layers.warp, scripts/models/layers.py:73-120, as used by the loss terms
 * scripts/models/losses.py:152-161): dflow [B,2,H,W] (NULL view = not wanted) and dimg [B,C,H,W] (NULL view = not wanted;
 * ACCUMULATED into with atomics - zero it first - and it must share img's row stride).                                 */
int ssm_warp_bilinear_bwd(ssm_view img, ssm_view flow, ssm_view dy, ssm_view dflow, ssm_view dimg, int B, int C, int H, int W,
                          void *stream);
int ssm_lrelu_bwd(ssm_view dy, ssm_view dpool, ssm_view y, ssm_view dz, int B, int C, int H, int W, float slope, int has_act,
                  void *stream);
/* ssm_lrelu_bwd that also writes dZ in the Q8 operand form (dz_q8: an even number of channel groups >= ceil(C/8), zero-initialised
 * beyond C) - the input of the data-gradient convolution when the training step runs the fp16 + fp8 kernels.               */
int ssm_lrelu_bwd_q8(ssm_view dy, ssm_view dpool, ssm_view y, ssm_view dz, ssm_hview dz_q8, int B, int C, int H, int W, float slope,
                     int has_act, void *stream);
int ssm_bias_grad(ssm_view dz, float *db, int B, int C, int H, int W, void *stream);
/* db[c] += sum dz (no zeroing): the training step zeroes one flat buffer holding every parameter gradient of a U-Net once. */
int ssm_bias_grad_acc(ssm_view dz, float *db, int B, int C, int H, int W, void *stream);
int ssm_conv2d_wgrad(ssm_view x, ssm_view dz, float *dw_oihw, int B, int Cin, int Cout, int H, int W, int k, int cin_total,
                     int ci_offset, int zero_first, void *stream);
/* ssm_conv2d_wgrad that also ACCUMULATES the bias gradient db_acc[co] += sum_{b,y,x} dz (what autograd computes for nn.Conv2d.bias,
 * scripts/models/layers.py:21-33): the sum over pixels is one more column of the same GEMM (dZ times a row of ones), so the separate
 * ssm_bias_grad pass over dZ - 48 launches of a training step - disappears.  Call it for ONE of a two-source layer's sources.        */
int ssm_conv2d_wgrad_bias(ssm_view x, ssm_view dz, float *dw_oihw, float *db_acc, int B, int Cin, int Cout, int H, int W, int k,
                          int cin_total, int ci_offset, int zero_first, void *stream);
/* Same contract as ssm_conv2d_wgrad (what autograd computes for nn.Conv2d.weight, scripts/models/layers.py:21-33 trained by
 * scripts/main.py:138-197) on the bf16 matrix cores with split operands: v = bf16(v) + bf16(v - bf16(v)) + r, products
 * hi*hi + hi*lo + lo*hi accumulated in fp32 (dropped terms ~2^-17 relative; no scaling needed, bf16 has fp32's exponent).
 * The weight gradient of the f16f8 training plan; the exact-fp32 plan keeps ssm_conv2d_wgrad.                                 */
int ssm_conv2d_wgrad_bf16x3(ssm_view x, ssm_view dz, float *dw_oihw, int B, int Cin, int Cout, int H, int W, int k, int cin_total,
                            int ci_offset, int zero_first, void *stream);
/* Weight gradient of a 3x3 layer in the Winograd domain, F(2x2,3x3), fp32 throughout (csrc/ssm_wgradw.hip) - the same quantity as
 * ssm_conv2d_wgrad(k = 3) (autograd of nn.Conv2d.weight, scripts/models/layers.py:21-33 under scripts/main.py:138-197) for 16 instead of
 * 36 multiplies per 2x2 output tile, in two launches:
 *   ssm_conv2d_wgrad_wino   du[16][Cout][cin_total] (+ ci_offset) += sum_tiles (A dZ A^T) (.) (B^T x B)     fp32 atomics; du must be
 *                           zero before a step's first launch (ssm_wgrad_wino_scratch_floats floats; the finishing launch re-zeroes
 *                           it); db_acc (may be NULL): += sum dz, as ssm_conv2d_wgrad_bias; two-source layers call it once per source
 *   ssm_wgrad_wino_finish   for each job: dw_oihw[co][ci][3][3] += scale * G^T du[.][co][ci] G, du := 0.  `jobs` is a DEVICE array of
 *                           n_jobs ssm_wgradw_finish_job records (one per layer: a training step finishes a gradient bucket's layers
 *                           in one launch); max_n = the largest n among them
 * ssm_wgrad_wino_supported: 3x3, Cin and Cout >= 32, maps of 40+ pixels (below that K = tiles is too short against the 16 x Cout x
 * Cin partial sums every workgroup adds; those layers keep ssm_conv2d_wgrad).                                                      */
typedef struct ssm_wgradw_finish_job {
    float *du;       /* [16][Cout][cin_total] scratch of the layer                 */
    float *dw;       /* [Cout][cin_total][3][3] gradient, accumulated into         */
    int n;           /* Cout * cin_total                                           */
    int pad_;
} ssm_wgradw_finish_job;
int ssm_wgrad_wino_supported(int Cin, int Cout, int H, int W, int k);
long long ssm_wgrad_wino_scratch_floats(int Cout, int cin_total);
int ssm_conv2d_wgrad_wino(ssm_view x, ssm_view dz, float *du, float *db_acc, int B, int Cin, int Cout, int H, int W, int cin_total,
                          int ci_offset, void *stream);
int ssm_wgrad_wino_finish(const void *jobs_dev, int n_jobs, int max_n, float scale, void *stream);
int ssm_upsample2x_cat_bwd(ssm_view du, ssm_view da, int Ca, ssm_view db, int Cb, int B, int h, int w, int acc_a, int acc_b,
                           void *stream);
/* ... with the LeakyReLU' of the layer that produced `a` applied to da (after the optional accumulation): da = dZ of that layer, ya = its
 * output [B, Ca, h, w] - the a-source of a decoder level has no other consumer (scripts/models/flow_computation.py:244-247).               */
int ssm_upsample2x_cat_bwd_mask(ssm_view du, ssm_view da, int Ca, ssm_view db, int Cb, ssm_view ya, float slope, int B, int h, int w,
                                int acc_a, int acc_b, void *stream);
int ssm_synthesize_bwd(ssm_view img6, ssm_view est4, ssm_view out5, ssm_view target, const float *t, const float *c_rec,
                       const float *c_warp, ssm_view dy_extra, ssm_view dout5, ssm_view dest4, int B, int H, int W, int stage2_terms,
                       void *stream);
int ssm_flowinterp_inputs_bwd(ssm_view img6, ssm_view flow4, ssm_view din16, ssm_view dest4, const float *t, const float *c_warp,
                              ssm_view dflow4, int B, int H, int W, int stage1_terms, void *stream);

/* ---- perceptual loss (PerceptualLoss: vgg16.features[:23] + MSELoss, scripts/models/losses.py:12-41,172-181,218-233) --
 * The VGG convolutions are ssm_conv2d_fwd launches with SSM_FLAG_LRELU and slope 0 (= ReLU); their data gradients are the
 * same kernel on the transposed filters, ReLU' is ssm_lrelu_bwd with slope 0.  fp32 planes.
 *  ssm_maxpool2_fwd   y[B,C,H/2,W/2] = nn.MaxPool2d(2, 2)(x)                       (H, W = input size, even)
 *  ssm_maxpool2_bwd   dx = dy routed to the first maximum of each 2x2 window (row-major scan, as torch's backward)
 *  ssm_sqdiff_grad    out = coef[b] * (a - b): gradient of the per-sample mean squared feature difference
 *  ssm_sqdiff_mean    out[b] = mean over C x H x W of (a - b)^2, the per-sample MSELoss(reduce=False).mean of losses.py:218,227
 *                     (two deterministic launches; scratch: 64 * B device floats owned by the caller)                    */
int ssm_maxpool2_fwd(ssm_view x, ssm_view y, int B, int C, int H, int W, void *stream);
int ssm_maxpool2_bwd(ssm_view x, ssm_view dy, ssm_view dx, int B, int C, int H, int W, void *stream);
int ssm_sqdiff_grad(ssm_view a, ssm_view b, const float *coef, ssm_view out, int B, int C, int H, int W, void *stream);
int ssm_sqdiff_mean(ssm_view a, ssm_view b, float *scratch, float *out, int B, int C, int H, int W, void *stream);
/* The L1 terms of SSMLosses.forward (scripts/models/losses.py:113-170 warp terms with the FREEZE gating of :159-167, :218-233
 * reconstruction) of one window as per-sample sums, out[b][0] = sum |pred - target|, out[b][1] = sum of the (up to four) warp maps -
 * the caller divides by 3 H W and applies lambda_r / lambda_w.  img6 = [I0 | I1], flow4 = stage 1's [F01 | F10], est4 = the approximated
 * flows [Ft1 | Ft0] (channels 6:10 of the stage-2 input), out5 = stage 2's output, pred = the synthesised frame.  Two deterministic
 * launches; scratch: 128 * B device floats owned by the caller.                                                                     */
int ssm_train_loss_sums(ssm_view img6, ssm_view flow4, ssm_view est4, ssm_view out5, ssm_view pred, ssm_view target, float *scratch,
                        float *out, int B, int H, int W, int stage1_terms, int stage2_terms, void *stream);

/* ---- recurrent bottleneck (BOTTLENECK=CLSTM|CGRU; BASELINE config 4) -----------------------------------------
 * Replaces ConvBLSTM / ConvBGRU(in_channels=512, hidden_channels=512, kernel_size=(3,3), num_layers=2,
 * batch_first=True) as constructed at scripts/models/flow_computation.py:73-88 / flow_interpolation.py:73-88 and
 * called at flow_computation.py:208-211 / flow_interpolation.py:284-287 (`conv6(x_fwd, x_rev)`).  Their source is
 * an un-vendored submodule (.gitmodules:1-3, SreenivasVRao/ConvGRU-ConvLSTM-PyTorch, commit not recorded); the
 * cells below restate that package's published equations - parity UNPINNED.
 * The gate convolutions are ssm_conv2d_fwd / ssm_conv2d_hl8_fwd launches without activation: `gates_x` = the
 * filter's input-channel part applied to x_t (+ bias; batched over the sequence), `gates_h` = its hidden-channel
 * part applied to h_{t-1} (NULL view at the first step: h_0 = c_0 = 0).  All tensors [B,*,H,W]; Hc % 8 == 0.
 * Outputs go to fp32 planes (h_f32) and/or an HL8 view (h_hl8), either may be a NULL view.
 *  ssm_convlstm_cell_fwd   gates [i|f|o|g] (4*Hc ch): c' = s(f)*c + s(i)*tanh(g);  h' = s(o)*tanh(c')
 *  ssm_convgru_reset_fwd   gates [gamma|beta] (2*Hc ch): rh = s(gamma) * h          (input of the candidate conv)
 *  ssm_convgru_update_fwd  u = s(beta); h' = (1-u)*h + u*tanh(cand_x + cand_h)      (cand_* = Hc ch)            */
/* flags: SSM_FLAG_Q8 = write the HL8 output in the Q8 operand form (view starting at an even channel group, Hc % 16 == 0) */
int ssm_convlstm_cell_fwd(ssm_view gates_x, ssm_view gates_h, ssm_view c_prev, ssm_view c_next, ssm_view h_f32,
                          ssm_hview h_hl8, int B, int Hc, int H, int W, int flags, void *stream);
int ssm_convgru_reset_fwd(ssm_view gates_x, ssm_view gates_h, ssm_view h_prev, ssm_view rh_f32, ssm_hview rh_hl8, int B,
                          int Hc, int H, int W, int flags, void *stream);
int ssm_convgru_update_fwd(ssm_view gates_x, ssm_view gates_h, ssm_view cand_x, ssm_view cand_h, ssm_view h_prev,
                           ssm_view h_f32, ssm_hview h_hl8, int B, int Hc, int H, int W, int flags, void *stream);

/* Adjoints of the cells above (training through the recurrent bottleneck; the gate convolutions' own gradients are the conv
 * backward entry points).  fp32 views.
 *  ssm_convlstm_cell_bwd   d h', d c' (NULL = 0) -> dgates [4*Hc] and d c          (gates_h / c_prev NULL at the first step)
 *  ssm_convgru_reset_bwd   d(rh) -> dgates[0:Hc] (gamma) and d h;  the caller zeroes / owns dgates[Hc:2Hc]
 *  ssm_convgru_update_bwd  d h' -> dgates[Hc:2Hc] (beta), d cand and d h (h_prev / dh_prev NULL at the first step);
 *                          `gates` = the summed pre-activations [gamma|beta], `cand` = the candidate pre-activation          */
int ssm_convlstm_cell_bwd(ssm_view gates_x, ssm_view gates_h, ssm_view c_prev, ssm_view dh, ssm_view dc_next, ssm_view dgates,
                          ssm_view dc_prev, int B, int Hc, int H, int W, void *stream);
int ssm_convgru_reset_bwd(ssm_view gates, ssm_view h_prev, ssm_view drh, ssm_view dgates, ssm_view dh_prev, int B, int Hc, int H, int W,
                          void *stream);
int ssm_convgru_update_bwd(ssm_view gates, ssm_view cand, ssm_view h_prev, ssm_view dh_next, ssm_view dgates, ssm_view dcand,
                           ssm_view dh_prev, int B, int Hc, int H, int W, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SSM_HIP_H */
