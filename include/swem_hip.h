/*
 * swem_hip.h -- C ABI of libswem_hip.so: the MI355X (gfx950) implementation of the
 * SWEM inference hot path (sequential weighted EM memory matching + the ResNet
 * key/value encoders and mask decoder around it).
 *
 * The reference (lmm077/SWEM) is pure Python/PyTorch; it has no FFI.  The entry
 * points below are what a Python binding (ctypes) of the
 * reference's own call sites binds instead of the ATen ops those call sites launch.
 * Every function cites the reference file:line whose arithmetic it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless stated; the library allocates nothing,
 *     frees nothing and keeps no pointer after it returns;
 *   - `stream` is a hipStream_t passed as void*; kernels are only enqueued, never
 *     synchronised; the library holds no mutable global state besides the
 *     thread-local error string;
 *   - return value 0 = ok, <0 = error (SWEM_E_*); text via swem_last_error();
 *   - activations are NHWC fp32 ("hwc"): [B][H][W][C], C contiguous.  The reference
 *     tensors are NCHW; the Python side exposes NHWC memory as NCHW-shaped
 *     channels_last views, so no copy happens at the module boundary;
 *   - EM state uses the reference's own layouts: kappa [N][2][C][L], nu [N][2][V][L],
 *     zita [N][2][L] (reference (B,N,2,C,L) with B folded into N).
 */
#ifndef SWEM_HIP_H
#define SWEM_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWEM_ABI_VERSION 2   /* round 5: explicit fault-word arguments (see "Asynchronous faults") */

enum {
  SWEM_OK = 0,
  SWEM_E_SHAPE = -1,     /* unsupported / inconsistent sizes */
  SWEM_E_WORKSPACE = -2, /* workspace too small */
  SWEM_E_HIP = -3,       /* a HIP call failed */
  SWEM_E_ARG = -4        /* null pointer / bad flag */
};

/* conv flags */
enum {
  SWEM_CONV_RELU_IN = 1,  /* relu applied to the input while loading (ResBlock, networks.py:26-27) */
  SWEM_CONV_RELU_OUT = 2, /* relu after scale/shift/residual.  Both ReLUs are `v_max_f32 0, x` (IEEE maxNum): a NaN maps to 0,
                           * like ATen's clamp-based ReLU on finite activations -- a NaN that an out-of-range fp16 operand pair
                           * produced is therefore NOT visible in the output; SWEM_FAULT_RANGE reports it at its source */
  SWEM_CONV_GLU = 4,      /* two filter banks f,a: y = f * sigmoid(a) (modules.py:25-26) */
  /* training (swem_hip_train.h): the same kernels compute the DATA GRADIENT of a convolution.  x = dY [B][H][W][c],
   * w = the filters transposed to [Cin][KH][KW][Cout], Cout(arg) = Cin; the output is dX with the forward input's size
   * (H-1)*stride + KH - 2*pad (+1 row / column with EH / EW when the strided filter left one uncovered); stride 1|2 */
  SWEM_CONV_DGRAD = 8,
  SWEM_CONV_DGRAD_EH = 16,
  SWEM_CONV_DGRAD_EW = 32,
  SWEM_CONV_MASK_POS = 64 /* res is a mask source, not an addend: y = res > 0 ? y : 0 (gradient of an input ReLU) */
};

int swem_version(void);
const char *swem_last_error(void);
/* number of compute units of the current device (for launch heuristics / reports) */
int swem_device_cus(void);

/* ------------------------------------------------------------------------------------
 * Dense convolution as implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 * Replaces every nn.Conv2d (+ frozen BatchNorm2d, + ReLU, + residual add) call on the
 * path: networks.py:22-32,139-170,176-216; mod_resnet.py:58-113; modules.py:16-26;
 * swem.py:33.
 *   input   : up to three NHWC sources concatenated on the channel axis (x1/x2 may be
 *             NULL); bsK = elements between consecutive batch items of source K,
 *             0 = the same image for every batch item (replaces .expand(), swem.py:52-53,94-95)
 *   w       : [Cout'][KH][KW][Cin], Cin = c0+c1+c2 (must be a multiple of 4);
 *             Cout' = Cout, or 2*Cout grouped [Cout/32][2][32] when SWEM_CONV_GLU
 *   w_bs    : 0 = one filter bank for the whole batch (every nn.Conv2d); else elements between the banks of
 *             consecutive batch items (a batched GEMM: the value readout of matching, modules.py:272-273);
 *             then Ho*Wo must be a multiple of the row tile (128 is always safe)
 *   scale   : [Cout'] or NULL (=1)     -- folded BatchNorm  gamma/sqrt(var+eps)
 *   shift   : [Cout'] or NULL (=0)     -- conv bias and folded BatchNorm shift
 *   res     : NHWC [B][Ho][Wo][Cout] added after scale/shift, or NULL; res_bs as bsK
 *   y       : NHWC [B][Ho][Wo][Cout],  Ho = (H + 2*pad - KH)/stride + 1
 *   plan    : tiling hint, 0 = built-in heuristic; else  wm | wn << 4 | nsplit << 8 | math << 16  with wave tile
 *             (32*wm) x (32*wn) in {1x1, 1x2, 2x2}, nsplit K-splits, math 0 = fp32 MFMA, 1 = "bf16x6": operands
 *             split exactly into three bf16 terms while they are staged, six bf16 MFMA products, fp32 accumulation
 *             (fp32-level error, 2.7x the fp32-MFMA rate; swem_conv2d_nhwc_bf16x3 is the pre-split form), 2 = plain bf16
 *             (pre-split entry only: plane 0 = the operands rounded to nearest-even bf16, ONE product, fp32 accumulation --
 *             the mixed-precision training mode, config.AMP).  Modes 0 and 1 give identical results up to fp32 rounding; callers may time candidates once
 *             per layer shape and pass the fastest
 *   ws      : workspace for split-K partial sums (swem_conv2d_workspace bytes for the same plan)
 */
size_t swem_conv2d_workspace(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                             int flags, int plan);
int swem_conv2d_nhwc_f32(void *stream, const float *x0, int c0, long long bs0, const float *x1, int c1,
                         long long bs1, const float *x2, int c2, long long bs2, int B, int H, int W,
                         const float *w, long long w_bs, const float *scale, const float *shift, const float *res,
                         long long res_bs, float *y, int Cout, int KH, int KW, int stride, int pad, int flags,
                         int plan, void *ws, size_t ws_bytes);

/* Pre-split operands for the bf16x6 math mode: x [npix][C] fp32 (C % 8 == 0) -> out: three bf16 planes, each
 * channel-group major [C/8][npix][8], with x = hi + mid + lo exactly (three round-to-nearest bf16 terms = 24
 * significant bits); relu != 0 splits relu(x) (the input ReLU of networks.py:26-27 then costs nothing in the conv). */
int swem_split_bf16x3_f32(void *stream, const float *x, void *out, long long npix, int C, int relu);
/* Operands for the "f16x3" arithmetic (round 4): x -> TWO fp16 planes in the same channel-group-major layout with
 * x = hi + mid, round-to-nearest residual: 11 + 11 significant bits and the residual's sign, i.e. x to half an fp32 ulp
 * wherever |x| >= 2^-2 (mid a normal fp16 number), to 2^-25 absolute below (mid subnormal; the converter and the f16 MFMA
 * keep subnormals).  |x| must stay below 65520, the fp16 range -- the range the reference's own mixed-precision mode
 * (fp16 autocast, basic_trainer.py:83-86) runs these activations in, but NOT the reference's fp32 inference
 * (networks.py:22-32 has no range limit).  Beyond it the planes hold inf / NaN, and a consumer's ReLU epilogue
 * (v_max_f32 0, x) maps the resulting NaN accumulators to 0 -- a finite, wrong feature map.  Therefore EVERY producer of
 * an fp16 pair takes a `fault` word and ORs SWEM_FAULT_RANGE into it when a value it splits has |x| >= 65520 (inf and NaN
 * included; tested on the value that is split, i.e. behind an input / output ReLU): see "Asynchronous faults" below.
 * Everywhere this header takes a plane count (`nplanes`: 2 or 3 bf16 planes) the value SWEM_PLANES_F16 asks for this fp16
 * pair instead, bit-identical to this function on the fp32 values.
 * A convolution reads such planes with plan bit 18 set (SWEM_PLAN_F16) and math 3: the three products hi.hi + hi.mid +
 * mid.hi on v_mfma_f32_*_f16 -- the LDS image, MFMA count and kernel of "bf16x3", with 23-bit instead of 16-bit operands: the
 * dropped mid.mid term is 2^-24 relative, so the result carries fp32-level error (the accumulation's own rounding dominates:
 * tests/test_gpu_ops.py::test_conv2d_f16x3_mode measures it against float64 beside the fp32-MFMA and bf16x6 kernels).
 * The FILTER planes of that mode are the fp16 pair of w[n][:] * 2^e[n], e[n] chosen per output column so that the column's
 * largest weight lands in [2^13, 2^14) (small weights would otherwise leave `mid` subnormal); the caller passes
 * scale[n] * 2^-e[n] as `scale` (exact: a power of two). */
#define SWEM_PLANES_F16 4
#define SWEM_PLAN_F16 (1 << 18)
int swem_split_f16x2_f32(void *stream, const float *x, void *out, long long npix, int C, int relu, void *fault);
/* The same convolution as swem_conv2d_nhwc_f32 in bf16x6 math with PRE-SPLIT sources and filters: xK = plane 0 of
 * source K in the layout above (npix = all pixels of its storage; bsK = fp32-element batch stride as before, a
 * multiple of cK), psK = elements between its three planes (npix * cK), cK % 32 == 0;
 * w_bf16x3 = three planes [K/8][Cout'][8], K = KH*KW*Cin ordered (ci / 32, ky, kx, ci % 32) over the concatenated input
 * channels ("channel-block major": the KH*KW taps of a 32-channel block are consecutive k-blocks, so a tile's activations
 * are fetched from HBM once instead of once per tap); KH*KW <= 64.
 * Every fragment goes HBM/L2 -> LDS by buffer-load-to-LDS (no register staging, no vector arithmetic in the k-loop).
 * Output, scale/shift/residual/ReLU-out/GLU, plan and workspace as swem_conv2d_nhwc_f32.  The plan's math field picks the
 * arithmetic on these planes: 1 (or 0) = bf16x6, all three planes, six products (fp32-level error); 3 = "bf16x3", planes
 * hi and mid only, the three products hi.hi + hi.mid + mid.hi (16 significant bits per operand, ~2^-16 relative per
 * product; half the MFMA work and two thirds of the LDS); 2 = plain bf16 (plane 0, one product: config.AMP).
 * Plan bits 20-23 pick a kernel variant: 0 = the tile's default LDS ring (three stages for 64x64, two otherwise),
 * 1 = the other stage count, 2 / 3 = the 128x128 tile on eight waves with two / three stages, 4 / 6 = variants 0 / 2 on
 * v_mfma_f32_16x16x32_bf16 instead of 32x32x16, 8 / 9 = the 128x128 tile on four waves with 16-k blocks and two / three stages
 * (half the LDS per stage: three / two blocks per CU instead of one); bits 24-27 = K-split factor of the last, partly filled round of tiles; bits 28-29 = log2 of the XCD partition of the
 * N tiles (1-3: 2 / 4 / 8 groups of N tiles, 8/groups XCDs per group, each XCD reading only its group's filters; 0: chosen
 * by the traffic model groups * activations + (8 / groups) * filters; ignored where the N tiles do not divide).
 * Plan tile wm = wn = 4 (f16x3 plans only: math 3 + SWEM_PLAN_F16; no per-batch filters, bits 24-27 = 0): the 256-COLUMN tiles of
 * conv_t256_kernel (round 5) -- eight waves, ONE block per CU, 128 KB of LDS, the next k-block's transfers requested a whole
 * k-block before anything waits for them, fragments read one step ahead of their MFMAs.  Bits 20-23 then give the tile HEIGHT:
 * 0 = 256 rows (wave grid 2 x 4; the only form with SWEM_CONV_GLU), 4 / 5 / 6 / 7 = 128 / 160 / 192 / 224 rows (wave grid 1 x 8),
 * chosen so that the tiles x K-split fill the chip's 256 CUs about once.  Same k order and products as the other tiles: without a
 * K-split the result is bit-identical to theirs.  swem_conv2d_workspace sizes the K-split workspace for the height given.
 * The same tile with math 1 (bf16x6, three planes, six products: all 24 operand bits) exists at ONE height: bits 20-23 = 4
 * (128 rows; no GLU) -- (128 + 256) rows x 3 planes fill the LDS. */
int swem_conv2d_nhwc_bf16x3(void *stream, const void *x0, int c0, long long bs0, long long ps0, const void *x1, int c1,
                            long long bs1, long long ps1, const void *x2, int c2, long long bs2, long long ps2, int B,
                            int H, int W, const void *w_bf16x3, const float *scale, const float *shift,
                            const float *res, long long res_bs, float *y, int Cout, int KH, int KW, int stride,
                            int pad, int flags, int plan, void *ws, size_t ws_bytes);

/* The two convolutions with a FUSED OPERAND SPLIT of their output: besides y, the epilogue writes y's bf16 planes in the
 * layout swem_split_bf16x3_f32 produces ([plane][Cout/8][M][8], M = B*Ho*Wo pixels, plane stride M*Cout elements), for the
 * convolutions that consume y pre-split -- no split launch and no re-read of y per consumer.
 *   planes       : planes of y, or NULL;  nplanes = 2 (hi, mid: every consumer runs bf16x3 / plain bf16) or 3
 *   planes_relu  : planes of relu(y) (for a consumer that applies its input ReLU while splitting, networks.py:26-27), or NULL
 * Bit-identical to swem_split_bf16x3_f32 on y.  Needs Cout % 8 == 0 (with SWEM_CONV_GLU: the planes of the gated output).
 * y may be NULL when at least one plane pointer is given: a PLANES-ONLY output, for a layer whose every consumer reads the
 * planes (conv1 -> conv2 -> conv3 inside a block, mod_resnet.py:58-113, networks.py:22-32): the fp32 map is not written. */
int swem_conv2d_nhwc_f32_planes(void *stream, const float *x0, int c0, long long bs0, const float *x1, int c1,
                                long long bs1, const float *x2, int c2, long long bs2, int B, int H, int W,
                                const float *w, long long w_bs, const float *scale, const float *shift, const float *res,
                                long long res_bs, float *y, int Cout, int KH, int KW, int stride, int pad, int flags,
                                int plan, void *ws, size_t ws_bytes, void *planes, int nplanes, void *planes_relu,
                                int nplanes_relu, void *fault);
int swem_conv2d_nhwc_bf16x3_planes(void *stream, const void *x0, int c0, long long bs0, long long ps0, const void *x1,
                                   int c1, long long bs1, long long ps1, const void *x2, int c2, long long bs2,
                                   long long ps2, int B, int H, int W, const void *w_bf16x3, const float *scale,
                                   const float *shift, const float *res, long long res_bs, float *y, int Cout, int KH,
                                   int KW, int stride, int pad, int flags, int plan, void *ws, size_t ws_bytes,
                                   void *planes, int nplanes, void *planes_relu, int nplanes_relu);
/* ... with caller-owned tile counters (round 3): `counters` (ncounters 32-bit words) must be ALL ZERO when the call is
 * enqueued and must not be used by another stream at the same time; the kernels leave every word zero again.  The K-split
 * (reduced by the last split of every tile, no reduce launch) and stream-K forms then need no memset launch per call.
 *
 * ASYNCHRONOUS FAULTS (round 4; round 5: an argument of its own).  The return code of a call only covers what is known when
 * it is enqueued.  `fault` (may be NULL: no reporting) points to ONE 32-bit word the CALLER owns, zero-initialised once and
 * never written by anything else: a kernel that detects one of the conditions below ORs the bit into it (agent-scope atomic)
 * and nothing on the device ever clears it -- it is deliberately NOT part of `counters`, which a caller may re-zero at the
 * head of a replayed HIP graph (a fault of replay k must still be there after replay k+1).  The caller reads it wherever it
 * synchronises anyway (swem_amd.ops.check_faults: every sequence boundary) and clears it itself.
 *   SWEM_FAULT_KSPLIT_WAIT / SWEM_FAULT_STREAMK_WAIT: a block's bounded wait for another block's partial tile expired (a
 *     producer that was preempted or never dispatched): the reduced tile is WRONG; the caller must also zero `counters`
 *     before the next call (a stale tile counter corrupts the next launch the same way).
 *   SWEM_FAULT_RANGE: a value written into an fp16 operand pair (SWEM_PLANES_F16) had |x| >= 65520: the planes hold inf / NaN
 *     and whatever consumes them is WRONG (possibly finite: a ReLU epilogue maps NaN to 0).  Every entry point that can write
 *     an fp16 pair takes `fault` for this; the caller re-runs the work in a bf16 / fp32 arithmetic (math 0 / 1), which has
 *     the reference's fp32 range. */
#define SWEM_FAULT_KSPLIT_WAIT 1   /* the reducing split of a tile gave up waiting for the other splits' partial tiles */
#define SWEM_FAULT_STREAMK_WAIT 2  /* a stream-K tile owner gave up waiting for a producer's partial tile */
#define SWEM_FAULT_RANGE 4         /* a value beyond the fp16 range went into an fp16 operand pair */
int swem_conv2d_nhwc_bf16x3_planes_ctr(void *stream, const void *x0, int c0, long long bs0, long long ps0, const void *x1,
                                   int c1, long long bs1, long long ps1, const void *x2, int c2, long long bs2,
                                   long long ps2, int B, int H, int W, const void *w_bf16x3, const float *scale,
                                   const float *shift, const float *res, long long res_bs, float *y, int Cout, int KH,
                                   int KW, int stride, int pad, int flags, int plan, void *ws, size_t ws_bytes,
                                   void *planes, int nplanes, void *planes_relu, int nplanes_relu, void *counters, size_t ncounters,
                                   void *fault);
/* ... and with the RESIDUAL given as operand planes instead of an fp32 map (round 4).  A ResNet block's output that only
 * convolutions and the NEXT block's residual add consume (mod_resnet.py:77-113: `out += identity`) need not exist as an fp32 map:
 * its producer writes planes only (y = NULL above) and this entry point reads the addend back from them -- hi + mid of the fp16
 * pair (exact in fp32: the value to 22-23 significant bits) or hi + mid + lo of three bf16 planes (exactly the value).
 *   res_planes : plane 0 of the residual, layout [Cout/8][res_npx][8]; res_ps = elements between its planes
 *   res_nplanes: SWEM_PLANES_F16 or 3 (two bf16 planes carry 16 bits only: refused)
 *   res_bs     : as `res_bs` of the fp32 form (elements between the batch items of the residual, 0 = one image for the batch)
 * No GLU, no SWEM_CONV_MASK_POS; everything else as swem_conv2d_nhwc_bf16x3_planes_ctr. */
int swem_conv2d_nhwc_bf16x3_planes_res(void *stream, const void *x0, int c0, long long bs0, long long ps0, const void *x1,
                                       int c1, long long bs1, long long ps1, const void *x2, int c2, long long bs2,
                                       long long ps2, int B, int H, int W, const void *w_bf16x3, const float *scale,
                                       const float *shift, const void *res_planes, long long res_ps, long long res_npx,
                                       int res_nplanes, long long res_bs, float *y, int Cout, int KH, int KW, int stride,
                                       int pad, int flags, int plan, void *ws, size_t ws_bytes, void *planes, int nplanes,
                                       void *planes_relu, int nplanes_relu, void *counters, size_t ncounters, void *fault);

/* A whole identity bottleneck block of the key encoder's ResNet-50 layer1 (mod_resnet.py:77-113 / torchvision v1.5
 * Bottleneck, networks.py:139-144: conv1x1 C -> C/4, bn, relu, conv3x3 C/4 -> C/4, bn, relu, conv1x1 C/4 -> C, bn, + x, relu;
 * stride 1, no down-sampling branch) in ONE launch, in the f16x3 arithmetic: the C/4-channel intermediates stay in the LDS.
 * As three launches of the convolution above the block runs at 0.3x of a plain copy of its input planes to its output planes
 * (profiles/r05_bottleneck_fusion_bound.json); csrc/bneck.hip.
 *   x_planes : the input as the fp16 pair of swem_split_f16x2_f32, [2][C/8][B*H*W][8], x_ps elements between the planes;
 *              also the identity (read back as hi + mid, like swem_conv2d_nhwc_bf16x3_planes_res)
 *   w1/w2/w3 : filter planes as swem_conv2d_nhwc_bf16x3 takes them with SWEM_PLAN_F16 (fp16 pairs [2][K/8][Cout][8], K ordered
 *              (ci / 32, ky, kx, ci % 32), columns scaled by powers of two that sK undo); sK / bK: folded BatchNorm scale / shift
 *   y        : fp32 NHWC output or NULL;  y_planes: the output's own fp16 pair (same layout as x_planes, y_ps apart) or NULL
 *   fault    : SWEM_FAULT_RANGE when an intermediate or the output leaves the fp16 range
 * C must be 256 (the layer1 geometry).  Same k order and product order as the convolution kernel's 16x16x32 form. */
int swem_bottleneck_f16x3(void *stream, const void *x_planes, long long x_ps, int B, int H, int W, int C, const void *w1,
                          const float *s1, const float *b1, const void *w2, const float *s2, const float *b2, const void *w3,
                          const float *s3, const float *b3, float *y, void *y_planes, long long y_ps, void *fault);

/* ------------------------------------------------------------------------------------
 * Pointwise / pooling / resampling kernels.
 */
/* (f - mean)/std, NCHW [B][3][H][W] -> NHWC [B][H][W][4] (4th channel 0).  networks.py:161
 * mean3/std3 are HOST pointers to 3 floats (the module's registered buffers, networks.py:157-158). */
int swem_prep_key_input_f32(void *stream, const float *frames, const float *mean3, const float *std3, float *out,
                            int B, int H, int W);
/* swem.py:48-53 + networks.py:115-117: per object n: [ (f-mean)/std, m[n+1], 1-m[n+1]-m[0], 0,0,0 ]
 * (single_obj: the "others" channel is 0 and the packed weight ignores it).
 * frame NCHW [B][3][H][W], masks [B][N+1][H][W] -> NHWC [B*N][H][W][8] */
int swem_prep_value_input_f32(void *stream, const float *frame, const float *masks, const float *mean3,
                              const float *std3, float *out, int B, int N, int H, int W, int single_obj);
/* The same inputs in SPACE-TO-DEPTH form, for conv1 as a 4x4 / stride-1 / pad-1 convolution on 32 channels (K = 512 for the
 * matrix cores instead of 7x7 / stride 2 on 3 or 5 channels): a 2x2 pixel block becomes one pixel of 4 phases x 8 channels,
 * channel (py*2 + px)*8 + c, c = (r, g, b, mask, others, 0, 0, 0) (masks == NULL: key form, (r, g, b, 0, ...), N = 1);
 * blocks stored behind one zero row and column: out NHWC [B*N][H/2+1][W/2+1][32] (may be NULL) and / or its bf16 planes
 * (layout of swem_split_bf16x3_f32, nplanes = 2 or 3 written).  Filters: w'[co][dy][dx][(py*2+px)*8 + c] =
 * w[co][2dy+py-1][2dx+px-1][c] (zero where the index is -1): swem_amd.ops.pack_stem_s2d.  H and W even. */
int swem_prep_input_s2d_f32(void *stream, const float *frame, const float *masks, const float *mean3, const float *std3,
                            float *out, void *planes, int nplanes, int B, int N, int H, int W, int single_obj, void *fault);
/* nn.MaxPool2d(3, 2, 1): mod_resnet.py:123.  NHWC, C % 4 == 0 */
int swem_maxpool3x3s2_nhwc_f32(void *stream, const float *x, float *y, int B, int H, int W, int C);
/* The same, and the result's bf16 planes for the pre-split convolutions that consume it (mod_resnet.py:123 -> layer1's
 * first block: conv1 and the downsample branch); arguments as swem_upsample_add_nhwc_f32_planes.  C % 8 == 0. */
int swem_maxpool3x3s2_nhwc_f32_planes(void *stream, const float *x, float *y, int B, int H, int W, int C, void *planes,
                                      int nplanes, void *planes_relu, int nplanes_relu, void *fault);
/* y = skip + bilinear(low -> Ho x Wo, align_corners=False): networks.py:193-194.  skip_bs 0 = shared skip */
int swem_upsample_add_nhwc_f32(void *stream, const float *skip, long long skip_bs, const float *low, float *y,
                               int B, int Hl, int Wl, int Ho, int Wo, int C);
/* The same, and the result's bf16 planes for the pre-split convolutions that consume it (layout of swem_split_bf16x3_f32):
 * planes = of y, planes_relu = of relu(y) (either may be NULL), nplanes* = 2 (hi, mid) or 3 planes written, each
 * B*Ho*Wo*C elements; C % 8 == 0.  Replaces a split launch (and its re-read of y) per consumer variant. */
int swem_upsample_add_nhwc_f32_planes(void *stream, const float *skip, long long skip_bs, const float *low, float *y,
                                      int B, int Hl, int Wl, int Ho, int Wo, int C, void *planes, int nplanes,
                                      void *planes_relu, int nplanes_relu, void *fault);
/* Both with `group` consecutive batch items sharing ONE skip image (item b adds skip image b / group; skip_bs = the stride between
 * skip images): the N objects of a clip share the clip's skip feature (swem.py:94-95 .expand) -- with several clips in one batch
 * (sequences in lock step, training clips) no per-object copy of the skip maps is materialised. */
int swem_upsample_add_grouped_nhwc_f32(void *stream, const float *skip, long long skip_bs, int group, const float *low, float *y,
                                       int B, int Hl, int Wl, int Ho, int Wo, int C);
int swem_upsample_add_grouped_nhwc_f32_planes(void *stream, const float *skip, long long skip_bs, int group, const float *low,
                                              float *y, int B, int Hl, int Wl, int Ho, int Wo, int C, void *planes, int nplanes,
                                              void *planes_relu, int nplanes_relu, void *fault);
/* F.interpolate / flip on NCHW planes; mode 0 = nearest (legacy), 1 = bilinear align_corners=False
 * (swem_evaluator.py:67,91), 2 = bicubic align_corners=False (swem_evaluator.py:43, basic_evaluator.py:160),
 * 3 = horizontal flip, same size (torch.flip(dims=[-1]), swem_evaluator.py:46-49) */
int swem_resize_planes_f32(void *stream, const float *x, float *y, int planes, int Hi, int Wi, int Ho, int Wo,
                           int mode);
/* swem.py:79-84: hard masks (int64 or float planes, channel 0 = background skipped) nearest ->
 * (h,w); soft masks bilinear -> (h,w); out [B*N][2][h*w] = {(1-h)(1-s), h*s} */
int swem_mask_prep_f32(void *stream, const void *hard, int hard_is_i64, int Hh, int Wh, const float *soft, int Hs,
                       int Ws, float *out, int B, int N, int h, int w);

/* CBAM (attentions.py:22-84) on NHWC x [B][H][W][C], fused with the residual use at networks.py:46-47:
 *   cscale[b][c] = sigmoid(mlp(avgpool(x)) + mlp(maxpool(x)))          ChannelGate
 *   sgate[b][p]  = sigmoid(conv7x7([max_c, mean_c](x*cscale)) + b7)    SpatialGate
 *   y = x + x * cscale * sgate                                         (block2 input = x + CBAM(x))
 * w1 [hid][C], b1 [hid], w2 [C][hid], b2 [C], w7 [2][7][7], b7 [1]; cscale [B][C] is also returned. */
size_t swem_cbam_workspace(int B, int H, int W, int C);
int swem_cbam_f32(void *stream, const float *x, const float *w1, const float *b1, const float *w2, const float *b2,
                  const float *w7, const float *b7, float *cscale, float *y, int B, int H, int W, int C, int hid,
                  void *ws, size_t ws_bytes);
/* The same, and y's bf16 planes for the pre-split convolution that consumes it (networks.py:46-48: block2 = ResBlock reads
 * relu(y) and y); plane arguments as swem_upsample_add_nhwc_f32_planes.  C % 8 == 0 when a plane pointer is given. */
int swem_cbam_f32_planes(void *stream, const float *x, const float *w1, const float *b1, const float *w2, const float *b2,
                         const float *w7, const float *b7, float *cscale, float *y, int B, int H, int W, int C, int hid,
                         void *ws, size_t ws_bytes, void *planes, int nplanes, void *planes_relu, int nplanes_relu, void *fault);

/* decoder.pred: conv3x3(relu(x)) -> 1 channel (networks.py:213).  x NHWC, w [3][3][C], logit [B][H][W] */
int swem_pred_head_f32(void *stream, const float *x, const float *w, const float *bias, float *logit, int B, int H,
                       int W, int C);
/* networks.py:215 + swem.py:99-106,110-116: bilinear -> (Ho,Wo), sigmoid, [prod(1-p), p] clamp, logit,
 * softmax over N+1, argmax.  logit4 [B*N][h4][w4]; valid [B][N+1] or NULL;
 * logits/prob [B][N+1][Ho][Wo]; argmax int64 [B][Ho][Wo] or NULL */
int swem_decode_head_f32(void *stream, const float *logit4, const float *valid, float *logits, float *prob,
                         long long *argmax, int B, int N, int h4, int w4, int Ho, int Wo);
/* swem_evaluator.py:83-87: argmax over N1 planes and its one-hot, both int64 */
int swem_argmax_onehot_i64(void *stream, const float *prob /*[B][N1][HW]*/, long long *argmax /*[B][HW]*/,
                           long long *onehot /*[B][N1][HW] or NULL*/, int B, int N1, long long HW);
/* torch.cat([x0, x1], dim=C) for NHWC x0 [B|1][P][c0], x1 [B|1][P][c1] (bs = 0: shared by the batch).
 * Only needed where a concatenated tensor is itself a residual (networks.py:44-45 with a ResNet-18 key
 * encoder); everywhere else the conv reads the sources directly. */
int swem_concat2_nhwc_f32(void *stream, const float *x0, int c0, long long bs0, const float *x1, int c1,
                          long long bs1, float *y, int B, long long P);
/* y = alpha*a + beta*b (b may be NULL): test-time-augmentation score averaging, swem_evaluator.py:49-53 */
int swem_lincomb_f32(void *stream, const float *a, float alpha, const float *b, float beta, float *y, long long n);
/* swem_evaluator.py:124-130 (YouTube-VOS: objects annotated from a later frame on): prob [B][N1][HW] scores,
 * new_masks [B][Nn1][HW] (channel 0 unused) -> out [B][N1+Nn1-1][HW]: scores zeroed where a new object is
 * annotated, new objects' masks appended as extra channels */
int swem_inject_objects_f32(void *stream, const float *prob, const float *new_masks, float *out, int B, int N1,
                            int Nn1, long long HW);
/* int64 index maps -> uint8 on the device, before the copy to the host (basic_evaluator.py:176) */
int swem_pack_u8_i64(void *stream, const long long *x, unsigned char *y, long long n);
/* batched 2-D transpose: in [batch][R][Cc] -> out [batch][Cc][ld] (ld >= R, pad columns zeroed) */
int swem_transpose_f32(void *stream, const float *in, float *out, int batch, int R, int Cc, int ld);

/* ------------------------------------------------------------------------------------
 * Sequential weighted EM (methods/SWEM/modules.py).  NK = 2*N (object-major, class minor).
 * Every operand is read as it lies in memory (pixel-major NHWC maps): there are no transposed copies.
 *   x     [P][C]      raw key of the frame, one row per pixel (reference x_t)
 *   v     [N][P][V]   value map per object, one row per pixel
 *   kp    [NK][C/4+1][L][4]  "packed keys": the key bases channel-group major, kp[nk][c/4][l][c%4] = kappa[nk][c][l]
 *                      (the GEMM kernels put one base on every lane: this keeps their loads coalesced), plus one more
 *                      group, kp[nk][C/4][l][0..3] = the squared norm of base l as C/32 partial sums (32 channels each).
 *                      The E/W and affinity kernels apply l2norm (modules.py:7-9) with them as the rows enter the
 *                      GEMM: kappa[:, l] / (|kappa[:, l]| + eps) -- so the M step, which writes both, is followed directly
 *                      by the next E step with no normalising kernel in between
 *   z     [N][Pz][2L] responsibilities, one row per PIXEL: z[n][p][cls*L + l], Pz = swem_em_pad(P) rows per object
 *                      (rows >= P are never read by the M step)
 * Key dimension C = 64 or 128 (the reference default KEYDIM is 128, configs/config.py:52); L = 64, 128 or 256; V a
 * multiple of 32.
 */
int swem_em_pad(int P); /* rows of a z buffer per object: P rounded up to a multiple of 128 */
/* l2norm over the channel dimension, modules.py:7-9: kn[nk][c/4][l][c%4] = kappa[nk][c][l] / (|kappa[nk][:][l]| + eps) */
int swem_em_norm_bases_f32(void *stream, const float *kappa /*[NK][C][L]*/, float *kn, int NK, int C, int L);
/* the packed keys kp of bases in the reference's layout ((C/4 + 1) * L * 4 floats per nk) */
int swem_em_pack_bases_f32(void *stream, const float *kappa /*[NK][C][L]*/, float *kp, int NK, int C, int L);
/* E and/or W step on one GEMM (they share x_t . l2norm(kappa)); kp = packed keys (normalised in the kernel):
 *   do_w: weights = masks * (1 - p_own)          modules.py:93-110  -> w_out [NK][P] (optional)
 *   do_e: z = softmax((s - rowmax)/tau) * weights modules.py:112-120 -> z
 *         (weights = result of do_w when set, else w_in [NK][P]) */
int swem_em_ew_f32(void *stream, const float *x, const float *kp, const float *masks /*[NK][P]*/,
                   const float *w_in, float *w_out, float *z, int N, int C, int P, int L, float tau, int do_w,
                   int do_e);
/* M step, modules.py:122-127 (R = C rows, A = x [P][C], a_per_object = 0) and the value update
 * modules.py:164-165 (R = V rows, A = v [N][P][V], a_per_object = 1), ONE launch:
 *   zita = zita_prev + sum_p z ;  out = (zita_prev * prev + A^T . z) / zita
 * prev/out [NK][R][L]; zita_prev/zita_out [NK][L]; kp_out (optional, key rows only): `out` as packed keys.
 * A block owns 16 bases x 32 rows over ALL pixels and sums its waves in a fixed order (deterministic, no atomics, no
 * partial sums in memory); R a multiple of 32.  The workspace query remains for callers of the earlier split-P version
 * and returns 0; ws may be NULL. */
size_t swem_em_mstep_workspace(int NK, int R, int P, int L);
int swem_em_mstep_f32(void *stream, const float *A, int a_per_object, const float *z, const float *prev,
                      const float *zita_prev, float *out, float *zita_out, float *kp_out, int NK, int R, int P,
                      int L, void *ws, size_t ws_bytes);
/* whole SWEMCore.swem() for one frame (modules.py:129-168): T x (E, M, W) + value update, two launches per iteration.
 *   v [N][P][V] NHWC value map; masks [N][2][P]; *_prev = prior bases (random_init output on frame 0) */
size_t swem_memorize_workspace(int N, int C, int V, int P, int L);
int swem_memorize_f32(void *stream, const float *x, const float *v, const float *masks, const float *kappa_prev,
                      const float *nu_prev, const float *zita_prev, float *kappa_out, float *nu_out,
                      float *zita_out, int N, int C, int V, int P, int L, int T, float tau, void *ws,
                      size_t ws_bytes);
/* The same with matching's PACKED banks kept current, so that swem_match_packed_f32 needs no per-frame repacking
 * (modules.py:295-306 `get_mem` concatenates the banks on every frame; here the caller owns one persistent pack):
 *   mkn [2N][C/4+1][2L][4] packed keys (see kp above) of both banks, rows [0,L) 'first', [L,2L) 'update'
 *   mvp [N][V][4L]        value bases, mvp[n][v][cls*2L + bank*L + l]
 *   mvq [N][2][4L/8][V][8] (optional, fp16) the same value bases pre-split for the readout GEMM: the fp16 pair hi, mid
 *                         of swem_split_f16x2_f32 (x = hi + mid to 22-23 significant bits; round 3: bf16, 16 bits),
 *                         k = cls*2L + bank*L + l in groups of 8 -- the filter layout of swem_conv2d_nhwc_bf16x3.
 *                         NULL: not kept (the readout then runs from mvp).  `fault`: SWEM_FAULT_RANGE when a value base
 *                         leaves the fp16 range (NULL: not reported -- a caller whose readout does not read mvq).
 * prior_packed != 0: the prior's packed keys are READ from the pack's 'update' half (written there by the previous
 * frame's call: kappa_prev must be that frame's kappa_out); the new bases are WRITTEN to bank `bank` (0 'first', 1 'update'). */
int swem_memorize_packed_f32(void *stream, const float *x, const float *v, const float *masks, const float *kappa_prev,
                             const float *nu_prev, const float *zita_prev, float *kappa_out, float *nu_out,
                             float *zita_out, float *mkn, float *mvp, void *mvq, int prior_packed, int bank, int N, int C,
                             int V, int P, int L, int T, float tau, void *ws, size_t ws_bytes, void *fault);

/* The same for the objects of SEVERAL clips in one call (round 6, sequences in lock step: evaluator.LockstepGraph): N objects in
 * total, N / clips per clip -- the reference's batch dimension, modules.py:129-168 with B = clips, every clip with its own key map:
 * x [clips][P][C].  v, masks, the bases and the pack are per object as above (the clips' objects back to back).  Per object the
 * launches run the same blocks on the same data as `clips` single-clip calls: identical results. */
int swem_memorize_packed_clips_f32(void *stream, const float *x, const float *v, const float *masks, const float *kappa_prev,
                                   const float *nu_prev, const float *zita_prev, float *kappa_out, float *nu_out,
                                   float *zita_out, float *mkn, float *mvp, void *mvq, int prior_packed, int bank, int N,
                                   int clips, int C, int V, int P, int L, int T, float tau, void *ws, size_t ws_bytes, void *fault);

/* The packed memorize in two calls (round 3).  modules.py:129-168 reads the value map only in its last statement
 * (:164-165, nu = (zita_ nu_ + mv z) / zita); everything before it -- every E, W and key M step, 2T - 1 of the 2T launches --
 * needs the key, the masks and the prior only, so a caller can run it BESIDE the value encoder that produces the value map
 * (swem_evaluator.py:91-93 runs encode_value and memorize one after the other).  `keys` leaves the last E step's
 * responsibilities in z [N][swem_em_pad(P)][2L] (caller-owned, not the workspace: the two calls may run on different
 * streams); `values` is the value update from them.  Same blocks on the same data as the one-call form: identical results. */
int swem_memorize_packed_keys_f32(void *stream, const float *x, const float *masks, const float *kappa_prev,
                                  const float *zita_prev, float *kappa_out, float *zita_out, float *mkn, float *z,
                                  int prior_packed, int bank, int N, int C, int P, int L, int T, float tau, void *ws,
                                  size_t ws_bytes);
int swem_memorize_packed_values_f32(void *stream, const float *v, const float *z, const float *nu_prev,
                                    const float *zita_prev, float *nu_out, float *mvp, void *mvq, int bank, int N, int V,
                                    int P, int L, void *fault);

/* ------------------------------------------------------------------------------------
 * Matching (modules.py:198-208, 232-289): l2norm, affinity, joint {bg,fg} softmax, value
 * readout and the top-l prefix-sum features, for all objects of one frame.
 *   qk [P][C] raw query key; banks: kappa_k [N][2][C][L], nu_k [N][2][V][L], k = first / update
 *   (update may be NULL on the first matched frame: Lm = L, else Lm = 2L)
 *   mem_out [N][Pm][V] with Pm = swem_match_pad(P) rows per object (rows >= P are zero),
 *   S [N][P][2*topl]  -- the NHWC sources of the fusion conv (modules.py:291)
 *   readout_plan: conv plan hint for the readout GEMM (0 = heuristic), see swem_conv2d_nhwc_f32
 */
int swem_match_pad(int P);
size_t swem_match_workspace(int N, int C, int V, int P, int L, int nbanks, int readout_plan);
int swem_match_f32(void *stream, const float *qk, const float *kappa_first, const float *nu_first,
                   const float *kappa_update, const float *nu_update, float *mem_out, float *S, int N, int C,
                   int V, int P, int L, int topl, float tau, int readout_plan, void *ws, size_t ws_bytes);

/* Matching on banks the caller keeps PACKED (layouts at swem_memorize_packed_f32): the reference concatenates and
 * normalises both banks on every frame (modules.py:282-283, 295-306); the 'first' bank never changes after frame 0 and the
 * 'update' bank's packed form is a by-product of memorize, so a persistent pack removes that work from the frame.
 * swem_match_pack_bank_f32 (re)builds one bank of a pack from the reference-layout bases (nbanks = 1: a pack of one bank;
 * mvq may be NULL).
 * Readout arithmetic (modules.py:272-273, mem_out = p . nu): readout_plan's math field 0 = fp32 matrix cores, 1 = bf16x6
 * (operands split in the kernel, fp32-level error), 3 (with or without SWEM_PLAN_F16) = the PRE-SPLIT readout in the f16x3
 * arithmetic -- the affinity kernel also writes the probabilities, times 2^14, as the fp16 pair hi / mid (p <= 1: unscaled,
 * most of a row would leave `mid` subnormal), the filters are the pack's mvq (required for this mode; without it the call
 * falls back to the fp32 kernel), three f16 products, the epilogue multiplies by 2^-14: 1e-7 from the fp32 readout. */
int swem_match_pack_bank_f32(void *stream, const float *kappa, const float *nu, float *mkn, float *mvp, void *mvq, int bank,
                             int nbanks, int N, int C, int V, int L, void *fault);
size_t swem_match_packed_workspace(int N, int C, int V, int P, int L, int readout_plan);
int swem_match_packed_f32(void *stream, const float *qk, const float *mkn, const float *mvp, const void *mvq, float *mem_out,
                          float *S, int N, int C, int V, int P, int L, int topl, float tau, int readout_plan, void *ws,
                          size_t ws_bytes);
/* The same, and the outputs' own bf16 planes for the pre-split convolution that consumes them (the fusion conv of
 * modules.py:286-291 reads cat([mem_out, qv, S])): mem_planes = planes of mem_out, [V/8][N*Pm][8] (Pm = swem_match_pad(P): the
 * padded rows are part of the pixel axis), N*Pm*V elements apart; s_planes = planes of S, [2 topl/8][N*P][8], N*P*2*topl apart;
 * *_nplanes = 2 or 3 written; either pointer may be NULL.  Bit-identical to swem_split_bf16x3_f32 on the fp32 outputs.
 * Only with the pre-split readout (mvq given and readout_plan's math field 3); V % 8 == 0, topl % 4 == 0. */
int swem_match_packed_f32_planes(void *stream, const float *qk, const float *mkn, const float *mvp, const void *mvq,
                                 float *mem_out, float *S, int N, int C, int V, int P, int L, int topl, float tau,
                                 int readout_plan, void *ws, size_t ws_bytes, void *mem_planes, int mem_nplanes, void *s_planes,
                                 int s_nplanes, void *fault);
/* Both for the objects of SEVERAL clips in one call (round 6): N objects in total, N / clips per clip, qk = one query key map per
 * clip [clips][P][C] (modules.py:232-293 with B = clips); packs and outputs per object.  Plane pointers may be NULL. */
int swem_match_packed_clips_f32(void *stream, const float *qk, const float *mkn, const float *mvp, const void *mvq, float *mem_out,
                                float *S, int N, int clips, int C, int V, int P, int L, int topl, float tau, int readout_plan,
                                void *ws, size_t ws_bytes, void *mem_planes, int mem_nplanes, void *s_planes, int s_nplanes,
                                void *fault);

#ifdef __cplusplus
}
#endif
#endif /* SWEM_HIP_H */
