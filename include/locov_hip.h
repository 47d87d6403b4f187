/*
 * locov_hip.h -- C ABI of the MI355X (gfx950) LSM ROI-head library, liblocov_hip.so.
 *
 * This is the drop-in boundary for LocOV's Localized-Semantic-Matching ROI-head hot
 * path.  LocOV itself is 100 % Python and reaches its kernels
 * through Detectron2 / torchvision / torch; every entry point below names the
 * reference interface (file:line under the LocOV tree) whose device arithmetic it
 * replaces.  The Python classes in locov_amd/ (same names, config keys and checkpoint
 * keys as the reference's) are the only callers; they bind these symbols with ctypes
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / HIP types in signatures.
 *   - Every pointer is a DEVICE pointer unless the parameter name ends in `_host`.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *     Calls only enqueue work on that stream: no allocation, no synchronisation, no
 *     hidden global state -> safe to capture in a hipGraph.  The ONE exception is the opt-in
 *     measurement aid locov_gemm_timing_* at the end of this header: a process-global,
 *     mutex-guarded list of HIP events, OFF by default, recording nothing while the launch
 *     stream is being captured; locov_gemm_timing_read() synchronises on those events.
 *   - Outputs are caller-allocated.  Tensors are dense row-major in the stated shape.
 *   - Return value: 0 on success, <0 on error (LOCOV_ERR_*); locov_last_error() returns
 *     a thread-local message for the last failing call on this thread.
 *   - Threading: re-entrant; one call stream per thread as with any HIP stream.
 */
#ifndef LOCOV_HIP_H
#define LOCOV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOCOV_ABI_VERSION 8

#define LOCOV_OK 0
#define LOCOV_ERR_INVALID_ARG (-1)
#define LOCOV_ERR_LAUNCH (-2)
#define LOCOV_ERR_UNSUPPORTED (-3)

/* element types for the mixed-precision entry points */
#define LOCOV_F32 0
#define LOCOV_BF16 1

/* row-normalisation modes (locov_rownorm_fwd) */
#define LOCOV_NORM_NONE 0
#define LOCOV_NORM_L2 1          /* normalize_vec   : logged_module.py:55-65 */
#define LOCOV_NORM_STANDARDIZE 2 /* standardize_vec : logged_module.py:68-72 */

/* epilogue flags of the GEMM entry points */
#define LOCOV_EPI_RELU 1u
/* locov_gemm_nt_f32_split*: x is ALREADY in the split layout of locov_split_f16x2_pack (written so by its producer),
 * scaled by x_scale: it is then staged by LDS DMA like W, with no conversion in the kernel */
#define LOCOV_GEMM_A_SPLIT 0x1000u
/* locov_gemm_nt_f32_split{,_segmean}: y is WRITTEN in the split layout of locov_split_f16x2_pack scaled by x_scale (the pre-split
 * x of the GEMM that consumes it: Res5 block outputs feed the next block's conv1, roi_emb_heads.py:217-245) / the residual
 * is READ from that layout (the same block output is the next block's identity shortcut).  hi + lo reproduces the fp32 value
 * to 2^-22 relative (exactly what the consuming GEMM's own split would keep); a finished value outside fp16's range raises the
 * range-guard word.  N and ldc must be multiples of 8. */
#define LOCOV_EPI_OUT_SPLIT 0x2000u
#define LOCOV_EPI_RES_SPLIT 0x4000u
#define LOCOV_SEGMEAN_RES_ROI_MAJOR 0x400u   /* locov_gemm_nt_f32_split_segmean: the residual rows are ROI-major */
/* locov_winograd_conv3x3_f32{,_split}: write the output rows ROI-major (row = roi * 49 + position) instead of
 * position-major (row = position * R + roi) -- the order locov_gemm_nt_f32_split_segmean consumes */
#define LOCOV_WINO_OUT_ROI_MAJOR 0x100u
/* ... and READ the input rows in that order */
#define LOCOV_WINO_IN_ROI_MAJOR 0x200u

typedef void *locov_stream_t;

int locov_abi_version(void);
const char *locov_last_error(void);

/* Kernel launches this library has enqueued in this process so far (every entry point checks its launches through one helper,
 * which counts them; an entry point that enqueues several kernels behind one check counts once per check).  A monotonic host-side
 * counter for diagnostics: locov_amd/sharding.GradientExchangeTrace places DistributedDataParallel's bucket-ready points on the
 * Res5 backward by it (ovr/engine/trainer.py:61-66), independent of what else shares the GPU. */
int64_t locov_launch_count(void);

/* Number of compute units / XCDs of the current device (for grid sizing in callers/bench). */
int locov_device_info(int *cu_count, int *wave_size, int *lds_bytes_per_cu);

/* ---------------------------------------------------------------------------------------
 * a-2  proposal -> FPN level assignment.
 * Replaces [D2-upstream] detectron2.modeling.poolers.assign_boxes_to_levels, reached from
 * ovr/modeling/roi_heads/roi_emb_heads.py:244 (self.pooler(features, boxes)) when the
 * pooler has >1 level.  boxes [R,4] XYXY fp32 -> levels [R] int64 in [0, max-min].
 * Bit-exact integer output (north_star).
 * ------------------------------------------------------------------------------------- */
int locov_level_assign(const float *boxes, int64_t R, int min_level, int max_level,
                       int canonical_box_size, int canonical_level, int64_t *levels,
                       locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-1  ROIAlign forward / backward, NCHW fp32 -- the stand-alone pooler contract.
 * Replaces [D2-upstream] ROIPooler -> ROIAlign -> torchvision.ops.roi_align, call site
 * ovr/modeling/roi_heads/roi_emb_heads.py:243-245 (_shared_roi_transform), pooler built
 * at :182-187 (output_size=14, scales=(1/16,), sampling_ratio=0, "ROIAlignV2").
 *   feat [N,C,H,W]; rois [R,5] = (batch_idx, x0, y0, x1, y1); out [R,C,pooled_h,pooled_w].
 *   sampling_ratio <= 0 -> adaptive ceil(roi/pooled) grid.  aligned != 0 -> ROIAlignV2.
 * Backward accumulates into grad_feat [N,C,H,W] (caller zeroes it) with fp32 atomics.
 * ------------------------------------------------------------------------------------- */
int locov_roi_align_fwd(const float *feat, int N, int C, int H, int W, const float *rois,
                        int64_t R, int pooled_h, int pooled_w, float spatial_scale,
                        int sampling_ratio, int aligned, float *out, locov_stream_t stream);

int locov_roi_align_bwd(const float *grad_out, int N, int C, int H, int W, const float *rois,
                        int64_t R, int pooled_h, int pooled_w, float spatial_scale,
                        int sampling_ratio, int aligned, float *grad_feat,
                        locov_stream_t stream);

/* Multi-level pooler in ONE launch: ROI r samples level levels[r] (output of
 * locov_level_assign).  Host arrays describe up to LOCOV_MAX_LEVELS feature maps that
 * share N and C.  Replaces the per-level nonzero / index_put_ loop of ROIPooler.forward. */
#define LOCOV_MAX_LEVELS 8
int locov_roi_align_levels_fwd(const float *const *feats_host, const int *H_host,
                               const int *W_host, const float *scales_host, int num_levels,
                               int N, int C, const float *rois, const int64_t *levels,
                               int64_t R, int pooled_h, int pooled_w, int sampling_ratio,
                               int aligned, float *out, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * MI355X-native layout path: channels-last feature map, channels on the
 * lanes, coalesced 16-byte gathers.
 *   locov_nchw_to_nhwc : [N,C,H,W] f32 -> [N,H,W,C] (f32 or bf16)
 *   locov_roi_align_nhwc_fwd : feat [N,H,W,C] -> out [R, ceil(ph/bin_stride), ceil(pw/bin_stride), C]
 *     bin_stride = 1: every bin.  bin_stride = 2: only even bins (ph, pw even) -- exactly
 *     the positions the stride-2 1x1 convs of Res5 block 0 read (STRIDE_IN_1X1=True,
 *     roi_emb_heads.py:217-241), so downstream results are unchanged.
 * ------------------------------------------------------------------------------------- */
int locov_nchw_to_nhwc(const float *in, int N, int C, int H, int W, void *out, int out_dtype,
                       locov_stream_t stream);

/* The pooler contract at HBM speed: same result as locov_roi_align_fwd, bit for bit
 * (out [R,C,pooled_h,pooled_w] fp32), but gathering from a channels-last COPY of the map
 * (feat_nhwc [N,H,W,C] fp32 from locov_nchw_to_nhwc; 17 MB per 1333x800 image, written once per
 * call) and transposing in LDS, so that both the taps and the output are fully coalesced. */
int locov_roi_align_from_nhwc_fwd(const float *feat_nhwc, int N, int H, int W, int C,
                                  const float *rois, int64_t R, int pooled_h, int pooled_w,
                                  float spatial_scale, int sampling_ratio, int aligned, float *out,
                                  locov_stream_t stream);

/* The same with a choice of arithmetic (the pooler of roi_emb_heads.py:182-187,243-245):
 *   LOCOV_ROIALIGN_EXACT : the call above -- torchvision's per-sample order, un-fused: bit-identical to the CPU oracle.
 *   LOCOV_ROIALIGN_FAST  : within 1e-5 of it (SURVEY.md 8d's ROIAlign gate).  Separable form: per bin and axis the samples'
 *     bilinear weights are summed per PIXEL, so a bin costs (gh+1)(gw+1) taps instead of 4 gh gw; and the proposal's
 *     output bins are cut into regions whose pixel rectangle x 32 channels is staged in LDS once ("LDS-staged proposal
 *     tiles"; a small box is one region) so that every tap is an LDS read.  A plan kernel computes both per ROI into
 *     `workspace` (locov_roi_align_plan_bytes(R) bytes, 16-byte aligned; unused by the exact mode).  ROIs whose samples lie
 *     more than a pixel apart (a forced sampling_ratio on a large box), whose grid exceeds 5 samples per axis, or pooled
 *     sizes above 16 take the exact arithmetic inside the same launch. */
#define LOCOV_ROIALIGN_EXACT 0
#define LOCOV_ROIALIGN_FAST 1
int64_t locov_roi_align_plan_bytes(int64_t R);
int locov_roi_align_from_nhwc_fwd_ex(const float *feat_nhwc, int N, int H, int W, int C,
                                     const float *rois, int64_t R, int pooled_h, int pooled_w,
                                     float spatial_scale, int sampling_ratio, int aligned, int mode,
                                     void *workspace, int64_t workspace_bytes, float *out,
                                     locov_stream_t stream);

int locov_roi_align_nhwc_fwd(const void *feat, int feat_dtype, int N, int H, int W, int C,
                             const float *rois, int64_t R, int pooled_h, int pooled_w,
                             float spatial_scale, int sampling_ratio, int aligned,
                             int bin_stride, int pos_major, void *out, int out_dtype,
                             locov_stream_t stream);

/* Same, with out_ld elements between consecutive output pixel rows (>= C): lets the rows be a
 * column block of a wider matrix (Res5 block 0 reads [conv2 output | pooled input] as one
 * K-concatenated GEMM operand). */
int locov_roi_align_nhwc_ld_fwd(const void *feat, int feat_dtype, int N, int H, int W, int C,
                                const float *rois, int64_t R, int pooled_h, int pooled_w,
                                float spatial_scale, int sampling_ratio, int aligned,
                                int bin_stride, int pos_major, void *out, int64_t out_ld,
                                int out_dtype, locov_stream_t stream);

/* Same again, for pooling a map that already went through a 1x1 convolution.  ROIAlign is linear, so
 * W . ROIAlign(F) == ROIAlign(W . F): Res5 block 0's conv1 and projection shortcut (both 1x1, stride 2 ==
 * the even bins; roi_emb_heads.py:217-241) are applied to the res4 MAP -- once per pixel instead of once
 * per ROI bin, 12x fewer rows at 1000 proposals per image -- and pooled here.
 *   feat_ld : elements between consecutive pixels of feat (>= C; the C channels pooled by this call may be
 *             a column block of a wider per-pixel vector, e.g. [conv1 | shortcut] outputs)
 *   ch_scale, ch_shift [C] (either may be null), relu: per-channel affine (FrozenBN) + ReLU applied to the
 *             pooled value, i.e. AFTER the pooling, exactly where the reference applies them. */
int locov_roi_align_nhwc_affine_fwd(const void *feat, int feat_dtype, int N, int H, int W, int C,
                                    int64_t feat_ld, const float *rois, int64_t R, int pooled_h,
                                    int pooled_w, float spatial_scale, int sampling_ratio,
                                    int aligned, int bin_stride, int pos_major,
                                    const float *ch_scale, const float *ch_shift, int relu,
                                    void *out, int64_t out_ld, int out_dtype,
                                    locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-4  spatial mean.  Replaces box_features.mean(dim=[2,3])
 * (roi_emb_heads.py:262,344,356).  x [R,C,HW] -> out [R,C].
 * channels_last == 1: x is [R,HW,C] (ROI-major pixel rows); == 2: x is [HW,R,C] (position-major).
 * ------------------------------------------------------------------------------------- */
int locov_spatial_mean_fwd(const float *x, int64_t R, int C, int HW, int channels_last,
                           float *out, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-5 / a-6 / a-8  y[M,N] = epi(x[M,K] . W[N,K]^T), fp32 in / fp32 accumulate on the
 * f32 MFMA pipe (exact fp32 products, no TF32).  nn.Linear layout (weight [out,in]).
 * Replaces self.bbox_pred(x) (box_emb_head.py:196), self.emb_pred(x) (:206) and, in fp32
 * mode, self.cls_score(x) (:211).
 *   epi: v = acc * scale[n] (if scale) + shift[n] (if shift) + residual[m,n] (if residual);
 *        ReLU if (flags & LOCOV_EPI_RELU).     shift == bias for nn.Linear.
 *   lda / ldc: row strides (elements) of x and y (>= K / >= N); residual uses ldc.
 * ------------------------------------------------------------------------------------- */
int locov_gemm_nt_f32(const float *x, int64_t lda, const float *W, const float *scale,
                      const float *shift, const float *residual, float *y, int64_t ldc,
                      int64_t M, int N, int K, unsigned flags, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-3  Res5 stage building blocks on channels-last pixel matrices.
 * Replace [D2-upstream] BottleneckBlock's conv2d + FrozenBatchNorm2d + ReLU (+ residual add)
 * as built by roi_emb_heads.py:217-241 and applied at :245,:323.
 *   - 1x1 convolutions are locov_gemm_nt_f32 on x [R*H*W, Cin] (weight [N,Cin,1,1] == [N,Cin]).
 *   - locov_conv3x3_nhwc_f32: 3x3 / pad 1 / stride 1 as an implicit GEMM (K = 9*Cin) over R
 *     independent HxW tiles; w_packed [N, 9*Cin] comes from locov_pack_conv3x3_weight
 *     ([N,Cin,3,3] -> k = (ky*3+kx)*Cin + c).  Same epilogue as locov_gemm_nt_f32.
 *     Row order of x / y / residual: pos_major == 0 -> ROI-major  row = r*H*W + (y*W + x);
 *                                    pos_major != 0 -> POSITION-major row = (y*W + x)*R + r.
 *     Position-major is the fast layout: tap validity is then uniform per workgroup and the
 *     zero-padding taps are skipped instead of multiplied (18 % of a 7x7 tile's MACs).
 *   - locov_frozen_bn_fold: scale = weight*rsqrt(var+eps), shift = bias - mean*scale, the
 *     per-channel affine FrozenBatchNorm2d applies; feeds the GEMM epilogue's scale/shift.
 * ------------------------------------------------------------------------------------- */
int locov_conv3x3_nhwc_f32(const float *x, int64_t R, int H, int W, int Cin, int pos_major,
                           const float *w_packed, const float *scale, const float *shift,
                           const float *residual, float *y, int N, unsigned flags,
                           locov_stream_t stream);

int locov_pack_conv3x3_weight(const float *w, int N, int Cin, void *out, int out_dtype,
                              locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-3  The same 3x3 / pad 1 / stride 1 convolution (BottleneckBlock.conv2 + FrozenBN + ReLU,
 * roi_emb_heads.py:217-241 applied at :245,:323) on 7x7 POSITION-major tiles, evaluated in
 * the minimal-filtering (Winograd) domain: each row of 7 outputs is an F(4,3) | F(3,3) pair,
 * 121 products per (ROI, cin, cout) instead of 361 real taps -> 3x fewer MFMA FLOPs.
 *   U  = locov_winograd_pack_weight(w [N,Cin,3,3])  -> [121, N, Cin] fp32 (transform in fp64)
 *   y[(p*R + r), n] = relu?( conv(x)[..] * scale[n] + shift[n] ),  x rows [(p*R + r), Cin], p = y*7+x
 *   ldy: elements between consecutive rows of y (>= N; y may be a column block of a wider matrix)
 *   workspace: locov_winograd_workspace_bytes(R, Cin, N) bytes of device memory, 16-B aligned
 *   flags: 0 or LOCOV_EPI_RELU.  Cin % 32 == 0, N % 4 == 0.
 * fp32 throughout; differs from the direct form by rounding only (~5e-6 of the activation
 * range over the whole stage; the 1e-4 logits gate holds, tests/test_gpu_winograd.py).
 * locov_gemm_nt_batched_f32 is the batched NT GEMM it runs on: problem b uses
 * x + b*stride_x, W + b*stride_w ([N,K] rows of K), y + b*stride_y (elements).
 * ------------------------------------------------------------------------------------- */
int64_t locov_winograd_workspace_bytes(int64_t R, int Cin, int N);

int locov_winograd_pack_weight(const float *w, int N, int Cin, float *U, locov_stream_t stream);

int locov_winograd_conv3x3_f32(const float *x, int64_t R, int Cin, const float *U,
                               const float *scale, const float *shift, float *y, int64_t ldy,
                               int N, unsigned flags, void *workspace, int64_t workspace_bytes,
                               locov_stream_t stream);

/* The same convolution with the 121 transform-domain GEMMs in split-operand arithmetic (see
 * locov_gemm_nt_f32_split below): U_split = locov_split_f16x2_pack of the [121*N, Cin] matrix U with
 * w_scale = u_scale; the transformed input is scaled by v_scale before its split. */
int locov_winograd_conv3x3_f32_split(const float *x, int64_t R, int Cin, const void *U_split,
                                     float u_scale, float v_scale, const float *scale,
                                     const float *shift, float *y, int64_t ldy, int N, unsigned flags,
                                     void *workspace, int64_t workspace_bytes, unsigned *overflow,
                                     locov_stream_t stream);

int locov_gemm_nt_batched_f32(const float *x, int64_t lda, int64_t stride_x, const float *W,
                              int64_t stride_w, float *y, int64_t ldc, int64_t stride_y,
                              int64_t M, int N, int K, int batch, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * f-4  Multi-token class scoring of the grounding predictor.  Replaces the part of
 * GroundingModule.forward after its token_score Linear (box_emb_grounding_head.py:163-225:
 * per-class padding, masked softmax / hardmax over the tokens, attention-weighted distance).
 *   sim [R, Ttot]      raw token similarities = emb . token_bank^T (locov_gemm_nt_f32)
 *   tok_off [K1]       first column of class k in sim;  num_tok [K1] its real token count
 *                      (0 = no tokens: attention 0, score -0, like the reference's bg row)
 *   gmin [1]           device scalar: minimum of the reference's padded similarity tensor
 *   scores [R, K1] = -sum_t att_t * dist_t;   att [R, K1, Tmax] (may be null)
 *   cosine: dist = (1 - sim)/T with NaN similarities zeroed; else dist = -sim/T.  Tmax <= 32.
 * ------------------------------------------------------------------------------------- */
int locov_token_attention_fwd(const float *sim, int64_t R, int Ttot, const int *tok_off,
                              const int *num_tok, int K1, int Tmax, float temperature,
                              int cosine, int hardmax, const float *gmin, float *scores,
                              float *att, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Opt-in reduced-precision form of the Res5 GEMMs (MODEL.ROI_BOX_HEAD.RES5_DTYPE: "bf16"; the default
 * and the parity path are fp32): bf16 operands on the bf16 MFMA pipe, fp32 accumulate, fp32 epilogue
 * (scale / shift / residual / ReLU as in locov_gemm_nt_f32) and fp32 output.  x / W / w_packed hold
 * bf16 bit patterns (locov_f32_to_bf16, locov_pack_conv3x3_weight(out_dtype = LOCOV_BF16)).
 * K % 8 == 0 (GEMM), Cin % 64 == 0 (conv).
 * ------------------------------------------------------------------------------------- */
int locov_gemm_nt_bf16(const uint16_t *x, int64_t lda, const uint16_t *W, const float *scale,
                       const float *shift, const float *residual, float *y, int64_t ldc,
                       int64_t M, int N, int K, unsigned flags, locov_stream_t stream);

int locov_conv3x3_nhwc_bf16(const uint16_t *x, int64_t R, int H, int W, int Cin, int pos_major,
                            const uint16_t *w_packed, const float *scale, const float *shift,
                            const float *residual, float *y, int N, unsigned flags,
                            locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Opt-in split-operand form of the fp32 Res5 GEMMs (MODEL.ROI_BOX_HEAD.RES5_DTYPE: "f16x2"; replaces the
 * same reference convolutions as locov_gemm_nt_f32, roi_emb_heads.py:217-245).  Every fp32 operand value v,
 * scaled by a power of two s that places the data in fp16's range, is represented as hi + lo with
 * hi = fp16(s v), lo = fp16(s v - hi) (22 significant bits; absolute floor 2^-25 / s), and x . W^T is formed
 * on the f16 matrix pipe as (hi.hi + hi.lo + lo.hi) / (s_x s_w) with fp32 accumulation: fp32 in, fp32 out,
 * a result within about two fp32 roundings per operand of the fp32-MFMA GEMM.  |s_x x| and |s_w w| must stay
 * below 65504 (x_scale = 16 covers |x| < 4094; choose w_scale from max |w|).  x is split on the fly; W is
 * split once by locov_split_f16x2_pack into `out`, a buffer of the SAME size as the fp32 matrix
 * (rows * K * 4 bytes; per row and group of 8 columns: 8 hi halves, then 8 lo halves).
 * K % 32 == 0; N, lda, ldc % 4 == 0; 16-byte aligned pointers.  Epilogue as locov_gemm_nt_f32.
 * RANGE GUARD: `overflow` (device pointer to one 32-bit word, or null) -- a launch in which any x value had
 * |x_scale * x| >= 65504 ORs 1 into it; the outputs such a value feeds are inf / NaN.  The word is only
 * ever set, never cleared: the caller zeroes it, enqueues any number of split launches and reads it once afterwards
 * (locov_amd's heads then repeat that call on the f32 MFMA).  Costs two v_max3_f32 per staged 16-byte chunk.
 * locov_split_f16x2_pack raises the same word when |w_scale * w| >= 65504, so that a caller may keep a weight's scale from
 * one optimizer step to the next (no host read of max |w| per step) and still learn when it stopped covering the data.
 * ------------------------------------------------------------------------------------- */
int locov_split_f16x2_pack(const float *w, int64_t rows, int K, int64_t ld, float w_scale, void *out,
                           unsigned *overflow, locov_stream_t stream);

int locov_gemm_nt_f32_split(const float *x, int64_t lda, const void *W_split, const float *scale,
                            const float *shift, const float *residual, float *y, int64_t ldc,
                            int64_t M, int N, int K, unsigned flags, float x_scale, float w_scale,
                            unsigned *overflow, locov_stream_t stream);

/* The last 1x1 convolution of Res5 fused with the spatial mean behind it (roi_emb_heads.py:245 -> :262,:344,:356):
 *   out[q, n] = mean over p < seg of relu?( scale[n] * (x[q*seg + p, :] . W[n, :]) + shift[n] + residual[p*R + q, n] )
 * with R = M / seg ROIs: x rows are ROI-major (LOCOV_WINO_OUT_ROI_MAJOR), the residual is the POSITION-major [M, N]
 * tensor of the previous block -- or, with LOCOV_SEGMEAN_RES_ROI_MAJOR in `flags`, a ROI-major one (row q*seg + p) --
 * and the [M, N] result is never written.  Per M-tile and ROI the kernel leaves column
 * sums in `workspace` (locov_gemm_segmean_workspace_bytes), a second small kernel adds the one or two partials of
 * each ROI in a fixed order (deterministic).  43 <= seg <= 128 (a 128-row tile holds at most four ROIs; Res5's 7x7
 * positions give 49), M % seg == 0, otherwise as locov_gemm_nt_f32_split.  M * N * 4 may pass 2^32 where the launch
 * takes the 256 x 256 tile (pre-split x, ROI-major residual, >= 1 024 tiles: the Res5 call of thousands of proposals); the
 * 128 x 128 form needs M * N * 4 < 2^32.  locov_gemm_segmean_supported says (1 / 0) whether a shape can be launched at
 * all -- the caller of roi_emb_heads.py:262,344,356 falls back to the unfused convolution + locov_spatial_mean_fwd. */
int locov_gemm_segmean_supported(int64_t lda, int64_t M, int N, int K, int seg, unsigned flags);
int64_t locov_gemm_segmean_workspace_bytes(int64_t M, int N);
int locov_gemm_nt_f32_split_segmean(const float *x, int64_t lda, const void *W_split, const float *scale,
                                    const float *shift, const float *residual, float *out, int64_t M,
                                    int N, int K, int seg, unsigned flags, float x_scale, float w_scale,
                                    void *workspace, int64_t workspace_bytes, unsigned *overflow,
                                    locov_stream_t stream);

/* `batch` independent problems (strides in fp32 elements; W_split problem b starts stride_w * 4 bytes * b in) */
int locov_gemm_nt_batched_f32_split(const float *x, int64_t lda, int64_t stride_x, const void *W_split,
                                    int64_t stride_w, float *y, int64_t ldc, int64_t stride_y,
                                    int64_t M, int N, int K, int batch, float x_scale, float w_scale,
                                    unsigned *overflow, locov_stream_t stream);

/* Measurement aid (bench.py's roofline block) -- process-global state, off by default (see Conventions): while
 * enabled, every GEMM-kernel launch made by this library (from any thread) is bracketed by HIP events on its launch
 * stream; launches on a stream that is being captured into a hipGraph are not recorded.  read() waits for them and
 * returns, for one kernel class, the number of launches, the sum of their durations (ms) and
 * the FLOPs they executed.  cls: 0 = gemm_nt_kernel<128x128, plain / batched> (1x1 convs, FCs,
 * Winograd-domain GEMMs), 1 = the position-major direct 3x3 conv, 2 = the other tile shapes,
 * 3 / 4 = classes 0 / 1 launched with bf16 operands, 5 = the split-operand GEMM on its 128x128 tile (gemm_split_kernel, every
 * form), 6 / 7 = the TN (weight-gradient) GEMM on the f32 MFMA / in split arithmetic; the split-operand GEMM on its 256x256 tile
 * (gemm_split_big_kernel) by launch kind: 8 = with the Winograd input transform in the epilogue
 * (locov_conv1x1_winograd_conv3x3_f32_split), 9 = one problem (the 1x1 convolutions), 10 = batched (the Winograd-domain GEMMs),
 * 11 = mean-fused (locov_gemm_nt_f32_split_segmean).
 * read_ex() also returns the ALGORITHMIC HBM bytes of those launches (every operand read once, every result written once;
 * stated for classes 5 and 8-11, 0 elsewhere); bytes may be null.
 * enable(on) clears what was recorded. */
int locov_gemm_timing_enable(int on);
int locov_gemm_timing_read(int cls, int64_t *launches, double *ms, double *flops);
int locov_gemm_timing_read_ex(int cls, int64_t *launches, double *ms, double *flops, double *bytes);

int locov_frozen_bn_fold(const float *weight, const float *bias, const float *running_mean,
                         const float *running_var, float eps, int C, float *scale, float *shift,
                         locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-7  row normalisation of [R,D] fp32: L2 (x / max(||x||, eps)) or standardise
 * ((x - mean) / (std_unbiased + eps)).  Replaces normalize_vec / standardize_vec
 * (logged_module.py:55-72) applied at box_emb_head.py:207-210 and to the bank at :223-232.
 * In-place allowed (y == x).
 * ------------------------------------------------------------------------------------- */
int locov_rownorm_fwd(const float *x, int64_t R, int D, int mode, float eps, float *y,
                      locov_stream_t stream);

/* Backward of locov_rownorm_fwd (training with a non-detached class predictor, box_emb_head.py:197-210):
 * grad_x from the forward INPUT x and grad_y.  L2 with |x| <= eps: the clamp is active, grad_x = grad_y / eps. */
int locov_rownorm_bwd(const float *x, const float *grad_y, int64_t R, int D, int mode, float eps,
                      float *grad_x, locov_stream_t stream);

/* fp32 -> bf16 (round-to-nearest-even, NaN preserved) for packing the text bank / embeddings */
int locov_f32_to_bf16(const float *x, int64_t n, uint16_t *y, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-8  region x text similarity GEMM, bf16 operands / fp32 accumulate on the bf16 MFMA pipe:
 *   logits[R,K1] = emb[R,D] . bank[K1,D]^T          (bias is identically zero,
 *   box_emb_head.py:234).  emb, bank are bf16 (uint16 storage).  Config 3 of BASELINE.json.
 * ------------------------------------------------------------------------------------- */
int locov_sim_gemm_bf16(const uint16_t *emb, const uint16_t *bank, int64_t R, int D, int K1,
                        float *logits, int64_t ldc, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-11  proposal labelling of a whole batch in one launch: the device half of SampleAllROIHeads.label_and_sample_proposals
 * (ovr/modeling/roi_heads/roi_emb_heads.py:25-118) -- [D2-upstream] pairwise_iou of every proposal with ITS image's ground truth,
 * Matcher (intervals [thr_lo[k], thr_hi[k]) -> thr_label[k] in {-1, 0, 1}; the first maximum wins, as torch.max), the class
 * labels of ROIHeads._sample_proposals (matched class / num_classes for background / -1 for ignored), the sort keys of the
 * sampler (rnd [2, total] uniform doubles: key = rnd + 2 [not in the population] + 4 image -- image-major, population first,
 * uniformly random inside it) and, per image, rows[i] = {foreground count, background count, entries of the quality matrix that
 * fail `>= 0` (the Matcher's assert), foreground boxes without positive width and height (Box2BoxTransform's assert)}.
 *   boxes [total, 4] / gt_boxes [sum M, 4] XYXY fp32 and gt_classes [sum M] int64 on the device, concatenated over the images;
 *   prop_offsets / gt_offsets: n_images + 1 HOST ints (rows of image i: [off[i], off[i+1])); rows [n_images, 4] int64, ZEROED by
 *   the caller; gt_index [total] int64 = matched row of the CONCATENATED ground truth (0 for an image without any);
 *   labels [total] int64; key_pos / key_neg [total] double.  At most LOCOV_LABEL_MAX_IMAGES images, LOCOV_LABEL_MAX_THRESHOLDS
 *   intervals.  Bit-identical to the torch-op form (tests/test_gpu_roi_heads.py).
 * ------------------------------------------------------------------------------------- */
#define LOCOV_LABEL_MAX_IMAGES 64
#define LOCOV_LABEL_MAX_THRESHOLDS 6
int locov_label_proposals(const float *boxes, const int *prop_offsets, const float *gt_boxes, const int64_t *gt_classes,
                          const int *gt_offsets, int n_images, const float *thr_lo, const float *thr_hi, const int *thr_label,
                          int n_thresholds, int64_t num_classes, const double *rnd, int64_t *gt_index, int64_t *labels,
                          double *key_pos, double *key_neg, int64_t *rows, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-11  the sampler behind that labelling, for a batch in which every image fills its budget: the rest of
 * SampleAllROIHeads.label_and_sample_proposals (ovr/modeling/roi_heads/roi_emb_heads.py:79-106: [D2-upstream]
 * ROIHeads._sample_proposals -> subsample_labels, then `proposals_per_image[sampled_idxs]`, gt_classes, the matched target's
 * fields, fg_proposal) in ONE launch, without a host read.  Image i takes num_pos = min(rows[i][0], max_pos) foreground proposals
 * in ascending key_pos, then budget - num_pos background proposals in ascending key_neg (the heads of the image's segments of
 * the two global argsorts the torch form takes), into slots [i * budget, (i + 1) * budget) of every output:
 *   picked (row of the concatenated proposals), out_boxes [.,4], out_classes (= labels[picked]), out_gt_boxes [.,4] (the matched
 *   ground-truth box; zeros for an image without ground truth), out_fg (class != num_classes), rois [.,5] = (image, box) -- the
 *   pooler's input format, convert_boxes_to_pooler_format of the sampled boxes -- and, when `field` is given, field_out =
 *   field[picked] (one more fp32 per-proposal field: objectness_logits).
 * All inputs are locov_label_proposals' outputs / inputs of the same batch, on the same stream.  A batch in which an image has
 * fewer candidates than its budget gets in-range but meaningless rows there: the caller learns it from `rows` (its one host read)
 * and samples that batch the host-driven way.  1..4096 proposals per image, ties between equal keys broken by row number.
 * ------------------------------------------------------------------------------------- */
int locov_sample_proposals(const double *key_pos, const double *key_neg, const int64_t *labels, const int64_t *gt_index,
                           const int64_t *rows, const float *boxes, const float *gt_boxes, const float *field,
                           const int *prop_offsets, const int *gt_offsets, int n_images, int budget, int max_pos,
                           int64_t num_classes, int64_t *picked, float *out_boxes, int64_t *out_classes, float *out_gt_boxes,
                           int64_t *out_fg, float *rois, float *field_out, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-11  the box-regression loss of the training heads in ONE launch.  Replaces the torch-op chain of [D2-upstream]
 * FastRCNNOutputLayers.box_reg_loss as the reference configures it (ovr/modeling/roi_heads/box_emb_grounding_head.py:278-279,
 * 370-374: "smooth_l1", SMOOTH_L1_BETA; Box2BoxTransform weights :368) -- get_deltas(proposal, matched ground truth) of the
 * foreground rows (0 <= gt_classes < num_classes), smooth-L1 (plain L1 below beta 1e-5) against the predicted deltas, summed in a
 * fixed order and divided by max(R, 1) -- and d loss / d pred_deltas from the same launch.
 *   proposal_boxes / gt_boxes [R, 4] XYXY fp32 (16-byte aligned), pred_deltas [R, ld] with ld = 4 (class-agnostic) or
 *   4 * num_classes, gt_classes [R] int64; loss [1]; dpred [R, ld] or NULL: written completely when ld == 4, otherwise only the four
 *   columns of a foreground row's class (the caller zeroes it).  Foreground boxes must have positive width and height (the
 *   labelling's validity bit, locov_label_proposals' rows[:, 3]).
 * ------------------------------------------------------------------------------------- */
int locov_box_reg_loss(const float *proposal_boxes, const float *gt_boxes, const float *pred_deltas, int64_t ld,
                       const int64_t *gt_classes, int64_t R, int64_t num_classes, float wx, float wy, float ww, float wh,
                       float smooth_l1_beta, float *loss, float *dpred, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-10  greedy NMS on the device.  Replaces [D2-upstream] torchvision.ops.nms as reached from
 * box_predictor.inference -> fast_rcnn_inference -> batched_nms
 * (ovr/modeling/roi_heads/roi_emb_heads.py:280,357).
 *   boxes_sorted [K,4] XYXY fp32, ALREADY in descending score order (class-aware NMS: the caller
 *   adds the per-class coordinate offsets first, exactly as torchvision's batched_nms does)
 *   keep [K] bytes (1 = kept), num_keep [1] int32 -- both on the device; no host sync.
 *   workspace: locov_nms_workspace_bytes(K) bytes of device memory.
 * ------------------------------------------------------------------------------------- */
int64_t locov_nms_workspace_bytes(int64_t K);

int locov_nms_sorted(const float *boxes_sorted, int64_t K, float iou_threshold, void *workspace,
                     unsigned char *keep, int *num_keep, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-10  the detection post-processing of the evaluation call, on the device, without a host read.  Replaces the torch-op chain
 * of [D2-upstream] FastRCNNOutputLayers.inference as the reference reaches it (ovr/modeling/roi_heads/roi_emb_heads.py:280,357;
 * evaluation runs it once per image: configs/coco_stt.yaml:50 TEST.IMS_PER_BATCH 1): Box2BoxTransform.apply_deltas with
 * class-agnostic deltas (ovr/modeling/roi_heads/box_emb_head.py:160-161 CLS_AGNOSTIC_BBOX_REG), Boxes.clip, `probs[:, :K] >
 * score_thresh`, class-wise NMS (batched_nms's coordinate shift, greedy in descending score order, IoU > nms_thresh dropped) and the
 * top `topk` detections per image in descending score order, ties in candidate (row, class) order.
 *   probs [R, ld_probs] fp32 = softmax of the logits (K foreground columns + background), deltas / proposal_boxes [R, 4] fp32
 *   (16-byte aligned), rows of image i = [row_offsets[i], row_offsets[i + 1]) (n_images + 1 HOST ints), image_hw: n_images HOST
 *   (height, width) pairs; wx..wh: Box2BoxTransform weights, scale_clamp its clamp of dw / dh.
 *   out_boxes [n_images, topk, 4], out_scores / out_classes / out_rows [n_images, topk] (rows: proposal index inside its image);
 *   counts_and_flags [n_images + 1] int32: detections per image, then LOCOV_DETECT_FLAG_* bits -- when one is set the outputs are
 *   not to be used (the caller runs the torch chain): NONFINITE = a decoded box or a probability is inf / NaN (the reference drops
 *   such proposals with a warning), OVERFLOW = an image has more than LOCOV_DETECT_MAX_CANDIDATES candidates.
 *   workspace: locov_detect_postprocess_workspace_bytes(R, n_images) bytes.  At most LOCOV_LABEL_MAX_IMAGES images, 16 383
 *   proposals per image, 32 767 classes.  Bit-identical to the torch chain (tests/test_gpu_postprocess.py).
 * ------------------------------------------------------------------------------------- */
#define LOCOV_DETECT_MAX_CANDIDATES 8192
#define LOCOV_DETECT_FLAG_NONFINITE 1
#define LOCOV_DETECT_FLAG_OVERFLOW 2
int64_t locov_detect_postprocess_workspace_bytes(int64_t R, int n_images);

int locov_detect_postprocess(const float *probs, int64_t ld_probs, int num_classes, const float *deltas, const float *proposal_boxes,
                             const int *row_offsets, const float *image_hw, int n_images, float wx, float wy, float ww, float wh,
                             float scale_clamp, float score_thresh, float nms_thresh, int topk, void *workspace, int64_t workspace_bytes,
                             float *out_boxes, float *out_scores, int64_t *out_classes, int64_t *out_rows, int *counts_and_flags,
                             locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-12  LSM grounding: word<->region alignment -> [caption, image] cost matrices.
 * Replaces the B^2-replicated chain of GroundingHead.forward
 * (ovr/modeling/mmss_heads/grounding_head.py:116-243) for LOCAL_METRIC "dot", ALIGNMENT
 * "softmax", GLOBAL_METRIC "aligned_local" (configs/coco_lsm.yaml).
 *   S [B*T, B*NR] = caption token embeddings [B*T, L] . region embeddings [B*NR, L]^T
 *     (one locov_gemm_nt_f32 call; the region embeddings are v2l_projection(region_features),
 *     grounding_head.py:111, itself a locov_gemm_nt_f32 call)
 *   caption_mask [B,T] = attention_mask * (1 - special_tokens_mask) as fp32, region_mask [B,NR] fp32
 *   cost_w2r / cost_r2w [B,B] (row = caption, column = image): global_dist_w2r / _r2w of :219-228,
 *     before the all-empty fix-up of :232-243 (a [B,B] select the host applies).
 * The backward returns dS given the gradients of the two cost matrices.
 * ------------------------------------------------------------------------------------- */
int locov_grounding_fwd(const float *S, int B, int T, int NR, const float *caption_mask,
                        const float *region_mask, float temperature, float *cost_w2r,
                        float *cost_r2w, locov_stream_t stream);

int locov_grounding_bwd(const float *S, int B, int T, int NR, const float *caption_mask,
                        const float *region_mask, float temperature, const float *grad_w2r,
                        const float *grad_r2w, float *grad_S, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-12  the cross-entropy tail of GroundingHead.forward on the [B, B] caption x image costs of locov_grounding_fwd, in ONE launch
 * (ovr/modeling/mmss_heads/grounding_head.py:239-251: pairs with neither words nor regions get max(cost) + 100; :273-290
 * diag(-log_softmax(-cost, dim 0 | 1)).mean(); :357-377 (cost.argmin(dim 0 | 1) == arange).mean()).
 *   cost_w2r / cost_r2w [B, B] (either may be NULL: ALIGN_WORDS_TO_REGIONS / ALIGN_REGIONS_TO_WORDS off), caption_mask [B, T],
 *   region_mask [B, NR] fp32; out8 = {CE choose caption, CE choose image, accuracy choose caption, accuracy choose image} for w2r,
 *   then for r2w.  _bwd: g_* = device scalars d L / d (each of the four CE values) or NULL (= 0); dcost_* [B, B].
 *   B <= LOCOV_GROUNDING_CE_MAX_B.
 * ------------------------------------------------------------------------------------- */
#define LOCOV_GROUNDING_CE_MAX_B 64
int locov_grounding_ce_fwd(const float *cost_w2r, const float *cost_r2w, const float *caption_mask, const float *region_mask, int B,
                           int T, int NR, float *out8, locov_stream_t stream);
int locov_grounding_ce_bwd(const float *cost_w2r, const float *cost_r2w, const float *caption_mask, const float *region_mask, int B,
                           int T, int NR, const float *g_w2r_caption, const float *g_w2r_image, const float *g_r2w_caption,
                           const float *g_r2w_image, float *dcost_w2r, float *dcost_r2w, locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Backward of the predictor's dense layers under autograd (locov_pool_fc_bwd, locov_sim_gemm_bwd; the
 * forward arithmetic they differentiate is box_emb_head.py:196 bbox_pred, :206 emb_pred, :211 cls_score).
 *
 * locov_pool_fc_bwd: x [R,C5] (the pooled features; the [R,C5,7,7] form composes with locov_spatial_mean_bwd),
 *   emb_w [D,C5], bbox_w [4,C5], grad_emb [R,D] and / or grad_deltas [R,4] (either may be null: a detached class
 *   predictor, roi_heads DETACH_CLASS_PREDICTOR, sends no grad_emb).  Outputs, each optional (null = not wanted):
 *   grad_x [R,C5] = grad_emb . emb_w + grad_deltas . bbox_w;  grad_emb_w [D,C5] = grad_emb^T x, grad_emb_b [D] = column
 *   sums; grad_bbox_w [4,C5], grad_bbox_b [4] likewise.  Deterministic (no atomics).  C5 % 4 == 0, D % 4 == 0.
 * locov_sim_gemm_bwd: grad_logits [R,K1], emb [R,D], bank [K1,D] -> grad_emb [R,D] = grad_logits . bank (any K1) and /
 *   or grad_bank [K1,D] = grad_logits^T emb (K1 % 4 == 0 only: the reference freezes the bank, box_emb_head.py:234-235,
 *   so no caller on the path asks for it).  fp32 (the bf16 similarity GEMM is an inference-only option).
 * workspace: device memory of at least *_workspace_bytes(...) bytes, 256-byte aligned (transposed weight copies and the
 * TN GEMM's partial tiles); R == 0 writes zero weight gradients and returns.
 * ------------------------------------------------------------------------------------- */
int64_t locov_pool_fc_bwd_workspace_bytes(int64_t R, int C5, int D);
int locov_pool_fc_bwd(const float *x, int64_t R, int C5, const float *emb_w, int D, const float *bbox_w,
                      const float *grad_emb, const float *grad_deltas, float *grad_x, float *grad_emb_w,
                      float *grad_emb_b, float *grad_bbox_w, float *grad_bbox_b, void *workspace,
                      int64_t workspace_bytes, locov_stream_t stream);
int64_t locov_sim_gemm_bwd_workspace_bytes(int64_t R, int D, int K1);
int locov_sim_gemm_bwd(const float *grad_logits, const float *emb, const float *bank, int64_t R, int D, int K1,
                       float *grad_emb, float *grad_bank, void *workspace, int64_t workspace_bytes,
                       locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-4...a-8 in one call: EmbeddingFastRCNNOutputLayers.forward (box_emb_head.py:179-212)
 * preceded by the spatial mean of its caller (roi_emb_heads.py:262,344,356).
 *   x        [R,C5,HW] fp32 (HW may be 1)            pooled  [R,C5]  (out; == x when HW==1 is allowed)
 *   emb_w    [D,C5], emb_b [D]                        emb     [R,D]   (out, after optional norm)
 *   bbox_w   [4,C5], bbox_b [4]                       deltas  [R,4]   (out)
 *   bank     [K1,D] fp32 (already normalised by set_class_embeddings if enabled)
 *   bank_bf16 optional packed copy; used when sim_dtype == LOCOV_BF16 (emb_bf16 [R,D] scratch)
 *   logits   [R,K1] (out)
 * Weights are read at call time (the reference re-assigns emb_pred.weight/bias to the
 * grounding head's parameters, distill_prop_mmss_gcnn.py:121-125).
 * ------------------------------------------------------------------------------------- */
int locov_box_head_fwd(const float *x, int64_t R, int C5, int HW, int channels_last,
                       const float *emb_w, const float *emb_b, const float *bbox_w,
                       const float *bbox_b, const float *bank, const uint16_t *bank_bf16,
                       int D, int K1, int norm_mode, int sim_dtype, float *pooled,
                       float *deltas, float *emb, uint16_t *emb_bf16, float *logits,
                       locov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a-3 under autograd: the backward pass of the Res5 stage (the LSM head TRAINS the Res5 convolution weights --
 * configs/coco_lsm.yaml:8 FREEZE_AT 0, FrozenBN only freezes the statistics -- through
 * roi_emb_heads.py:323 (whole grid) and :343-347 (sampled proposals); the reference gets these gradients from
 * cuDNN via torch autograd).  All tensors are channels-last pixel-row matrices as in the forward entry points.
 *
 * locov_gemm_nt_f32_ex       : locov_gemm_nt_f32 with an explicit row pitch of W (ldb) and an optional `mask`
 *                              [M, ldc]: the finished value is kept where mask > 0 and zeroed elsewhere -- the ReLU
 *                              backward of a saved activation fused into the data-gradient GEMM  dx = g . (s*W)
 *                              (W given transposed, locov_weight_transpose_scale).
 * locov_conv3x3_nhwc_f32_ex  : locov_conv3x3_nhwc_f32 with the same mask (3x3 data gradient = convolution with the
 *                              flipped, transposed filter, locov_conv3x3_weight_flip).
 * locov_winograd_conv3x3_f32_ex : the same for 7x7 tiles in the Winograd domain.
 * locov_gemm_tn_f32          : out[b][N,K] = row_scale[n] * sum_m a_b[m,n] * b_b[m,k] -- the weight gradient
 *                              dW = s * g^T x of a 1x1 convolution (contraction over the pixel rows; split over M,
 *                              partial tiles reduced in a fixed order: deterministic).  batch problems with strides
 *                              (elements); workspace: locov_gemm_tn_workspace_bytes(M, N, K, batch).
 * locov_winograd_wgrad_f32   : dw [N,Cin,3,3] of a 3x3 convolution over R 7x7 tiles, in the Winograd domain:
 *                              dU_f = ((A (x) A) g)_f^T ((BT (x) BT) x)_f (121 TN GEMMs), dw = row_scale * (G (x) G)^T dU.
 *                              flags: 0 (position-major rows) or LOCOV_WINO_IN_ROI_MAJOR (both x and g).
 * locov_im2col3x3_nhwc / locov_conv3x3_wgrad_unpack : the general-grid form of the same gradient
 *                              (dw_packed [N, 9*Cin] = g^T . im2col(x) through locov_gemm_tn_f32).
 * amax_out (locov_relu_mask, locov_spatial_mean_bwd, locov_gemm_nt_f32_split_ex, locov_winograd_conv3x3_f32_split_ex): optional
 *                              16-byte operand-scale slot, ZEROED by the caller; the launch folds max |the tensor it writes| into
 *                              word 2 (an atomic max on the bit pattern), and a later split GEMM that takes the slot as its
 *                              x_scale_dev / a_scale_dev derives the tensor's power-of-two scale from that word (words 0-1 stay
 *                              zero: "not reduced by locov_split_scale_from_amax") -- the gradient is not read a second time
 *                              just to find its range.
 * locov_relu_mask, locov_spatial_mean_bwd, locov_rows_stride2, locov_roi_align_nhwc_bwd : element-wise pieces --
 *                              ReLU backward, the mean's broadcast (roi_emb_heads.py:344) fused with it, block 0's
 *                              stride-2 pixel selection on the whole grid and its adjoint, and the adjoint of
 *                              locov_roi_align_nhwc_fwd (fp32 atomics into a zeroed channels-last map gradient).
 * ------------------------------------------------------------------------------------- */
int locov_gemm_nt_f32_ex(const float *x, int64_t lda, const float *W, int64_t ldb, const float *scale,
                         const float *shift, const float *residual, const float *mask, float *y, int64_t ldc,
                         int64_t M, int N, int K, unsigned flags, locov_stream_t stream);

int locov_conv3x3_nhwc_f32_ex(const float *x, int64_t R, int H, int W, int Cin, int pos_major,
                              const float *w_packed, const float *scale, const float *shift,
                              const float *residual, const float *mask, float *y, int N, unsigned flags,
                              locov_stream_t stream);

int locov_winograd_conv3x3_f32_ex(const float *x, int64_t R, int Cin, const float *U, const float *scale,
                                  const float *shift, const float *mask, float *y, int64_t ldy, int N,
                                  unsigned flags, void *workspace, int64_t workspace_bytes,
                                  locov_stream_t stream);

int64_t locov_gemm_tn_workspace_bytes(int64_t M, int N, int K, int batch);
int locov_gemm_tn_f32(const float *a, int64_t lda, int64_t stride_a, const float *b, int64_t ldb,
                      int64_t stride_b, float *out, int64_t ldo, int64_t stride_o, int64_t M, int N, int K,
                      int batch, const float *row_scale, void *workspace, int64_t workspace_bytes,
                      locov_stream_t stream);

/* The same gradients in the split-operand arithmetic of the forward (3 f16 MFMAs per fp32 product block instead of 16 f32
 * ones).  A gradient tensor has no a-priori range, so its operand scale is chosen ON THE DEVICE:
 *   locov_split_scale_from_amax : scale_out[0] = s = the power of two with max |s x| in [2^(target_log2-1), 2^target_log2), scale_out[1] = 1/s,
 *                                 scale_out[2..3] = scratch (max bits, arrival tickets; zero again afterwards); a 16-byte memset and
 *                                 ONE kernel (the workgroup drawing the last ticket writes the scale), no host read.  scale_out: 16 bytes.
 *   locov_split_scale_from_amax_zeroed : the same without the memset, for 16 bytes whose words 2..3 are ALREADY zero (fresh from a
 *                                 zeroed pool, or last used by one of these two calls).
 *   locov_gemm_nt_f32_split_ex  : locov_gemm_nt_f32_split with the epilogue `mask` of locov_gemm_nt_f32_ex and, when
 *                                 x_scale_dev is non-null, the operand scale of x read from it (x_scale is then ignored).
 *   locov_gemm_tn_f32_split     : locov_gemm_tn_f32 with split operands: a (the gradient) scaled by a_scale_dev[0], b (the
 *                                 activation) by b_scale; both are converted on their way into LDS.  Same workspace,
 *                                 same fixed-order chunk reduction, same range-guard word.
 *   locov_winograd_wgrad_f32_split : locov_winograd_wgrad_f32 with its 121 TN GEMMs in that arithmetic (the transformed
 *                                 gradient's scale is chosen on the device, the transformed activation is scaled by 0.25). */
int locov_split_scale_from_amax(const float *x, int64_t n, float target_log2, float *scale_out,
                                locov_stream_t stream);
/* An upper bound of max |.| of a tensor that is about to be derived from SMALL ones, without a pass over the large tensor:
 * folds  max_i (muls[i] * max |xs[i]|)  (count <= LOCOV_AMAX_BOUND_MAX device tensors of ns[i] elements, ns[i] % 4 == 0; host
 * arrays) into word 2 of `slot`, a ZEROED 16-byte operand-scale slot as for amax_out.  The training step's head of the
 * backward chain: the masked broadcast of the pooled gradient (locov_spatial_mean_bwd: g = grad / 49 where the activation is
 * positive) and the masked whole-grid gradient (locov_relu_mask) are bounded by their small inputs (roi_emb_heads.py:344,:323). */
#define LOCOV_AMAX_BOUND_MAX 4
int locov_amax_bound(const float *const *xs, const int64_t *ns, const float *muls, int count, float *slot,
                     locov_stream_t stream);
int locov_split_scale_from_amax_zeroed(const float *x, int64_t n, float target_log2, float *scale_out,
                                       locov_stream_t stream);
int locov_gemm_nt_f32_split_ex(const float *x, int64_t lda, const void *W_split, const float *scale,
                               const float *shift, const float *residual, const float *mask, float *y,
                               int64_t ldc, int64_t M, int N, int K, unsigned flags, float x_scale,
                               const float *x_scale_dev, float w_scale, unsigned *overflow,
                               float *amax_out, locov_stream_t stream);
int locov_gemm_tn_f32_split(const float *a, int64_t lda, int64_t stride_a, const float *b, int64_t ldb,
                            int64_t stride_b, float *out, int64_t ldo, int64_t stride_o, int64_t M, int N,
                            int K, int batch, const float *row_scale, const float *a_scale_dev,
                            float b_scale, unsigned *overflow, void *workspace, int64_t workspace_bytes,
                            locov_stream_t stream);
/* locov_winograd_conv3x3_f32_split with the output `mask` of locov_winograd_conv3x3_f32_ex and, with v_scale_auto != 0, the
 * scale of the transformed input chosen on the device (the input is a gradient: the data gradient of a 3x3 convolution).
 * y_split_scale > 0: y is written in the split layout of locov_split_f16x2_pack scaled by y_split_scale (N % 32 == 0, ldy == N,
 * no mask) -- the pre-split A operand (LOCOV_GEMM_A_SPLIT) of the 1x1 convolution that follows; out-of-range values raise
 * *overflow here.  (With a fixed v_scale the transformed input is handed to the batched GEMM in that layout as well.)
 * y_split_scale < 0 (no mask): y stays fp32 and is RANGE-CHECKED for the split GEMM that will read it at operand scale
 * -y_split_scale (|scale * y| >= 65504 raises *overflow here, one launch before that GEMM's own check: a caller that waits for
 * the guard word -- roi_emb_heads.py:343-347 under autograd -- can record its event in front of the stage's last launches). */
int locov_winograd_conv3x3_f32_split_ex(const float *x, int64_t R, int Cin, const void *U_split, float u_scale,
                                        float v_scale, int v_scale_auto, const float *scale, const float *shift,
                                        const float *mask, float *y, int64_t ldy, int N, unsigned flags,
                                        float y_split_scale, void *workspace, int64_t workspace_bytes,
                                        unsigned *overflow, float *amax_out, locov_stream_t stream);
int locov_winograd_wgrad_f32_split(const float *x, const float *g, int64_t R, int Cin, int N, unsigned flags,
                                   const float *row_scale, float *dw, unsigned *overflow, void *workspace,
                                   int64_t workspace_bytes, locov_stream_t stream);
/* ... with the transformed activation the FORWARD already made: v_split = the first 121 * R * Cin floats of the workspace that
 * locov_winograd_conv3x3_f32_split[_ex] was given for the same x (split layout x 0.25, one pass: R proposals) -- a training step
 * that keeps that workspace per bottleneck (roi_emb_heads.py:343-347 under autograd) saves the second input transform, and the TN
 * GEMMs stage the operand as it is (locov_gemm_tn_f32_split_b).  Cin % 8 == 0.  The same bits as locov_winograd_wgrad_f32_split.
 *   locov_gemm_tn_f32_split_b : locov_gemm_tn_f32_split whose b is ALREADY in the split layout of locov_split_f16x2_pack at scale
 *   b_scale (K, ldb, stride_b multiples of 8): transposed on its way into LDS, not converted; the same bits. */
int locov_winograd_wgrad_f32_split_v(const float *v_split, const float *g, int64_t R, int Cin, int N, unsigned flags,
                                     const float *row_scale, float *dw, unsigned *overflow, void *workspace,
                                     int64_t workspace_bytes, locov_stream_t stream);
int locov_gemm_tn_f32_split_b(const float *a, int64_t lda, int64_t stride_a, const float *b_split, int64_t ldb,
                              int64_t stride_b, float *out, int64_t ldo, int64_t stride_o, int64_t M, int N,
                              int K, int batch, const float *row_scale, const float *a_scale_dev,
                              float b_scale, unsigned *overflow, void *workspace, int64_t workspace_bytes,
                              locov_stream_t stream);
/* A bottleneck's first two convolutions in ONE call (roi_emb_heads.py:217-245: conv1 1x1 + FrozenBN + ReLU, then conv2 3x3 + FrozenBN
 * + ReLU?) in split arithmetic:  y = locov_winograd_conv3x3_f32_split_ex(relu(x . W1^T * scale1 + shift1), ...)  -- the same bits as
 * locov_gemm_nt_f32_split (LOCOV_GEMM_A_SPLIT | LOCOV_EPI_RELU) followed by that call, which is what it falls back to.  Where the shapes
 * fill the chip with 256x256 tiles (both operands pre-split, C % 256 == 0, >= 1 024 tiles) the 1x1 convolution's epilogue applies the
 * Winograd INPUT transform itself: the [49 R, C] pixel tensor between the two convolutions is never written or re-read.
 *   x_split [49 R, K] (row pitch ldx): split layout x x_scale, ROI-major rows (r*49 + position); W1_split [C, K] x w1_scale;
 *   U_split [121, N, C] x u_scale; v_scale: operand scale of the transformed pixels; flags: LOCOV_WINO_IN_ROI_MAJOR (required)
 *   | LOCOV_WINO_OUT_ROI_MAJOR | LOCOV_EPI_RELU (of the 3x3); y, ldy, y_split_scale as in locov_winograd_conv3x3_f32_split_ex.
 *   workspace: locov_conv1x1_winograd_workspace_bytes_for(R, K, C, N, ldx) bytes -- the transform-domain workspace, plus the
 *   [49 R, C] pixel scratch only when these shapes take the two-launch fallback; locov_conv1x1_winograd_workspace_bytes(R, C, N)
 *   is the upper bound over both forms (always sufficient). */
int64_t locov_conv1x1_winograd_workspace_bytes(int64_t R, int C, int N);
int64_t locov_conv1x1_winograd_workspace_bytes_for(int64_t R, int K, int C, int N, int64_t ldx);
/* The same for block 0, whose 1x1 convolution ran on the feature MAP (ROIAlign is linear): the pooler + FrozenBN + ReLU + conv2 in one
 * call -- locov_roi_align_nhwc_affine_fwd(bin_stride 2, ROI-major, relu) followed by locov_winograd_conv3x3_f32_split_ex, and those
 * bits.  With C % 64 == 0 the ROIAlign workgroup (one ROI x 64 channels) keeps its 49 pooled rows in LDS and writes their Winograd
 * input transform itself; otherwise the two launches.  feat_nhwc: fp32 [Nimg, H, W, C] with pixel pitch feat_ld; pooled: 14 (the
 * even bins form the 7 x 7 tile); workspace: locov_roi_align_winograd_workspace_bytes(R, C, N) (the pixel scratch only for the
 * two-launch fallback; locov_conv1x1_winograd_workspace_bytes is the upper bound); the other arguments as above. */
int64_t locov_roi_align_winograd_workspace_bytes(int64_t R, int C, int N);
int locov_roi_align_winograd_conv3x3_f32_split(const float *feat_nhwc, int Nimg, int H, int W, int C, int64_t feat_ld,
                                               const float *rois, int64_t R, int pooled, float spatial_scale,
                                               int sampling_ratio, int aligned, const float *scale1, const float *shift1,
                                               const void *U_split, float u_scale, float v_scale, const float *scale2,
                                               const float *shift2, float *y, int64_t ldy, int N, unsigned flags,
                                               float y_split_scale, void *workspace, int64_t workspace_bytes,
                                               unsigned *overflow, locov_stream_t stream);
int locov_conv1x1_winograd_conv3x3_f32_split(const float *x_split, int64_t ldx, int K, float x_scale, const void *W1_split,
                                             float w1_scale, const float *scale1, const float *shift1, int64_t R, int C,
                                             const void *U_split, float u_scale, float v_scale, const float *scale2,
                                             const float *shift2, float *y, int64_t ldy, int N, unsigned flags,
                                             float y_split_scale, void *workspace, int64_t workspace_bytes,
                                             unsigned *overflow, locov_stream_t stream);

int64_t locov_winograd_wgrad_workspace_bytes(int64_t R, int Cin, int N);
int locov_winograd_wgrad_f32(const float *x, const float *g, int64_t R, int Cin, int N, unsigned flags,
                             const float *row_scale, float *dw, void *workspace, int64_t workspace_bytes,
                             locov_stream_t stream);

/* Every split-operand form of the Res5 convolution weights a training step needs, in ONE launch (csrc/weight_prep.hip).
 * The LSM configuration trains the Res5 convolutions (configs/coco_lsm.yaml:8), so each step re-derives, per 1x1 convolution,
 * W [N,K] (forward, roi_emb_heads.py:323,343) and (s W)^T [K,N] (data gradient); per 3x3 convolution the Winograd-domain
 * filter of the proposals' 7x7 tiles, the im2col filter of the whole-grid call, and both forms of the flipped filter
 * flip(s w) -- each in the (hi, lo) f16 split layout of locov_split_f16x2_pack, scaled by `scale` (a power of two).  One job =
 * one operand; results are bit-identical to locov_weight_transpose_scale / locov_conv3x3_weight_flip /
 * locov_winograd_pack_weight / locov_pack_conv3x3_weight followed by locov_split_f16x2_pack.  `overflow` as there. */
#define LOCOV_WEIGHT_PREP_MAX_JOBS 32
enum {
    LOCOV_PREP_PLAIN = 0,       /* w [N,K]                -> [N,K]                        (K % 32 == 0)   */
    LOCOV_PREP_TRANSPOSE = 1,   /* w [N,K], s[N]          -> [K,N] = (s w)^T              (N % 32 == 0)   */
    LOCOV_PREP_IM2COL = 2,      /* w [N,Cin,3,3]          -> [N, 9 Cin], column = tap * Cin + c   (Cin % 32 == 0) */
    LOCOV_PREP_IM2COL_FLIP = 3, /* w [N,Cin,3,3], s[N]    -> [Cin, 9 N] of flip(s w)      (N % 32 == 0)   */
    LOCOV_PREP_WINO = 4,        /* w [N,Cin,3,3]          -> U [121, N, Cin]              (Cin % 32 == 0) */
    LOCOV_PREP_WINO_FLIP = 5    /* w [N,Cin,3,3], s[N]    -> U [121, Cin, N] of flip(s w) (N % 32 == 0)   */
};
typedef struct locov_weight_prep_job {
    const float *w;             /* the convolution weight as stored (device) */
    const float *row_scale;     /* FrozenBN scale s[n] (device) or NULL */
    void *out;                  /* destination, same byte size as the fp32 operand would have (device) */
    float scale;                /* power-of-two operand scale */
    int kind, N, K;             /* K = Cin for the 3x3 kinds */
} locov_weight_prep_job;
int locov_res5_weight_prep(const locov_weight_prep_job *jobs, int n_jobs, unsigned *overflow, locov_stream_t stream);

int locov_weight_transpose_scale(const float *w, int N, int K, const float *row_scale, float *out,
                                 locov_stream_t stream);
int locov_conv3x3_weight_flip(const float *w, int N, int Cin, const float *row_scale, float *out,
                              locov_stream_t stream);
int locov_im2col3x3_nhwc(const float *x, int64_t R, int H, int W, int C, float *col, locov_stream_t stream);
int locov_conv3x3_wgrad_unpack(const float *dw_packed, int N, int Cin, const float *row_scale, float *dw,
                               locov_stream_t stream);
int locov_relu_mask(const float *g, const float *act, int64_t n, float *out, float *amax_out, locov_stream_t stream);
/* locov_zero_if_raised : the GradScaler-style skip of a backward pass in split arithmetic, decided ON THE DEVICE.  `flag` is the
 *                          pass's range-guard word (Res5RowsFn.backward -- the gradients of roi_emb_heads.py:323,343-347's Res5
 *                          calls, which the reference's cuDNN autograd computes in fp32 and cannot overflow): when it is non-zero
 *                          each of the n_tensors (<= LOCOV_ZERO_LIST_MAX) fp32 gradient tensors (`tensors[i]`, `counts[i]` elements;
 *                          host arrays of device pointers / counts) is zero-filled, so that no inf / NaN produced beyond fp16's range
 *                          reaches an optimizer step or DDP's all-reduce; when it is zero nothing is read or written.  One launch,
 *                          no host read -- the caller learns of the skip from the word whenever it next reads it. */
#define LOCOV_ZERO_LIST_MAX 24
int locov_zero_if_raised(float *const *tensors, const int64_t *counts, int n_tensors, const unsigned *flag,
                         locov_stream_t stream);

int locov_spatial_mean_bwd(const float *g, const float *act, int64_t R, int C, int HW, float *out,
                           float *amax_out, locov_stream_t stream);
int locov_rows_stride2(const float *src, int N, int H, int W, int C, int forward, float *dst,
                       locov_stream_t stream);
int locov_roi_align_nhwc_bwd(const float *grad_rows, int64_t grad_ld, int N, int H, int W, int C,
                             const float *rois, int64_t R, int pooled_h, int pooled_w, float spatial_scale,
                             int sampling_ratio, int aligned, int bin_stride, int pos_major, float *grad_feat,
                             locov_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LOCOV_HIP_H */
