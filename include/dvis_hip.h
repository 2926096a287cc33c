/*
 * dvis_hip.h — C ABI of libdvis_hip.so: the MI355X (gfx950) hot path of DVIS++.
 *
 * Every entry point takes plain DEVICE pointers + sizes + a hipStream_t (passed
 * as void*; NULL = the null stream), enqueues asynchronously on that stream,
 * allocates nothing, keeps no reference to its arguments and is re-entrant.
 * Return value: 0 on success, a negative DVIS_E_* code otherwise;
 * dvis_last_error() returns a thread-local human-readable message.  Launch
 * failures are reported (the reference only printf()s them:
 * ops/src/cuda/ms_deform_im2col_cuda.cuh:953-957).
 *
 * Reference interfaces replaced (paths relative to /root/reference/DVIS_Plus):
 *   dvis_msda_forward        <- ms_deform_attn_forward,  mask2former/modeling/pixel_decoder/ops/src/ms_deform_attn.h:25-45
 *                               (kernel ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304, host ms_deform_attn_cuda.cu:25-85)
 *   dvis_msda_backward       <- ms_deform_attn_backward, ops/src/ms_deform_attn.h:47-66 (ms_deform_attn_cuda.cu:88-158)
 *   dvis_msda_fused_forward  <- softmax + location arithmetic + op of MSDeformAttn.forward,
 *                               ops/modules/ms_deform_attn.py:101-117 (no materialised loc / weights)
 *   dvis_mask_logits         <- einsum("bqc,bchw->bqhw"), dvis_Plus/video_mask2former_transformer_decoder.py:363
 *   dvis_attn_mask           <- einsum + F.interpolate + (sigmoid < 0.5) of forward_prediction_heads, ibid. :363-371,
 *                               plus the "fully masked row" reset at :297
 *   dvis_attention_forward   <- nn.MultiheadAttention core (softmax(QK^T/sqrt(d) [+mask]) V) as used by
 *                               CrossAttentionLayer / SelfAttentionLayer (mask2former_video/.../video_mask2former_transformer_decoder.py:18-136),
 *                               ReferringCrossAttentionLayer (dvis_Plus/tracker.py:8-92), TemporalRefiner (dvis_Plus/refiner.py:104-139)
 *   dvis_add_layernorm       <- `norm(tgt + tgt2)` of every post-norm residual block (msdeformattn.py:125-131,
 *                               video_mask2former_transformer_decoder.py:47-50,108-111,166-170, tracker.py:51-53)
 *   dvis_nchw_to_tokens      <- src.flatten(2).transpose(1, 2) + torch.cat over levels, msdeformattn.py:64-79
 *   dvis_bias_act            <- FrozenBN shift + shortcut add + ReLU after each backbone convolution (detectron2 BottleneckBlock)
 *   dvis_conv1x1_bias_act    <- 1x1 convolution + that epilogue in one pass (conv1 / conv3 / stride-1 shortcut of the bottleneck)
 *   dvis_conv3x3_winograd    <- 3x3 / stride 1 / pad 1 convolution (+ bias + ReLU): the FPN output convolution of the pixel decoder,
 *                               mask2former/modeling/pixel_decoder/msdeformattn.py:262-270 (built), :343-349 (applied), and conv2 of
 *                               the R50 bottlenecks (detectron2 BottleneckBlock)
 *   dvis_conv3x3s2           <- the stride-2 3x3 convolutions of the R50 (conv2 of the first res3 / res4 / res5 bottleneck)
 *   dvis_conv7x7s2           <- the 7x7 / stride 2 stem convolution of the R50 (detectron2 BasicStem.conv1)
 *   dvis_bias_relu_maxpool   <- FrozenBN shift + ReLU + max_pool2d(3, stride 2, padding 1) of the ResNet stem (detectron2 BasicStem)
 *   dvis_upsample_add        <- `cur_fpn + F.interpolate(out[-1], size=..., mode="bilinear")`, msdeformattn.py:347
 *   dvis_group_norm_affine / dvis_scale_shift_act / dvis_upsample_add_affine
 *                            <- the GroupNorm (+ReLU) of detectron2's Conv2d wrapper around the FPN lateral / output
 *                               convolutions, msdeformattn.py:280-300 (built) and :343-349 (applied)
 *   dvis_vps_argmax          <- two-stage resize + sigmoid + score-weighted argmax + segment areas of inference_video_vps,
 *                               dvis_Plus/meta_architecture.py:890-925
 *   dvis_resize2_gt0         <- the two F.interpolate calls + `> 0.` of inference_video_vis, dvis_Plus/meta_architecture.py:843-853
 *   dvis_vss_argmax          <- two-stage resize + sigmoid + einsum("qc,qthw->cthw") + max(0) of inference_video_vss,
 *                               dvis_Plus/meta_architecture.py:954-979
 *   dvis_lsap_solve          <- scipy.optimize.linear_sum_assignment as called by Noiser.match_embds, dvis_Plus/noiser.py:43-56
 *   dvis_match_chain         <- the frame-by-frame matching loop of ReferringTracker_noiser.forward, dvis_Plus/tracker.py:210-291
 *   dvis_gemm_nt             <- the projections around every attention / FFN block of the tracker and the refiner
 *                               (nn.MultiheadAttention in/out_proj, linear1/2, MLP: dvis_Plus/tracker.py:293-318,
 *                               dvis_Plus/refiner.py:104-139), the cosine matrices of Noiser.match_embds (noiser.py:43-56)
 *                               and the refiner's nn.Conv1d layers as im2col GEMMs (refiner.py:42-54,116-119)
 *   dvis_gemm_ln             <- projection + the LayerNorm seam in front of it: `tgt = norm(identity + attn)`, `norm(tgt + ffn(tgt))`
 *                               followed by the next in_proj / linear1 / ref_proj layer, dvis_Plus/tracker.py:45-52, 277-318
 *   dvis_x3_linear / dvis_x3_linear_ln / dvis_x3_ffn_ln
 *                            <- the dense layers of MSDeformAttnTransformerEncoderLayer over all pixels of all frames:
 *                               value_proj / sampling_offsets / attention_weights / output_proj
 *                               (ops/modules/ms_deform_attn.py:96-117), `norm1(src + src2)`, forward_ffn =
 *                               `norm2(src + linear2(relu(linear1(src))))` (mask2former/modeling/pixel_decoder/msdeformattn.py:103-131)
 *   dvis_conv1x1_x3          <- the compute-bound 1x1 convolutions of the R50 bottlenecks (as dvis_conv1x1_mfma) in that arithmetic
 *   dvis_bneck_x3            <- the res2 bottlenecks (detectron2 BottleneckBlock.forward: conv2 -> conv3 + shortcut -> the next block's
 *                               conv1) as one kernel per block, 64-channel maps as pre-split operand images
 */
#ifndef DVIS_HIP_H
#define DVIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of floating tensors */
enum { DVIS_F32 = 0, DVIS_F64 = 1, DVIS_F16 = 2, DVIS_BF16 = 3 };

/* error codes */
enum {
  DVIS_OK = 0,
  DVIS_E_ARG = -1,      /* bad argument (null pointer, non-positive size, unsupported dtype) */
  DVIS_E_LAUNCH = -2,   /* hipLaunchKernel / hipGetLastError failed */
  DVIS_E_UNSUPPORTED = -3
};

const char *dvis_last_error(void);
int dvis_version(void);

/*
 * Multi-scale deformable attention, forward.
 *   value   (N, S, M, D)          dtype
 *   shapes  (L, 2) int64 [H_l, W_l]    DEVICE memory (reference contract, ms_deform_attn_cuda.cu:39-41)
 *   level_start (L,) int64              DEVICE memory
 *   loc     (N, Lq, M, L, P, 2)   dtype, normalised (x, y)
 *   w       (N, Lq, M, L, P)      dtype
 *   out     (N, Lq, M*D)          dtype, fully overwritten (no need to pre-zero)
 * out[n,q,m,:] = sum_{l,p} w * bilinear(value[n, level l, :, m, :], (x*W_l - 0.5, y*H_l - 0.5)), zero padding.
 * Any N is accepted (the reference's N % im2col_step constraint is not needed).
 */
int dvis_msda_forward(int dtype, const void *value, const int64_t *shapes, const int64_t *level_start,
                      const void *loc, const void *w, int N, int S, int M, int D, int L, int Lq, int P,
                      void *out, void *stream);

/*
 * Backward.  grad_value (N,S,M,D), grad_loc (N,Lq,M,L,P,2), grad_w (N,Lq,M,L,P): grad_value MUST be
 * zero-filled by the caller (it is accumulated with atomics); grad_loc / grad_w are fully overwritten.
 */
int dvis_msda_backward(int dtype, const void *value, const int64_t *shapes, const int64_t *level_start,
                       const void *loc, const void *w, const void *grad_out,
                       int N, int S, int M, int D, int L, int Lq, int P,
                       void *grad_value, void *grad_loc, void *grad_w, void *stream);

/*
 * The fp32 backward with a run-to-run REPRODUCIBLE grad_value.  The reference accumulates grad_value with float
 * atomics (ops/src/cuda/ms_deform_im2col_cuda.cuh: atomicAdd in every col2im kernel), so two runs of its backward differ in their
 * last bits; dvis_msda_backward does the same.  Here every contribution is rounded on its own to a multiple of 2^-e (e from the
 * launch's max |grad_out| x max |attention weight|: 42 bits below the largest possible contribution) and added as a 64-bit INTEGER
 * atomic — integer addition is associative, the order the atomics land in does not matter — then converted back.  grad_loc / grad_w
 * are per-sample reductions in a fixed order in both entry points.  ws: dvis_msda_backward_det_ws_bytes(N, S, M, D) bytes
 * (8 per value element + 16), 16-byte aligned, zeroed inside.  ABSOLUTE resolution 2^-42 of max |grad_out| x max |weight| (fp32 carries
 * 2^-24 relative to each sum): cells whose whole sum is below 2^-18 of that bound are less precise than with float atomics.
 */
int64_t dvis_msda_backward_det_ws_bytes(int N, int S, int M, int D);
int dvis_msda_backward_det(const float *value, const int64_t *shapes, const int64_t *level_start, const float *loc, const float *w,
                           const float *grad_out, int N, int S, int M, int D, int L, int Lq, int P, float *grad_value, float *grad_loc,
                           float *grad_w, void *ws, void *stream);

/*
 * Fused forward (fp32): takes the RAW outputs of the sampling_offsets / attention_weights linears.
 *   offsets  rows of (M, L, P, 2) floats, row stride `off_stride` floats, one row per (n, q)
 *   logits   rows of (M, L*P)     floats, row stride `logit_stride` floats
 *   ref      (Nref, Lq, L, 2) reference points; Nref == 1 broadcasts over the batch
 * loc = ref[:, :, None, :, None, :] + offsets / (W_l, H_l);  w = softmax over (L*P) — ms_deform_attn.py:101-109.
 *   shapes_host: NULL, or a HOST copy of `shapes`; when given and sum(H_l*W_l) == Lq == S (encoder self-attention:
 *   the queries are the pixels) blocks own 8x8 pixel tiles instead of 64 consecutive queries (cache locality only;
 *   results are identical).
 */
int dvis_msda_fused_forward(const float *value, const int64_t *shapes, const int64_t *level_start,
                            const float *ref, int Nref, const float *offsets, int64_t off_stride,
                            const float *logits, int64_t logit_stride,
                            int N, int S, int M, int D, int L, int Lq, int P, float *out,
                            const int64_t *shapes_host, void *stream);

/*
 * Same, plus the projection of the queries' POSITION embedding: MSDeformAttn is called with query = src + pos
 * (pixel_decoder/msdeformattn.py:124-126) and only the two linears above read the query, so
 * linear(src + pos) = linear(src) + (pos W^T).  pos_offsets / pos_logits: one row per QUERY (shared by all N frames,
 * row stride `pos_stride` floats) of (pos W_off^T) / (pos W_logit^T) WITHOUT bias; added to the raw rows inside the
 * kernel — the (N, Lq, C) `src + pos` tensor is never formed.  Both NULL = dvis_msda_fused_forward.
 */
int dvis_msda_fused_forward_pos(const float *value, const int64_t *shapes, const int64_t *level_start,
                                const float *ref, int Nref, const float *offsets, int64_t off_stride,
                                const float *logits, int64_t logit_stride, const float *pos_offsets,
                                const float *pos_logits, int64_t pos_stride, int N, int S, int M, int D, int L, int Lq,
                                int P, float *out, const int64_t *shapes_host, void *stream);

/*
 * dvis_msda_fused_forward on HALF-PRECISION storage: `value`, the raw offset / logit rows and `out` in `dtype` (DVIS_F16 /
 * DVIS_BF16 — what the projections of MSDeformAttn.forward, ops/modules/ms_deform_attn.py:97-105, produce under
 * torch.autocast, how the reference evaluates: train_net_video.py:259), reference points fp32, arithmetic fp32, 8 channels
 * per lane.  The reference's own op dispatches float / double only (ms_deform_attn_cuda.cu:69) and its module falls into the
 * grid_sample path there.  Row strides in ELEMENTS; the reference's row layout (all heads' offsets, then all heads' logits).
 */
int dvis_msda_fused_forward_h(int dtype, const void *value, const int64_t *shapes, const int64_t *level_start,
                              const float *ref, int Nref, const void *offsets, int64_t off_stride, const void *logits,
                              int64_t logit_stride, int N, int S, int M, int D, int L, int Lq, int P, void *out,
                              const int64_t *shapes_host, void *stream);

/*
 * Same, with the per-head layout of a projection row made explicit: head m's 2*L*P offsets start `off_head_stride` floats
 * after head m-1's, its L*P logits `logit_head_stride` floats after head m-1's (0 = the reference's layout: L*P*2 and
 * L*P, i.e. all heads' offsets, then all heads' logits).  With the rows of the fused projection permuted into per-head
 * SLOTS [2LP offsets | LP logits | pad] (offsets = proj, logits = proj + 2LP, both head strides = the slot size) a
 * (query, head) pair reads one contiguous run of its row: fewer 128-byte lines shared between the heads' XCDs.
 * value_head_major != 0: `value` is laid out (M, N, S, D) — head outermost, as dvis_gemm_nt_hm writes the value projection —
 * instead of the reference's (N, S, M, D): neighbouring pixels of a head are adjacent 128-byte lines.
 */
int dvis_msda_fused_forward_slots(const float *value, const int64_t *shapes, const int64_t *level_start,
                                  const float *ref, int Nref, const float *offsets, int64_t off_stride,
                                  const float *logits, int64_t logit_stride, int off_head_stride, int logit_head_stride,
                                  int value_head_major, const float *pos_offsets, const float *pos_logits, int64_t pos_stride, int N, int S,
                                  int M, int D, int L, int Lq, int P, float *out, const int64_t *shapes_host,
                                  void *stream);

/*
 * Mask logits: out[b, q, p] = sum_c embed[b, q, c] * feat[b, c, p]     (fp32, exact-fp32 MFMA)
 *   embed (B, Q, C), feat (B, C, HW), out (B, Q, HW)
 */
int dvis_mask_logits(const float *embed, const float *feat, int B, int Q, int C, int64_t HW,
                     float *out, void *stream);

/*
 * Attention mask of the masked-attention decoder, one launch:
 *   logits = embed x feat at the stride-4 map (never written to HBM), bilinear-resized
 *   (align_corners=False) to (h, w) where H % h == 0, W % w == 0 and the factor is even
 *   (=> the mean of the 2x2 centre pixels of every block, evaluated in torch's order),
 *   blocked = logits < 0  (== sigmoid < 0.5).
 *   embed (B, Q, C), feat (B, C, H, W)  ->  mask (B, Q, h*w) uint8 (1 = blocked), one copy (not per head);
 *   allowed_count (B, Q) int32 = number of un-blocked keys per row (zeroed inside).  A row with count 0 is
 *   "blocked everywhere": dvis_attention_forward then ignores the mask for that row, which is the reference's
 *   reset `attn_mask[where(attn_mask.sum(-1) == HW)] = False` (:297) without its host sync.
 */
int dvis_attn_mask(const float *embed, const float *feat, int B, int Q, int C, int H, int W, int h, int w,
                   uint8_t *mask, int32_t *allowed_count, void *stream);
/*
 * The same attention mask from a feature PYRAMID (round 5).  The bilinear down-sizing of forward_prediction_heads
 * (dvis_Plus/video_mask2former_transformer_decoder.py:367, F.interpolate(..., align_corners=False) by an even integer factor) is
 * linear: out = 0.25 ((l_a + l_b) + (l_c + l_d)) over the four centre pixels of a block = the contraction of
 * 0.25 ((f_a + f_b) + (f_c + f_d)).  dvis_center_pool3 forms those averages for the factors 2 / 4 / 8 of a stride-4 map in ONE read
 * (feat: `planes` maps of H x W, H % 8 == W % 8 == 0 -> p2 (H/2 x W/2), p4, p8), once per clip; dvis_attn_mask_pooled contracts a
 * level's pooled map (B, C, h, w) and thresholds: mask (B, Q, h*w) uint8 (1 = blocked), allowed_count as above.  One quarter of
 * dvis_attn_mask's products per layer and no re-read of the stride-4 map in any of the nine decoder layers.  Equal to
 * dvis_attn_mask except where |logit| is of the order of its rounding error (the two orders round differently).
 */
int dvis_center_pool3(const float *feat, int64_t planes, int H, int W, float *p2, float *p4, float *p8, void *stream);
int dvis_attn_mask_pooled(const float *embed, const float *pooled, int B, int Q, int C, int h, int w, uint8_t *mask,
                          int32_t *allowed_count, void *stream);

/*
 * softmax(Q K^T * scale [masked]) V for B batch entries x `heads` heads, fp32 in/out, exact-fp32 MFMA.
 *   q (B, heads, Lq, d), k / v (B, heads, Lk, d), out (B, heads, Lq, d) as STRIDED views: x_strides[3] =
 *   {batch, head, row} strides in floats, d contiguous, d in {32, 64}.  (Lets q/k/v be slices of fused
 *   projections and out be the (Lq, B, heads*d) buffer the out-projection reads.)
 *   mask: NULL or uint8 (B, Lq, Lk), 1 = blocked, shared by the heads of a batch entry (4-byte aligned).
 *   allowed_count: NULL or int32 (B, Lq) from dvis_attn_mask; rows with count 0 ignore the mask.
 *   ws: workspace of dvis_attention_ws_bytes(B*heads, Lq, Lk, d) bytes (split-K partials); may be NULL when 0.
 */
int64_t dvis_attention_ws_bytes(int BH, int Lq, int Lk, int d);
int dvis_attention_forward(const float *q, const int64_t *q_strides, const float *k, const int64_t *k_strides,
                           const float *v, const int64_t *v_strides, float *out, const int64_t *o_strides,
                           const uint8_t *mask, const int32_t *allowed_count, int B, int heads, int Lq, int Lk,
                           int d, float scale, void *ws, void *stream);
/* Same with the kernel chosen by the CALLER instead of from the sizes: 0 = dvis_attention_forward's choice, 1 = the
 * short-key latency kernel (Lk <= 128: one workgroup per (batch, head, 16-query tile), K / V staged in one memory round
 * trip).  A (batch, head)'s result depends on the kernel, so a caller that needs the same bits for a clip whether it runs
 * alone or batched with others (the referring tracker's recurrence, dvis_Plus/tracker.py:277-318) pins it. */
int dvis_attention_forward_k(const float *q, const int64_t *q_strides, const float *k, const int64_t *k_strides,
                           const float *v, const int64_t *v_strides, float *out, const int64_t *o_strides,
                           const uint8_t *mask, const int32_t *allowed_count, int B, int heads, int Lq, int Lk,
                           int d, float scale, void *ws, void *stream, int kernel);
/* kernel 2: long self-attention at d = 64 without a mask (the ViT blocks of the DINOv2 / ViT-Adapter backbones,
 * dvis_Plus/../vit_adapter: 3681 tokens x 16 heads at 720p) on split-f16 matrix-core products — Q, K, V as two f16 terms each
 * (x3_common.h: three products per fp32 product, fp32 accumulation), softmax in fp32, the probabilities split the same way.
 * Two launches: the operands' two-term images are written once into `ws` (dvis_attention_ws_bytes_k(.., 2) bytes: what Q, K, V
 * take in fp32), then one workgroup per (batch-head, 128 queries) streams K / V from there.  Operands must stay below 4094 in
 * magnitude (the range guard of dvis_x3_set_range_flag reports a violation). */
int64_t dvis_attention_ws_bytes_k(int BH, int Lq, int Lk, int d, int kernel);

/*
 * out[r, :] = LayerNorm(x[r, :] + res[r, :]) * gamma + beta, rows x C fp32 (C % 4 == 0, C <= 1024); `res` may be NULL
 * and may have its own row stride (floats).  One pass: 2 reads + 1 write (torch: add kernel + layer_norm kernel).
 */
int dvis_add_layernorm(const float *x, const float *res, int64_t res_row_stride, const float *gamma, const float *beta,
                       float *out, int64_t rows, int C, float eps, void *stream);

/*
 * dvis_add_layernorm with a second output  out_pos[r, :] = out[r, :] + pos[r mod pos_rows, :]  (pos: pos_rows x C): the
 * deformable encoder's next-layer query `with_pos_embed(src, pos)` (msdeformattn.py:121-123 / ms_deform_attn.py:108) is
 * written while the row is still in registers instead of by an add kernel of its own.
 */
int dvis_add_layernorm_pos(const float *x, const float *res, int64_t res_row_stride, const float *gamma, const float *beta,
                           float *out, const float *pos, int64_t pos_rows, float *out_pos, int64_t rows, int C, float eps,
                           void *stream);

/*
 * out = lateral + F.interpolate(top, size=(H, W), mode="bilinear", align_corners=False) for `planes` = N*C planes;
 * lateral / out (planes, H, W), top (planes, h, w); W % 4 == 0.  (FPN top-down step, msdeformattn.py:347.)
 */
int dvis_upsample_add(const float *lateral, const float *top, float *out, int64_t planes, int H, int W, int h, int w,
                      void *stream);

/*
 * out[n][row0 + p][c] = x[n][c][p]: lays an (N, C, HW) map down as HW rows of an (N, S, C) token matrix — the
 * flatten(2).transpose(1, 2) + torch.cat over levels of MSDeformAttnTransformerEncoderOnly.forward
 * (mask2former/modeling/pixel_decoder/msdeformattn.py:64-79).  x contiguous (N, C, HW), out contiguous (N, S, C).
 */
int dvis_nchw_to_tokens(const float *x, float *out, int64_t N, int C, int64_t HW, int64_t S, int64_t row0, void *stream);

/*
 * dvis_nchw_to_tokens with the map read as x[n][c] * scale[n*C + c] + shift[n*C + c] (the input projection's GroupNorm,
 * msdeformattn.py:231-247, from dvis_group_norm_affine; NULL pair = identity) and an optional second output
 * out_pos[n][row0 + p][c] = out[n][row0 + p][c] + pos[row0 + p][c] (pos: (S, C) token-major position + level embedding;
 * the first encoder layer's query, msdeformattn.py:121-123; NULL pair = not written).
 */
/* The way back for one level: out (N, C, HW) = tok[:, row0 : row0 + HW, :] transposed (tok: (N, S, C)) — the encoder output
 * as the map the FPN's top-down path reads, msdeformattn.py:333-339 (+ :347). */
int dvis_tokens_to_nchw(const float *tok, float *out, int64_t N, int C, int64_t HW, int64_t S, int64_t row0, void *stream);
/* Input normalisation + zero padding of a clip in one pass (dvis_Plus/meta_architecture.py:1310-1311, :186-187, :638-639:
 * `(x - pixel_mean) / pixel_std`, then ImageList.from_tensors(images, size_divisibility)):
 * out (planes, Hp, Wp) = (float(in (planes, H, W)) - mean[p % C]) / std[p % C] inside the image, 0 in the padding; in: uint8
 * (is_u8 = 1) or fp32 planes, planes = frames * C.  The same fp32 subtract and divide as the torch expression: same bits. */
int dvis_normalize_pad(const void *in, int is_u8, float *out, int64_t planes, int C, int H, int W, int Hp, int Wp, const float *mean,
                       const float *stdv, void *stream);
int dvis_nchw_to_tokens_affine(const float *x, const float *scale, const float *shift, const float *pos, float *out,
                               float *out_pos, int64_t N, int C, int64_t HW, int64_t S, int64_t row0, void *stream);

/*
 * 1x1 stride-1 convolution on NCHW with its epilogue in one pass (exact-fp32 MFMA):
 *   out[n, m, p] = relu?( sum_k w[m, k] x[n, k, p] + bias[m] (+ res[n, m, p]) )
 *   x (N, K, HW), w (M, K) (FrozenBN scale folded in), bias (M) or NULL, res (N, M, HW) or NULL, out (N, M, HW); fp32,
 *   16-byte aligned, K % 4 == 0, HW % 4 == 0.  Replaces conv -> dvis_bias_act for the bottleneck's conv1 / conv3 /
 *   stride-1 shortcut (detectron2 BottleneckBlock, FrozenBN folded by the caller) where the layer is memory-bound.
 *   dvis_conv1x1_supported(K, M, HW) != 0 tells whether the shape is served (M <= 64: K <= 512; M <= 128: K <= 256;
 *   wider: K <= 128); other shapes return DVIS_E_ARG — the caller keeps the library contraction for them.
 */
int dvis_conv1x1_supported(int K, int M, int64_t HW);
/*
 * The same operation for the COMPUTE-bound 1x1 layers (many input channels: C % 128 == 0, K % 64 == 0, HW even and >= 64): weights
 * streamed through the MFMA operand layout `uf` (K * C floats, written by dvis_conv1x1_mfma_pack), 64 pixels x 64 output channels
 * per workgroup.  y (N, K, HW) = relu?(w (K, C) x (N, C, HW) + bias[k] + res).
 */
int dvis_conv1x1_mfma_supported(int C, int K, int64_t HW);
int dvis_conv1x1_mfma_pack(const float *w, float *uf, int K, int C, void *stream);
int dvis_conv1x1_mfma(const float *x, const float *uf, const float *bias, const float *res, float *y, int N, int C, int K, int64_t HW,
                      int relu, void *stream);
/* stride 2 (the 1x1 shortcut of the first res3 / res4 / res5 bottleneck): y (N, K, ceil(H/2), ceil(W/2)) = relu?(w x[:, :, ::2, ::2] + bias + res);
 * served when dvis_conv1x1_mfma_supported(C, K, ceil(H/2) * ceil(W/2)) and 2 input images stay below 2 GiB. */
int dvis_conv1x1s2_mfma(const float *x, const float *uf, const float *bias, const float *res, float *y, int N, int C, int K, int H, int W,
                        int relu, void *stream);
int dvis_conv1x1_bias_act(const float *x, const float *w, const float *bias, const float *res, float *out,
                          int N, int K, int M, int64_t HW, int relu, void *stream);

/*
 * y (N, K, H, W) = relu?(conv2d(x (N, C, H, W), w (K, C, 3, 3), stride 1, padding 1) + bias[k]) as Winograd F(2x2, 3x3) on the fp32
 * matrix cores: 4 multiplies per output instead of 9, accumulated in a fixed order (bit-reproducible).  `uf` = the transformed
 * weights (16 * K * C floats), written once per weight by dvis_conv3x3_winograd_pack(w, uf, K, C).  bias (K) or NULL.  fp32, NCHW
 * contiguous, 16-byte aligned.  dvis_conv3x3_winograd_supported(C, K, H, W) != 0 tells whether the shape is served (C % 16 == 0,
 * K % 64 == 0, W even, at least 64 2x2 tiles per image, 2 images < 2 GiB); other shapes return DVIS_E_ARG — the caller keeps the library
 * convolution for them.  Rounding differs from a direct convolution by the usual F(2x2, 3x3) factor (a few fp32 ulps of the
 * accumulated magnitude; tests/test_winograd_gpu.py bounds it against fp64).
 */
int dvis_conv3x3_winograd_supported(int C, int K, int H, int W);
/*
 * The same for stride 2 (padding 1): y (N, K, ceil(H/2), W/2) = relu?(conv2d(x, w, stride 2, padding 1) + bias[k]), direct on the fp32
 * matrix cores (conv2 of the first bottleneck of res3 / res4 / res5 in detectron2's ResNet with STRIDE_IN_1X1 = False).  `uf` =
 * the weights in the MFMA operand layout (12 * K * C floats), written by dvis_conv3x3s2_pack.  Same shape rules (W even).
 */
int dvis_conv3x3s2_supported(int C, int K, int H, int W);
/*
 * The ResNet stem: y (N, 64, H/2, W/2) = relu?(conv2d(x (N, 3, H, W), w (64, 3, 7, 7), stride 2, padding 3) + bias[k]), direct on
 * the fp32 matrix cores (detectron2 BasicStem.conv1).  `uf` = 10 240 floats written by dvis_conv7x7s2_pack(w, uf).  H, W even.
 */
int dvis_conv7x7s2_supported(int C, int K, int H, int W);
int dvis_conv7x7s2_pack(const float *w, float *uf, void *stream);
int dvis_conv7x7s2(const float *x, const float *uf, const float *bias, float *y, int N, int H, int W, int relu, void *stream);
int dvis_conv3x3s2_pack(const float *w, float *uf, int K, int C, void *stream);
int dvis_conv3x3s2(const float *x, const float *uf, const float *bias, float *y, int N, int C, int K, int H, int W, int relu,
                   void *stream);
int dvis_conv3x3_winograd_pack(const float *w, float *uf, int K, int C, void *stream);
int dvis_conv3x3_winograd(const float *x, const float *uf, const float *bias, float *y, int N, int C, int K, int H, int W,
                          int relu, void *stream);

/*
 * In place on `planes` = N*C contiguous planes of HW floats (NCHW): x = relu?(x + bias[c] + res).  bias (C,) or NULL,
 * res same shape as x or NULL.  The folded-FrozenBN bias, bottleneck shortcut add and ReLU after a library convolution
 * in one pass.  HW % 4 == 0.
 */
int dvis_bias_act(float *x, const float *bias, const float *res, int64_t planes, int C, int64_t HW, int relu,
                  void *stream);

/*
 * out (planes, H/2, W/2) = relu(max_pool2d(x (planes, H, W), kernel 3, stride 2, padding 1) + bias[plane % C]) — equal, bit
 * for bit, to max_pool2d(relu(x + bias)) (fp32 add and ReLU are monotonic): the stem's folded-FrozenBN shift, ReLU and
 * pooling in one pass over the stem convolution's output.  bias (C,) or NULL.  H even, W % 8 == 0, x / out 16-byte aligned.
 */
int dvis_bias_relu_maxpool(const float *x, const float *bias, float *out, int64_t planes, int C, int H, int W,
                           void *stream);

/*
 * GroupNorm as two cheap steps (nn.GroupNorm(G, C) on (N, C, HW) fp32, eps as given):
 *   dvis_group_norm_affine  statistics of every (sample, group) in ONE read of x (fp64 sums), returned as the per-plane
 *                           affine  scale[n*C + c] = rstd * gamma[c],  shift[n*C + c] = beta[c] - mean * scale
 *                           (gamma / beta may be NULL = 1 / 0);  (C / G) * HW % 4 == 0, x 16-byte aligned;
 *   dvis_scale_shift_act    in place x[plane] = relu?(x[plane] * scale[plane] + shift[plane]) — GroupNorm (+ReLU) applied;
 *   dvis_upsample_add_affine  as dvis_upsample_add with the lateral operand read as lateral * scale + shift — the
 *                           lateral convolution's GroupNorm applied on the fly (msdeformattn.py:345-347).
 */
int dvis_group_norm_affine(const float *x, const float *gamma, const float *beta, float *scale, float *shift, int64_t N,
                           int C, int G, int64_t HW, float eps, void *stream);
int dvis_scale_shift_act(float *x, const float *scale, const float *shift, int64_t planes, int64_t HW, int relu,
                         void *stream);
int dvis_upsample_add_affine(const float *lateral, const float *lat_scale, const float *lat_shift, const float *top,
                             float *out, int64_t planes, int H, int W, int h, int w, void *stream);

/*
 * Depthwise 3 x 3 convolution (padding 1) + bias (+ exact GELU) on a TOKEN-major map: the ConvFFN of the ViT-Adapter extractors
 * (mask2former/modeling/backbones_vitAdapter/adapter_modules.py, DWConv + the GELU that follows it in ConvFFN.forward).  The
 * reference transposes each of the three pyramid levels of its (B, N, C) token tensor to NCHW, runs a grouped Conv2d, transposes
 * back and concatenates; here one call per level reads and writes the token tensor in place of all that:
 *   x / out: level base pointers, token (b, y, x) at base + b * batch_stride + (y * w + x) * C; weight (C, 1, 3, 3) = C x 9; bias C or
 *   NULL; C % 4 == 0; out must not alias x.
 */
int dvis_dwconv3x3_tokens(const float *x, float *out, int64_t batch_stride, int B, int h, int w, int C, const float *weight,
                          const float *bias, int gelu, void *stream);

/*
 * The stride-4 output of the ViT-Adapter backbone (mask2former/modeling/backbones_vitAdapter/adapter.py, forward:
 * `c1 = self.up(c2) + c1; c1 = c1 + F.interpolate(x1, scale_factor=4, mode="bilinear", align_corners=False); f1 = self.norm1(c1)`).
 * `up` = ConvTranspose2d(C, C, 2, 2) is a GEMM over the stride-8 tokens with 4 C output features ordered (dy, dx, co); the caller
 * runs it with the eval-BatchNorm scale folded into the weights and hands the TOKEN-major result in:
 *   out[b, co, 2y+dy, 2x+dx] = g[(b, y, x), (dy, dx, co)] + scale[co] * (c1[b, co, 2y+dy, 2x+dx] + up4(x1)[b, co, 2y+dy, 2x+dx]) + shift[co]
 *   g (B * h8 * w8, 4 C); c1 / out (B, C, 2 h8, 2 w8) NCHW; x1 (B, (h8 / 2) * (w8 / 2), C) tokens of the stride-16 ViT grid or NULL;
 *   shift = scale * up.bias + (BN beta - mean * scale).  C % 64 == 0, h8 and w8 even, B * h8 <= 65535.
 * One pass over the output (7.2 GB at 30 frames x 1024 channels x 184 x 320) instead of six.
 */
int dvis_adapter_res2(const float *g, const float *c1, const float *x1, const float *scale, const float *shift, float *out, int B, int C,
                      int h8, int w8, void *stream);

/*
 * Panoptic arg-max of a clip in one pass (inference_video_vps, dvis_Plus/meta_architecture.py:890-925):
 *   prob_k = resize2(sigmoid(resize1(logits_k)[:img_h, :img_w])), both resizes bilinear align_corners=False
 *   (stride-4 map (h,w) -> padded input (first_h, first_w) -> crop (img_h, img_w) -> output (out_h, out_w));
 *   ids = argmax_k scores[k] * prob_k (first maximum wins), conf = prob_ids >= 0.5,
 *   areas (3, K) int32 = [ (ids == k).sum(), (prob_k >= 0.5).sum(), ((ids == k) & conf).sum() ]   (zeroed inside).
 *   logits: K x T maps of h*w floats, logits[k][t] at logits + k*stride_k + t*stride_t; 1 <= K <= 256.
 *   ids (T, out_h, out_w) int32, conf (T, out_h, out_w) uint8.
 */
int dvis_vps_argmax(const float *logits, int64_t stride_k, int64_t stride_t, const float *scores, int K, int T,
                    int h, int w, int first_h, int first_w, int img_h, int img_w, int out_h, int out_w,
                    int32_t *ids, uint8_t *conf, int32_t *areas, void *stream);

/*
 * Instance masks of a clip in one pass (inference_video_vis, dvis_Plus/meta_architecture.py:843-853; MinVIS.inference_video
 * :390-399):  out[k][t] = resize2(resize1(logits[k][t])[:img_h, :img_w]) > 0,  both resizes bilinear align_corners=False,
 * evaluated in torch's CPU operation order (csrc/torch_cpu_math.h).  logits: K x T maps of h*w floats at
 * logits + k*stride_k + t*stride_t;  out (K, T, out_h, out_w) uint8 (0 / 1), 4-byte aligned.
 */
int dvis_resize2_gt0(const float *logits, int64_t stride_k, int64_t stride_t, int K, int T, int h, int w, int first_h,
                     int first_w, int img_h, int img_w, int out_h, int out_w, uint8_t *out, void *stream);

/*
 * The float values behind dvis_resize2_gt0 / dvis_vps_argmax:  out[k][t] = resize2(f(resize1(logits[k][t])[:img_h, :img_w])),
 * f = sigmoid when `sigmoid` != 0, identity otherwise — the reference's interpolate -> crop -> (sigmoid) -> interpolate
 * sequence (dvis_Plus/meta_architecture.py:843-853, :899-905) in torch's CPU operation order.  out (K, T, out_h, out_w) float.
 */
int dvis_resize2(const float *logits, int64_t stride_k, int64_t stride_t, int K, int T, int h, int w, int first_h,
                 int first_w, int img_h, int img_w, int out_h, int out_w, int sigmoid, float *out, void *stream);

/*
 * Semantic arg-max of a clip in one pass (inference_video_vss, dvis_Plus/meta_architecture.py:954-979):
 *   prob_q as in dvis_vps_argmax; out[t][y][x] = argmax_c sum_q cls[q][c] * prob_q[t][y][x]   (first maximum wins).
 *   logits: Q x T maps of h*w floats at logits + q*stride_q + t*stride_t; cls: Q rows of cls_row_stride floats, the
 *   first C valid and the rest ZERO (cls_row_stride = C rounded up to a multiple of 32, <= 128), 16-byte aligned.
 *   out (T, out_h, out_w) int64 class indices.
 */
int dvis_vss_argmax(const float *logits, int64_t stride_q, int64_t stride_t, const float *cls, int cls_row_stride, int Q,
                    int C, int T, int h, int w, int first_h, int first_w, int img_h, int img_w, int out_h, int out_w,
                    int64_t *out, void *stream);

/*
 * HOST function: minimum-cost assignment of an nr x nc (nr <= nc) row-major double cost matrix,
 * shortest-augmenting-path (Jonker-Volgenant / Crouse 2016) like scipy.optimize.linear_sum_assignment;
 * col4row[i] = column assigned to row i.  Returns 0, or DVIS_E_ARG for nan/-inf entries or nr > nc.
 */
int dvis_lsap_solve(const double *cost, int nr, int nc, int64_t *col4row);

/*
 * HOST function: the tracker's per-clip matching recurrence (ReferringTracker_noiser.forward, dvis_Plus/tracker.py:283-291
 * + Noiser.match_embds) in one call.  cost (T, Q, Q) fp32 HOST memory, cost[i][c][r] = 1 - cos(cur_i[c], ref_i[r]) with
 * ref_i = the UN-permuted embeddings of frame i-1 (frame 0: the carried-over / own embeddings); the column
 * permutation by the previous frame's assignment is applied inside.  indices (T, Q) int64 out.
 */
int dvis_match_chain(const float *cost, int T, int Q, int64_t *indices);

/*
 * Deterministic exact-fp32 GEMM (v_mfma_f32_16x16x4_f32, no atomics, no inter-workgroup waiting):
 *   C[b][m][n] = act( sum_k A[b][m][k] * W[b][n][k] + bias[n] + res[b][m][n] ),   act = 0: identity, 1: ReLU
 * A (M x K, row stride lda), W (N x K, row stride ldw) — i.e. F.linear's weight layout —, C (M x N, row stride ldc),
 * res optional (row stride ldres), bias optional (N); strides in floats; `batch` independent problems at the batch
 * strides (0 = operand shared by the batch).  Needs K, lda, ldw, strideA, strideW multiples of 4 and 16-byte aligned
 * A / W (returns DVIS_E_ARG otherwise).  The summation order of an output element depends on (M, N, K, batch, config)
 * only: two calls with the same arguments return bit-identical results.  config: -1 = chosen from the sizes, else an
 * index < dvis_gemm_num_configs() (tile / K-split variants; tuning aid).
 */
int dvis_gemm_nt(const float *A, int64_t lda, int64_t strideA, const float *W, int64_t ldw, int64_t strideW,
                 const float *bias, const float *res, int64_t ldres, int64_t strideRes, float *C, int64_t ldc,
                 int64_t strideC, int M, int N, int K, int batch, int act, int config, void *stream);
/* Same with a HEAD-MAJOR output: column n of row m is stored at C[(n / head_d) * head_stride + m * head_d + n % head_d], i.e. C
 * is (N / head_d) matrices of M x head_d — the value layout dvis_msda_fused_forward_slots(value_head_major = 1) gathers from
 * (the value projection of MSDeformAttn, ops/modules/ms_deform_attn.py:97-100, written in the layout its consumer wants).
 * head_d == 0: dvis_gemm_nt.  Needs batch == 1, no residual. */
int dvis_gemm_nt_hm(const float *A, int64_t lda, int64_t strideA, const float *W, int64_t ldw, int64_t strideW,
                    const float *bias, const float *res, int64_t ldres, int64_t strideRes, float *C, int64_t ldc,
                    int64_t strideC, int M, int N, int K, int batch, int act, int config, int head_d, int64_t head_stride,
                    void *stream);
/* Same as dvis_gemm_nt with a bias PER BATCH ENTRY (strideBias floats apart; 0 = dvis_gemm_nt): `batch` projections with
 * their own weights and biases as one launch — the out-projections of the referring tracker's six cross-attention layers,
 * whose attention does not depend on the layer chain (dvis_Plus/tracker.py:293-318: q = reference, k / v = the frame's queries
 * in every layer), batched per frame. */
int dvis_gemm_nt_bb(const float *A, int64_t lda, int64_t strideA, const float *W, int64_t ldw, int64_t strideW,
                    const float *bias, int64_t strideBias, const float *res, int64_t ldres, int64_t strideRes, float *C,
                    int64_t ldc, int64_t strideC, int M, int N, int K, int batch, int act, int config, void *stream);
int dvis_gemm_num_configs(void);
/* The configuration dvis_gemm_nt(config = -1) would choose for these sizes.  A row's result depends on (N, K, config) only,
 * so a caller that wants the SAME bits for a row whether it is computed alone or stacked with other rows (the tracker run
 * for one clip or for two clips at once) pins the configuration of the smaller problem. */
int dvis_gemm_pick_config(int M, int N, int K, int batch);
/* Configuration for these sizes INSIDE the K-split family of `nw` waves (1, 4 or 8; -1 otherwise): all configurations of a
 * family sum an output element's products in the same order, so a row's bits depend on (N, K, nw) only — not on M.  The
 * per-frame segmenter folds the frames of a clip into the batch (dvis_Plus/video_mask2former_transformer_decoder.py:327-335);
 * taking its GEMMs from one family makes a frame's result independent of how many frames share the call (a rank's shard of a
 * clip vs the whole clip: north_star's frame sharding).  dvis_gemm_config_waves: the family of a configuration. */
int dvis_gemm_pick_config_nw(int M, int N, int K, int batch, int nw);
int dvis_gemm_config_waves(int config);

/*
 * GEMM with the LayerNorm(s) of the post-norm transformer blocks folded into its A-operand prologue:
 *   X = LN2( LN1(A) + ADD ),   C = act( X W^T + bias + res ),   a_out = X
 * LN1 / LN2 = LayerNorm over the K columns of a row (two-pass mean / centred variance, eps1 / eps2) with affine
 * (gamma, beta); each is skipped when its gamma is NULL, ADD (M x K, row stride ldadd) when NULL; a_out (M x K, optional)
 * receives the normalised rows.  Replaces the `tgt = norm(identity + attn(...))` / `tgt = norm(tgt + ffn(tgt))` seams of
 * ReferringCrossAttentionLayer / SelfAttentionLayer / FFNLayer (dvis_Plus/tracker.py:45-52,
 * mask2former_video/.../video_mask2former_transformer_decoder.py:47-50,166-170) between two projections: the producing
 * dvis_gemm_nt writes the raw residual sum, the consumer normalises it while its weight fragments are in flight — no
 * LayerNorm launch, no inter-workgroup hand-off.  Forms: none / LN1 / ADD + LN2 / LN1 + ADD + LN2.  K % 16 == 0, K <= 512 (2048
 * without norms), N % 4 == 0, row strides % 4 == 0, 16-byte aligned
 * operands (dvis_gemm_ln_supported; DVIS_E_ARG otherwise).  Deterministic: a row's result depends on (N, K, config) only.
 */
int dvis_gemm_ln(const float *A, int64_t lda, const float *add, int64_t ldadd, const float *gamma1, const float *beta1,
                 float eps1, const float *gamma2, const float *beta2, float eps2, float *a_out, int64_t ldaout,
                 const float *W, int64_t ldw, const float *bias, const float *res, int64_t ldres, float *C, int64_t ldc,
                 int M, int N, int K, int act, int config, void *stream);
/* norms != 0: the call carries LayerNorms / ADD (K <= 512); a plain call (the "everything in one memory round trip" form of
 * dvis_gemm_nt for launch-bound chains) reaches K <= 2048. */
int dvis_gemm_ln_supported(int M, int N, int K, int norms);
int dvis_gemm_ln_num_configs(void);
int dvis_gemm_ln_pick_config(int M, int N, int K);

/*
 * Tall fp32 GEMMs (M = every pixel of every frame, K = 256) on the F16 matrix cores with fp32-grade results: each fp32
 * operand v is carried as two f16 terms hi = rn16(v 2^e), lo = rn16(v 2^e - hi) (22 significand bits) and each product as
 * three matrix-core products lo*hi + hi*lo + hi*hi accumulated in fp32; the result is scaled back by 2^-(xexp + wexp).
 * Measured against fp64 the error is that of an fp32 GEMM (its accumulation rounding dominates; tests/test_gemm_x3_gpu.py).
 * xexp / wexp / hexp: the power-of-two exponents the activations / weights / hidden activations are scaled by before the
 * split; |v 2^e| must stay below 65504 (f16 range; values beyond saturate to +-inf in the hi term).
 *
 * dvis_x3_pack: W (N x K, row stride ldw) -> `packed` (dvis_x3_packed_bytes(N, K) bytes), the kernels' LDS image (1 KB
 * fragments [pass][k-step][row block][hi, lo][lane][8 halves]; N = 288: ten row blocks per k-step, the tenth zero, so that an item's
 * pieces divide among the streaming kernel's 8 waves — dvis_x3_packed_bytes(288, K) = 320 K 4).  Once per weight.
 */
int64_t dvis_x3_packed_bytes(int N, int K);
/* The dvis_x3_* / dvis_conv*_x3 kernels are persistent (one workgroup per CU for the length of the launch).  A host that runs a
 * second stream of small kernels next to them (DVIS_Plus_offline.stream(): the previous clip's tracker chain) leaves `cus` CUs
 * (a multiple of 8) out of their grids; returns the previous setting.  Default 0 (env DVIS_X3_RESERVE). */
int dvis_x3_set_reserve(int cus);
/* Range guard of the split-f16 kernels.  The reference computes this path in fp32 (an explicit fp32 island,
 * mask2former/modeling/pixel_decoder/msdeformattn.py:314,320): no activation magnitude breaks it.  Here an activation with
 * |v 2^xexp| >= 65520 becomes (inf, -inf) in the split and its output row non-finite — which a following ReLU would turn into
 * plain zeros.  Every dvis_x3_* / dvis_conv*_x3 launch therefore tests its PRE-activation outputs and, on a non-finite value,
 * stores the launch's tag (dvis_x3_set_tag, per host thread, > 0; returns the previous tag) into the int32 device word
 * registered for the current device (dvis_x3_set_range_flag; NULL = guard off, the default).  The word is sticky: the host
 * zeroes it, reads it once per clip and maps the tag back to the layer (dvis_plus_amd/functions.py: X3RangeError). */
int dvis_x3_set_range_flag(int32_t *device_word);
int dvis_x3_set_tag(int tag);
int dvis_x3_pack(const float *W, int64_t ldw, int N, int K, int wexp, void *packed, void *stream);
/* which (N, K) the projection kernels serve (ln != 0: the LayerNorm form) */
int dvis_x3_linear_supported(int N, int K, int ln);
/* out (M x N, row stride ldo) = act( x (M x K, row stride ldx) W^T + bias ),  act = ReLU if relu != 0; bias may be NULL.
 * Replaces nn.Linear over the flattened (frames x pixels) batch: ms_deform_attn.py:96-99 (value_proj), :101-102
 * (sampling_offsets | attention_weights as one stacked weight), and the masked-attention decoder's key / value projections of
 * every pixel for all layers of a level at once (mask2former_video/.../video_mask2former_transformer_decoder.py:81-88,
 * nn.MultiheadAttention in_proj), and the DINOv2 blocks' qkv / proj / fc1 / fc2 over every token of every frame
 * (mask2former/modeling/backbones_vitAdapter/.../vision_transformer blocks).  K % 64 == 0; N in {128, 192, 256} or N % 256 == 0
 * (output features in passes of 256 over one tile of x); N = 288 at K = 256. */
int dvis_x3_linear(const float *x, int64_t ldx, int64_t M, int K, const void *packed, int N, int xexp, int wexp,
                   const float *bias, int relu, float *out, int64_t ldo, void *stream);
/* The same with the activations x[t] + xadd[t mod xadd_rows] (xadd: xadd_rows x K, the position embedding shared by the frames;
 * the sum is formed in registers and never written): `query = with_pos_embed(src, pos)` in front of sampling_offsets /
 * attention_weights, msdeformattn.py:99-101,122, and the decoder's keys `with_pos_embed(memory, pos)`,
 * video_mask2former_transformer_decoder.py:81-88.  K = 256 (the row's fragments stay in registers for every pass over N). */
int dvis_x3_linear_add(const float *x, int64_t ldx, int64_t M, int K, const void *packed, int N, int xexp, int wexp,
                       const float *xadd, int64_t xadd_rows, const float *bias, int relu, float *out, int64_t ldo, void *stream);
/* dvis_x3_linear with the activation chosen by `act` (0 none, 1 ReLU, 2 the exact GELU of nn.GELU(): 0.5 t (1 + erf(t / sqrt 2)))
 * and an optional residual added AFTER it: out = act(x W^T + bias) + res (res: M x N, row stride ldres floats, 16-byte aligned; may
 * alias nothing the kernel writes).  The ViT blocks of the DINOv2 / ViT-Adapter backbones (mask2former/modeling/
 * backbones_vitAdapter: `x = x + ls1(attn(norm1(x)))`, `Mlp: fc2(act(fc1(x)))`) take their GELU and their residual adds here
 * instead of in two more passes over the 452 MB token tensor of a 30-frame clip. */
int dvis_x3_linear_res(const float *x, int64_t ldx, int64_t M, int K, const void *wp, int N, int xexp, int wexp, const float *bias, int act,
                       const float *res, int64_t ldres, float *out, int64_t ldo, void *stream);

/*
 * The same `out = act(x W^T + bias) + res` for the LARGE tall GEMMs of the ViT blocks (K = 1024 .. 4096; qkv / proj / fc1 / fc2 of
 * mask2former/modeling/backbones_vitAdapter at 110 430 tokens per 30-frame clip), csrc/gemm_x3_tile.hip: workgroup tiles of
 * 128 tokens x 256 features with BOTH operands staged through LDS (the rows split into their two f16 terms once per tile), so a
 * row tile is read once per 256 output features from the L2 of the XCD that owns it — the streaming kernels above re-read and
 * re-split it per pass from HBM.  N % 256 == 0, K % 32 == 0 (dvis_x3_tile_supported); weights packed once with dvis_x3_tile_pack
 * into dvis_x3_tile_packed_bytes(N, K) bytes (the LDS image per (feature tile, k-step of 16), scaled by 2^wexp).
 * A row's result does not depend on M.  Same range guard as the other split-f16 kernels.
 */
int dvis_x3_tile_supported(int N, int K);
int64_t dvis_x3_tile_packed_bytes(int N, int K);
int dvis_x3_tile_pack(const float *w, int64_t ldw, int N, int K, int wexp, void *packed, void *stream);
int dvis_x3_tile_linear(const float *x, int64_t ldx, int64_t M, int K, const void *wp, int N, int xexp, int wexp, const float *bias, int act,
                        const float *res, int64_t ldres, float *out, int64_t ldo, void *stream);
/* The ViT blocks' fused qkv projection + attention operands (backbones_vitAdapter attention: `qkv = self.qkv(x)`, then softmax(q k^T
 * / sqrt(d)) v per head): dvis_x3_tile_linear_qkv computes x W^T + bias for N = 3 * heads * 64 columns ordered (q | k | v, head, dim)
 * and writes, instead of the fp32 qkv tensor, the two-term f16 operand images of the split-f16 attention kernel into `ws`
 * (dvis_attention_ws_bytes_k(B * heads, L, L, 64, 2) bytes; rows of x = B batch entries of L tokens; qscale = softmax scale x log2(e)
 * x 2^4); dvis_attention_x3_packed then runs that kernel's main launch on `ws` (out: (B, heads, L, 64) view, strides in floats as in
 * dvis_attention_forward).  Saves the 1.36 GB qkv tensor's write + re-read and the pack launch per block. */
int dvis_x3_tile_linear_qkv(const float *x, int64_t ldx, int64_t M, int K, const void *wp, int N, int xexp, int wexp, const float *bias,
                            int heads, int L, float qscale, void *ws, void *stream);
int dvis_attention_x3_packed(const void *ws, float *out, const int64_t *o_strides, int B, int heads, int L, void *stream);
/*
 * ROW IMAGES (round 6): the row operand of the tiled GEMM pre-split by its PRODUCER — [row tile of 128][k-tile of 16][hi, lo][chunk of
 * 8 k][128 rows][8 halves], values x 2^xexp, dvis_x3_rows_image_bytes(M, K) bytes (the fp32 tensor's size, rows padded to 128) — so
 * that both operands of the ViT blocks' GEMMs (block.py:36-104: norm1 -> qkv -> attention -> proj, norm2 -> fc1 -> GELU -> fc2) are
 * LDS-DMA streams and no fp32 activation is written between a normalisation / activation / attention and the GEMM behind it.
 *   producers: dvis_x3_rows_image (fp32 rows; k order 0), dvis_layernorm_rows_image (LayerNorm of fp32 rows, C <= 1024; order 0),
 *              dvis_x3_tile_linear_image with act = 2 (GELU; order 1), dvis_attention_x3_packed_image (order 2);
 *   consumers: dvis_x3_tile_linear_image (act 0 / 1: fp32 rows out, + residual), dvis_x3_tile_linear_qkv_image; their weights
 *              packed with dvis_x3_tile_pack_order(order of the image's producer): the k slots of a k-tile are a free permutation
 *              as long as both operands use the same one.
 * Rows past M of the last row tile are written as zeros by every producer.  Results per row as dvis_x3_tile_linear (the same three
 * products per k-tile, fp32 accumulate; orders 1 / 2 add the k-tile's terms in another order).
 */
int64_t dvis_x3_rows_image_bytes(int64_t M, int K);
int dvis_x3_rows_image(const float *x, int64_t ldx, int64_t M, int K, int xexp, void *image, void *stream);
int dvis_layernorm_rows_image(const float *x, const float *gamma, const float *beta, int64_t M, int C, float eps, int xexp, void *image,
                              void *stream);
int dvis_x3_tile_pack_order(const float *w, int64_t ldw, int N, int K, int wexp, int order, void *packed, void *stream);
int dvis_x3_tile_linear_image(const void *ximg, int64_t M, int K, const void *wp, int N, int xexp, int wexp, const float *bias, int act,
                              const float *res, int64_t ldres, float *out, int64_t ldo, void *oimg, int oexp, void *stream);
int dvis_x3_tile_linear_qkv_image(const void *ximg, int64_t M, int K, const void *wp, int N, int xexp, int wexp, const float *bias, int heads,
                                  int L, float qscale, void *ws, void *stream);
int dvis_attention_x3_packed_image(const void *ws, void *image, int B, int heads, int L, int xexp, void *stream);
/* out = LayerNorm( x W^T + bias + res ) over the N = 256 features (gamma, beta, eps; two-pass statistics as torch);
 * pos (pos_rows x N, optional): out2[t] = out[t] + pos[t mod pos_rows] (the next layer's `with_pos_embed(src, pos)`,
 * msdeformattn.py:99-101,122).  Replaces output_proj + `src = norm1(src + dropout1(src2))`, msdeformattn.py:124-125. */
int dvis_x3_linear_ln(const float *x, int64_t ldx, int64_t M, int K, const void *packed, int N, int xexp, int wexp,
                      const float *bias, const float *res, int64_t ldres, const float *gamma, const float *beta, float eps,
                      const float *pos, int64_t pos_rows, float *out, float *out2, int64_t ldo, void *stream);
/* The FFN block in one kernel: out = LayerNorm( x + linear2( relu( linear1(x) ) ) ), K = N = 256, H % 128 == 0; the
 * (M x H) hidden tensor is never written.  Replaces forward_ffn, msdeformattn.py:116-120 (+ :130).
 * dvis_x3_ffn_pack interleaves W1 (H x K) and W2 (N x H) in the order the kernel streams them (W2's k in accumulator order). */
int64_t dvis_x3_ffn_packed_bytes(int K, int H, int N);
/*
 * The compute-bound 1x1 convolutions of the R50 bottlenecks in the same arithmetic, on NCHW:
 *   y (N, K, OH, OW) = relu?( w (K, C) x (N, C, H, W)[:, :, ::stride, ::stride] + bias[k] + res ),  stride 1 or 2,
 * bias (K) / res (N, K, OH, OW) optional.  Replaces conv1 / conv3 / the stride-2 shortcut of detectron2's BottleneckBlock with
 * the FrozenBN folded by the caller (`build_resnet_backbone`, configs/dvis_Plus/.../Base-*.yaml:2-16).
 * C % 64 == 0, K = 64 / 128 or K % 256 == 0, every tensor below 2 GiB (dvis_conv1x1_x3_supported).  `packed`: dvis_conv1x1_x3_pack
 * of the (K x C) weight, dvis_conv1x1_x3_packed_bytes(C, K) bytes.
 */
int dvis_conv1x1_x3_supported(int C, int K, int64_t N, int64_t HW_in, int64_t HW_out);
int64_t dvis_conv1x1_x3_packed_bytes(int C, int K);
int dvis_conv1x1_x3_pack(const float *w, int K, int C, int wexp, void *packed, void *stream);
int dvis_conv1x1_x3(const float *x, const void *packed, const float *bias, const float *res, float *y, int N, int C, int K, int H,
                    int W, int stride, int xexp, int wexp, int relu, void *stream);
/* The last 1x1 convolution of a down-sampling bottleneck TOGETHER with its shortcut (detectron2 BottleneckBlock with a
 * projection shortcut, SURVEY.md App. B: out = relu(conv3(a) + shortcut(x))): one accumulation over the concatenated channels
 * [a (N, C, H, W) | x2 (N, C2, H2, W2) sampled with stride2], `packed` = dvis_conv1x1_x3_pack of the (K, C + C2) matrix
 * [W3 | Ws], bias = b3 + bs.  The shortcut's (N, K, H, W) map is neither written nor read back. */
int dvis_conv1x1_x3_dual(const float *x, const float *x2, const void *packed, const float *bias, const float *res, float *y, int N,
                         int C, int C2, int K, int H, int W, int H2, int W2, int stride2, int xexp, int wexp, int relu, void *stream);
/* The 3x3 / padding 1 / stride 1 or 2 convolution through the same kernel: nine taps = nine times the input channels, each
 * chunk of 64 channels read at the tap's pixel (outside the image: an out-of-range buffer offset, i.e. exact zero padding).
 * w (K, C, 3, 3).  Serves the stride-2 conv2 of the first res3 / res4 / res5 bottleneck (detectron2 BottleneckBlock). */
int64_t dvis_conv3x3_x3_packed_bytes(int C, int K);
int dvis_conv3x3_x3_pack(const float *w, int K, int C, int wexp, void *packed, void *stream);
int dvis_conv3x3_x3(const float *x, const void *packed, const float *bias, const float *res, float *y, int N, int C, int K, int H,
                    int W, int stride, int xexp, int wexp, int relu, void *stream);
/*
 * The res2 bottlenecks as a chain (csrc/bneck_x3.hip; detectron2 BottleneckBlock, 64 middle / 256 output channels):
 * one launch = conv2 (3x3) + ReLU -> conv3 + shortcut + ReLU -> the NEXT block's conv1 + ReLU, per 32-pixel group, without
 * conv2's map or a second read of the block output touching memory.  The 64-channel maps between blocks are OPERAND IMAGES:
 * per group (n, y, x / 32) 8 KB = [k-step][hi, lo f16 term][32 g + x % 32][8 halves] in accumulator channel order
 * (dvis_bneck_x3_image_bytes(N, H, W) bytes), written by dvis_conv1x1_x3_image (the stage's first conv1) or by the previous
 * dvis_bneck_x3.  res: the identity shortcut (N, 256, H, W); or NULL with x2 = the block input (N, 64, H, W) and a stream packed
 * with the projection shortcut ws (first block).  out: NULL (last block: no chained conv1, packed without w1) or the next image.
 * y (N, 256, H, W); b2 / b3 (+ bs) / b1: folded-BN shifts (64 / 256 / 64).  e2 / e3 / e1: the weights' scaling exponents.
 */
int dvis_bneck_x3_supported(int64_t N, int H, int W);
int64_t dvis_bneck_x3_image_bytes(int64_t N, int H, int W);
int64_t dvis_bneck_x3_packed_bytes(int tail, int dual);
int dvis_bneck_x3_pack(const float *w2, const float *w3, const float *ws, const float *w1, int e2, int e3, int e1, void *packed,
                       void *stream);
int dvis_bneck_x3(const void *a1, const float *res, const float *x2, const void *packed, const float *b2, const float *b3,
                  const float *b1, float *y, void *out, int N, int H, int W, int xexp, int e2, int e3, int e1, void *stream);
int dvis_conv1x1_x3_image(const float *x, const void *packed, const float *bias, void *image, int N, int C, int H, int W, int xexp,
                          int wexp, int oexp, int relu, void *stream);
/*
 * Operand images on either side of csrc/conv1x1_x3.hip (the maps INSIDE the res3 - res5 bottlenecks and the FPN output convolution's
 * input): C channels = C / 64 chunks of 8 KB per 32-pixel group (n, y, x / 32), each [k-step][hi, lo f16 term][32 g + x % 32][8 halves]
 * in accumulator channel order.  The producer's epilogue splits once; a 3x3 consumer's nine taps re-read the image: 8 loads of 16 bytes
 * per 64-channel chunk and lane instead of 32 of 4 bytes, no split arithmetic in the loop.
 *   dvis_conv_x3_pack_image  weights (K, C, taps 1 | 9) for an image-INPUT launch (channels of a chunk in accumulator order)
 *   dvis_conv_x3_image       relu?(conv + bias + res): ximg | x in, image | y out (at least one image); K = 128 or K % 256 == 0
 *   dvis_upsample_add_image  dvis_upsample_add[_affine] with the sum written as an operand image (the FPN's top-down path)
 */
int64_t dvis_conv_x3_image_bytes(int64_t N, int C, int H, int W);
int dvis_conv_x3_pack_image(const float *w, int K, int C, int taps, int wexp, void *packed, void *stream);
int dvis_conv_x3_image(const void *ximg, const float *x, const void *packed, const float *bias, const float *res, float *y, void *image,
                       int N, int C, int K, int H, int W, int stride, int taps, int xexp, int wexp, int oexp, int relu, void *stream);
int dvis_upsample_add_image(const float *lateral, const float *lat_scale, const float *lat_shift, const float *top, void *image, int N,
                            int C, int H, int W, int h, int w, int oexp, void *stream);
int dvis_x3_ffn_pack(const float *W1, int64_t ldw1, const float *W2, int64_t ldw2, int K, int H, int N, int w1exp, int w2exp,
                     void *packed, void *stream);
int dvis_x3_ffn_ln(const float *x, int64_t ldx, int64_t M, int K, int H, int N, const void *packed, int xexp, int w1exp,
                   int hexp, int w2exp, const float *b1, const float *b2, const float *gamma, const float *beta, float eps,
                   const float *pos, int64_t pos_rows, float *out, float *out2, int64_t ldo, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DVIS_HIP_H */
