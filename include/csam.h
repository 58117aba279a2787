/* csam.h -- C ABI of libcsam_hip.so: MI355X (gfx950) kernels for Crowd-SAM's dense-prompt path.
 *
 * The reference (FelixCaae/CrowdSAM, pure Python/PyTorch) has no FFI: the boundary the hot path
 * sits behind is the Python API `segment_anything_cs.{sam_model_registry,SamPredictor}` /
 * `crowdsam.model.CrowdSAM`.  This header is the boundary one level below
 * it: what the build's own `segment_anything_cs/` + `crowdsam/` modules bind through ctypes
 * (crowdsam_amd/hip.py), and what a maintainer of the reference would bind to replace the
 * `nn.Module.forward` bodies cited at each entry (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C, no torch types; every pointer is a DEVICE pointer owned by the caller unless noted
 *     "host"; the library never frees or retains them; no hidden allocations: scratch is passed in
 *     (see *_workspace_bytes);
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous w.r.t. the host and
 *     performs no device synchronisation;
 *   - returns 0 on success, <0 on error (CSAM_ERR_*); csam_last_error() returns a thread-local
 *     message; no C++ exception crosses the ABI;
 *   - "f16" = IEEE half, "f32" = float.  GEMM-shaped work: fp16 operands, fp32 MFMA accumulate.
 *   - reference file:line citations are relative to /root/reference.
 *
 * Roles (round 6).  Every entry below is one of:
 *   PRODUCT     what CrowdSAM.generate / SamPredictor run by default.  Decoder routes depend on the batch size only:
 *               >= 256 prompts per batch (the dense sweep): csam_i2t_t2i_fold, csam_upscale_stream, csam_t2i_shared,
 *               csam_pool_adjoint_mfma, csam_mask_post_scored, csam_post_finalize_compact, csam_mask_write;
 *               < 256 prompts (the shipped 32-prompt EPS batches): csam_token_block_a / _b, csam_token_heads, csam_i2t_stream,
 *               csam_t2i_fused (+ csam_t2i_merge_launch, csam_t2i_fused_parts / _workspace_bytes), csam_splitk_reduce;
 *               everything in the encoder, prompt, selection, NMS, small-region, RLE, evaluator and resize groups.
 *   ROUTE       the kernel a PRODUCT kernel replaced on some batch range, still reachable through a DecoderPlan attribute
 *               (crowdsam_amd/decoder.py: i2t_t2i, i2t_rank, i2t_rank_l1, t2i_rank, t2i_stream, i2t_stream, up_stream,
 *               up_stream_small, i2t_fold, token_block, splitk -- attributes, not environment switches) and pinned against its
 *               successor by a parity test: csam_i2t_t2i (unfolded), csam_i2t_rank, csam_i2t_rank_proj, csam_t2i_rank,
 *               csam_t2i_stream, csam_i2t_fused, csam_upscale_fused, csam_token_self_attn, csam_mask_post.
 *   COMPARATOR  reachable only through DecoderPlan(fused=False) -- round 1's unfused kernel chain, kept because the fused
 *               kernels' tests compare against it -- or from a test directly: csam_attn_t2i, csam_attn_i2t,
 *               csam_attn_t2i_workspace_bytes, csam_hyper_masks, csam_ln64_gelu, csam_softmax_stats, csam_pool_adjoint,
 *               csam_post_finalize, csam_i2t_rank_workspace_bytes.
 * Entries that neither a product path nor a test called were dropped in round 6: csam_rle_count, csam_rle_write,
 * csam_rle_count_idx, csam_rle_write_idx (the *_box forms with boxes == NULL are they), csam_pool_adjoint_v2.
 */
#ifndef CSAM_H
#define CSAM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CSAM_OK 0
#define CSAM_ERR_ARG (-1)
#define CSAM_ERR_HIP (-2)
#define CSAM_ERR_WORKSPACE (-3)
#define CSAM_ACT_NONE 0
#define CSAM_ACT_GELU 1 /* erf GELU (nn.GELU default): fp32 outputs use erf (A&S 7.1.26, |err| <= 1.5e-7); fp16 outputs and
                          * the fused upscaler use a minimax polynomial of Phi (|err| <= 4.5e-5, below the fp16 rounding of the
                          * value it produces; csam_common.h csam_gelu_poly2, tools/fit_gelu_poly.py) */
#define CSAM_ACT_RELU 2
#define CSAM_DT_F16 0
#define CSAM_DT_F32 1

int csam_abi_version(void);
const char* csam_last_error(void);

/* ---- GEMM with fused epilogue: C[M,N] = act(A[M,K] W[N,K]^T + bias[N]) * colscale[N] + R[m,N].
 * Replaces every nn.Linear / 1x1 conv / im2col'd conv / ConvTranspose2d(k2,s2) on the path:
 * image_encoder.py:227,238 (qkv, proj), common.py:25-26 (MLP), image_encoder.py:88-104 (neck),
 * transformer.py:228-254 (decoder projections), mask_decoder.py:56-62 (upscaler), :187 (dino_proj).
 * W is the PyTorch Linear layout.  N % 128 == 0, K % 64 == 0, any M.
 * Kernel selection is a function of the arguments only (csrc/gemm_f16.hip gemm_launch): the hand-scheduled four-wave 256 x 256 x 64
 * kernel (gemm4w, generated assembly main loop) for fp16 outputs with N >= 2048 and for fp32 outputs + fp32 residual from 192 tiles,
 * K % 128 == 0; 128-column tiles of 64 / 96 / 128 / 256 rows otherwise.  Every kernel accumulates an output element over the same
 * ascending chain of 32-wide MFMA steps and rounds its epilogue the same way: results do not depend on the kernel chosen. */
int csam_gemm_f16(void* stream, const void* A_f16, long lda, const void* W_f16, long ldw, void* C, long ldc,
                  int c_dtype, const float* bias, const float* colscale, const void* residual, long ldr,
                  int r_dtype, int act, int M, int N, int K);
/* LayerNorm FOLDED into the two GEMMs around it (common.py:38-43 LayerNorm between the residual projections and the
 * qkv / mlp.lin1 projections of image_encoder.py:166-182, and the same pattern in DINOv2's blocks):
 *   LN(x) W^T + b = rstd * (x (gamma (.) W)^T - mean * colsum(gamma (.) W)) + (W beta + b)
 * PRODUCER call (the projection that writes the fp32 residual stream x, c_dtype CSAM_DT_F32): C16_out receives fp16(x),
 * rowstats_out [M][N/128][2] the (sum, sum of squares) of every row's 128-column slices -- no atomics, deterministic.
 * CONSUMER call (A_f16 = that fp16 copy, W_f16 = gamma-folded weight, bias = W beta + b): rowstats_in = the producer's
 * partials with n_partials = K / 128, colsum [N] fp32 = column sums of the fp16 weight actually multiplied, eps as in
 * the LayerNorm.  Either side may be NULL; with all of them NULL this is csam_gemm_f16. */
int csam_gemm_f16_ln(void* stream, const void* A_f16, long lda, const void* W_f16, long ldw, void* C, long ldc, int c_dtype,
                     const float* bias, const float* colscale, const void* residual, long ldr, int r_dtype, int act,
                     int M, int N, int K, void* C16_out, long ldc16, float* rowstats_out, const float* rowstats_in,
                     int n_partials, float eps, const float* colsum);
/* residual row = m % res_mod: adds a per-image [res_mod,N] constant to every prompt's slab of a
 * prompt-stacked GEMM ((keys + key_pe) W = keys W + key_pe W, transformer.py:173-175,186-188). */
int csam_gemm_f16_resmod(void* stream, const void* A_f16, long lda, const void* W_f16, long ldw, void* C, long ldc,
                         int c_dtype, const float* bias, const void* residual, long ldr, int r_dtype, int res_mod,
                         int act, int M, int N, int K);
/* `batch` independent GEMMs (grid.z) with element strides (4 hyper-MLPs, mask_decoder.py:175-179). */
int csam_gemm_f16_batched(void* stream, const void* A_f16, long lda, long strideA, const void* W_f16, long ldw,
                          long strideW, void* C, long ldc, long strideC, int c_dtype, const float* bias,
                          long strideBias, int act, int M, int N, int K, int batch);
/* fp32 VALU linear for the decision-driving last layers of the small heads
 * (mask_decoder.py:175-198 hyper / IoU / parallel IoU / classifier, predictor.py:113-121). */
int csam_linear_f32(void* stream, const float* A, long lda, const float* W, long ldw, const float* bias,
                    const float* residual, long ldr, float* C, long ldc, int M, int N, int K, int act);

int csam_linear_f32_batched(void* stream, const float* A, long lda, long strideA, const float* W, long ldw,
                            long strideW, const float* bias, long strideBias, float* C, long ldc, long strideC,
                            int M, int N, int K, int act, int batch);

/* ---- encoder-side streaming kernels */
/* nn.LayerNorm / LayerNorm2d on token-major rows (image_encoder.py:168,180; common.py:38-43;
 * transformer.py norm1-4). One wave per row, D % 4 == 0, D <= 1280. */
int csam_layernorm(void* stream, const void* x, long ldx, int x_dtype, void* y, long ldy, int y_dtype,
                   const float* gamma, const float* beta, int M, int D, float eps);
/* LayerNorm of contiguous fp32 rows with the casts of its consumers folded in (the decoder's token rows,
 * transformer.py:164-190): y_f32 = LN(x); y_f16 = fp16(y) and ype_f16 = fp16(y + pe) when the pointers are given
 * (pe fp32 [M,D]).  Same arithmetic, in the same order, as csam_layernorm followed by csam_add_cast. */
int csam_layernorm_cast(void* stream, const float* x, const float* gamma, const float* beta, int M, int D, float eps,
                        float* y_f32, void* y_f16_or_null, const float* pe_or_null, void* ype_f16_or_null);
/* Sam.preprocess (sam.py:163-173) + PatchEmbed im2col (image_encoder.py:387-395): raw f32 CHW image
 * (h,w <= 1024) -> A f16 [4096,768]. mean3/std3 are HOST pointers to 3 floats. */
int csam_sam_im2col(void* stream, const float* img_chw, int h, int w, const float* mean3, const float* std3,
                    void* out_f16);
/* preprocess + bilinear 1024->1022 (predictor.py:104) + DINOv2 14x14 patch im2col -> f16 [5329,640];
 * frame = 1024 (raw SAM frame) or 1022 (already resized + normalised tensor). */
int csam_dino_im2col(void* stream, const float* img_chw, int h, int w, int frame, const float* mean3,
                     const float* std3, void* out_f16);
int csam_im2col3x3(void* stream, const void* in_f16, void* out_f16, int C); /* neck 3x3 conv, image_encoder.py:96-102 */
int csam_add_cast(void* stream, const float* a, const float* b, long b_row_stride, void* y_f16, float* y_f32,
                  long M, int N);
int csam_preprocess_pad(void* stream, const float* img_chw, int h, int w, const float* mean3, const float* std3,
                        float* out_3x1024x1024); /* Sam.preprocess materialised (API path) */
int csam_sigmoid_max(void* stream, const float* x, int C, int N, float* out); /* crowdsam/model.py:203 */
/* utils.resize_image -> cv2.resize(image, (w, h)) (crowdsam/utils.py:141-149; INTER_LINEAR, uint8 HWC, 3 channels) on
 * the uploaded frame.  OpenCV's generic fixed-point algorithm restated (third-party, unpinned): xofs int32 [dw],
 * xcoef int16 [dw][2], yofs int32 [dh][2] (both source rows, clamped), ycoef int16 [dh][2] are the 11-bit tables
 * cv::resize builds (crowdsam_amd/resize.py); area2x != 0 selects the exact-2x INTER_AREA route (tables unused).
 * Writes the uint8 HWC frame and / or the fp32 CHW tensor (0..255) the encoder kernels read; either may be NULL. */
int csam_resize_linear_u8(void* stream, const uint8_t* src_hwc, int sh, int sw, const int* xofs, const short* xcoef,
                          const int* yofs, const short* ycoef, int dh, int dw, int area2x, uint8_t* dst_hwc,
                          float* dst_chw_f32);
/* One separable pass of Pillow's Image.resize(size, BILINEAR) for uint8 HWC frames (3 channels): ResizeLongestSide.
 * apply_image (segment_anything_cs/utils/transforms.py:26-31, reached when the frame's long side is 1023, SURVEY.md
 * trap 9).  xmin int32 [n_out], ntap int32 [n_out] (<= 8), coef int32 [n_out][8]: Pillow's 22-bit tables
 * (crowdsam_amd/resize.py, pinned against Pillow itself).  axis 0 resizes x (dh == sh), axis 1 resizes y (dw == sw);
 * dst_chw_f32 (optional) receives the fp32 CHW copy. */
int csam_pil_resample_u8(void* stream, const uint8_t* src_hwc, int sh, int sw, const int* xmin, const int* ntap,
                         const int* coef, int dh, int dw, int axis, uint8_t* dst_hwc, float* dst_chw_f32);
/* SamPredictor.set_image's HWC uint8 -> 1x3xHxW layout change + cast (predictor.py:52-56) */
int csam_u8hwc_to_f32chw(void* stream, const uint8_t* src_hwc, int h, int w, float* dst_chw);

/* ---- attention */
/* 14x14 windowed attention + decomposed rel-pos; window partition/unpartition folded into the
 * addressing; pad tokens are REAL keys whose q/k/v equal the qkv bias (image_encoder.py:224-289,
 * 325-361; SURVEY.md trap 4). qkv f16 [4096,3D] laid out [3][nH][hd]; out f16 [4096,D]; head_dim hd = D / nH is 64
 * (ViT-B / L) or 80 (ViT-H). */
/* relcat_f16 [64,hd]: rows 0..26 rel_pos_h, 27..53 rel_pos_w, rest zero (fp16 like the other weights) */
int csam_win_attn(void* stream, const void* qkv_f16, const float* qkv_bias, const void* relcat_f16, void* out_f16,
                  int D, int nH, float scale);
/* the same over n_images images in one launch (image b = rows b*4096 .. b*4096+4095 of qkv_f16 / out_f16): the batch axis of
 * ImageEncoderViT.forward's x [B,H,W,C] (image_encoder.py:106-116), which the reference's multi-crop loop
 * (crowdsam/model.py:151-178) feeds one crop at a time */
int csam_win_attn_batched(void* stream, const void* qkv_f16, const float* qkv_bias, const void* relcat_f16, void* out_f16,
                          int D, int nH, float scale, int n_images);
/* generic head_dim (<= 128, multiple of 8; ViT-H: 80) attention = gather / batched GEMMs / softmax / scatter, the
 * reference's materialised formulation (image_encoder.py:224-289,325-361).  Groups G = nH (global, Tp = 4096) or
 * 25 windows x nH (Tp = 256, T_valid = 196); operands zero-padded to 128 dims; window pad tokens take the qkv bias. */
int csam_head_gather(void* stream, const void* qkv_f16, const float* qkv_bias, void* Qs_f16, void* K_f16, void* VT_f16,
                     int D, int nH, int head_dim, int Tp, int T_valid, int window, float scale);
/* P = softmax over the T_valid keys of S + (Th[q,kh] + Tw[q,kw]) * inv_scale; relpos_raw f32 [G][Tp][256] = Qs . relcat^T
 * (Th at [q][qh - kh + side - 1], Tw at [q][128 + qw - kw + side - 1]); side = 64 (global) or 14 (window) */
int csam_softmax_relpos(void* stream, const float* S, const float* relpos_raw, void* P_f16, int G, int Tp, int T_valid,
                        int side, float inv_scale);
int csam_head_scatter(void* stream, const void* O_f16, void* out_f16, int D, int nH, int head_dim, int Tp, int T_valid,
                      int window);
/* flash-style global attention, head_dim 64: every DINOv2 block (relpos_raw == NULL, ragged T) and SAM's
 * global blocks: relpos_raw fp32 [nH][4096][256] = q . [rel_pos_h(127) | 0 | rel_pos_w(127) | 0]^T, produced by
 * one csam_gemm_f16_batched over the heads (image_encoder.py:349-350).
 * q_prescaled != 0: the q columns of qkv_f16 (and therefore relpos_raw) already carry scale * log2(e), folded into the
 * qkv projection's weights by the caller -- the scores then need no per-element scaling; 0: plain q, scaled in-kernel. */
long csam_flash_attn_workspace_bytes(int T, int nH); /* per-head V^T scratch; zero-initialise it once */
int csam_flash_attn(void* stream, const void* qkv_f16, long ld, int q_off, int k_off, int v_off,
                    const float* relpos_raw, void* out_f16, long ldo, int T, int nH, float scale, void* vt_workspace,
                    long vt_workspace_bytes, int q_prescaled);
/* the same over n_images sequences of T tokens each in one launch (image b = rows b*T .. b*T+T-1 of qkv_f16 / out_f16;
 * relpos_raw [n_images][nH][4096][256]; V^T scratch n_images x csam_flash_attn_workspace_bytes(T, nH)): the batch axis B of
 * Attention.forward (image_encoder.py:224-240) and of DINOv2's blocks */
int csam_flash_attn_batched(void* stream, const void* qkv_f16, long ld, int q_off, int k_off, int v_off,
                            const float* relpos_raw, void* out_f16, long ldo, int T, int nH, float scale, void* vt_workspace,
                            long vt_workspace_bytes, int q_prescaled, int n_images);
/* the same at head_dim 80 (ViT-H global blocks): qkv heads 80 wide, V^T scratch nH x 80 x Tpad, q scaled in-kernel; relpos_raw
 * as above, from plain q (e.g. one batched GEMM with K = 128 over a [256,128] table whose columns 80.. are zero). */
long csam_flash_attn80_workspace_bytes(int T, int nH);
int csam_flash_attn80(void* stream, const void* qkv_f16, long ld, int q_off, int k_off, int v_off,
                      const float* relpos_raw, void* out_f16, long ldo, int T, int nH, float scale, void* vt_workspace,
                      long vt_workspace_bytes);

/* ---- prompt encoder + two-way decoder (all prompts of a batch at once) */
/* prompt_encoder.py:75-93,189-218 + mask_decoder.py:153-155: tokens f32 [B,7,256] =
 * [iou; mask0..3; PE(point)+point_embed[1]; not_a_point]; coords f32 [B,2] in the 1024 frame. */
int csam_point_tokens(void* stream, const float* coords, const float* gauss_2x128, const float* out_tokens5,
                      const float* point_embed1, const float* not_a_point, float* tokens, int B);
/* the same with a label per prompt (prompt_encoder.py:88-92): 1 foreground, 0 background (point_embed[0]), -1 not-a-point */
int csam_point_tokens_labeled(void* stream, const float* coords, const int* labels, const float* gauss_2x128,
                              const float* out_tokens5, const float* point_embed0, const float* point_embed1,
                              const float* not_a_point, float* tokens, int B);
/* box prompts (prompt_encoder.py:95-102,152-163 with points == None; predictor.py:214-292 `boxes`): tokens f32 [B,7,256] =
 * [iou; mask0..3; PE(x0,y0)+point_embed[2]; PE(x1,y1)+point_embed[3]] -- two corner tokens, no padding point: the same seven
 * tokens per prompt, so the decoder entries below serve box prompts unchanged.  boxes f32 [B,4] XYXY in the 1024 frame. */
int csam_box_tokens(void* stream, const float* boxes_xyxy, const float* gauss_2x128, const float* out_tokens5,
                    const float* point_embed2, const float* point_embed3, float* tokens, int B);
int csam_pe_points(void* stream, const float* coords, const float* gauss_2x128, float* out, int P); /* dense PE */
int csam_token_self_attn(void* stream, const void* qk_f16, const void* v_f16, void* out_f16, int B); /* transformer.py:164-169 */
long csam_attn_t2i_workspace_bytes(int B, int nsplit);
/* token->image attention (transformer.py:173-177,105-112): 7 queries x T keys x 8 heads x 16 */
int csam_attn_t2i(void* stream, const void* q_f16, const void* K_f16, const void* V_f16, long ldkv,
                  long kv_prompt_stride, void* out_f16, int B, int T, int nsplit, void* workspace,
                  long workspace_bytes);
/* image->token attention (transformer.py:186-190): T image queries x 7 token keys */
int csam_attn_i2t(void* stream, const void* Qi_f16, long ldq, long q_prompt_stride, const void* k_f16,
                  const void* v_f16, void* out_f16, int B, int T, int nsplit);
int csam_ln64_gelu(void* stream, void* x_f16, const float* gamma, const float* beta, long rows, float eps); /* mask_decoder.py:58-59 */
int csam_hyper_masks(void* stream, const void* up2_f16, const float* hyper, float* masks, int B); /* mask_decoder.py:181 */
/* PWD-Net pooling (mask_decoder.py:186-190), re-associated through the adjoint of the 73->256 bilinear */
int csam_softmax_stats(void* stream, const float* masks, float* stats, int rows);
int csam_adj_taps_bytes(void);
int csam_pool_adjoint(void* stream, const float* masks, const float* stats, const void* taps_dev, void* w_f16,
                      long ldw, int rows);
/* the same on the matrix cores: out = U^T exp(x - max) U with the banded U packed into MFMA fragment blocks by the
 * host (csam_adj_mfma_bytes() bytes, layout in decoder.hip); one wave per plane, bound by the fp32 read of the logits */
int csam_adj_mfma_bytes(void);
int csam_pool_adjoint_mfma(void* stream, const float* masks, float* stats, const void* tables_dev, void* w_f16,
                           long ldw, int rows);
int csam_rowscale_bias(void* stream, const float* P, const float* stats, const float* bias, float* out, int rows,
                       int N);
/* Epilogue of a split-K GEMM (round 4; the small-batch decoder's skinny GEMMs -- mask_decoder.py's MLP second layer
 * transformer.py:215-216 at K = 2048 and the PWD-Net pooling product at K = 5376 -- run as `splits` K-slices through
 * csam_gemm_f16_batched and are summed here in slice order, bit-repeatably):
 * out[r,c] = (sum_s partials[s * slab_stride + r * N + c]) * (stats ? 1 / stats[2 r + 1] : 1) + bias[c] + residual[r * ldr + c];
 * stats / bias / residual may be NULL.  N % 4 == 0. */
/* The token side of a two-way decoder block for small prompt batches (round 4; csrc/token_block.hip): one 16-wave workgroup
 * per two prompts walks the launch sequence it replaces with the activations in LDS.
 * _a: transformer.py:164-170 token self-attention (q | k and v projections, 7 x 7 attention per head, out projection,
 *     + residual unless residual_or_null is NULL = layer 0's skip_first_layer_pe form, norm1) and the q projection of the
 *     token->image attention (:173-177).  from_tokens != 0: both operands are fp16(tokens0) and src_* are ignored.
 *     Writes queries fp32 [B*7,256], q16 = fp16(queries), qpe16 = fp16(queries + tokens0), t2i_q fp16 [B*7,128].
 * _b: out projection of the token->image attention + residual + norm2 (:175-177), MLP 256 -> 2048 ReLU -> 256 + residual +
 *     norm3 (:180-183), the k / v projections of the image->token attention (:186-190) and, when next_q_w is given, the q
 *     projection of the NEXT token->image attention.  queries is read (residual) and rewritten.  With t2i_partials the
 *     attention output is merged from csam_t2i_fused's partial records here (attn_o_f16 is ignored), in csam_t2i_merge's arithmetic.
 * fp16 weights in FRAGMENT ORDER: element (n, k) of a row-major [N][K] matrix at
 *   ((n / 16 * (K / 32) + k / 32) * 64 + (k % 32 / 8) * 16 + n % 16) * 8 + k % 8
 * (one 1 KB wave load per MFMA A-fragment); biases / LayerNorm parameters fp32.  Bit-identical to the launch sequence they
 * replace (csam_gemm_f16, csam_token_self_attn, csam_layernorm_cast ... in single-pass form). */
int csam_token_block_a(void* stream, const void* src_qk_f16, const void* src_v_f16, const float* tokens0, int from_tokens,
                       const float* residual_or_null, const void* qk_w_f16, const float* qk_b, const void* v_w_f16,
                       const float* v_b, const void* o_w_f16, const float* o_b, const float* norm_g, const float* norm_b,
                       float eps, const void* q_w_f16, const float* q_b, float* queries, void* q16, void* qpe16,
                       void* t2i_q_f16, int B);
int csam_token_block_b(void* stream, const void* attn_o_f16, const float* t2i_partials_or_null, int nparts, float* queries, const float* tokens0, const void* o_w_f16,
                       const float* o_b, const float* norm2_g, const float* norm2_b, const void* mlp1_w_f16,
                       const float* mlp1_b, const void* mlp2_w_f16, const float* mlp2_b, const float* norm3_g,
                       const float* norm3_b, const void* k_w_f16, const float* k_b, const void* v_w_f16, const float* v_b,
                       const void* next_q_w_f16_or_null, const float* next_q_b_or_null, float eps, void* q16, void* qpe16,
                       void* i2t_k_f16, void* i2t_v_f16, void* t2i_q_f16_or_null, int B);
/* Small batches: everything between the final token->image attention and the upscaler in one launch -- out projection +
 * residual + final LayerNorm (transformer.py:105-112), the four hyper-network MLPs (mask_decoder.py:175-179: 256 -> 256 -> 256 -> 32
 * on mask tokens 1..4; third layer fp32; fp16 weights in the fragment order of csam_token_block_a, per matrix), the IoU head (:184, on token 0; third layer fp32, 4 outputs) and the parallel residual
 * IoU head (:194-198: [token 0 | mask token l] 512 -> 256 -> 256 -> 1, + the IoU head's output l).  hyper_w0 / w1 fp16 [4][256][256],
 * hyper_w2 fp32 [4][32][256]; iou_w0 / w1 fp16 [256][256], iou_w2 fp32 [4][256]; par_w0 fp16 [256][512], par_w1 fp16 [256][256],
 * par_w2 fp32 [1][256].  Writes hyper fp32 [B][4][32], iou0 fp32 [B][4], res_iou fp32 [B][4].  IoU outputs bit-identical to the
 * launch sequence csam_gemm_f16 / _batched + csam_layernorm_cast + csam_linear_f32 / _batched it replaces, hyper outputs to
 * the last fp32 bit. */
int csam_token_heads(void* stream, const void* attn_o_f16, const float* t2i_partials_or_null, int nparts, const float* queries, const void* o_w_f16, const float* o_b,
                     const float* norm_g, const float* norm_b, float eps, const void* hyper_w0_f16, const float* hyper_b0,
                     const void* hyper_w1_f16, const float* hyper_b1, const float* hyper_w2, const float* hyper_b2,
                     const void* iou_w0_f16, const float* iou_b0, const void* iou_w1_f16, const float* iou_b1,
                     const float* iou_w2, const float* iou_b2, const void* par_w0_f16, const float* par_b0,
                     const void* par_w1_f16, const float* par_b1, const float* par_w2, const float* par_b2, float* hyper_out,
                     float* iou0_out, float* res_iou_out, int B);
int csam_splitk_reduce(void* stream, const float* partials, int splits, long slab_stride, const float* stats_or_null,
                       const float* bias_or_null, const float* residual_or_null, long ldr, float* out, long ldo, int rows,
                       int N);

/* ---- fused decoder kernels: one pass over the per-prompt key state each (SURVEY.md 8d's R1..R4) */
/* whole image->token half-block (transformer.py:186-190): [Q-proj] -> 7-key attention -> out-proj ->
 * +residual -> LayerNorm4.  Give either Q (hoisted layer-0 projection) or Wq+qpe. Wo columns k-permuted. */
int csam_i2t_fused(void* stream, const void* X_f16, long x_prompt_stride, const void* Q_f16, long q_prompt_stride,
                   const void* Wq_f16, const float* qpe, const void* k_f16, const void* v_f16,
                   const void* Wo_perm_f16, const float* bo, const float* gamma, const float* beta, float eps,
                   void* out_f16, int B, int T);
/* token->image attention with the K/V projections fused in (csam_t2i_fused with keys + weights) as a persistent,
 * weight-stationary flash pass: 4-wave workgroups walk whole prompts, Wk/Wv slices in registers, 32-key tiles LDS-DMA'd a
 * tile ahead, online softmax, out [B,7,128] written once per prompt (no workspace, no merge). transformer.py:173-177,105-112 */
/* The same attention in rank-56 form (7 queries per prompt: transformer.py:173-177,105-112 with the K / V projections folded
 * into the token side): scores = X (Wk^T q) + (pe Wk^T) q, output = softmax-weighted sums of the RAW key rows, Y f16
 * [B,7,8 heads,256]; Wv and out_proj follow as one GEMM over K = 8 x 256 (weights folded by the caller).  q_scaled f16
 * [B,7,128] = (queries + pe) Wq^T + bq, pre-multiplied by 0.25 log2(e); Wk f16 [128,256]; kpe f16 [T,128];
 * Qp_workspace >= B*64*256*2 bytes (back-projected queries, built by the prologue kernel). */
int csam_t2i_rank(void* stream, const void* X_f16, const void* Wk_f16, const void* kpe_f16, const void* q_scaled_f16,
                  void* Qp_workspace, long workspace_bytes, void* Y_f16, int B, int T);
int csam_t2i_stream(void* stream, const void* X_f16, const void* Wkv_f16, const float* kpe, const float* bv,
                    const void* q_f16, void* out_f16, int B, int T);
/* same half-block as a persistent, weight-stationary stream (one 8-wave workgroup per CU walks 128-token tiles; the
 * projection weights stay in registers, key tiles are LDS-DMA'd one tile ahead).  k_scaled_f16 [B,7,128] is the
 * token-side k projection PRE-MULTIPLIED by 0.25*log2(e); Wo_f16 is the plain [256,128] out-proj weight. */
int csam_i2t_stream(void* stream, const void* X_f16, long x_prompt_stride, const void* Q_f16, long q_prompt_stride,
                    const void* Wq_f16, const float* qpe, const void* k_scaled_f16, const void* v_f16,
                    const void* Wo_f16, const float* bo, const float* gamma, const float* beta, float eps,
                    void* out_f16, int B, int T);
/* the hoisted-Q (layer 0) form of the same half-block in its rank-56 formulation: out_proj(softmax(q K_b^T) V_b) = P_b M_b
 * with M_b[(head, key), :] = Wo[:, head] v_b[key, head] built per prompt into `workspace` (csam_i2t_rank_workspace_bytes).
 * Every wave owns 16 tokens with all 256 channels, so there is no barrier inside a prompt.  Q_f16 [T,128] is the hoisted
 * projection (bias + pe included), k_scaled_f16 as for csam_i2t_stream, Wo_f16 the plain [256,128] out-proj weight. */
long csam_i2t_rank_workspace_bytes(int B);
int csam_i2t_rank(void* stream, const void* X_f16, long x_prompt_stride, const void* Q_f16, long q_prompt_stride,
                  const void* k_scaled_f16, const void* v_f16, const void* Wo_f16, const float* bo, const float* gamma,
                  const float* beta, float eps, void* out_f16, int B, int T, void* workspace, long workspace_bytes);
/* csam_i2t_rank for a layer whose keys differ per prompt (layer 1, transformer.py:186-190 with q = (keys + pe) Wq^T + bq):
 * the image-side q projection is never formed -- qpe_f16 [T,128] = pe Wq^T + bq is the shared part of the scores and
 * X . Kp_b^T with Kp_b[8 h + j] = Wq_h^T k_b[j, h] (56 back-projected token keys per prompt, a prologue kernel) the
 * per-prompt part.  workspace: csam_i2t_rank_proj_workspace_bytes(B). */
long csam_i2t_rank_proj_workspace_bytes(int B);
int csam_i2t_rank_proj(void* stream, const void* X_f16, long x_prompt_stride, const void* qpe_f16, const void* Wq_f16,
                       const void* k_scaled_f16, const void* v_f16, const void* Wo_f16, const float* bo,
                       const float* gamma, const float* beta, float eps, void* out_f16, int B, int T, void* workspace,
                       long workspace_bytes);
/* csam_i2t_rank / csam_i2t_rank_proj (Wq_f16 null: hoisted-Q form, Q_f16 = the shared image-side queries [T,128]; else
 * the projected form, Q_f16 = qpe_f16) with the token->image attention of the NEXT block (transformer.py:173-177 of layer
 * L+1, or :105-112) folded in: the reader half of each workgroup takes the new key rows from LDS instead of HBM.  The reader
 * operands are csam_t2i_rank's (t2i_Wk_f16 [128,256], t2i_kpe_f16 [T,128], t2i_qs_f16 [B*7,128] scaled queries of the NEXT
 * block -- they depend on the token side only); Y_f16 [B*7, 8*256] is bit-identical to csam_i2t_rank[_proj] followed by
 * csam_t2i_rank.  workspace: csam_i2t_t2i_workspace_bytes(B). */
long csam_i2t_t2i_workspace_bytes(int B);
int csam_i2t_t2i(void* stream, const void* X_f16, long x_prompt_stride, const void* Q_f16, long q_prompt_stride,
                 const void* Wq_f16, const void* k_scaled_f16, const void* v_f16, const void* Wo_f16, const float* bo,
                 const float* gamma, const float* beta, float eps, void* out_f16, const void* t2i_Wk_f16,
                 const void* t2i_kpe_f16, const void* t2i_qs_f16, void* Y_f16, int B, int T, void* workspace,
                 long workspace_bytes);
/* csam_i2t_t2i with per-tile constants folded out of the producer's loop (transformer.py:186-190): fold bit 0 -- the
 * out-projection bias rides in M_b (bo / 8 per valid slot; a head's 7 softmax weights sum to 1), so no bias rows are read per
 * tile; fold = 3 (projected form only) additionally stores the PLAIN normalised keys: the caller has folded the LayerNorm's
 * gamma / beta into the consumers of these keys -- the first conv of csam_upscale_* (W1 (.) gamma, b1 + W1 beta) and the final
 * attention (t2i_Wk_f16 = Wk (.) gamma here; Wc (.) gamma and bc + Wc beta in the GEMM after Y).  fold = 0 is csam_i2t_t2i. */
int csam_i2t_t2i_fold(void* stream, const void* X_f16, long x_prompt_stride, const void* Q_f16, long q_prompt_stride,
                      const void* Wq_f16, const void* k_scaled_f16, const void* v_f16, const void* Wo_f16, const float* bo,
                      const float* gamma, const float* beta, float eps, void* out_f16, const void* t2i_Wk_f16,
                      const void* t2i_kpe_f16, const void* t2i_qs_f16, void* Y_f16, int B, int T, void* workspace,
                      long workspace_bytes, int fold);
/* mask_decoder.py:172-181: ConvT -> LayerNorm2d -> GELU -> ConvT -> GELU -> hyper-network product */
int csam_upscale_fused(void* stream, const void* keys_f16, const void* W1_f16, const float* b1,
                       const float* ln_gamma, const float* ln_beta, float eps, const void* W2_perm_f16,
                       const float* b2, const float* hyper, float* masks, float* stats_or_null, int B);
/* same as a persistent, weight-stationary stream: 4-wave workgroups walk whole prompts in 32-token tiles, W1 slices in
 * registers, key tiles LDS-DMA'd a tile ahead; stats column 0 (per-plane max) written once per prompt, no atomics.  With fewer
 * prompts than resident workgroups (B < 2 x CUs) the workgroups walk RANGES of tiles instead and the maxima go to stats by atomic
 * max after an initialising launch (csam_upscale_fused's protocol) */
int csam_upscale_stream(void* stream, const void* keys_f16, const void* W1_f16, const float* b1,
                        const float* ln_gamma, const float* ln_beta, float eps, const void* W2_perm_f16,
                        const float* b2, const float* hyper, float* masks, float* stats_or_null, int B);
/* token->image attention with the K/V projections fused in (transformer.py:173-177,105-112).  out_f16 NULL: the partial
 * records stay in the workspace -- fp32 [B][csam_t2i_fused_parts()][8 heads][7 queries][18] = (running max, sum, 16 accumulators)
 * of an online softmax in base 2 -- for a consumer that merges them itself (csam_token_block_b, csam_token_heads) */
int csam_t2i_fused_parts(void);
long csam_t2i_fused_workspace_bytes(int B);
int csam_t2i_fused(void* stream, const void* X_f16, const void* Wkv_f16, const float* kpe, const float* bv,
                   const void* K0_f16, const void* V0T_f16, const void* q_f16, void* out_f16, int B,
                   void* workspace, long workspace_bytes);
/* layer-0 token->image attention over the SHARED (per-image) K / V: q f16 [B,7,128]; Kh f16 [8][256][16 keys][16 d] and
 * Vh f16 [8][256][16 d][16 keys] = per-head 16-key tiles of the hoisted projections; out f16 [B,7,128] */
int csam_t2i_shared(void* stream, const void* q_f16, const void* Kh_f16, const void* Vh_f16, void* out_f16, int B);
int csam_t2i_merge_launch(void* stream, const float* part, void* out_f16, int B, int nparts);

/* ---- PWD-Net selection, fused mask post-processing, EPS occupancy */
/* crowdsam/model.py:351,325,354: s = clamp(iou,0)*sigmoid(cls[...,0]); sel = first argmax over 4 */
int csam_select_masks(void* stream, const float* iou, const float* cls, int n_class, int* sel, float* score,
                      int* category, float* fused_or_null, int B);
/* sam.py:153-161 + amg.py:156-176,303-346 on the SELECTED candidate only: bilinear x4 (+ second resize
 * when original != input size), >thr mask bytes, stability counts, bbox extents. */
int csam_mask_post(void* stream, const float* lowres, const int* sel, int B, int in_h, int in_w, int out_h,
                   int out_w, float thr, float off, void* out_mask_u8, int* inter, int* uni, int* box,
                   float* tmp_f32);
/* statistics pass of csam_mask_post that skips the prompts the predicted-IoU filter drops anyway (score[b] <= score_thr;
 * score_thr <= 0: none skipped), as the reference filters on iou_preds before it computes stability
 * (crowdsam/model.py:371-381) */
int csam_mask_post_scored(void* stream, const float* lowres, const int* sel, const float* score, float score_thr, int B,
                          int in_h, int in_w, int out_h, int out_w, float thr, float off, int* inter, int* uni,
                          int* box, float* tmp_f32);
/* two-pass mode: csam_mask_post with out_mask_u8 == NULL gives the statistics only; after csam_post_finalize
 * the bytes of the surviving prompts (keep[b] != 0) are produced by csam_mask_write. */
int csam_mask_write(void* stream, const float* lowres, const int* sel, const void* keep_u8, const int* slot_or_null,
                    int B, int in_h, int in_w, int out_h, int out_w, float thr, void* out_mask_u8, float* tmp_f32);
/* finalize + in-kernel compaction of the survivors into an image-level store (MaskData.cat of
 * crowdsam/model.py:247 without host sync or gather copies): slot[b] = store index or -1.
 * edge10_host (HOST pointer, NULL = off): {crop_box[4], orig_box[4], downscale, atol} of utils.is_box_near_crop_edge
 * (crowdsam/utils.py:213-223), applied per batch BEFORE the occupancy flags as crowdsam/model.py:386-389 does.
 * n_valid_or_null (DEVICE pointer): with the device-resident sampler only the first *n_valid slots of the batch hold a
 * prompt (csam_eps_select); the others are dropped here. */
int csam_post_finalize_compact(void* stream, const float* score, const int* inter, const int* uni, const int* box,
                               const int* category, const int* points_xy, float pred_iou_thresh,
                               float stability_thresh, float filter_thresh, void* keep_u8, void* occ_u8, int* slot,
                               int* counter, float* out_score, float* out_stability, int* out_box, int* out_category,
                               int* out_points, int B, int capacity, const float* edge10_host,
                               const int* n_valid_or_null);
int csam_bilinear_f32(void* stream, const float* src, int n, int sh, int sw, float* dst, int H, int W);
/* crowdsam/model.py:371-389,246: keep / occupancy flags, stability = inter/union, empty box -> 0 */
int csam_post_finalize(void* stream, const float* score, const int* inter, const int* uni, int* box,
                       float pred_iou_thresh, float stability_thresh, float filter_thresh, float* stability,
                       void* keep_u8, void* occ_u8, int B);
/* crowdsam/model.py:238: occupancy bits of the remaining points (never the mask itself goes D2H) */
int csam_occupancy_lookup(void* stream, const int* points_xy, int P, const void* masks_u8, const void* occ_u8,
                          const int* slot_or_null, int B, int H, int W, void* out_u8);
/* Efficient Prompt Sampler without a host round trip per batch (crowdsam/model.py:233-249): the shuffled point list and
 * one alive flag per point live on the device.  csam_eps_select: the next batch = the first min(B, #alive) alive points in
 * list order -- what points[:batch_size] is on the host -- written to out_points_xy [B,2] and, through
 * ResizeLongestSide.apply_coords in float64 (transforms.py:33-41: x * scale_x, y * scale_y, then the fp32 cast), to
 * out_coords [B,2]; their flags are cleared; counts2[0] = number of valid slots, counts2[1] = points still alive.
 * csam_occupancy_prune: alive[p] &= !(point p lies under a mask of this batch with its occupancy flag set), the
 * points[~occupy_mask[y, x]] of :238-239. */
int csam_eps_select(void* stream, const int* points_xy, void* alive_u8, int P, int B, double scale_x, double scale_y,
                    int* out_points_xy, float* out_coords, int* counts2);
int csam_occupancy_prune(void* stream, const int* points_xy, int P, const void* masks_u8, const void* occ_u8,
                         const int* slot_or_null, int B, int H, int W, void* alive_u8);

/* ---- NMS + RLE */
long csam_box_nms_workspace_bytes(int N);
/* torchvision.ops.batched_nms semantics (crowdsam/model.py:171,257,429): stable descending scores,
 * suppress IoU > thr; kept indices (int64) in descending-score order; N <= 16384. */
int csam_box_nms(void* stream, const float* boxes, const float* scores, int N, float thr, long* out_keep,
                 int* out_count, void* workspace, long workspace_bytes);
/* crowdsam/utils.py:422-467 mask_iou_nms + coverage (opt-in test.mask_nms_thresh): masks nearest-resampled to
 * 150x150, greedy in descending score, drop when max(inter/|A|, inter/|B|) > thr against a kept mask; N <= 16384 */
long csam_mask_nms_workspace_bytes(int N);
int csam_mask_nms(void* stream, const void* masks_u8, const float* scores, int N, int H, int W, float thr,
                  long* out_keep, int* out_count, void* workspace, long workspace_bytes);
/* amg.py:107-135 mask_to_rle_pytorch: column-major change positions, two passes, over the masks masks_u8[idx[i]] (idx NULL: i) --
 * the run-length encoder reads the kept masks in their store slots instead of a gathered copy (crowdsam/model.py:288-296).
 * masks_u8 must hold strict 0 / 1 bytes (what csam_mask_write produces and what a torch.bool tensor is): the W % 4 == 0 kernels
 * compare bit 0 of packed bytes.  (Round 6 dropped csam_rle_{count,write}[_idx]: the box forms with boxes == NULL are they.)
 * With the masks' bounding boxes (int32 [N,4] = x0, y0, x1, y1, inclusive maxima; NULL = scan everything): a column
 * changes value only inside the box rows (and at row 0 against the column to its left), so the passes read the boxes instead
 * of the frames -- person-sized masks of a crowd frame cover a few percent of it. */
int csam_rle_count_box(void* stream, const void* masks_u8, const int* idx_or_null, const int* boxes_or_null, int N, int H, int W,
                       int* col_offsets, int* totals);
int csam_rle_write_box(void* stream, const void* masks_u8, const int* idx_or_null, const int* boxes_or_null, int N, int H, int W,
                       const int* col_offsets, const long* mask_offsets, uint32_t* out_positions);
/* COCO compressed-RLE strings of N masks ON THE DEVICE from the change positions above (amg.py:294-300 coco_encode_rle ->
 * pycocotools rleToString: run k coded as cnt[k] - (k > 2 ? cnt[k-2] : 0) in 5-bit groups + 48): positions uint32, mask i =
 * [pos_offsets[i], pos_offsets[i+1]) (device, N + 1 entries); first_pixel u8 [N] = the mask's pixel (0, 0) (a set first pixel
 * = a leading zero-length run); hw = H * W.  max_counts >= total positions + 2 N bounds the launch.  Strings go to out_chars
 * back to back, string i = [str_offsets[i], str_offsets[i+1]) (device, N + 1 entries; [N] = total length -- when it exceeds
 * out_cap the characters past out_cap were dropped and the caller must retry with a larger buffer; 5 bytes per count suffice
 * up to H * W = 2^24).  No host synchronisation. */
long csam_coco_rle_pack_workspace_bytes(int N, long max_counts);
int csam_coco_rle_pack(void* stream, const uint32_t* positions, const long* pos_offsets, const uint8_t* first_pixel, int N,
                       long hw, long max_counts, void* workspace, long workspace_bytes, char* out_chars, long out_cap,
                       long* str_offsets);

/* ---- small-region clean-up (amg.py:267-291 remove_small_regions mode "holes" then "islands", 8-connected,
 * as driven by crowdsam/model.py:394-443): masks u8 [n,H,W] -> out u8 (may alias masks), changed int32 [n]
 * (either pass modified the mask), boxes f32 [n,4] XYXY of the edited masks (amg.py:293-324) */
/* Compact form of csam_small_regions (round 3): the same hole filling + island removal on the masks masks_base[idx[i]]
 * (idx NULL: i) written to out_base[idx[i]] (out_base may equal masks_base: in place) -- the NMS survivors are cleaned up
 * in their store slots, and no per-pixel label array exists (ring-forest labelling, csrc/regions.hip).  Replaces the host
 * loop of crowdsam/model.py:394-443 over amg.py:267-291 exactly like csam_small_regions. */
long csam_small_regions_idx_workspace_bytes(int n, int H, int W);
int csam_small_regions_idx(void* stream, const uint8_t* masks_base, const int* idx_or_null, uint8_t* out_base, int* changed,
                           float* boxes, int n, int H, int W, int min_area, void* workspace, long workspace_bytes);
/* Windows of masks <-> a dense [n, Hc, Wc] stack, for the bounding-box-restricted form of the clean-up above (amg.py:267-291
 * sees the whole frame, but every component of a mask lies inside its box: with a ring of background around the box the
 * clean-up of the window IS the clean-up of the frame -- crowdsam/model.py::postprocess_small_regions).  windows int32
 * [n,6] = (x0, y0, w, h, ox, oy): the window inside the H x W store slot and its place in the stack (ox + w <= Wc,
 * oy + h <= Hc), Wc % 4 == 0.  to_store 0: gather (stack zero outside
 * the window, bytes normalised to 0 / 1); 1: scatter back (only_u8 [n] non-NULL: only the masks flagged there). */
int csam_mask_window_copy(void* stream, void* store_u8, const int* slots_or_null, const int* windows, const void* only_u8_or_null,
                          void* crop_u8, int n, int H, int W, int Hc, int Wc, int to_store);
long csam_small_regions_workspace_bytes(int n, int H, int W);
int csam_small_regions(void* stream, const uint8_t* masks, uint8_t* out, int* changed, float* boxes, int n, int H,
                       int W, int min_area, void* workspace, long workspace_bytes);

/* ---- CrowdHuman evaluator, Caltech matching (tools/crowdhuman_eval.py:113-143 compare_caltech with :215-236
 * box_overlap_opr), float64, one wave per image.  dt [Nd,5] = x0,y0,x1,y1,score in descending score per image,
 * gt [Ng,5] = x0,y0,x1,y1,tag with the gt_npos[i] positive boxes of image i first; offsets are CSR rows.
 * label[d] = 1 matched, 0 false positive, -1 dropped (ignore region / image without GT); pos[d] as the reference. */
int csam_caltech_match(void* stream, const double* dt, const long* dt_off, const double* gt, const long* gt_off,
                       const int* gt_npos, int n_img, int max_pos, double thres, signed char* label,
                       unsigned char* pos);

/* fuse_simmap (crowdsam/model.py:273-286): sum / count over every mask's pixels of the [fh,fw] prior map resized
 * bilinearly (align_corners False) to H x W; masks u8 [n,H,W], sim f32 rows of stride ld_sim, sum f64 [n], count i32 [n] */
int csam_mask_mean_bilinear(void* stream, const uint8_t* masks, int n, int H, int W, const float* sim, int fh, int fw,
                            int ld_sim, double* sum, int* count);

/* host helper (HOST pointers): COCO compressed-RLE string of run lengths (amg.py:294-300 / pycocotools
 * rleToString); returns the length or -1 when cap is too small (13 chars per run always suffice) */
long csam_coco_rle_string(const long long* counts, long n, char* out, long cap);
/* the strings of n masks in one call (a crowded frame keeps hundreds of masks): counts back to back with boundaries
 * offs[n + 1]; strings back to back in out with boundaries out_offs[n + 1]; returns the total length or -1 */
long csam_coco_rle_strings(const long long* counts, const long long* offs, long n, char* out, long cap,
                           long long* out_offs);

#ifdef __cplusplus
}
#endif
#endif /* CSAM_H */
