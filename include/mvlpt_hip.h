/* libmvlpt_hip.so — C ABI of the MI355X-native prompted-CLIP hot path.
 *
 * Drop-in boundary for the reference's `CustomCLIP.forward` + loss/backward as driven by
 * `MVLPT.forward_backward` (reference: trainers/mvlpt.py:540-583 and :910-932).  The reference has no FFI
 * (it is pure Python on PyTorch ops), so each entry point below names the Python call site it replaces;
 * `INTEGRATION.md` shows the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain C, raw DEVICE pointers + sizes, no torch types.  The caller (PyTorch-ROCm host code) owns every
 *     tensor buffer it passes; the library owns only its handle (packed frozen weights, saved activations,
 *     workspace).
 *   - every call only ENQUEUES work on `stream` (a hipStream_t): no hidden synchronisation, except that the
 *     first call at a new, larger problem size grows the workspace (hipMalloc).
 *   - return 0 on success, <0 on error; `mvlpt_last_error()` gives the message.  One handle per process per
 *     GPU, not thread-safe (the reference drives the device from a single Python thread).
 *   - prompt tensors and features cross the boundary as fp32; the towers compute in `compute_dtype`
 *     (fp16 or bf16 MFMA inputs, fp32 accumulation, fp32 residual stream, fp32 LayerNorm/softmax/CE).
 *   - token / class / task indexing is integer and exact.
 */
#ifndef MVLPT_HIP_H
#define MVLPT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* mvlpt_stream_t; /* hipStream_t */

enum { MVLPT_DT_F32 = 0, MVLPT_DT_F16 = 1, MVLPT_DT_BF16 = 2 };
enum { MVLPT_LABEL_INT64 = 0, MVLPT_LABEL_PROB_F32 = 1 };
enum { MVLPT_ERR_ARG = -1, MVLPT_ERR_HIP = -2, MVLPT_ERR_STATE = -3, MVLPT_ERR_UNSUPPORTED = -4 };

/* Architecture of the frozen CLIP (same quantities clip.model.build_model infers, clip/model.py:395-418). */
typedef struct MvlptArch {
  int image_resolution, patch_size, vision_width, vision_layers, vision_heads;
  int context_length, text_width, text_layers, text_heads;
  int embed_dim;
  int compute_dtype; /* MVLPT_DT_F16 (reference default PREC, train.py:131) or MVLPT_DT_BF16 */
} MvlptArch;

/* Precision mode of the towers (reference knob: TRAINER.MVLPT.PREC, trainers/mvlpt.py:835-836, 848-850).
 * All modes: 16-bit MFMA operands, fp32 accumulation, fp32 residual stream / LayerNorm / softmax / cross-entropy.
 *   MVLPT_PREC_FAST       single 16-bit operands everywhere (prompt gradients within ~4e-3 of the fp32 CPU path)
 *   MVLPT_PREC_SPLIT_GRAD default: a tower whose forward is saved for a backward runs with SPLIT operands — every GEMM A
 *                         operand is hi = round16(x) plus its rounding residual as one e5m2 byte (the "mixed pair" below: the
 *                         residual term runs on the fp8 MFMA against an e4m3 copy of the frozen weight: e5m2 keeps 2 mantissa bits
 *                         of a residual that is <= 2^-11 |x| (fp16) / 2^-8 |x| (bf16), e4m3 3 bits of the weight, so an operand is
 *                         carried to ~2^-14 (fp16) / ~2^-11 (bf16) of its value — the per-element bounds tests/test_hip_mixed_pair.py
 *                         asserts; a K = 768 product lands at ~1e-5 relative against 2e-4 with single operands),
 *                         the attention core takes 16-bit hi+lo pairs with three-term products — so prompt gradients match the
 *                         fp32 CPU path to 1e-3; forward-only towers (e.g. the image tower under CoOp, inference) stay fast.
 *                         Environment MVLPT_SPLIT_LO8=0 selects 16-bit hi+lo pairs for the GEMMs as well (twice the matrix time)
 *   MVLPT_PREC_SPLIT_ALL  split operands in every tower, 16-bit hi+lo pairs (~22 bits) everywhere (PREC = "fp32") */
enum { MVLPT_PREC_FAST = 0, MVLPT_PREC_SPLIT_GRAD = 1, MVLPT_PREC_SPLIT_ALL = 2 };

int mvlpt_create(const MvlptArch* arch, void** handle);
/* switch the precision mode (takes effect at the next tower forward) */
int mvlpt_set_precision(void* handle, int mode);
/* LayerNorm folding (no counterpart in the reference, which calls nn.LayerNorm as its own op, clip/model.py:186-187): inside a
 * tower the GEMM in front of a LayerNorm (out-projection / MLP down-projection + fp32 residual) also writes round16(x * gamma)
 * and per-row partial sums, and the GEMM behind it (QKV / MLP up-projection) applies mean and rstd in its epilogue — the
 * stand-alone LayerNorm pass over the residual stream disappears.  mode 0: off, 1: image tower, 2: both towers (default;
 * environment MVLPT_LN_FOLD); towers with fewer than `min_rows` token rows (default 4096) keep the stand-alone kernel. */
int mvlpt_set_ln_fold(void* handle, int mode, int min_rows);
/* Packed residual stream (ON by default since round 6; environment MVLPT_RESID_PACKED = 0 switches it off): an fp16 image tower that has
 * no prompt rows and keeps nothing for a backward (the CoOp configurations, BASELINE configs[0..1]; clip/model.py:185-188 `x = x + ...`)
 * carries the residual stream as hi = round16(x) + one byte with the next 8 bits of x instead of fp32, and hi is at the same time the
 * 16-bit operand of the GEMM behind every LayerNorm (its gamma folded into the frozen weight): 6 instead of 10 bytes of memory
 * traffic per element and residual update, x carried to 2^-20 (|x| saturates at 65504); image tower alone -1.7 %, headline step -0.9 %.
 * Round 5 kept it off because ~1 tower in 1 000 - 6 000 returned ONE image 1e-3 off under a concurrent text tower: that was the gfx950
 * packed-fp32 hazard in the tower entry kernel (NOTES_experiments.md round 6), gone since the library is built without packed fp32
 * instructions: 0 in 20 000 towers.  0: the fp32 stream everywhere. */
int mvlpt_set_resid_packed(void* handle, int on);
/* `vpt_dropout` of the reference (trainers/mvlpt.py:165, 424 and :77): the visual prompt rows are expanded over the batch and THEN
 * dropped out, so every image has its own mask.  masks = fp32 [n_layers, B, n_vpt, width] on the device, 0 or 1 / (1 - p): layer 0
 * belongs to the shallow prompts, layer l >= 1 to the deep prompts spliced in front of block l.  ONE-SHOT: the NEXT mvlpt_image_fwd
 * checks the extents against its own (n_layers >= 1 + n_deep, batch, n_vpt; a mismatch is MVLPT_ERR_ARG), multiplies the prompt rows
 * it writes with the masks, hands the pointer to ITS mvlpt_image_bwd (which multiplies the gradients it sums over the batch) and
 * clears the setting — a later forward without a new call runs without dropout.  The caller keeps the buffer alive until that
 * backward.  masks = NULL clears a pending setting. */
int mvlpt_set_vpt_dropout(void* handle, const float* masks, int n_layers, int batch, int n_vpt, int width);
/* Workspaces only grow, and a block that was outgrown is retired (not freed) so that no step ever meets a device-wide sync.
 * mvlpt_trim synchronises the device and releases the retired blocks: call it at an epoch boundary (e.g. after a one-off large
 * evaluation batch or class list). */
int mvlpt_trim(void* handle);
/* DEBUG (tools/tower_stage_probe.py; off by default, no cost when off): while enabled, mvlpt_image_fwd adds one 64-bit fingerprint
 * (position-weighted sum of the 32-bit words) per intermediate — patches, patch embedding, token assembly, and per packed block qkv,
 * attention output, the stream + row statistics behind each of the two updates, the MLP activations — to a device array it clears at
 * its start.  The call synchronises the device, copies up to max_out fingerprints of the LAST forward to host_out (may be NULL),
 * switches the recording on / off and returns the number copied.  Two runs on the same input agree entry by entry; the first entry
 * that differs names the kernel. */
int mvlpt_debug_checksums(void* handle, int enable, unsigned long long* host_out, int max_out);
int mvlpt_destroy(void* handle);
const char* mvlpt_last_error(void* handle); /* handle may be NULL for create() failures */
const char* mvlpt_version(void);

/* Streams confined to a partition of the compute units.  The reference runs the two towers one after the other on one
 * stream (`image_features = self.image_encoder(...)`, `text_features = self.text_encoder(...)`, trainers/mvlpt.py:543-548);
 * here they are independent until the logits and run side by side.  The image tower's persistent GEMM workgroups hold
 * every CU they are given for the whole launch, so the text tower's short, wide kernels would only ever run in their
 * tails: a stream from mvlpt_stream_create_cus owns logical compute units [cu_first, cu_first + cu_count)
 * (hipExtStreamCreateWithCUMask; logical CU i sits on XCD i % 8, so a multiple of 8 is the same share of every XCD), and
 * every launcher of this library sizes persistent / grid-stride grids by the stream's partition (mvlpt_stream_cus: the
 * partition size, or the device's CU count for any other stream).  Any hipStream_t still works everywhere. */
int mvlpt_stream_create_cus(int cu_first, int cu_count, mvlpt_stream_t* stream);
int mvlpt_stream_destroy(mvlpt_stream_t stream);
int mvlpt_stream_cus(mvlpt_stream_t stream);

/* Frozen weights (replaces `self.model.to(self.device)` for the CLIP towers, trainers/mvlpt.py:867, with a
 * one-time pack: 16-bit copy for the forward GEMM and a pre-transposed copy for the dX GEMM — legal because
 * every non-prompt parameter is frozen, trainers/mvlpt.py:855-858).  `name` is the key of
 * clip.model.CLIP.state_dict() ("visual.conv1.weight", "transformer.resblocks.3.mlp.c_fc.bias", ...).
 * `dev_ptr` is a contiguous device tensor of `dtype` (fp32/fp16/bf16).  Unknown names return MVLPT_ERR_ARG. */
int mvlpt_load_frozen(void* handle, const char* name, const void* dev_ptr, int dtype, const int64_t* shape, int ndim,
                      mvlpt_stream_t stream);
/* 0 when every tensor the towers need has been loaded; otherwise <0 and last_error names the first missing. */
int mvlpt_frozen_ready(void* handle);

/* ImageEncoder.forward (trainers/mvlpt.py:52-93).  image [B,3,R,R] of `image_dtype`; vpt [n_vpt,dv] fp32 or
 * NULL (shallow prompts, forward_vpt :416-437); vpt_deep [n_deep,n_vpt,dv] fp32 or NULL (deep prompts for
 * layers 1..n_deep, :73-83).  feat_out [B,embed] fp32.  save_for_bwd != 0 keeps activations for image_bwd. */
int mvlpt_image_fwd(void* handle, const void* image, int image_dtype, const float* vpt, const float* vpt_deep, int n_vpt,
                    int n_deep, int B, float* feat_out, int save_for_bwd, mvlpt_stream_t stream);
/* dX-only backward of the image tower: dfeat [B,embed] fp32 -> dvpt [n_vpt,dv], dvpt_deep [n_deep,n_vpt,dv]
 * (sum over the batch: prompts are `expand`ed, :75,:424).  Must follow image_fwd(save_for_bwd=1), same B. */
int mvlpt_image_bwd(void* handle, const float* dfeat, float* dvpt, float* dvpt_deep, mvlpt_stream_t stream);

/* forward_coop + TextEncoder.forward (trainers/mvlpt.py:439-515, 105-130).
 * prefix [C,1,dt], suffix [C,L-1-n_ctx,dt] fp32 (the `token_prefix` / `token_suffix` buffers, :312-316);
 * ctx [n_ctx,dt] (ctx_per_class=0) or [C,n_ctx,dt] (CSC, ctx_per_class=1) fp32, NULL when n_ctx == 0;
 * layout int32 [C,L]: source row per position — 0 = prefix, e>0 = suffix row e-1, e<0 = ctx row -e-1
 * (encodes the end/middle/front layouts); eot int32 [C] = argmax of tokenized_prompts (:128).
 * L <= context_length (CUT_CONTEXTLEN gives L < 77, :111-117).  feat_out [C,embed] fp32. */
int mvlpt_text_fwd(void* handle, const float* prefix, const float* suffix, const float* ctx, int ctx_per_class, int n_ctx,
                   const int32_t* layout, const int32_t* eot, int C, int L, float* feat_out, int save_for_bwd,
                   mvlpt_stream_t stream);
/* dfeat [C,embed] fp32 -> dctx (same shape as ctx).  Must follow text_fwd(save_for_bwd=1). */
int mvlpt_text_bwd(void* handle, const float* dfeat, float* dctx, mvlpt_stream_t stream);

/* Cosine logits (trainers/mvlpt.py:550-554) with the multiplicative per-task mask (:573-581):
 * logits[b,c] = exp(logit_scale) * <img_b/|img_b|, txt_c/|txt_c|> * [task_lo[b] <= c < task_hi[b]].
 * task_lo/task_hi int32 [B] or NULL (no mask).  fp32 throughout. */
int mvlpt_logits_fwd(void* handle, const float* img_feat, const float* txt_feat, float logit_scale_exp, const int32_t* task_lo,
                     const int32_t* task_hi, int B, int C, float* logits, mvlpt_stream_t stream);
/* dlogits [B,C] -> dimg [B,embed], dtxt [C,embed] (either may be NULL).  Uses the features of the last
 * mvlpt_logits_fwd call on this handle. */
int mvlpt_logits_bwd(void* handle, const float* dlogits, float* dimg, float* dtxt, mvlpt_stream_t stream);

/* F.cross_entropy(output, label) with mean reduction (trainers/mvlpt.py:931) and its gradient.
 * labels: int64 [B] (MVLPT_LABEL_INT64) or fp32 probabilities [B,C] (MVLPT_LABEL_PROB_F32, rows already
 * normalised as in :914-916).  loss [1]; dlogits [B,C] or NULL; ncorrect [1] or NULL (top-1 hits, with
 * argmax(label) as the target for soft labels, :935-936). */
int mvlpt_cross_entropy(void* handle, const float* logits, const void* labels, int label_kind, int B, int C, float* loss,
                        float* dlogits, float* ncorrect, mvlpt_stream_t stream);

/* ---- kernel-level entry points (what the parity tests call; same kernels the towers use) ------------------ */
/* C[M,N] = A[M,K] * Bt[N,K]^T with epilogue `epi` (0 store16(+bias), 1 bias+QuickGELU (out2 = pre-activation),
 * 2 fp32 out = acc+bias+resid32, 3 out16 = acc*QuickGELU'(aux16), 4 fp32 store).  K % 64 == 0, N % 128 == 0. */
int mvlpt_op_gemm(int dtype, int epi, const void* A, const void* Bt, int M, int N, int K, const float* bias, const void* aux,
                  const float* resid, void* out, void* out2, mvlpt_stream_t stream);
/* same with a split-precision A operand: A is [M, 2K] = [A_hi | A_lo] (16-bit pair), C = (A_hi + A_lo) * Bt^T; additional
 * epilogues 5 (out [M,2N] = hi|lo pair of QuickGELU(acc+bias), out2 = pre-activation) and 6 (pair of acc*QuickGELU'(aux)).
 * Note: the saved pre-activation `out2` is ONE 16-bit value per element in every mode (QuickGELU' is evaluated on it in the
 * backward): the one operand of the split towers that is not a pair; the parity budget carries it (gradients 1-7e-4 of 1e-3). */
int mvlpt_op_gemm_split(int dtype, int epi, const void* A, const void* Bt, int M, int N, int K, const float* bias, const void* aux,
                        const float* resid, void* out, void* out2, mvlpt_stream_t stream);
/* LayerNorm with the 16-bit output written as a hi|lo pair [rows, 2d] */
int mvlpt_op_layernorm_fwd_split(int out_dtype, const float* x, const float* gamma, const float* beta, void* y, int rows, int d,
                                 mvlpt_stream_t stream);
int mvlpt_op_layernorm_bwd_split(int dtype, const void* dy, const float* x, const float* gamma, const float* resid, float* out32,
                                 void* out16, int rows, int d, mvlpt_stream_t stream);
/* attention core of the split-precision mode; every operand is a 16-bit hi|lo pair [rows, 2*cols] = [hi(cols) | lo(cols)]:
 * qkv [N*L, 6*H*64] -> out [N*L, 2*H*64], lse;  backward: dout [N*L, 2*H*64] -> dqkv [N*L, 6*H*64] (delta: [N*H*L] scratch) */
int mvlpt_op_attention32_fwd(int dtype, const void* qkv, void* out, float* lse, int N, int L, int H, int causal, int q_rows,
                             mvlpt_stream_t stream);
int mvlpt_op_attention32_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                             void* dqkv, int N, int L, int H, int causal, mvlpt_stream_t stream);
/* ---- mixed pair: the default split format of MVLPT_PREC_SPLIT_GRAD.  A row holds [hi (cols x 16 bit) | residual bytes (cols) |
 * unused] at the pair's pitch of 2*cols 16-bit elements: hi = round16(x), byte = e5m2((x - hi) * 2^10) (bf16: 2^7).  A GEMM with
 * such an A operand multiplies hi with the 16-bit weight on v_mfma_f32_16x16x32 and the residual bytes with the weight's e4m3
 * copy on v_mfma_scale_f32_16x16x128_f8f6f4 (twice the rate): 1.5x the matrix time of a single-operand GEMM instead of 2x.
 * pack_weight_mixed: w32 [rows, cols] fp32 -> out [R, 3K/2] 16-bit elements = [W16 (K) | e4m3(W * 2^*w8_exp) (K bytes)] with
 * (R, K) = (rows, cols), or (cols, rows) when `transposed`; synchronises the stream (the exponent is returned to the host).
 * gemm_mixed: epilogues 2 / 4 (fp32 outputs), 7 (16-bit pair for the attention core), 5 / 6 (mixed-pair outputs). K % 128 == 0. */
int mvlpt_op_pack_weight_mixed(int dtype, const float* w32, int rows, int cols, int transposed, void* out, int* w8_exp,
                               mvlpt_stream_t stream);
int mvlpt_op_gemm_mixed(int dtype, int epi, const void* A, const void* Bt, int ldb, int w8_exp, int M, int N, int K, const float* bias,
                        const void* aux, const float* resid, void* out, void* out2, mvlpt_stream_t stream);
int mvlpt_op_cast_mixed(int dtype, const float* in, void* out, int64_t rows, int d, mvlpt_stream_t stream);
/* ---- LayerNorm folding at kernel level (see mvlpt_set_ln_fold): LN(x) W^T + b = rstd_r (x gamma) W^T - rstd_r mean_r (W gamma) + (b + W beta).
 * fold_vectors: colsum = W gamma, bias2 = b + W beta from the PACKED 16-bit weight W16 [N, ld] (load time).
 * gemm_ln_producer: out32 = A Bt^T + bias + resid (as epilogue 2) AND x16 = round16(out32 * gamma) in the A-operand format
 *   x16_split (0 [M,N], 1 hi|lo pair [M,2N], 2 mixed pair) AND part[(row * ntp + j) * 2 ..] = {sum, sum of squares} of the row over
 *   output columns 128 j .. 128 j + 127, *nt = N / 128 slots whatever tile geometry the launch uses (ntp: slots per row, even, >= *nt, <= 8).  A: a_split 0 / 1 / 2 as in op_gemm*.
 * gemm_folded: epilogue `epi` (0, 1, 5, 7) on A16 = x16 with the normalisation applied from `part` (K = length of the rows). */
int mvlpt_op_fold_vectors(int dtype, const void* W16, int ld, const float* gamma, const float* beta, const float* b, float* colsum,
                          float* bias2, int N, int K, mvlpt_stream_t stream);
int mvlpt_op_gemm_ln_producer(int dtype, const void* A, int a_split, const void* Bt, int ldb, int w8_exp, int M, int N, int K,
                              const float* bias, const float* resid, const float* gamma, int x16_split, float* out32, void* x16,
                              float* part, int ntp, int* nt, mvlpt_stream_t stream);
int mvlpt_op_gemm_folded(int dtype, int epi, const void* A16, int a_split, const void* Bt, int ldb, int w8_exp, int M, int N, int K,
                         const float* colsum, const float* bias2, const float* part, int ntp, int nt, void* out, void* out2,
                         mvlpt_stream_t stream);
/* ---- packed residual stream at kernel level (see mvlpt_set_resid_packed; fp16 only).  An element x is stored as hi = round16(x)
 * and lo = clamp((bits(x) - bits(float(hi))) >> 5, -128, 127) (int8): x' = bits(float(hi)) + (lo << 5).
 * fold_weight: Wg = round16(W16 * gamma) [N, ldg], colsum = row sums of Wg (load time).  respk_pack: fp32 rows -> (hi, lo) and,
 *   when part != NULL, {sum, sum of squares} of each row in slot 0 of its ntp slots (the others 0).  respk_unpack: rows r * row_mul
 *   of (hi, lo) -> fp32.  gemm_residp: (hi_out, lo_out) = pack(A Bt^T + bias + unpack(hi_in, lo_in)) (in place allowed) and the
 *   row statistics of the fp32 value as in gemm_ln_producer; the consumer is gemm_folded on A16 = hi_out with Bt = Wg. */
int mvlpt_op_fold_weight(const void* W16, int ld, const float* gamma, void* Wg16, int ldg, float* colsum, int N, int K,
                         mvlpt_stream_t stream);
int mvlpt_op_respk_pack(const float* x, void* hi, uint8_t* lo, float* part, int ntp, int rows, int d, mvlpt_stream_t stream);
/* the tower entry of the packed stream (ImageEncoder.forward's class token + positional embedding + ln_pre, trainers/mvlpt.py:60-66,
 * straight into the packed format): rows [batch, 1 + grid2, d] from patch_emb [batch * grid2, d], cls [d], pos [1 + grid2, d];
 * part as in respk_pack (the statistics of block 0's ln_1) */
int mvlpt_op_assemble_packed(const float* patch_emb, const float* cls, const float* pos, const float* ln_g, const float* ln_b, void* hi,
                             uint8_t* lo, float* part, int ntp, int batch, int grid2, int d, mvlpt_stream_t stream);
int mvlpt_op_respk_unpack(const void* hi, const uint8_t* lo, int row_mul, float* out, int rows, int d, mvlpt_stream_t stream);
int mvlpt_op_gemm_residp(const void* A, const void* Bt, int ldb, int M, int N, int K, const float* bias, const void* hi_in,
                         const uint8_t* lo_in, void* hi_out, uint8_t* lo_out, float* part, int ntp, int* nt, mvlpt_stream_t stream);
int mvlpt_op_layernorm_fwd_mixed(int out_dtype, const float* x, const float* gamma, const float* beta, void* y, int rows, int d,
                                 mvlpt_stream_t stream);
int mvlpt_op_layernorm_bwd_mixed(int dtype, const void* dy, const float* x, const float* gamma, const float* resid, float* out32,
                                 void* out16, int rows, int d, mvlpt_stream_t stream);
/* attention core with 16-bit pair inputs (qkv, dout) and mixed-pair tensors on the GEMM side (out; dqkv) */
int mvlpt_op_attention32_fwd_mixed(int dtype, const void* qkv, void* out, float* lse, int N, int L, int H, int causal, int q_rows,
                                   mvlpt_stream_t stream);
int mvlpt_op_attention32_bwd_mixed(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                                   void* dqkv, int N, int L, int H, int causal, mvlpt_stream_t stream);
int mvlpt_op_layernorm_fwd(int out_dtype, const float* x, const float* gamma, const float* beta, void* y, int rows, int d,
                           mvlpt_stream_t stream);
int mvlpt_op_layernorm_bwd(int dtype, const void* dy, const float* x, const float* gamma, const float* resid, float* out32,
                           void* out16, int rows, int d, mvlpt_stream_t stream);
/* qkv [N*L,3*H*64] 16-bit -> out [N*L,H*64], lse [N*H*L] (may be NULL) */
int mvlpt_op_attention_fwd(int dtype, const void* qkv, void* out, float* lse, int N, int L, int H, int causal,
                           mvlpt_stream_t stream);
int mvlpt_op_attention_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                           void* dqkv, int N, int L, int H, int causal, mvlpt_stream_t stream);
int mvlpt_op_cast(int dtype, const float* in, void* out, int64_t n, mvlpt_stream_t stream);

/* ---- input pipeline ("next" row f3 of the scope table) -------------------------------------------------------
 * Replaces the per-image CPU transform the reference runs in DataLoader workers: Dassl `build_transform` with
 * INPUT.TRANSFORMS = random_resized_crop / random_flip / normalize, INTERPOLATION bicubic, CLIP PIXEL_MEAN/STD
 * (configs/trainers/MVLPT/vit_b16.yaml:8-13), and the ELEVATER eval transform Resize(BICUBIC) [+ CenterCrop] +
 * ToTensor + Normalize (trainers/vision_benchmark/evaluation/feature.py:538-553).  Both are torchvision ops on PIL
 * images: `img.crop(box).resize((rw, rh), BICUBIC)` [+ window] [+ horizontal flip], u8/255, (x - mean)/std.
 * The random parameters (crop box, flip) are drawn by the caller (host RNG, as torchvision does).
 * Results are bit-identical to Pillow (8-bit resample) and torch CPU (fp32 arithmetic). */
typedef struct MvlptImageDesc {
  int64_t offset;                 /* byte offset of pixel (0,0) in `src`; image = uint8 HWC, 3 channels, row stride 3*width */
  int32_t height, width;
  int32_t crop_top, crop_left, crop_height, crop_width;   /* PIL crop box, inside the image */
  int32_t resize_height, resize_width;                    /* size the crop is resampled to */
  int32_t out_top, out_left;                              /* window [out_top, out_top+out_h) x [out_left, out_left+out_w)
                                                             of the resized image that is produced (CenterCrop); 0, 0 if none */
  int32_t flip;                                           /* 1 = horizontal flip of the produced window */
  int32_t reserved;
} MvlptImageDesc;
/* src: device, packed decoded images; descs: HOST array of B descriptors (uploaded on `stream`);
 * out: device [B,3,out_h,out_w] of out_dtype (MVLPT_DT_*), may be NULL; out_u8: device [B,out_h,out_w,3] resized 8-bit
 * image before ToTensor (may be NULL; parity tests).  mean/std: 3 host floats each. */
int mvlpt_preprocess(void* handle, const uint8_t* src, int64_t src_bytes, const MvlptImageDesc* descs, int B, int out_h, int out_w,
                     const float* mean, const float* std, void* out, int out_dtype, uint8_t* out_u8, mvlpt_stream_t stream);

/* ---- per-kernel timing with HIP events on the launch stream (bench.py's roofline leg) --------------------- */
typedef struct MvlptKernelStat {
  char name[32];
  int64_t launches;
  double ms;    /* sum of event-timed launch durations */
  double flops; /* algorithmic FLOPs summed over launches (2*M*N*K for GEMM, 4*L*L*64 per head for attention) */
  double bytes; /* algorithmic HBM bytes summed over launches */
  double busy_ms; /* length of the UNION of the launch intervals: equals `ms` when launches never overlap; smaller
                     when the same kernel runs concurrently on two streams (image and text tower) */
  double flops_executed; /* FLOPs the launches actually issued: 2x `flops` for GEMMs with split-precision operands */
} MvlptKernelStat;
/* all_kernels == 0: only the dominant kernel (gemm_bt) is timed, through its own dispatch timestamps (no marker
 * packets on the stream); != 0: every kernel class is bracketed by marker events (adds ~1.5 us per event). */
int mvlpt_profile_begin(void* handle, int all_kernels);
/* paused != 0: launches are not timed until resumed (bench.py samples every 4th step to keep the overhead ~1 %) */
int mvlpt_profile_pause(void* handle, int paused);
/* synchronises the recorded events, fills up to `max_stats` entries, returns the number written (or <0): first one entry per kernel
 * class (busy_ms = union of the launch intervals), then the GEMM launches once more per problem, longest total time first, named
 * "g<M>x<N>x<K> e<epilogue> s<operand format> f<folded consumer>" (busy_ms = ms = sum of the launch durations) */
int mvlpt_profile_end(void* handle, MvlptKernelStat* stats, int max_stats);

#ifdef __cplusplus
}
#endif
#endif /* MVLPT_HIP_H */
