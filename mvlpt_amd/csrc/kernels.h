// Launcher declarations for the hand-written gfx950 kernels (gemm.hip, norm.hip, attention.hip, glue.hip).
// Every launcher only enqueues on the given stream; no hidden synchronisation.
#pragma once
#include "common.h"

namespace mvlpt {

// ---------------------------------------------------------------- streams with a compute-unit partition (engine.hip)
// Compute units the kernels of `s` can run on: the partition size for a stream made by mvlpt_stream_create_cus, the whole
// device otherwise.  Persistent / grid-stride launchers size their grids from it.
int stream_cus(hipStream_t s);

// ---------------------------------------------------------------- GEMM  C = A * Bt^T (+ epilogue)
enum GemmEpi {
  EPI_STORE16 = 0,  // out16 = acc (+bias)
  EPI_GELU = 1,     // u = acc+bias ; out2 (optional, 16-bit) = u ; out16 = QuickGELU(u)
  EPI_RESID32 = 2,  // out32 = acc + bias + resid32          (fp32 residual stream, may alias out)
  EPI_GELUBWD = 3,  // out16 = acc * QuickGELU'(aux16)       (aux = saved pre-activation u)
  EPI_STORE32 = 4,  // out32 = acc (+bias)
  // split-precision outputs: a 16-bit pair (hi = round16(v), lo = round16(v - hi)) stored as ONE row of 2N elements
  // [hi(0..N) | lo(0..N)], i.e. directly the `a_split` A operand of the next GEMM
  EPI_GELU_SPLIT = 5,     // u = acc+bias ; out2 (optional, 16-bit [M,N]) = u ; out [M,2N] = split(QuickGELU(u))
  EPI_GELUBWD_SPLIT = 6,  // out [M,2N] = split(acc * QuickGELU'(aux16))
  EPI_STORE_SPLIT = 7,    // out [M,2N] = split(acc (+bias))
  // EPI_RESID32 that also PRODUCES the input of the LayerNorm behind it in folded form (GemmArgs::ln_*, "LayerNorm folding")
  EPI_RESID32_LN = 8,
  // internal: EPI_STORE16 / EPI_GELU / EPI_STORE_SPLIT / EPI_GELU_SPLIT with the consumer side of the folding compiled in
  // (launch_gemm selects them when GemmArgs::fold_part is set; callers pass the plain values)
  EPI_STORE16_FOLD = 9, EPI_GELU_FOLD = 10, EPI_STORE_SPLIT_FOLD = 11, EPI_GELU_SPLIT_FOLD = 12,
  // EPI_RESID32_LN on the PACKED residual stream (fp16, single operands; GemmArgs::rp_*): the stream is read and written as
  // hi (16-bit, [M,N]) + lo (one byte, [M,N]) — common.h respk_* — and the hi plane IS the consumer's A operand (the LayerNorm's
  // gamma is folded into the consumer's weight instead of into the activation): no fp32 stream, no separate 16-bit copy
  EPI_RESIDP_LN = 13,
};
constexpr int epi_base(int epi) {
  return epi == EPI_STORE16_FOLD ? EPI_STORE16 : epi == EPI_GELU_FOLD ? EPI_GELU : epi == EPI_STORE_SPLIT_FOLD ? EPI_STORE_SPLIT
         : epi == EPI_GELU_SPLIT_FOLD ? EPI_GELU_SPLIT : epi;
}
constexpr bool epi_folds(int epi) { return epi >= EPI_STORE16_FOLD && epi <= EPI_GELU_SPLIT_FOLD; }
constexpr bool epi_ln_producer(int epi) { return epi == EPI_RESID32_LN || epi == EPI_RESIDP_LN; }
struct GemmArgs {
  const void* A;       // [M,K] 16-bit
  const void* Bt;      // [N,K] 16-bit
  int M, N, K;
  const float* bias;   // [N] or null
  const void* aux;     // [M,N] 16-bit (EPI_GELUBWD)
  const float* resid;  // [M,N] fp32   (EPI_RESID32)
  void* out;           // [M,N]
  void* out2;          // [M,N] 16-bit, optional (EPI_GELU)
  // Split-precision A operand: A is [M, 2K] = [A_hi | A_lo] (16-bit pair, A ~ A_hi + A_lo to ~22 bits) and the
  // product is A_hi*Bt^T + A_lo*Bt^T in the same fp32 accumulators: the K loop runs over 2K with the Bt K-index wrapping.
  // The frozen weights are exactly representable in 16 bits (they are stored so, clip/model.py:371-392), so the product
  // then carries the activations at ~fp32 precision.  Costs twice the MFMA work.
  // a_split == 2 (mixed pair): A is [M, 2K] 16-bit slots per row (same pitch) holding [A_hi (K x 16 bit) | A_lo8 (K bytes) | unused],
  // A_lo8 = e5m2(A_lo * 2^Lo8<T>::EXP) (common.h), and Bt rows hold [W16 (K x 16 bit) | W8 (K bytes)] with pitch ldb >= 3K/2
  // elements, W8 = e4m3(W * 2^w8_exp): the product is A_hi*W16^T on v_mfma_f32_16x16x32 plus A_lo8*W8^T on
  // v_mfma_scale_f32_16x16x128_f8f6f4 (twice the rate, the e8m0 scale operands undo the two exponents) in the same
  // accumulators: 1.5x the matrix time of a single-operand GEMM instead of 2x; the operand is carried to ~2^-14 (fp16) / ~2^-11
  // (bf16) of its value (e5m2 residual x e4m3 weight copy; tests/test_hip_mixed_pair.py asserts these per-element bounds).
  // K % 128 == 0.
  int a_split = 0;
  int ldb = 0;         // Bt row pitch in 16-bit elements (0: K)
  int w8_exp = 0;      // a_split == 2: exponent of the weight's fp8 plane
  int out_lo8 = 0;     // EPI_GELU_SPLIT / EPI_GELUBWD_SPLIT: store the pair as [hi | lo8] (mixed pair) instead of [hi | lo]
  // Row pitches in 16-bit elements (0: dense).  lda: rows of A (dense: K, or 2K with a_split); ldo: rows of `out` for the pair-producing
  // epilogues 5 / 6 / 7 (dense: 2N).  A pitch that is a multiple of 4 KiB puts the same K-stage of every row on the same memory
  // channels (M = 7 700, K = 2048 pair rows of 8 KiB: 56 vs 36 us next to K = 1920 / 2176, profiles/r04_pitch_probe.txt): the engine
  // pads such rows by 128 bytes.
  int lda = 0, ldo = 0;
  int xcd_order = 0;   // set by the one-tile-per-workgroup launcher (gemm_pc_kernel): XCD-aware tile order
  // ---- LayerNorm folding: LN(x) W^T = rstd_r * ((x * gamma) W^T)[r,n] - rstd_r * mean_r * (W gamma)[n] + (W beta)[n], so the
  // GEMM in front of a LayerNorm hands the un-normalised row to the GEMM behind it and the LayerNorm pass (one read of the fp32
  // residual stream + one 16-bit write per LayerNorm) disappears.  No atomics, no extra launch: every output tile owns its slots.
  // Producer (EPI_RESID32_LN): besides out32 = acc + bias + resid32 the epilogue stores ln_x16 = round16(out32 * ln_gamma) — the A
  // operand of the consumer, format ln_split: 0 [M,N], 1 hi|lo pair [M,2N], 2 mixed pair (same pitch) — and, per row and block j of
  // 128 output columns (independent of the tile geometry), the partial sums {sum out32, sum out32^2} at ln_part[(row * ln_ntp + j) * 2].
  const float* ln_gamma = nullptr;
  void* ln_x16 = nullptr;
  int ln_split = 0;
  float* ln_part = nullptr;
  int ln_ntp = 0;               // slots per row (>= N / 128, even)
  // Consumer (EPI_STORE16 / EPI_GELU / EPI_STORE_SPLIT / EPI_GELU_SPLIT with fold_part != null): A holds round16(x * gamma);
  // mean / rstd of row r are rebuilt from its fold_nt partials (summed in slot order: deterministic) and the epilogue computes
  // rstd_r * acc - rstd_r * mean_r * fold_colsum[n] + bias[n], with fold_colsum = W gamma and bias = b + W beta precomputed
  // (frozen weights).  K is the length of the normalised rows.  The partials of a tile's rows ride along with its last K-stage
  // (LDS-DMA into a 16 KiB region behind the ring), so the epilogue reads them from LDS.
  const float* fold_part = nullptr;
  const float* fold_colsum = nullptr;
  int fold_ntp = 0, fold_nt = 0;
  // ---- packed residual stream (EPI_RESIDP_LN): the stream in front of the GEMM is rp_hi_in [M,N] fp16 + rp_lo_in [M,N] bytes,
  // the updated stream goes to `out` [M,N] fp16 (hi) + rp_lo_out [M,N] bytes; ln_part / ln_ntp as for EPI_RESID32_LN (the
  // statistics are those of the fp32 value before it is packed).  In place (in == out) is allowed: a tile reads what it rewrites.
  const void* rp_hi_in = nullptr;
  const uint8_t* rp_lo_in = nullptr;
  uint8_t* rp_lo_out = nullptr;
#ifdef MVLPT_GEMM_TRACE
  long long* trace = nullptr;   // debug builds only: per-wave (point id << 56 | s_memtime) records of workgroup 0
#endif
};
// N-tile width the launcher picks for this problem on this stream
int gemm_tile_n(int dtype, int epi, const GemmArgs& g, hipStream_t s);
// ev_start/ev_stop (optional): recorded by the dispatch itself (hipExtLaunchKernelGGL): kernel-exact timing with no
// extra marker packets on the stream.
hipError_t launch_gemm(int dtype, int epi, const GemmArgs& g, hipStream_t s, hipEvent_t ev_start = nullptr,
                       hipEvent_t ev_stop = nullptr);

// gemm_duo.hip: two half-tile wave groups per workgroup, the epilogue of one under the matrix work of the other (launch_gemm
// routes the problems `gemm_duo_takes` accepts there)
bool gemm_duo_takes(int epi, const GemmArgs& g, hipStream_t s);
hipError_t launch_gemm_duo(int dtype, int epi, const GemmArgs& g, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop);

// fp32 GEMM on the f32-input MFMA (exact f32 products, v_mfma_f32_16x16x4_f32) for the two tiny projections
// next to the logits (CLS / EOT rows only): C[M,N] = alpha * A[M,K] * Bt[N,K]^T.  K % 16 == 0, N % 4 == 0.
hipError_t launch_sgemm_bt(const float* A, const float* Bt, float* C, int M, int N, int K, const float* alpha_dev, hipStream_t s);

// ---------------------------------------------------------------- LayerNorm (fp32 statistics)
// Input row r is read at  x + in_row(r) * d  with in_row(r) = row_idx ? row_idx[r] : r * row_mul.
struct LnFwdArgs {
  const float* x; const int32_t* row_idx; int row_mul;
  const float* gamma; const float* beta;
  void* y;          // [rows,d] contiguous, 16-bit (out_dtype = compute dtype) or fp32 (out_dtype = DT_F32)
  int rows, d;
  int split = 0;    // 16-bit outputs only: y is [rows, 2d] = [hi | lo] (split-precision A operand, see GemmArgs::a_split);
                    // 2: same pitch, [hi | lo8 bytes | unused] (mixed pair)
};
hipError_t launch_ln_fwd(int out_dtype, const LnFwdArgs& a, hipStream_t s);

struct LnBwdArgs {
  const void* dy;        // [rows,d] contiguous, 16-bit (dy_dtype = compute dtype) or fp32 (dy_dtype = DT_F32)
  int dy_dtype;
  const float* x; const int32_t* row_idx; int row_mul;   // LN input rows (same mapping as forward)
  const float* gamma;
  const float* resid;    // fp32, same row mapping as x, or null:  out32 = resid + dx
  float* out32;          // fp32, same row mapping as x (may alias resid)
  void* out16;           // optional 16-bit copy, same row mapping
  int rows, d;
  int split = 0;         // out16 is [*, 2d] = [hi | lo]; 2: [hi | lo8 bytes | unused] (mixed pair)
};
hipError_t launch_ln_bwd(int dtype, const LnBwdArgs& a, hipStream_t s);

// ---------------------------------------------------------------- attention (head_dim 64, L <= 256)
// qkv: [N*L, 3*d] 16-bit, token = n*L + i, columns [q | k | v], head h at h*64 inside each third.
struct AttnArgs {
  const void* qkv; void* out /*[N*L,d]*/; float* lse /*[N*H*L] or null*/;
  int N, L, H; int causal;
  int q_rows = 0;   // > 0: only queries 0..q_rows-1 of every sequence are computed (last layer: only CLS is consumed)
  int flags = 0;    // experiment bits: 1 = non-temporal K/V staging, 2 = non-temporal output stores
};
// ea / eb (optional): start / stop events of the kernel's own dispatch (hipExtLaunchKernelGGL), for mvlpt_profile_* — resident kernels only
hipError_t launch_attn_fwd(int dtype, const AttnArgs& a, hipStream_t s, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
struct AttnBwdArgs {
  const void* qkv; const void* out; const void* dout; const float* lse;
  float* delta /*[N*H*L] scratch*/; void* dqkv /*[N*L,3d]*/;
  int N, L, H; int causal;
};
hipError_t launch_attn_bwd(int dtype, const AttnBwdArgs& a, hipStream_t s);
int attn_max_len();
// backward when only query 0 of every sequence carries a gradient: o_cls / do_cls are compact [N, H*64] rows
hipError_t launch_attn_bwd_cls(int dtype, const void* qkv, const void* o_cls, const void* do_cls, const float* lse, void* dqkv,
                               int N, int L, int H, hipStream_t s);
// streaming variants (attention_stream.hip): any L, 32 KiB LDS ring; launch_attn_* picks between the two families
hipError_t launch_attn_fwd_stream(int dtype, const AttnArgs& a, hipStream_t s);
hipError_t launch_attn_bwd_stream(int dtype, const AttnBwdArgs& a, hipStream_t s);

// ---------------------------------------------------------------- split-precision attention (attention32.hip), any L
// Split-precision mode of the towers that carry a gradient: every operand is a 16-bit hi|lo pair [rows, 2*cols]
// (hi = round16(x), lo = round16(x - hi)), products keep hi*hi + hi*lo + lo*hi on the 16-bit MFMA (fp32 accumulation);
// the outputs are pairs again: directly the `a_split` A operand of the next GEMM.
struct Attn32Args {
  const void* qkv_split; // [N*L, 6d] 16-bit: [hi(q|k|v) | lo(q|k|v)], head h at h*64 inside each d-wide block
  void* out_split;       // [N*L, 2d] 16-bit: [hi(d) | lo(d)]
  float* lse;            // [N*H*L] or null
  int N, L, H; int causal;
  int q_rows = 0;        // > 0: only queries 0..q_rows-1 of every sequence are computed
  int out_lo8 = 0;       // out_split rows are [hi(d) | lo8 (d bytes) | unused] (mixed pair: the out-projection's A operand)
#ifdef MVLPT_ATTN_TRACE
  long long* trace = nullptr;   // debug builds only (tools/attn_trace.py): (point id << 56 | s_memtime) records of one workgroup
#endif
};
hipError_t launch_attn32_fwd(int dtype, const Attn32Args& a, hipStream_t s);
struct Attn32BwdArgs {
  const void* qkv_split; const void* out_split; const void* dout_split /*[N*L, 2d]*/; const float* lse;
  float* delta /*[N*H*L] scratch*/; void* dqkv_split /*[N*L, 6d] 16-bit: [hi(3d) | lo(3d)]*/;
  int N, L, H; int causal;
  int lo8 = 0;           // out_split rows are [hi | lo8] (read for delta) and dqkv_split rows are written as [hi(3d) | lo8 (3d bytes) | unused]
#ifdef MVLPT_ATTN_TRACE
  long long* trace = nullptr;   // debug builds only (tools/attn_trace.py bwd): records of one workgroup of attn32r_bwd_kernel
#endif
};
hipError_t launch_attn32_bwd(int dtype, const Attn32BwdArgs& a, hipStream_t s);

// ---------------------------------------------------------------- input pipeline (preprocess.hip)
// same layout as MvlptImageDesc (include/mvlpt_hip.h)
struct PpDesc {
  int64_t offset;                            // byte offset of pixel (0,0) of this image in the packed uint8 HWC source
  int32_t height, width;
  int32_t crop_top, crop_left, crop_h, crop_w;
  int32_t resize_h, resize_w;                // size the crop box is resampled to
  int32_t out_top, out_left;                 // window of the resized image that is produced (CenterCrop); 0,0 for training
  int32_t flip, reserved;
};
hipError_t launch_preprocess(const uint8_t* src, const PpDesc* descs_dev, int B, int max_crop_h, int ks_max, int out_h, int out_w,
                             int32_t* tables, size_t table_stride, uint8_t* tmp, size_t tmp_stride, const float* mean,
                             const float* stdv, void* out, int out_dtype, uint8_t* out_u8, hipStream_t s);

// ---------------------------------------------------------------- glue
hipError_t launch_cast_f32_to16(int dtype, const float* in, void* out, size_t n, const float* scale_dev, hipStream_t s);
// fp32 [rows,d] (* scale_dev[0]) -> 16-bit pair [rows, 2d] = [hi | lo]
// (lo8 = 1: [hi | lo8 bytes | unused], the mixed pair)
hipError_t launch_cast_f32_split(int dtype, const float* in, void* out, size_t rows, int d, const float* scale_dev, hipStream_t s, int lo8 = 0);
hipError_t launch_cast_any_to_f32(int in_dtype, const void* in, float* out, size_t n, hipStream_t s);
// W [rows, cols] (fp32) -> out16 [rows, ld_out] zero padded (cols <= ld_out)
hipError_t launch_pack_weight(int dtype, const float* w, void* out, int rows, int cols, int ld_out, hipStream_t s);
// fp8 plane of a packed weight: out8[r, c] = e4m3(W[r, c] * scale_dev[0]) (transposed = 1: out8[c, r]), zero padded to `cols_out`
// bytes per row; `out8` points at the plane inside the 16-bit rows, pitch_bytes apart
hipError_t launch_pack_weight8(const float* w, uint8_t* out8, int rows, int cols, int transposed, int cols_out, size_t pitch_bytes,
                               const float* scale_dev, hipStream_t s);
// W [rows, cols] (fp32) -> out16 [cols, rows] (transposed)
hipError_t launch_pack_weight_t(int dtype, const float* w, void* out, int rows, int cols, hipStream_t s, int ld_out = 0);
// image [B,3,R,R] (fp32 / f16 / bf16) -> patches16 [B*g*g, Kp], column order (c,ky,kx), zero padded to Kp
hipError_t launch_patchify(int dtype, const void* image, int image_dtype, void* out, int B, int R, int P, int Kp, hipStream_t s);
// tokens: row0 = LN(cls + pos0); rows 1..n = vpt; rest = LN(patch + pos)   (trainers/mvlpt.py:56-62)
hipError_t launch_assemble_tokens(const float* patch_emb, const float* cls, const float* pos, const float* g, const float* b,
                                  const float* vpt, int n_vpt, float* x, int B, int G2, int d, hipStream_t s,
                                  const float* vmask = nullptr /* [B, n_vpt, d] dropout mask of the prompt rows, or null */);
// the same rows (no prompts) straight into the packed residual stream (hi [B*(1+G2), d] fp16, lo bytes) + the rows' {sum, sum of
// squares} in slot 0 of `part` [rows][ntp][2]
hipError_t launch_assemble_tokens_packed(const float* patch_emb, const float* cls, const float* pos, const float* g, const float* b,
                                         void* hi, uint8_t* lo, float* part, int ntp, int B, int G2, int d, hipStream_t s);
// x[b, 1+j, :] = rows[j, :]   (deep prompt overwrite, trainers/mvlpt.py:78-82)
hipError_t launch_overwrite_rows(const float* rows, int n, float* x, int B, int L, int d, hipStream_t s, const float* vmask = nullptr);
// prompts (forward_coop + positional embedding, trainers/mvlpt.py:439-515, 107/112)
hipError_t launch_assemble_prompts(const float* prefix, const float* suffix, const float* ctx, int ctx_per_class, int n_ctx,
                                   const int32_t* layout, const float* pos, float* x, int C, int L, int d, hipStream_t s);
hipError_t launch_build_ctx_pos(const int32_t* layout, int32_t* ctx_pos, int C, int L, int n_ctx, hipStream_t s);
hipError_t launch_eot_rows(const int32_t* eot, int32_t* rows, int C, int L, hipStream_t s);
// dst[r] = src[idx[r]] (scatter = 0) or dst[idx[r]] = src[r] (scatter = 1); row_bytes % 16 == 0
hipError_t launch_copy_rows(const void* src, void* dst, const int32_t* idx, int rows, int row_bytes, int scatter, hipStream_t s);
// rows of row_bytes (multiple of 16) from src + r*src_pitch to dst + r*dst_pitch
hipError_t launch_copy_rows_strided(const void* src, void* dst, int rows, size_t src_pitch, size_t dst_pitch, int row_bytes, hipStream_t s);
// out[j,:] = inv_scale * sum_b dx[b, row0+j, :]; optionally zero those rows of dx32/dx16 afterwards
hipError_t launch_reduce_prompt_rows(int dtype, float* dx32, void* dx16, int B, int L, int d, int row0, int n, float* out,
                                     const float* scale_dev, int zero_after, hipStream_t s, int split16 = 0, const float* vmask = nullptr);
// dctx (generic) [n,d] = inv_scale * sum_c dx[c, ctx_pos[c,j], :]  or (per class) [C,n,d]
hipError_t launch_gather_ctx_grad(const float* dx, const int32_t* ctx_pos, int C, int L, int d, int n_ctx, int per_class,
                                  float* dctx, const float* scale_dev, hipStream_t s);
// scale_dev[0] = 2^k with amax(|v|)*2^k ~ target ; scale_dev[1] = 1/scale_dev[0]
hipError_t launch_grad_scale(const float* v, size_t n, float target, float* scale_dev, hipStream_t s);
hipError_t launch_zero(void* p, size_t bytes, hipStream_t s);
hipError_t launch_checksum(const void* p, size_t bytes, unsigned long long* out, hipStream_t s);      // debug (mvlpt_debug_checksums)
// LayerNorm folding: colsum[n] = sum_k W16[n,k] gamma[k], bias2[n] = b[n] + sum_k W16[n,k] beta[k]  (W16 [N, ld] packed weight)
hipError_t launch_fold_vectors(int dtype, const void* W16, int ld, const float* gamma, const float* beta, const float* b,
                               float* colsum, float* bias2, int N, int K, hipStream_t s);

// packed residual stream (common.h respk_*, EPI_RESIDP_LN; fp16 only): gamma folded into the consumer's weight
// (Wg = round16(W16 * gamma), colsum = its row sums), fp32 rows -> packed rows (+ LayerNorm-folding partials in slot 0 of ntp,
// or part = null), packed rows r * row_mul -> fp32 rows
hipError_t launch_fold_weight(const void* W16, int ld, const float* gamma, void* Wg16, int ldg, float* colsum, int N, int K, hipStream_t s);
hipError_t launch_respk_pack_rows(const float* x, void* hi, uint8_t* lo, float* part, int ntp, int rows, int d, hipStream_t s);
hipError_t launch_respk_unpack_rows(const void* hi, const uint8_t* lo, int row_mul, float* out, int rows, int d, hipStream_t s);

// ---------------------------------------------------------------- head: cosine logits + cross-entropy (fp32)
hipError_t launch_normalize_rows(const float* x, float* xn, float* norm, int rows, int d, hipStream_t s);
hipError_t launch_logits(const float* imn, const float* txn, float scale, const int32_t* lo, const int32_t* hi,
                         float* logits, int B, int C, int e, hipStream_t s);
hipError_t launch_cross_entropy(const float* logits, const void* labels, int label_kind, int B, int C, float* row_loss,
                                float* loss, float* dlogits, float* ncorrect, hipStream_t s);
hipError_t launch_logits_bwd(const float* dlogits, const float* imn, const float* txn, const float* inorm, const float* tnorm,
                             float scale, const int32_t* lo, const int32_t* hi, float* dimg, float* dtxt,
                             int B, int C, int e, hipStream_t s);

}  // namespace mvlpt
