// Host side of libmvlpt_hip.so: the handle (packed frozen CLIP weights, workspaces, saved activations), the
// layer-by-layer orchestration of the hand-written kernels, and the C ABI declared in include/mvlpt_hip.h.
//
// Data layout in HBM (per tower; N sequences of L tokens, width d, token = n*L + i, batch-major):
//   residual stream x      fp32  [N*L, d]      one buffer per LayerNorm input when activations are saved
//                                              (2*layers+1 buffers), a single in-place buffer otherwise
//   h16 (LN output)        16bit [N*L, d]      reused
//   qkv16                  16bit [N*L, 3d]     per layer when saved (attention backward needs Q,K,V)
//   attn16 (attention out) 16bit [N*L, d]      per layer when saved (delta = rowsum(dO*O))
//   lse                    fp32  [N*H*L]       per layer when saved
//   u16 (pre-GELU)         16bit [N*L, 4d]     per layer when saved;  a16 (GELU out) reused
//   backward: dx32 fp32 [N*L,d] (in place through the layers), dx16, dh16, dO16 [N*L,d], du16 [N*L,4d],
//             dqkv16 [N*L,3d], delta [N*H*L]
// Frozen weights: W16 [out,in] (forward Bt operand) and W16^T [in,out] (dX Bt operand), fp32 biases/LN params.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <array>
#include <map>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mvlpt_hip.h"
#include "kernels.h"

using namespace mvlpt;

// ------------------------------------------------------------------------------------------------ CU-partitioned streams
// A stream made by mvlpt_stream_create_cus only ever gets the compute units of its mask; everything that sizes a grid by
// "resident workgroups" (persistent GEMM, grid-stride LayerNorm) asks stream_cus() instead of the device.  A handful of
// streams per process: a linear scan under a mutex costs nothing next to a launch.

namespace {
struct StreamPart { hipStream_t s; int cus; };
std::mutex g_part_mu;
std::vector<StreamPart> g_parts;
int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  return cus;
}
}  // namespace
namespace mvlpt {
int stream_cus(hipStream_t s) {
  if (s) {
    std::lock_guard<std::mutex> lk(g_part_mu);
    for (const StreamPart& p : g_parts) if (p.s == s) return p.cus;
  }
  return device_cus();
}
}  // namespace mvlpt

namespace {

// ------------------------------------------------------------------------------------------------ small utils
// Grow-only workspace.  Growth (a larger batch / class count than anything seen so far) must not stall the streams: work
// that was enqueued on ANY stream may still be using the old block, so it is neither synchronised on nor freed here — it
// is retired and released with the handle (1.5x geometric growth bounds the retired total by 2x the final size; the
// 288 GB of HBM are not the constraint, a device-wide sync in the middle of a step would be).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  std::vector<void*> retired;
  ~DevBuf() {
    if (p) (void)hipFree(p);
    for (void* q : retired) (void)hipFree(q);
  }
  // release the retired blocks; the caller guarantees that nothing enqueued before still uses them (mvlpt_trim)
  size_t trim() {
    size_t n = retired.size();
    for (void* q : retired) (void)hipFree(q);
    retired.clear();
    return n;
  }
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    size_t got = std::max(bytes, cap + cap / 2);
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, got);
    if (e != hipSuccess && got > bytes) { got = bytes; e = hipMalloc(&q, got); }      // the slack did not fit: exact size
    if (e != hipSuccess) return e;
    if (p) retired.push_back(p);
    p = q; cap = got;
    return hipSuccess;
  }
};
struct Bump {
  char* base = nullptr; size_t off = 0, cap = 0;
  template <typename T> T* take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    T* r = (T*)(base + off); off += bytes; return r;
  }
  void* take_bytes(size_t bytes) { return take<char>(bytes); }
};
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// Packed frozen weight of one nn.Linear: rows of [W16 (K x 16 bit) | W8 (K bytes, e4m3(W * 2^e8))] with a pitch of 3K/2
// elements, in both orientations (GemmArgs::a_split == 2 reads the fp8 plane, every other GEMM only the 16-bit one).
struct WRef { const void* p; int ld; int e8; };
struct Linear {
  void* w = nullptr; void* wt = nullptr; float* b = nullptr; int out = 0, in = 0;
  int ldw = 0, ldwt = 0, e8 = 0;
  float *fold_s = nullptr, *fold_b = nullptr;   // LayerNorm folding: W gamma and b + W beta of the LayerNorm in front (qkv: ln_1, fc: ln_2)
  // packed residual stream (fp16 image tower without a gradient): round16(W * gamma) [out, in] and its row sums
  void* wg = nullptr; float* fold_sg = nullptr;
  WRef fw() const { return WRef{w, ldw, e8}; }      // forward Bt operand  [out, in]
  WRef fwg() const { return WRef{wg, in, 0}; }
  WRef bw() const { return WRef{wt, ldwt, e8}; }    // dX Bt operand       [in, out]
};
struct LNp { float* g = nullptr; float* b = nullptr; };
struct Block { LNp ln1, ln2; Linear qkv, o, fc, pr; };
struct TowerW { int width = 0, layers = 0, heads = 0; std::vector<Block> blocks; };

struct TowerState {
  bool valid = false, saved = false, causal = false;
  bool exact = false;                    // split-precision operands + pair-product attention (see DESIGN.md "Precision modes")
  int wide_pitch = 0;                    // split towers: row pitch of a16 / du16 (0: dense)
  int xs = 0;                            // GEMM A operands: 0 single 16-bit, 1 16-bit pair, 2 mixed pair (hi + e5m2 residual byte)
  int N = 0, L = 0, d = 0, H = 0, layers = 0;
  std::vector<float*> x;                 // 2*layers+1 entries (all equal when !saved)
  std::vector<void*> qkv, attn, u;       // per layer (all equal when !saved)
  std::vector<float*> lse;
  void *h16 = nullptr, *a16 = nullptr;
  float* dx32 = nullptr; void *dx16 = nullptr, *du16 = nullptr, *dO16 = nullptr, *dqkv16 = nullptr; float* dh32 = nullptr;
  float* delta = nullptr; float* scale_dev = nullptr;
  std::vector<char> skip;                // layer skipped (reference deep-prompt quirk, Appendix A.3)
  // LayerNorm folding (kernels.h): per-row, per-N-tile {sum, sum of squares} of the residual stream in front of ln_1 / ln_2,
  // written by the FC2 / out-projection epilogues; h16 then holds round16(x * gamma) instead of LN(x)
  bool fold = false;
  float* part[2] = {nullptr, nullptr};   // [T][FOLD_NTP][2]
  int nt[2] = {0, 0}, ntp[2] = {0, 0};   // slots in use / slots per row, as the last producer wrote them
};
constexpr int FOLD_NTP = 8;              // slots per row the buffers are sized for (gemm.hip: FOLD_MAX_NTP)

enum ProfClass { PC_GEMM = 0, PC_ATTN_FWD, PC_ATTN_BWD, PC_LN_FWD, PC_LN_BWD, PC_GLUE, PC_HEAD, PC_ATTN_FWD_IMG, PC_COUNT };
static const char* kProfNames[PC_COUNT] = {"gemm_bt", "attention_fwd", "attention_bwd", "layernorm_fwd", "layernorm_bwd",
                                           "glue", "head_logits_ce", "attention_fwd_image"};
struct ProfRec { int cls; hipEvent_t a, b; double flops, bytes, flops_exec; int M = 0, N = 0, K = 0, epi = -1, split = 0, fold = 0; };

struct Engine {
  MvlptArch arch{};
  int dt = DT_F16;
  int prec_mode = MVLPT_PREC_SPLIT_GRAD;
  int fold_mode = 2;    // LayerNorm folding: 0 off, 1 image tower, 2 both towers (MVLPT_LN_FOLD, mvlpt_set_ln_fold)
  int fold_min_rows = 1024;   // towers with fewer token rows keep the stand-alone LayerNorm (a handful of tiles: nothing to win).  4096 until round 6: the small-tile
                              // consumer geometries were where the packed-fp32 hazard first showed (NOTES round 6); cfg1 (1 600 image rows): -1.5 %
  bool fold_ready = false;
  // packed residual stream: the fp16 image tower without prompts and without a backward carries x as hi (fp16, at the
  // same time the A operand behind every LayerNorm) + one byte instead of fp32 + a 16-bit copy.  MVLPT_RESID_PACKED / mvlpt_set_resid_packed
  int resid_packed = 1;      // on (round 6): the rare 1e-3 glitch of round 5 was the packed-fp32 hazard in the tower entry, gone with the NOPK build
                             // (0 / 20 000 towers under a concurrent text tower, profiles/r06_packed_stream_determinism.txt); MVLPT_RESID_PACKED=0: fp32 stream
  // mvlpt_set_vpt_dropout: [layers, B, n_vpt, d] masks of the visual prompt rows for the NEXT image_fwd (one-shot: the forward
  // validates the geometry, takes the setting over as v_mask — what ITS backward uses — and clears it, so a stale pointer never
  // reaches another forward or another user of the engine)
  const float* vpt_mask = nullptr;
  int vpt_mask_layers = 0, vpt_mask_B = 0, vpt_mask_n = 0, vpt_mask_d = 0;
  const float* v_mask = nullptr;     // the masks of the forward whose activations are saved (null: none)
  bool lo8 = true;      // split towers of MVLPT_PREC_SPLIT_GRAD use the mixed pair (hi + e5m2 residual byte; MVLPT_SPLIT_LO8=0: 16-bit pairs)
  std::string err;
  TowerW vis, txt;
  // vision extras
  void* conv_w = nullptr; int Kp = 0; float* cls_emb = nullptr; float* vpos = nullptr; LNp ln_pre, ln_post;
  float *vproj = nullptr, *vproj_t = nullptr;  // [dv,e] and [e,dv] fp32 (the projections next to the logits stay fp32)
  // text extras
  float* tpos = nullptr; LNp ln_final; float *tproj = nullptr, *tproj_t = nullptr;
  std::vector<void*> owned;               // every weight allocation (freed in destroy)
  DevBuf vis_ws, txt_ws, head_ws, ce_ws, tmp, pp_ws;
  TowerState vs, ts;
  // vision fwd extras (carved from vis_ws)
  int vB = 0, v_nvpt = 0, v_ndeep = 0; float* cls32 = nullptr; float* dcls32 = nullptr;
  float* xc32 = nullptr; void *ac16 = nullptr, *hc16 = nullptr, *gc16 = nullptr;   // CLS-only last layer (compact [B,·])
  // ... and what its backward needs (VPT / UPT): saved LN inputs, pre-GELU, and the compact gradient buffers
  bool v_cls_last = false; float *xcm32 = nullptr, *xco32 = nullptr, *dxc32 = nullptr, *dhc32 = nullptr;
  void *uc16 = nullptr, *duc16 = nullptr, *dxc16 = nullptr, *dOc16 = nullptr;
  // text fwd extras
  int tC = 0, tL = 0, t_nctx = 0, t_per_class = 0; int32_t* eot_rows = nullptr; int32_t* ctx_pos = nullptr;
  // EOT-only last text block (compact [C,·] rows; the text-side twin of the CLS-only last image block)
  bool t_eot_last = false; float *txc32 = nullptr, *txm32 = nullptr, *txo32 = nullptr, *tdxc32 = nullptr, *tdhc32 = nullptr;
  void *tac16 = nullptr, *thc16 = nullptr, *tgc16 = nullptr, *tuc16 = nullptr, *tduc16 = nullptr, *tdxc16 = nullptr, *tdOc16 = nullptr;
  float* eot32 = nullptr; float* deot32 = nullptr;
  // head state
  int hB = 0, hC = 0; float h_scale = 0.f; const int32_t *h_lo = nullptr, *h_hi = nullptr;
  float *imn = nullptr, *txn = nullptr, *inorm = nullptr, *tnorm = nullptr;
  // profiling
  // mvlpt_debug_checksums: one 64-bit fingerprint per intermediate of the image tower (debug; off by default)
  unsigned long long* dbg_ck = nullptr; int dbg_ck_n = 0; bool dbg_ck_on = false;
  bool prof_on = false, prof_all = false; std::vector<ProfRec> prof; std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
};

thread_local std::string g_create_err;

int fail(Engine* E, int code, const std::string& msg) { if (E) E->err = msg; else g_create_err = msg; return code; }
int hipfail(Engine* E, hipError_t e, const char* what) {
  return fail(E, MVLPT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(E, call) do { hipError_t _e = (call); if (_e != hipSuccess) return hipfail((E), _e, #call); } while (0)

struct ProfScope {
  Engine* E; hipStream_t s; bool on = false; hipEvent_t b{};
  ProfScope(Engine* E_, hipStream_t s_, int cls, double flops, double bytes) : E(E_), s(s_) {
    if (!E || !E->prof_on || !E->prof_all) return;   // marker events cost ~1.5 us each: only in the full-breakdown mode
    if (E->ev_used + 2 > E->ev_pool.size()) {
      if (E->ev_pool.size() >= 65536) return;
      for (int i = 0; i < 512; ++i) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) return; E->ev_pool.push_back(ev); }
    }
    hipEvent_t a = E->ev_pool[E->ev_used++]; b = E->ev_pool[E->ev_used++];
    (void)hipEventRecord(a, s);
    E->prof.push_back(ProfRec{cls, a, b, flops, bytes, flops});
    on = true;
  }
  ~ProfScope() { if (on) (void)hipEventRecord(b, s); }
};

constexpr int DBG_CK_MAX = 256;
static void dbg_ck(Engine* E, const void* p, size_t bytes, hipStream_t s) {
  if (!E->dbg_ck_on || !E->dbg_ck || E->dbg_ck_n >= DBG_CK_MAX) return;
  (void)launch_checksum(p, bytes, E->dbg_ck + E->dbg_ck_n++, s);
}

// the single-operand attention forward (the headline's image tower), timed like the GEMMs by its own dispatch timestamps whenever
// profiling is on (class "attention_fwd_image": bench.py's roofline.image_attention_fwd)
static hipError_t attn_fwd_timed(Engine* E, const AttnArgs& a, double flops, double bytes, hipStream_t s) {
  hipEvent_t ea = nullptr, eb = nullptr;
  if (E && E->prof_on && !E->prof_all) {
    if (E->ev_used + 2 > E->ev_pool.size() && E->ev_pool.size() < 65536)
      for (int i = 0; i < 512; ++i) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) break; E->ev_pool.push_back(ev); }
    if (E->ev_used + 2 <= E->ev_pool.size()) {
      ea = E->ev_pool[E->ev_used++]; eb = E->ev_pool[E->ev_used++];
      E->prof.push_back(ProfRec{PC_ATTN_FWD_IMG, ea, eb, flops, bytes, flops});
    }
  }
  return launch_attn_fwd(E->dt, a, s, ea, eb);
}

// ------------------------------------------------------------------------------------------------ kernel wrappers
// LayerNorm folding of one GEMM call (GemmArgs::ln_* / fold_*)
struct Fold {
  const float* ln_gamma = nullptr; void* ln_x16 = nullptr; int ln_split = 0; float* ln_part = nullptr; int ln_ntp = 0;   // producer
  const float* fold_part = nullptr; const float* fold_colsum = nullptr; int fold_ntp = 0, fold_nt = 0;                   // consumer
  const void* rp_hi_in = nullptr; const uint8_t* rp_lo_in = nullptr; uint8_t* rp_lo_out = nullptr;                       // packed stream (EPI_RESIDP_LN)
};
hipError_t gemm(Engine* E, int epi, const void* A, WRef Bt, int M, int N, int K, const float* bias, const void* aux,
                const float* resid, void* out, void* out2, hipStream_t s, int dtype = -1, int a_split = 0, const Fold* f = nullptr,
                int lda = 0, int ldo = 0) {
  GemmArgs g{A, Bt.p, M, N, K, bias, aux, resid, out, out2};
  g.a_split = a_split; g.ldb = Bt.ld; g.w8_exp = Bt.e8; g.lda = lda; g.ldo = ldo;
  if (f) {
    g.ln_gamma = f->ln_gamma; g.ln_x16 = f->ln_x16; g.ln_split = f->ln_split; g.ln_part = f->ln_part; g.ln_ntp = f->ln_ntp;
    g.fold_part = f->fold_part; g.fold_colsum = f->fold_colsum; g.fold_ntp = f->fold_ntp; g.fold_nt = f->fold_nt;
    g.rp_hi_in = f->rp_hi_in; g.rp_lo_in = f->rp_lo_in; g.rp_lo_out = f->rp_lo_out;
  }
  g.out_lo8 = (a_split == 2 && (epi == EPI_GELU_SPLIT || epi == EPI_GELUBWD_SPLIT)) ? 1 : 0;
  const int dt = dtype >= 0 ? dtype : E->dt;
  const double ob = (epi == EPI_RESIDP_LN) ? 6.0 : (epi == EPI_RESID32) ? 8.0 : (epi == EPI_RESID32_LN) ? 8.0 + 2.0 * (a_split ? 1.5 : 1.0) : ((epi == EPI_STORE32 || epi == EPI_STORE_SPLIT) ? 4.0 : ((epi == EPI_GELUBWD || epi == EPI_GELU_SPLIT) ? 4.0 :
                    (epi == EPI_GELUBWD_SPLIT ? 6.0 : 2.0)));
  // the dominant kernel is timed by its own dispatch (start/stop timestamps of the AQL packet): no marker packets
  hipEvent_t ea = nullptr, eb = nullptr;
  if (E && E->prof_on) {
    if (E->ev_used + 2 > E->ev_pool.size() && E->ev_pool.size() < 65536)
      for (int i = 0; i < 512; ++i) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) break; E->ev_pool.push_back(ev); }
    if (E->ev_used + 2 <= E->ev_pool.size()) {
      ea = E->ev_pool[E->ev_used++]; eb = E->ev_pool[E->ev_used++];
      // flops = ALGORITHMIC (what the reference's fp32 GEMM does: 2MNK); the split-precision modes execute twice that
      // (16-bit pair) or 1.5x in 16-bit-MFMA time (mixed pair: the residual term runs on the fp8 MFMA at twice the rate)
      const double xa = a_split == 2 ? 1.5 : (a_split ? 2.0 : 1.0);
      E->prof.push_back(ProfRec{PC_GEMM, ea, eb, 2.0 * M * N * K,
                                2.0 * ((double)M * K * xa + (double)N * K * (a_split == 2 ? 1.5 : 1.0)) + ob * M * N + (out2 ? 2.0 * M * N : 0),
                                2.0 * M * N * K * xa, M, N, K, epi, a_split, (f && f->fold_part) ? 1 : 0});
    }
  }
  return launch_gemm(dt, epi, g, s, ea, eb);
}
hipError_t ln_fwd(Engine* E, int out_dt, const float* x, const int32_t* idx, int row_mul, const LNp& p, void* y, int rows, int d,
                  hipStream_t s, int split = 0) {
  LnFwdArgs a{x, idx, row_mul, p.g, p.b, y, rows, d};
  a.split = (split && out_dt != DT_F32) ? split : 0;
  ProfScope ps(E, s, PC_LN_FWD, 8.0 * rows * d, (double)rows * d * (4.0 + (out_dt == DT_F32 ? 4.0 : 2.0)));
  return launch_ln_fwd(out_dt, a, s);
}
hipError_t ln_bwd(Engine* E, const void* dy, int dy_dtype, const float* x, const int32_t* idx, int row_mul, const LNp& p,
                  const float* resid, float* out32, void* out16, int rows, int d, hipStream_t s, int dtype = -1, int split = 0) {
  LnBwdArgs a{dy, dy_dtype, x, idx, row_mul, p.g, resid, out32, out16, rows, d};
  a.split = split;
  ProfScope ps(E, s, PC_LN_BWD, 14.0 * rows * d, (double)rows * d * ((dy_dtype == DT_F32 ? 4.0 : 2.0) + 4.0 + (resid ? 4.0 : 0.0) + 4.0 + (out16 ? 2.0 : 0.0)));
  return launch_ln_bwd(dtype >= 0 ? dtype : E->dt, a, s);
}

// ------------------------------------------------------------------------------------------------ tower workspace
// `exact` (split-precision mode): every 16-bit operand buffer holds a hi|lo pair (twice the columns) and QKV / dO are fp32
size_t tower_bytes(const TowerW& W, int N, int L, bool save, bool exact) {
  const size_t T = (size_t)N * L, d = W.width, H = W.heads, nl = W.layers, X = exact ? 2 : 1;
  size_t b = 0;
  b += (save ? (2 * nl + 1) : 1) * align256(T * d * 4);             // x
  b += align256(T * d * 2 * X);                                      // h16
  b += (save ? nl : 1) * align256(T * 3 * d * 2 * X);                // qkv
  b += (save ? nl : 1) * align256(T * d * 2 * X);                    // attn
  b += (save ? nl : 1) * align256((size_t)N * H * L * 4);            // lse
  b += (save ? nl : 1) * align256(T * 4 * d * 2);                    // u
  b += align256(T * (4 * d * 2 * X + 128));                          // a16 (pair rows padded: wide_pitch)
  b += 2 * align256(T * FOLD_NTP * 8);                               // LayerNorm-folding partials
  if (save) {
    b += 2 * align256(T * d * 4) + 2 * align256(T * d * 2 * X) + (align256(T * 4 * d * 2) + align256(T * 3 * d * 2)) * X + align256(T * 128);
    b += align256((size_t)N * H * L * 4) + 256;
  }
  return b + 4096;
}
// row pitch (16-bit elements) of the pair tensors between the two MLP GEMMs ([T, 2 * 4d]: a16, du16; GemmArgs::lda / ldo): rows
// whose dense pitch is a multiple of 4 KiB get 128 bytes more (every CLIP width: 16 d bytes)
int wide_pitch(int cols) {
  static const int on = getenv("MVLPT_WIDE_PITCH") ? atoi(getenv("MVLPT_WIDE_PITCH")) : 1;
  return on && (4 * cols) % 4096 == 0 ? 2 * cols + 64 : 0;
}
void carve_tower(Bump& bp, const TowerW& W, TowerState& st, int N, int L, bool save, bool causal, bool exact, int xs) {
  const size_t T = (size_t)N * L, d = W.width, H = W.heads, X = exact ? 2 : 1; const int nl = W.layers;
  st = TowerState();
  st.N = N; st.L = L; st.d = (int)d; st.H = (int)H; st.layers = nl; st.saved = save; st.causal = causal; st.exact = exact;
  st.xs = exact ? xs : 0;
  st.x.resize(2 * nl + 1); st.qkv.resize(nl); st.attn.resize(nl); st.u.resize(nl); st.lse.resize(nl); st.skip.assign(nl, 0);
  float* x0 = bp.take<float>(T * d);
  for (int i = 0; i < 2 * nl + 1; ++i) st.x[i] = (save && i > 0) ? bp.take<float>(T * d) : x0;
  st.h16 = bp.take_bytes(T * d * 2 * X);
  for (int l = 0; l < nl; ++l) {
    const bool fresh = save || l == 0;
    st.qkv[l] = fresh ? bp.take_bytes(T * 3 * d * 2 * X) : st.qkv[0];
    st.attn[l] = fresh ? bp.take_bytes(T * d * 2 * X) : st.attn[0];
    st.lse[l] = fresh ? bp.take<float>((size_t)N * H * L) : st.lse[0];
    st.u[l] = fresh ? bp.take_bytes(T * 4 * d * 2) : st.u[0];
  }
  st.a16 = bp.take_bytes(T * (4 * d * 2 * X + 128));
  st.wide_pitch = exact ? wide_pitch(4 * (int)d) : 0;
  st.part[0] = bp.take<float>(T * FOLD_NTP * 2); st.part[1] = bp.take<float>(T * FOLD_NTP * 2);
  if (save) {
    st.dx32 = bp.take<float>(T * d);
    st.dx16 = bp.take_bytes(T * d * 2 * X); st.dh32 = bp.take<float>(T * d); st.dO16 = bp.take_bytes(T * d * 2 * X);
    st.du16 = bp.take_bytes(T * (4 * d * 2 * X + 128)); st.dqkv16 = bp.take_bytes(T * 3 * d * 2 * X);
    st.delta = bp.take<float>((size_t)N * H * L);
    st.scale_dev = bp.take<float>(4);   // {scale, 1/scale, amax scratch, pad}
  }
  st.valid = true;
}

// which pair format a split tower uses: the default mode carries the residual as one e5m2 byte (GemmArgs::a_split == 2);
// MVLPT_PREC_SPLIT_ALL (PREC = "fp32": as exact as this engine gets) keeps 16-bit pairs everywhere
int split_kind(const Engine* E) { return (E->prec_mode == MVLPT_PREC_SPLIT_ALL || !E->lo8) ? 1 : 2; }

// attention core of the split-precision mode: qkv pair [T,6d] -> O as a hi|lo pair [T,2d]
int attn32_fwd(Engine* E, TowerState& st, int l, int q_rows, hipStream_t s) {
  Attn32Args a{st.qkv[l], st.attn[l], st.saved ? st.lse[l] : nullptr, st.N, st.L, st.H, st.causal ? 1 : 0, q_rows};
  a.out_lo8 = st.xs == 2 ? 1 : 0;
  const double rows = q_rows > 0 ? (double)q_rows : (double)st.L;
  ProfScope ps(E, s, PC_ATTN_FWD, 4.0 * rows * st.L * 64.0 * st.N * st.H * (st.causal ? 0.5 : 1.0), (double)st.N * st.L * st.d * 16.0);
  HIPCHK(E, launch_attn32_fwd(E->dt, a, s));
  return 0;
}
int attn32_bwd(Engine* E, TowerState& st, int l, hipStream_t s) {
  Attn32BwdArgs a{st.qkv[l], st.attn[l], st.dO16, st.lse[l], st.delta, st.dqkv16, st.N, st.L, st.H, st.causal ? 1 : 0};
  a.lo8 = st.xs == 2 ? 1 : 0;
  ProfScope ps(E, s, PC_ATTN_BWD, 14.0 * st.L * st.L * 64.0 * st.N * st.H * (st.causal ? 0.5 : 1.0), (double)st.N * st.L * st.d * 32.0);
  HIPCHK(E, launch_attn32_bwd(E->dt, a, s));
  return 0;
}

// ---- LayerNorm folding (kernels.h GemmArgs::ln_* / fold_*): the residual GEMM in front of a LayerNorm writes
// round16(x * gamma) into h16 and per-row partial sums; the GEMM behind it applies mean / rstd in its epilogue.
// which = 0: the LayerNorm is ln_1 (producer: FC2 of the previous block), 1: ln_2 (producer: this block's out-projection).
bool fold_producer(Engine* E, TowerState& st, int which, int M, int N, int K, const LNp& ln, hipStream_t s, Fold* f) {
  if (!st.fold) return false;
  GemmArgs q{};
  q.M = M; q.N = N; q.K = K; q.a_split = st.xs;
  const int bn = gemm_tile_n(E->dt, EPI_RESID32_LN, q, s);      // the launcher's N-tile for this problem on this stream
  if (bn <= 0 || N % bn) return false;
  const int nt = N / 128, ntp = (nt + 1) & ~1;          // one slot per 128 output columns, whatever the tile geometry
  if (ntp > FOLD_NTP) return false;
  *f = Fold();
  f->ln_gamma = ln.g; f->ln_x16 = st.h16; f->ln_split = st.xs; f->ln_part = st.part[which]; f->ln_ntp = ntp;
  st.nt[which] = nt; st.ntp[which] = ntp;
  return true;
}
Fold fold_consumer(const TowerState& st, int which, const Linear& L) {
  Fold f;
  f.fold_part = st.part[which]; f.fold_colsum = L.fold_s; f.fold_ntp = st.ntp[which]; f.fold_nt = st.nt[which];
  return f;
}

// Split-precision variant of block_fwd / block_bwd below: every GEMM A operand is a 16-bit hi|lo pair (~22 bits), the
// attention core runs in fp32.  Used for towers that carry a gradient (and for every tower under MVLPT_PREC_SPLIT_ALL).
// ln1_folded: h16 / part[0] already hold this block's ln_1 input in folded form (written by the previous block's FC2);
// next_ln1: the LayerNorm the block's output goes into next, when that one may be folded; *produced: it was.
int block_fwd_x(Engine* E, const TowerW& W, TowerState& st, int l, hipStream_t s, bool ln1_folded, const LNp* next_ln1, bool* produced) {
  const Block& B = W.blocks[l];
  const int T = st.N * st.L, d = st.d;
  float* xin = st.x[2 * l]; float* xmid = st.x[2 * l + 1]; float* xout = st.x[2 * l + 2];
  Fold fq, fo, ff, fp;
  if (ln1_folded) fq = fold_consumer(st, 0, B.qkv);
  else HIPCHK(E, ln_fwd(E, E->dt, xin, nullptr, 1, B.ln1, st.h16, T, d, s, st.xs));
  HIPCHK(E, gemm(E, EPI_STORE_SPLIT, st.h16, B.qkv.fw(), T, 3 * d, d, ln1_folded ? B.qkv.fold_b : B.qkv.b, nullptr, nullptr, st.qkv[l], nullptr, s, -1, st.xs,
                 ln1_folded ? &fq : nullptr));
  if (int rc = attn32_fwd(E, st, l, 0, s)) return rc;
  const bool p2 = fold_producer(E, st, 1, T, d, d, B.ln2, s, &fo);
  HIPCHK(E, gemm(E, p2 ? EPI_RESID32_LN : EPI_RESID32, st.attn[l], B.o.fw(), T, d, d, B.o.b, nullptr, xin, xmid, nullptr, s, -1, st.xs, p2 ? &fo : nullptr));
  if (p2) ff = fold_consumer(st, 1, B.fc);
  else HIPCHK(E, ln_fwd(E, E->dt, xmid, nullptr, 1, B.ln2, st.h16, T, d, s, st.xs));
  HIPCHK(E, gemm(E, EPI_GELU_SPLIT, st.h16, B.fc.fw(), T, 4 * d, d, p2 ? B.fc.fold_b : B.fc.b, nullptr, nullptr, st.a16, st.saved ? st.u[l] : nullptr, s, -1, st.xs,
                 p2 ? &ff : nullptr, 0, st.wide_pitch));
  const bool p1 = next_ln1 && fold_producer(E, st, 0, T, d, 4 * d, *next_ln1, s, &fp);
  HIPCHK(E, gemm(E, p1 ? EPI_RESID32_LN : EPI_RESID32, st.a16, B.pr.fw(), T, d, 4 * d, B.pr.b, nullptr, xmid, xout, nullptr, s, -1, st.xs, p1 ? &fp : nullptr,
                 st.wide_pitch, 0));
  if (produced) *produced = p1;
  return 0;
}
int block_bwd_x(Engine* E, const TowerW& W, TowerState& st, int l, hipStream_t s) {
  const Block& B = W.blocks[l];
  const int T = st.N * st.L, d = st.d;
  HIPCHK(E, gemm(E, EPI_GELUBWD_SPLIT, st.dx16, B.pr.bw(), T, 4 * d, d, nullptr, st.u[l], nullptr, st.du16, nullptr, s, -1, st.xs, nullptr, 0, st.wide_pitch));
  HIPCHK(E, gemm(E, EPI_STORE32, st.du16, B.fc.bw(), T, d, 4 * d, nullptr, nullptr, nullptr, st.dh32, nullptr, s, -1, st.xs, nullptr, st.wide_pitch, 0));
  HIPCHK(E, ln_bwd(E, st.dh32, DT_F32, st.x[2 * l + 1], nullptr, 1, B.ln2, st.dx32, st.dx32, st.dx16, T, d, s, -1, st.xs));
  HIPCHK(E, gemm(E, EPI_STORE_SPLIT, st.dx16, B.o.bw(), T, d, d, nullptr, nullptr, nullptr, st.dO16, nullptr, s, -1, st.xs));
  if (int rc = attn32_bwd(E, st, l, s)) return rc;
  HIPCHK(E, gemm(E, EPI_STORE32, st.dqkv16, B.qkv.bw(), T, d, 3 * d, nullptr, nullptr, nullptr, st.dh32, nullptr, s, -1, st.xs));
  HIPCHK(E, ln_bwd(E, st.dh32, DT_F32, st.x[2 * l], nullptr, 1, B.ln1, st.dx32, st.dx32, st.dx16, T, d, s, -1, st.xs));
  return 0;
}

// ResidualAttentionBlock.forward (clip/model.py:185-188) on token buffers (LayerNorm folding: see block_fwd_x)
int block_fwd(Engine* E, const TowerW& W, TowerState& st, int l, hipStream_t s, bool ln1_folded = false, const LNp* next_ln1 = nullptr,
              bool* produced = nullptr) {
  if (produced) *produced = false;
  if (st.exact) return block_fwd_x(E, W, st, l, s, ln1_folded, next_ln1, produced);
  const Block& B = W.blocks[l];
  const int T = st.N * st.L, d = st.d;
  float* xin = st.x[2 * l]; float* xmid = st.x[2 * l + 1]; float* xout = st.x[2 * l + 2];
  Fold fq, fo, ff, fp;
  if (ln1_folded) fq = fold_consumer(st, 0, B.qkv);
  else HIPCHK(E, ln_fwd(E, E->dt, xin, nullptr, 1, B.ln1, st.h16, T, d, s));
  HIPCHK(E, gemm(E, EPI_STORE16, st.h16, B.qkv.fw(), T, 3 * d, d, ln1_folded ? B.qkv.fold_b : B.qkv.b, nullptr, nullptr, st.qkv[l], nullptr, s, -1, 0,
                 ln1_folded ? &fq : nullptr));
  {
    AttnArgs a{st.qkv[l], st.attn[l], st.saved ? st.lse[l] : nullptr, st.N, st.L, st.H, st.causal ? 1 : 0};
    const double fl = 4.0 * st.L * st.L * 64.0 * st.N * st.H * (st.causal ? 0.5 : 1.0);
    ProfScope ps(E, s, PC_ATTN_FWD, fl, (double)T * d * 2.0 * 4.0);
    HIPCHK(E, attn_fwd_timed(E, a, fl, (double)T * d * 2.0 * 4.0, s));
  }
  const bool p2 = fold_producer(E, st, 1, T, d, d, B.ln2, s, &fo);
  HIPCHK(E, gemm(E, p2 ? EPI_RESID32_LN : EPI_RESID32, st.attn[l], B.o.fw(), T, d, d, B.o.b, nullptr, xin, xmid, nullptr, s, -1, 0, p2 ? &fo : nullptr));
  if (p2) ff = fold_consumer(st, 1, B.fc);
  else HIPCHK(E, ln_fwd(E, E->dt, xmid, nullptr, 1, B.ln2, st.h16, T, d, s));
  HIPCHK(E, gemm(E, EPI_GELU, st.h16, B.fc.fw(), T, 4 * d, d, p2 ? B.fc.fold_b : B.fc.b, nullptr, nullptr, st.a16, st.saved ? st.u[l] : nullptr, s, -1, 0,
                 p2 ? &ff : nullptr));
  const bool p1 = next_ln1 && fold_producer(E, st, 0, T, d, 4 * d, *next_ln1, s, &fp);
  HIPCHK(E, gemm(E, p1 ? EPI_RESID32_LN : EPI_RESID32, st.a16, B.pr.fw(), T, d, 4 * d, B.pr.b, nullptr, xmid, xout, nullptr, s, -1, 0, p1 ? &fp : nullptr));
  if (produced) *produced = p1;
  return 0;
}

// block_fwd on the PACKED residual stream (common.h respk_*, kernels.h EPI_RESIDP_LN): fp16 tower, single operands, nothing saved,
// both LayerNorms folded.  The stream lives in st.x[0] as hi [T,d] fp16 | lo [T,d] bytes and is updated in place; the hi plane is
// the A operand of the QKV / MLP-up GEMMs, whose weights carry the LayerNorm's gamma (Linear::wg).  part[0] holds the row
// statistics of the stream on entry (the previous block's FC2, or the pack kernel in front of block 0).
int block_fwd_packed(Engine* E, const TowerW& W, TowerState& st, int l, hipStream_t s) {
  const Block& B = W.blocks[l];
  const int T = st.N * st.L, d = st.d;
  void* hi = st.x[0];
  uint8_t* lo = (uint8_t*)st.x[0] + (size_t)T * d * 2;
  Fold fq = fold_consumer(st, 0, B.qkv), fo, ff, fp;
  fq.fold_colsum = B.qkv.fold_sg;
  HIPCHK(E, gemm(E, EPI_STORE16, hi, B.qkv.fwg(), T, 3 * d, d, B.qkv.fold_b, nullptr, nullptr, st.qkv[l], nullptr, s, -1, 0, &fq));
  dbg_ck(E, st.qkv[l], (size_t)T * 3 * d * 2, s);
  {
    AttnArgs a{st.qkv[l], st.attn[l], nullptr, st.N, st.L, st.H, st.causal ? 1 : 0};
    const double fl = 4.0 * st.L * st.L * 64.0 * st.N * st.H * (st.causal ? 0.5 : 1.0);
    ProfScope ps(E, s, PC_ATTN_FWD, fl, (double)T * d * 2.0 * 4.0);
    HIPCHK(E, attn_fwd_timed(E, a, fl, (double)T * d * 2.0 * 4.0, s));
  }
  dbg_ck(E, st.attn[l], (size_t)T * d * 2, s);
  if (!fold_producer(E, st, 1, T, d, d, B.ln2, s, &fo)) return fail(E, MVLPT_ERR_STATE, "packed residual stream: out-projection cannot produce the ln_2 statistics");
  fo.rp_hi_in = hi; fo.rp_lo_in = lo; fo.rp_lo_out = lo;
  HIPCHK(E, gemm(E, EPI_RESIDP_LN, st.attn[l], B.o.fw(), T, d, d, B.o.b, nullptr, nullptr, hi, nullptr, s, -1, 0, &fo));
  dbg_ck(E, hi, (size_t)T * d * 3, s); dbg_ck(E, fo.ln_part, (size_t)T * fo.ln_ntp * 8, s);
  ff = fold_consumer(st, 1, B.fc);
  ff.fold_colsum = B.fc.fold_sg;
  HIPCHK(E, gemm(E, EPI_GELU, hi, B.fc.fwg(), T, 4 * d, d, B.fc.fold_b, nullptr, nullptr, st.a16, nullptr, s, -1, 0, &ff));
  dbg_ck(E, st.a16, (size_t)T * 4 * d * 2, s);
  if (!fold_producer(E, st, 0, T, d, 4 * d, B.ln1, s, &fp)) return fail(E, MVLPT_ERR_STATE, "packed residual stream: MLP down-projection cannot produce the ln_1 statistics");
  fp.rp_hi_in = hi; fp.rp_lo_in = lo; fp.rp_lo_out = lo;
  HIPCHK(E, gemm(E, EPI_RESIDP_LN, st.a16, B.pr.fw(), T, d, 4 * d, B.pr.b, nullptr, nullptr, hi, nullptr, s, -1, 0, &fp));
  dbg_ck(E, hi, (size_t)T * d * 3, s); dbg_ck(E, fp.ln_part, (size_t)T * fp.ln_ntp * 8, s);
  return 0;
}

// dX-only backward of one block: dx32/dx16 hold d(block output) on entry and d(block input) on exit
int block_bwd(Engine* E, const TowerW& W, TowerState& st, int l, hipStream_t s) {
  if (st.exact) return block_bwd_x(E, W, st, l, s);
  const Block& B = W.blocks[l];
  const int T = st.N * st.L, d = st.d;
  HIPCHK(E, gemm(E, EPI_GELUBWD, st.dx16, B.pr.bw(), T, 4 * d, d, nullptr, st.u[l], nullptr, st.du16, nullptr, s));
  // the dX GEMMs that feed a LayerNorm backward store fp32 (one 16-bit rounding less per half layer)
  HIPCHK(E, gemm(E, EPI_STORE32, st.du16, B.fc.bw(), T, d, 4 * d, nullptr, nullptr, nullptr, st.dh32, nullptr, s));
  HIPCHK(E, ln_bwd(E, st.dh32, DT_F32, st.x[2 * l + 1], nullptr, 1, B.ln2, st.dx32, st.dx32, st.dx16, T, d, s));
  HIPCHK(E, gemm(E, EPI_STORE16, st.dx16, B.o.bw(), T, d, d, nullptr, nullptr, nullptr, st.dO16, nullptr, s));
  {
    AttnBwdArgs a{st.qkv[l], st.attn[l], st.dO16, st.lse[l], st.delta, st.dqkv16, st.N, st.L, st.H, st.causal ? 1 : 0};
    const double fl = 14.0 * st.L * st.L * 64.0 * st.N * st.H * (st.causal ? 0.5 : 1.0);
    ProfScope ps(E, s, PC_ATTN_BWD, fl, (double)T * d * 2.0 * 8.0);
    HIPCHK(E, launch_attn_bwd(E->dt, a, s));
  }
  HIPCHK(E, gemm(E, EPI_STORE32, st.dqkv16, B.qkv.bw(), T, d, 3 * d, nullptr, nullptr, nullptr, st.dh32, nullptr, s));
  HIPCHK(E, ln_bwd(E, st.dh32, DT_F32, st.x[2 * l], nullptr, 1, B.ln1, st.dx32, st.dx32, st.dx16, T, d, s));
  return 0;
}

// ------------------------------------------------------------------------------------------------ weight loading
bool parse_block_name(const char* rest, int* layer, const char** tail) {
  // rest = "<l>.<tail>"
  char* endp = nullptr;
  long l = strtol(rest, &endp, 10);
  if (endp == rest || *endp != '.') return false;
  *layer = (int)l; *tail = endp + 1; return true;
}

int upload_f32(Engine* E, const float* src32, size_t n, float** dst, hipStream_t s) {
  void* p = nullptr;
  HIPCHK(E, hipMalloc(&p, n * 4));
  E->owned.push_back(p);
  HIPCHK(E, hipMemcpyAsync(p, src32, n * 4, hipMemcpyDeviceToDevice, s));
  *dst = (float*)p; return 0;
}
int pack_linear(Engine* E, const float* w32, int out, int in, Linear* L, hipStream_t s) {
  // rows of [16-bit plane | fp8 plane]: pitch 3K/2 elements (K = the contraction length of that orientation)
  const int ldw = in + in / 2, ldwt = out + out / 2;
  void *w = nullptr, *wt = nullptr;
  HIPCHK(E, hipMalloc(&w, (size_t)out * ldw * 2)); E->owned.push_back(w);
  HIPCHK(E, hipMalloc(&wt, (size_t)in * ldwt * 2)); E->owned.push_back(wt);
  HIPCHK(E, launch_pack_weight(E->dt, w32, w, out, in, ldw, s));
  HIPCHK(E, launch_pack_weight_t(E->dt, w32, wt, out, in, s, ldwt));
  // fp8 planes: W8 = e4m3(W * 2^e8), amax(|W|) * 2^e8 in [128, 256) (e4m3 tops out at 448); the exponent comes back to the
  // host once per tensor (load time), the GEMM gets it by value
  HIPCHK(E, E->ce_ws.reserve(256));
  float* sc = (float*)E->ce_ws.p;
  HIPCHK(E, launch_grad_scale(w32, (size_t)out * in, 128.0f, sc, s));
  HIPCHK(E, launch_pack_weight8(w32, (uint8_t*)w + (size_t)in * 2, out, in, 0, in, (size_t)ldw * 2, sc, s));
  HIPCHK(E, launch_pack_weight8(w32, (uint8_t*)wt + (size_t)out * 2, out, in, 1, out, (size_t)ldwt * 2, sc, s));
  float sc_host = 1.0f;
  HIPCHK(E, hipMemcpyAsync(&sc_host, sc, sizeof(float), hipMemcpyDeviceToHost, s));
  HIPCHK(E, hipStreamSynchronize(s));
  int e8 = 0;
  (void)frexpf(sc_host, &e8);      // sc = 0.5 * 2^e8' -> exponent e8' - 1
  L->w = w; L->wt = wt; L->out = out; L->in = in; L->ldw = ldw; L->ldwt = ldwt; L->e8 = e8 - 1; return 0;
}

int load_block_tensor(Engine* E, TowerW& W, int layer, const char* tail, const float* p, const int64_t* shape, int ndim,
                      hipStream_t s) {
  if (layer < 0 || layer >= W.layers) return fail(E, MVLPT_ERR_ARG, "layer index out of range");
  Block& B = W.blocks[layer];
  const int d = W.width;
  auto is2 = [&](int r, int c) { return ndim == 2 && shape[0] == r && shape[1] == c; };
  auto is1 = [&](int r) { return ndim == 1 && shape[0] == r; };
  std::string t(tail);
  if (t == "attn.in_proj_weight") { if (!is2(3 * d, d)) goto bad; return pack_linear(E, p, 3 * d, d, &B.qkv, s); }
  if (t == "attn.in_proj_bias") { if (!is1(3 * d)) goto bad; return upload_f32(E, p, 3 * d, &B.qkv.b, s); }
  if (t == "attn.out_proj.weight") { if (!is2(d, d)) goto bad; return pack_linear(E, p, d, d, &B.o, s); }
  if (t == "attn.out_proj.bias") { if (!is1(d)) goto bad; return upload_f32(E, p, d, &B.o.b, s); }
  if (t == "mlp.c_fc.weight") { if (!is2(4 * d, d)) goto bad; return pack_linear(E, p, 4 * d, d, &B.fc, s); }
  if (t == "mlp.c_fc.bias") { if (!is1(4 * d)) goto bad; return upload_f32(E, p, 4 * d, &B.fc.b, s); }
  if (t == "mlp.c_proj.weight") { if (!is2(d, 4 * d)) goto bad; return pack_linear(E, p, d, 4 * d, &B.pr, s); }
  if (t == "mlp.c_proj.bias") { if (!is1(d)) goto bad; return upload_f32(E, p, d, &B.pr.b, s); }
  if (t == "ln_1.weight") { if (!is1(d)) goto bad; return upload_f32(E, p, d, &B.ln1.g, s); }
  if (t == "ln_1.bias") { if (!is1(d)) goto bad; return upload_f32(E, p, d, &B.ln1.b, s); }
  if (t == "ln_2.weight") { if (!is1(d)) goto bad; return upload_f32(E, p, d, &B.ln2.g, s); }
  if (t == "ln_2.bias") { if (!is1(d)) goto bad; return upload_f32(E, p, d, &B.ln2.b, s); }
  return fail(E, MVLPT_ERR_ARG, std::string("unknown block tensor: ") + tail);
bad:
  return fail(E, MVLPT_ERR_ARG, std::string("shape mismatch for block tensor ") + tail);
}

const char* first_missing(Engine* E) {
  static thread_local std::string m;
  auto chk = [&](const void* p, const std::string& n) { if (!p && m.empty()) m = n; };
  m.clear();
  chk(E->conv_w, "visual.conv1.weight"); chk(E->cls_emb, "visual.class_embedding"); chk(E->vpos, "visual.positional_embedding");
  chk(E->ln_pre.g, "visual.ln_pre.weight"); chk(E->ln_pre.b, "visual.ln_pre.bias");
  chk(E->ln_post.g, "visual.ln_post.weight"); chk(E->ln_post.b, "visual.ln_post.bias"); chk(E->vproj, "visual.proj");
  chk(E->tpos, "positional_embedding"); chk(E->ln_final.g, "ln_final.weight"); chk(E->ln_final.b, "ln_final.bias");
  chk(E->tproj, "text_projection");
  for (int t = 0; t < 2; ++t) {
    TowerW& W = t ? E->txt : E->vis;
    const std::string pre = t ? "transformer.resblocks." : "visual.transformer.resblocks.";
    for (int l = 0; l < W.layers; ++l) {
      const Block& B = W.blocks[l]; const std::string q = pre + std::to_string(l) + ".";
      chk(B.qkv.w, q + "attn.in_proj_weight"); chk(B.qkv.b, q + "attn.in_proj_bias");
      chk(B.o.w, q + "attn.out_proj.weight"); chk(B.o.b, q + "attn.out_proj.bias");
      chk(B.fc.w, q + "mlp.c_fc.weight"); chk(B.fc.b, q + "mlp.c_fc.bias");
      chk(B.pr.w, q + "mlp.c_proj.weight"); chk(B.pr.b, q + "mlp.c_proj.bias");
      chk(B.ln1.g, q + "ln_1.weight"); chk(B.ln1.b, q + "ln_1.bias"); chk(B.ln2.g, q + "ln_2.weight"); chk(B.ln2.b, q + "ln_2.bias");
    }
  }
  return m.empty() ? nullptr : m.c_str();
}

// LayerNorm folding: the per-column vectors of every linear layer that follows a LayerNorm (qkv <- ln_1, fc <- ln_2), once,
// when all frozen tensors are there (the first tower forward)
int prepare_fold(Engine* E, hipStream_t s) {
  if (E->fold_ready) return 0;
  // A REBUILD (frozen tensors reloaded, packed stream toggled) rewrites vectors the other tower — on another stream — may still be
  // reading: drain the device first.  Rare and outside any step; the first build has no readers.
  if (!E->vis.blocks.empty() && E->vis.blocks[0].qkv.fold_s) HIPCHK(E, hipDeviceSynchronize());
  for (TowerW* W : {&E->vis, &E->txt}) {
    const int d = W->width;
    for (Block& B : W->blocks) {
      for (int k = 0; k < 2; ++k) {
        Linear& L = k ? B.fc : B.qkv;
        const LNp& ln = k ? B.ln2 : B.ln1;
        if (!L.fold_s) {      // (kept across reloads of the frozen tensors: the shapes are fixed by the architecture)
          void* p = nullptr;
          HIPCHK(E, hipMalloc(&p, (size_t)L.out * 2 * sizeof(float)));
          E->owned.push_back(p);
          L.fold_s = (float*)p; L.fold_b = (float*)p + L.out;
        }
        HIPCHK(E, launch_fold_vectors(E->dt, L.w, L.ldw, ln.g, ln.b, L.b, L.fold_s, L.fold_b, L.out, d, s));
        if (W == &E->vis && E->dt == DT_F16 && E->resid_packed) {      // packed residual stream: gamma inside the weight
          if (!L.wg) {
            void* p = nullptr;
            HIPCHK(E, hipMalloc(&p, (size_t)L.out * L.in * 2 + (size_t)L.out * sizeof(float)));
            E->owned.push_back(p);
            L.wg = p; L.fold_sg = (float*)((char*)p + (size_t)L.out * L.in * 2);
          }
          HIPCHK(E, launch_fold_weight(L.w, L.ldw, ln.g, L.wg, L.in, L.fold_sg, L.out, d, s));
        }
      }
    }
  }
  // Both towers' vectors are computed on the stream of whichever tower gets here first; the other tower runs on another
  // stream and reads them without any dependency on `s`.  One host-side wait, once per (re)load of the frozen tensors.
  HIPCHK(E, hipStreamSynchronize(s));
  E->fold_ready = true;
  return 0;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

#ifndef MVLPT_SRC_HASH
#define MVLPT_SRC_HASH "unknown"
#endif
#ifndef MVLPT_GIT_HASH
#define MVLPT_GIT_HASH "nogit"
#endif
// "mvlpt_hip 0.1 (gfx950) src:<sha256 of csrc/*.hip csrc/*.h include/mvlpt_hip.h, 12 hex> git:<commit the tree was built in>"
const char* mvlpt_version(void) { return "mvlpt_hip 0.1 (gfx950) src:" MVLPT_SRC_HASH " git:" MVLPT_GIT_HASH; }

const char* mvlpt_last_error(void* h) { return h ? ((Engine*)h)->err.c_str() : g_create_err.c_str(); }

int mvlpt_create(const MvlptArch* a, void** handle) {
  if (!a || !handle) return fail(nullptr, MVLPT_ERR_ARG, "null argument");
  if (a->compute_dtype != MVLPT_DT_F16 && a->compute_dtype != MVLPT_DT_BF16)
    return fail(nullptr, MVLPT_ERR_ARG, "compute_dtype must be MVLPT_DT_F16 or MVLPT_DT_BF16");
  if (a->vision_width != a->vision_heads * 64 || a->text_width != a->text_heads * 64)
    return fail(nullptr, MVLPT_ERR_UNSUPPORTED, "head_dim must be 64 (CLIP ViT: heads = width // 64)");
  if (a->vision_width % 128 || a->text_width % 128 || a->embed_dim % 64)
    return fail(nullptr, MVLPT_ERR_UNSUPPORTED, "widths must be multiples of 128 and embed_dim of 64");
  if (a->patch_size <= 0 || a->image_resolution % a->patch_size)
    return fail(nullptr, MVLPT_ERR_ARG, "image_resolution must be a multiple of patch_size");
  if (a->vision_layers <= 0 || a->text_layers <= 0 || a->context_length <= 0)
    return fail(nullptr, MVLPT_ERR_ARG, "bad layer count / context length");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, MVLPT_ERR_HIP, "no HIP device visible: libmvlpt_hip needs an AMD GPU (there is no CPU fallback)");
  Engine* E = new Engine();
  E->arch = *a; E->dt = a->compute_dtype;
  if (const char* v = getenv("MVLPT_SPLIT_LO8")) E->lo8 = atoi(v) != 0;
  if (const char* v = getenv("MVLPT_LN_FOLD")) E->fold_mode = atoi(v);
  if (const char* v = getenv("MVLPT_RESID_PACKED")) E->resid_packed = atoi(v) ? 1 : 0;
  if (const char* v = getenv("MVLPT_LN_FOLD_MIN_ROWS")) E->fold_min_rows = atoi(v) > 0 ? atoi(v) : 1;
  E->vis.width = a->vision_width; E->vis.layers = a->vision_layers; E->vis.heads = a->vision_heads;
  E->vis.blocks.resize(a->vision_layers);
  E->txt.width = a->text_width; E->txt.layers = a->text_layers; E->txt.heads = a->text_heads;
  E->txt.blocks.resize(a->text_layers);
  const int K = 3 * a->patch_size * a->patch_size;
  E->Kp = (K + 63) / 64 * 64;
  *handle = E;
  return 0;
}

int mvlpt_set_precision(void* h, int mode) {
  Engine* E = (Engine*)h;
  if (!E) return MVLPT_ERR_ARG;
  if (mode != MVLPT_PREC_FAST && mode != MVLPT_PREC_SPLIT_GRAD && mode != MVLPT_PREC_SPLIT_ALL)
    return fail(E, MVLPT_ERR_ARG, "set_precision: unknown mode");
  E->prec_mode = mode;
  return 0;
}

int mvlpt_stream_create_cus(int cu_first, int cu_count, mvlpt_stream_t* stream) {
  if (!stream) return fail(nullptr, MVLPT_ERR_ARG, "stream_create_cus: null argument");
  const int total = device_cus();
  if (cu_first < 0 || cu_count <= 0 || cu_first + cu_count > total)
    return fail(nullptr, MVLPT_ERR_ARG, "stream_create_cus: CU range outside the device (" + std::to_string(total) + " compute units)");
  // bit i of the mask = logical compute unit i; the driver deals logical CUs round-robin over the XCDs (bit i -> XCD i % 8),
  // so a contiguous range of 8k bits is k compute units on EVERY XCD: each partition keeps all eight L2s
  std::vector<uint32_t> mask((total + 31) / 32, 0u);
  for (int i = cu_first; i < cu_first + cu_count; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) return hipfail(nullptr, e, "hipExtStreamCreateWithCUMask");
  { std::lock_guard<std::mutex> lk(g_part_mu); g_parts.push_back(StreamPart{s, cu_count}); }
  *stream = (mvlpt_stream_t)s;
  return 0;
}

int mvlpt_stream_destroy(mvlpt_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!s) return 0;
  {
    std::lock_guard<std::mutex> lk(g_part_mu);
    for (size_t i = 0; i < g_parts.size(); ++i) if (g_parts[i].s == s) { g_parts.erase(g_parts.begin() + i); break; }
  }
  hipError_t e = hipStreamDestroy(s);
  if (e != hipSuccess) return hipfail(nullptr, e, "hipStreamDestroy");
  return 0;
}

int mvlpt_stream_cus(mvlpt_stream_t stream) { return stream_cus((hipStream_t)stream); }

int mvlpt_set_resid_packed(void* h, int on) {
  Engine* E = (Engine*)h;
  if (!E) return MVLPT_ERR_ARG;
  if ((on != 0) != (E->resid_packed != 0)) E->fold_ready = false;      // the gamma-folded weights are built by prepare_fold
  E->resid_packed = on ? 1 : 0;
  return 0;
}
int mvlpt_set_ln_fold(void* h, int mode, int min_rows) {
  Engine* E = (Engine*)h;
  if (!E) return MVLPT_ERR_ARG;
  if (mode < 0 || mode > 2 || min_rows < 1) return fail(E, MVLPT_ERR_ARG, "set_ln_fold: mode in {0, 1, 2}, min_rows >= 1");
  E->fold_mode = mode; E->fold_min_rows = min_rows;
  return 0;
}

int mvlpt_set_vpt_dropout(void* h, const float* masks, int n_layers, int batch, int n_vpt, int width) {
  Engine* E = (Engine*)h;
  if (!E) return MVLPT_ERR_ARG;
  if (masks && (n_layers <= 0 || batch <= 0 || n_vpt <= 0 || width != E->arch.vision_width))
    return fail(E, MVLPT_ERR_ARG, "set_vpt_dropout: masks are [n_layers, batch, n_vpt, vision_width], every extent positive");
  E->vpt_mask = masks;
  E->vpt_mask_layers = masks ? n_layers : 0;
  E->vpt_mask_B = masks ? batch : 0; E->vpt_mask_n = masks ? n_vpt : 0; E->vpt_mask_d = masks ? width : 0;
  return 0;
}

int mvlpt_debug_checksums(void* h, int enable, unsigned long long* host_out, int max_out) {
  Engine* E = (Engine*)h;
  if (!E) return MVLPT_ERR_ARG;
  int n = 0;
  if (host_out && max_out > 0 && E->dbg_ck) {
    HIPCHK(E, hipDeviceSynchronize());
    n = E->dbg_ck_n < max_out ? E->dbg_ck_n : max_out;
    HIPCHK(E, hipMemcpy(host_out, E->dbg_ck, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  }
  if (enable && !E->dbg_ck) {
    void* p = nullptr;
    HIPCHK(E, hipMalloc(&p, DBG_CK_MAX * sizeof(unsigned long long)));
    E->owned.push_back(p);
    E->dbg_ck = (unsigned long long*)p;
  }
  E->dbg_ck_on = enable != 0;
  return n;
}

int mvlpt_trim(void* h) {
  Engine* E = (Engine*)h;
  if (!E) return MVLPT_ERR_ARG;
  HIPCHK(E, hipDeviceSynchronize());       // epoch boundary: a host-side pause is acceptable here, never inside a step
  for (DevBuf* b : {&E->vis_ws, &E->txt_ws, &E->head_ws, &E->ce_ws, &E->tmp, &E->pp_ws}) (void)b->trim();
  return 0;
}

int mvlpt_destroy(void* h) {
  if (!h) return 0;
  Engine* E = (Engine*)h;
  (void)hipDeviceSynchronize();
  for (void* p : E->owned) (void)hipFree(p);
  for (hipEvent_t ev : E->ev_pool) (void)hipEventDestroy(ev);
  delete E;
  return 0;
}

int mvlpt_load_frozen(void* h, const char* name, const void* dev_ptr, int dtype, const int64_t* shape, int ndim,
                      mvlpt_stream_t stream) {
  Engine* E = (Engine*)h;
  if (!E || !name || !dev_ptr || !shape || ndim < 0 || ndim > 4) return fail(E, MVLPT_ERR_ARG, "null/invalid argument");
  hipStream_t s = (hipStream_t)stream;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  std::string nm(name);
  if (nm == "logit_scale" || nm == "token_embedding.weight") return 0;  // not used by the towers
  E->fold_ready = false;      // W gamma / b + W beta of the LayerNorm folding are rebuilt from the new tensors (prepare_fold)
  // stage as fp32
  const float* p32 = (const float*)dev_ptr;
  if (dtype != MVLPT_DT_F32) {
    HIPCHK(E, E->tmp.reserve(n * 4));
    HIPCHK(E, launch_cast_any_to_f32(dtype, dev_ptr, (float*)E->tmp.p, n, s));
    p32 = (const float*)E->tmp.p;
  }
  const MvlptArch& A = E->arch;
  const int dv = A.vision_width, dtw = A.text_width, e = A.embed_dim;
  const int G2 = (A.image_resolution / A.patch_size) * (A.image_resolution / A.patch_size);
  auto is1 = [&](int r) { return ndim == 1 && shape[0] == r; };
  auto is2 = [&](int r, int c) { return ndim == 2 && shape[0] == r && shape[1] == c; };
  int rc = 0;
  const char* vb = "visual.transformer.resblocks.";
  const char* tb = "transformer.resblocks.";
  if (nm.rfind(vb, 0) == 0 || nm.rfind(tb, 0) == 0) {
    const bool isv = nm.rfind(vb, 0) == 0;
    int layer; const char* tail;
    if (!parse_block_name(name + strlen(isv ? vb : tb), &layer, &tail)) return fail(E, MVLPT_ERR_ARG, "malformed block name: " + nm);
    rc = load_block_tensor(E, isv ? E->vis : E->txt, layer, tail, p32, shape, ndim, s);
  } else if (nm == "visual.conv1.weight") {
    if (!(ndim == 4 && shape[0] == dv && shape[1] == 3 && shape[2] == A.patch_size && shape[3] == A.patch_size))
      return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm);
    void* w = nullptr;
    HIPCHK(E, hipMalloc(&w, (size_t)dv * E->Kp * 2)); E->owned.push_back(w);
    HIPCHK(E, launch_pack_weight(E->dt, p32, w, dv, 3 * A.patch_size * A.patch_size, E->Kp, s));
    E->conv_w = w;
  } else if (nm == "visual.class_embedding") { if (!is1(dv)) return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm); rc = upload_f32(E, p32, n, &E->cls_emb, s);
  } else if (nm == "visual.positional_embedding") { if (!is2(G2 + 1, dv)) return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm); rc = upload_f32(E, p32, n, &E->vpos, s);
  } else if (nm == "visual.ln_pre.weight") { if (!is1(dv)) return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm); rc = upload_f32(E, p32, n, &E->ln_pre.g, s);
  } else if (nm == "visual.ln_pre.bias") { if (!is1(dv)) return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm); rc = upload_f32(E, p32, n, &E->ln_pre.b, s);
  } else if (nm == "visual.ln_post.weight") { if (!is1(dv)) return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm); rc = upload_f32(E, p32, n, &E->ln_post.g, s);
  } else if (nm == "visual.ln_post.bias") { if (!is1(dv)) return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm); rc = upload_f32(E, p32, n, &E->ln_post.b, s);
  } else if (nm == "visual.proj") {
    if (!is2(dv, e)) return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm);
    // proj is [dv,e]: forward Bt = proj^T [e,dv]; dX Bt = proj [dv,e]; both kept in fp32
    void *w = nullptr, *wt = nullptr;
    HIPCHK(E, hipMalloc(&w, n * 4)); E->owned.push_back(w);
    HIPCHK(E, hipMalloc(&wt, n * 4)); E->owned.push_back(wt);
    HIPCHK(E, launch_pack_weight(DT_F32, p32, w, dv, e, e, s));
    HIPCHK(E, launch_pack_weight_t(DT_F32, p32, wt, dv, e, s));
    E->vproj = (float*)w; E->vproj_t = (float*)wt;
  } else if (nm == "positional_embedding") { if (!is2(A.context_length, dtw)) return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm); rc = upload_f32(E, p32, n, &E->tpos, s);
  } else if (nm == "ln_final.weight") { if (!is1(dtw)) return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm); rc = upload_f32(E, p32, n, &E->ln_final.g, s);
  } else if (nm == "ln_final.bias") { if (!is1(dtw)) return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm); rc = upload_f32(E, p32, n, &E->ln_final.b, s);
  } else if (nm == "text_projection") {
    if (!is2(dtw, e)) return fail(E, MVLPT_ERR_ARG, "shape mismatch: " + nm);
    void *w = nullptr, *wt = nullptr;
    HIPCHK(E, hipMalloc(&w, n * 4)); E->owned.push_back(w);
    HIPCHK(E, hipMalloc(&wt, n * 4)); E->owned.push_back(wt);
    HIPCHK(E, launch_pack_weight(DT_F32, p32, w, dtw, e, e, s));
    HIPCHK(E, launch_pack_weight_t(DT_F32, p32, wt, dtw, e, s));
    E->tproj = (float*)w; E->tproj_t = (float*)wt;
  } else {
    return fail(E, MVLPT_ERR_ARG, "unknown frozen tensor name: " + nm);
  }
  if (rc) return rc;
  if (dtype != MVLPT_DT_F32) HIPCHK(E, hipStreamSynchronize(s));  // tmp is reused by the next call
  return 0;
}

int mvlpt_frozen_ready(void* h) {
  Engine* E = (Engine*)h;
  if (!E) return MVLPT_ERR_ARG;
  const char* m = first_missing(E);
  if (m) return fail(E, MVLPT_ERR_STATE, std::string("frozen tensor not loaded: ") + m);
  return 0;
}

// ------------------------------------------------------------------------------------------------ image tower
int mvlpt_image_fwd(void* h, const void* image, int image_dtype, const float* vpt, const float* vpt_deep, int n_vpt, int n_deep,
                    int B, float* feat_out, int save_for_bwd, mvlpt_stream_t stream) {
  Engine* E = (Engine*)h;
  if (!E || !image || !feat_out || B <= 0) return fail(E, MVLPT_ERR_ARG, "image_fwd: null/invalid argument");
  if (int rc = mvlpt_frozen_ready(h)) return rc;
  if ((n_vpt > 0) != (vpt != nullptr)) return fail(E, MVLPT_ERR_ARG, "image_fwd: vpt pointer and n_vpt disagree");
  if ((n_deep > 0) != (vpt_deep != nullptr) || (n_deep > 0 && n_vpt <= 0)) return fail(E, MVLPT_ERR_ARG, "image_fwd: deep prompts need vpt");
  hipStream_t s = (hipStream_t)stream;
  const MvlptArch& A = E->arch;
  const int G = A.image_resolution / A.patch_size, G2 = G * G, dv = A.vision_width, e = A.embed_dim;
  const int Lv = 1 + n_vpt + G2;
  if (Lv > attn_max_len()) return fail(E, MVLPT_ERR_UNSUPPORTED, "image_fwd: sequence length > 256 not supported yet");
  const bool save = save_for_bwd != 0;
  const bool exact = E->prec_mode == MVLPT_PREC_SPLIT_ALL || (E->prec_mode == MVLPT_PREC_SPLIT_GRAD && save);
  const size_t X = exact ? 2 : 1;
  const size_t npatch = (size_t)B * G2;
  size_t need = tower_bytes(E->vis, B, Lv, save, exact) + align256(npatch * E->Kp * 2) + align256(npatch * dv * 4) +
                7 * align256((size_t)B * dv * 4) + 4 * align256((size_t)B * dv * 2 * X) + 3 * align256((size_t)B * dv * 8 * X) + 4096;
  E->vs.valid = false;
  HIPCHK(E, E->vis_ws.reserve(need));
  Bump bp; bp.base = (char*)E->vis_ws.p; bp.cap = E->vis_ws.cap;
  void* patches = bp.take_bytes(npatch * E->Kp * 2);
  float* pe = bp.take<float>(npatch * dv);
  E->cls32 = bp.take<float>((size_t)B * dv);
  E->dcls32 = bp.take<float>((size_t)B * dv);
  E->xc32 = bp.take<float>((size_t)B * dv);
  E->ac16 = bp.take_bytes((size_t)B * dv * 2 * X); E->hc16 = bp.take_bytes((size_t)B * dv * 2 * X);
  E->gc16 = bp.take_bytes((size_t)B * dv * 4 * 2 * X);
  E->xcm32 = bp.take<float>((size_t)B * dv); E->xco32 = bp.take<float>((size_t)B * dv);
  E->dxc32 = bp.take<float>((size_t)B * dv); E->dhc32 = bp.take<float>((size_t)B * dv);
  E->dxc16 = bp.take_bytes((size_t)B * dv * 2 * X); E->dOc16 = bp.take_bytes((size_t)B * dv * 2 * X);
  E->uc16 = bp.take_bytes((size_t)B * dv * 4 * 2); E->duc16 = bp.take_bytes((size_t)B * dv * 4 * 2 * X);
  E->v_cls_last = false;
  carve_tower(bp, E->vis, E->vs, B, Lv, save, false, exact, split_kind(E));
  TowerState& st = E->vs;
  E->vB = B; E->v_nvpt = n_vpt; E->v_ndeep = n_deep;
  st.fold = E->fold_mode >= 1 && (size_t)B * Lv >= (size_t)E->fold_min_rows && dv >= 256;
  if (st.fold) if (int rc = prepare_fold(E, s)) return rc;

  // vpt_dropout masks of layer l's prompt rows (mvlpt_set_vpt_dropout), or null
  // the setting belongs to THIS forward (and its backward) only
  const float* const masks = E->vpt_mask;
  const int mask_layers = E->vpt_mask_layers, mask_B = E->vpt_mask_B, mask_n = E->vpt_mask_n;
  E->vpt_mask = nullptr; E->vpt_mask_layers = 0; E->v_mask = nullptr;
  if (masks && (n_vpt <= 0 || mask_layers < 1 + n_deep || mask_B != B || mask_n != n_vpt))
    return fail(E, MVLPT_ERR_ARG, "image_fwd: the prompt dropout masks do not match this forward (layers >= 1 + n_deep, batch, n_vpt)");
  E->v_mask = masks;
  auto vmask = [&](int l) -> const float* { return masks ? masks + (size_t)l * B * n_vpt * dv : nullptr; };
  { ProfScope ps(E, s, PC_GLUE, 0, (double)npatch * E->Kp * 6.0);
    HIPCHK(E, launch_patchify(E->dt, image, image_dtype, patches, B, A.image_resolution, A.patch_size, E->Kp, s)); }
  if (E->dbg_ck_on) { E->dbg_ck_n = 0; if (E->dbg_ck) (void)hipMemsetAsync(E->dbg_ck, 0, DBG_CK_MAX * sizeof(unsigned long long), s); }
  dbg_ck(E, patches, (size_t)npatch * E->Kp * 2, s);
  HIPCHK(E, gemm(E, EPI_STORE32, patches, WRef{E->conv_w, 0, 0}, (int)npatch, dv, E->Kp, nullptr, nullptr, nullptr, pe, nullptr, s));
  dbg_ck(E, pe, (size_t)npatch * dv * 4, s);
  // Packed residual stream (block_fwd_packed): fp16 tower, no prompts, nothing saved, every LayerNorm folded.  assemble_tokens
  // writes the rows in the packed format together with the row statistics of ln_1 of block 0.
  const int nt_d = dv / 128, ntp_d = (nt_d + 1) & ~1;
  const bool packed = E->resid_packed && E->dt == DT_F16 && !save && !exact && st.fold && n_vpt == 0 && dv % 128 == 0 &&
                      ntp_d <= FOLD_NTP && E->vis.blocks[0].qkv.wg != nullptr;
  void* const xhi = st.x[0];
  uint8_t* const xlo = (uint8_t*)st.x[0] + (size_t)B * Lv * dv * 2;
  if (packed) {
    ProfScope ps(E, s, PC_GLUE, 0, (double)B * Lv * dv * 7.0);
    HIPCHK(E, launch_assemble_tokens_packed(pe, E->cls_emb, E->vpos, E->ln_pre.g, E->ln_pre.b, xhi, xlo, st.part[0], ntp_d, B, G2, dv, s));
    st.nt[0] = nt_d; st.ntp[0] = ntp_d;
    dbg_ck(E, xhi, (size_t)B * Lv * dv * 3, s); dbg_ck(E, st.part[0], (size_t)B * Lv * ntp_d * 8, s);
  } else {
    ProfScope ps(E, s, PC_GLUE, 0, (double)B * Lv * dv * 8.0);
    HIPCHK(E, launch_assemble_tokens(pe, E->cls_emb, E->vpos, E->ln_pre.g, E->ln_pre.b, vpt, n_vpt, st.x[0], B, G2, dv, s, vmask(0)));
  }
  bool cls_only_last = false;
  bool ln1_ready = packed;      // LayerNorm folding: the previous block's FC2 left this block's ln_1 input in folded form
  // ln_1 of block l can be folded when nothing touches the residual stream between FC2 of block l-1 and it: not behind a
  // deep-prompt overwrite, not behind a skipped block
  auto next_foldable = [&](int l) -> const LNp* {
    const int n = l + 1;
    if (!st.fold || n >= E->vis.layers) return nullptr;
    if (n_deep > 0) return nullptr;                       // rows 1..n_vpt are overwritten (or the block is skipped) in front of every later ln_1
    return &E->vis.blocks[n].ln1;
  };
  for (int l = 0; l < E->vis.layers; ++l) {
    if (l > 0 && n_deep > 0) {
      if (l <= n_deep) {
        ProfScope ps(E, s, PC_GLUE, 0, (double)B * n_vpt * dv * 4.0);
        HIPCHK(E, launch_overwrite_rows(vpt_deep + (size_t)(l - 1) * n_vpt * dv, n_vpt, st.x[2 * l], B, Lv, dv, s, vmask(l)));
      } else {
        // reference quirk (trainers/mvlpt.py:71-83 has no `else`): the layer is skipped entirely
        st.skip[l] = 1;
        if (st.saved) {
          HIPCHK(E, hipMemcpyAsync(st.x[2 * l + 2], st.x[2 * l], (size_t)B * Lv * dv * 4, hipMemcpyDeviceToDevice, s));
        }
        continue;
      }
    }
    if (l == E->vis.layers - 1) { cls_only_last = true; break; }
    if (packed) { if (int rc = block_fwd_packed(E, E->vis, st, l, s)) return rc; continue; }
    bool produced = false;
    if (int rc = block_fwd(E, E->vis, st, l, s, ln1_ready, next_foldable(l), &produced)) return rc;
    ln1_ready = produced;
  }
  if (cls_only_last) {
    // Only x[:, 0, :] of the last block is consumed (trainers/mvlpt.py:88): keys/values are still needed for every
    // token, but queries, out-proj, ln_2 and the MLP are evaluated for the CLS row only (B rows instead of B*Lv: ~20/24
    // of the layer's GEMM FLOPs vanish) — in the forward and, when the tower has a backward, in the backward as well
    // (mvlpt_image_bwd: only the CLS query carries a gradient into this block's attention).
    const int l = E->vis.layers - 1;
    const Block& Bk = E->vis.blocks[l];
    const int T = B * Lv;
    float* xin = st.x[2 * l];
    float* xmid = save ? E->xcm32 : E->xc32;     // kept for the backward: LN2 input ...
    float* xout = save ? E->xco32 : E->xc32;     // ... and ln_post input
    E->v_cls_last = save;
    const int xs = st.xs;                 // split-precision operands (hi|lo pairs, twice the columns) + pair-product attention
    Fold fq;
    if (ln1_ready) fq = fold_consumer(st, 0, Bk.qkv);
    else HIPCHK(E, ln_fwd(E, E->dt, xin, nullptr, 1, Bk.ln1, st.h16, T, dv, s, xs));
    if (packed) fq.fold_colsum = Bk.qkv.fold_sg;       // A = the stream's hi plane, gamma inside the weight
    HIPCHK(E, gemm(E, xs ? EPI_STORE_SPLIT : EPI_STORE16, packed ? xhi : st.h16, packed ? Bk.qkv.fwg() : Bk.qkv.fw(), T, 3 * dv, dv,
                   ln1_ready ? Bk.qkv.fold_b : Bk.qkv.b, nullptr, nullptr, st.qkv[l], nullptr, s, -1, xs, ln1_ready ? &fq : nullptr));
    dbg_ck(E, st.qkv[l], (size_t)T * 3 * dv * 2 * X, s);
    if (xs) {
      // only the CLS rows of the attention output are produced: the backward (delta = rowsum(dO * O) over EVERY row, with
      // dO = 0 off the CLS rows) must not meet uninitialised memory there
      if (save) HIPCHK(E, launch_zero(st.attn[l], (size_t)T * dv * 4, s));
      if (int rc = attn32_fwd(E, st, l, 1, s)) return rc;
    } else {
      AttnArgs a{st.qkv[l], st.attn[l], save ? st.lse[l] : nullptr, st.N, st.L, st.H, 0, 1};
      ProfScope ps(E, s, PC_ATTN_FWD, 4.0 * st.L * 64.0 * st.N * st.H, (double)T * dv * 2.0 * 2.0);
      HIPCHK(E, launch_attn_fwd(E->dt, a, s));
    }
    { ProfScope ps(E, s, PC_GLUE, 0, (double)B * dv * 12.0);
      HIPCHK(E, launch_copy_rows_strided(st.attn[l], E->ac16, B, (size_t)Lv * dv * 2 * X, (size_t)dv * 2 * X, (int)(dv * 2 * X), s));
      if (packed) HIPCHK(E, launch_respk_unpack_rows(xhi, xlo, Lv, E->xc32, B, dv, s));
      else HIPCHK(E, launch_copy_rows_strided(xin, E->xc32, B, (size_t)Lv * dv * 4, (size_t)dv * 4, dv * 4, s)); }
    dbg_ck(E, E->ac16, (size_t)B * dv * 2 * X, s); dbg_ck(E, E->xc32, (size_t)B * dv * 4, s);
    HIPCHK(E, gemm(E, EPI_RESID32, E->ac16, Bk.o.fw(), B, dv, dv, Bk.o.b, nullptr, E->xc32, xmid, nullptr, s, -1, xs));
    HIPCHK(E, ln_fwd(E, E->dt, xmid, nullptr, 1, Bk.ln2, E->hc16, B, dv, s, xs));
    HIPCHK(E, gemm(E, xs ? EPI_GELU_SPLIT : EPI_GELU, E->hc16, Bk.fc.fw(), B, 4 * dv, dv, Bk.fc.b, nullptr, nullptr, E->gc16, save ? E->uc16 : nullptr, s, -1, xs));
    HIPCHK(E, gemm(E, EPI_RESID32, E->gc16, Bk.pr.fw(), B, dv, 4 * dv, Bk.pr.b, nullptr, xmid, xout, nullptr, s, -1, xs));
    HIPCHK(E, ln_fwd(E, DT_F32, xout, nullptr, 1, E->ln_post, E->cls32, B, dv, s));
  } else
  // ln_post on the CLS row, then @ proj   (trainers/mvlpt.py:88-91)
  HIPCHK(E, ln_fwd(E, DT_F32, st.x[2 * E->vis.layers], nullptr, Lv, E->ln_post, E->cls32, B, dv, s));
  { ProfScope ps(E, s, PC_HEAD, 2.0 * B * e * dv, 4.0 * ((double)B * dv + (double)e * dv + (double)B * e));
    HIPCHK(E, launch_sgemm_bt(E->cls32, E->vproj_t, feat_out, B, e, dv, nullptr, s)); }
  return 0;
}

int mvlpt_image_bwd(void* h, const float* dfeat, float* dvpt, float* dvpt_deep, mvlpt_stream_t stream) {
  Engine* E = (Engine*)h;
  if (!E || !dfeat) return fail(E, MVLPT_ERR_ARG, "image_bwd: null argument");
  TowerState& st = E->vs;
  if (!st.valid || !st.saved) return fail(E, MVLPT_ERR_STATE, "image_bwd: call image_fwd(save_for_bwd=1) first");
  if ((E->v_nvpt > 0 && !dvpt) || (E->v_ndeep > 0 && !dvpt_deep)) return fail(E, MVLPT_ERR_ARG, "image_bwd: missing gradient output");
  hipStream_t s = (hipStream_t)stream;
  const MvlptArch& A = E->arch;
  const int B = E->vB, dv = A.vision_width, e = A.embed_dim, Lv = st.L, n = E->v_nvpt;
  const size_t T = (size_t)B * Lv;
  { ProfScope ps(E, s, PC_GLUE, 0, (double)T * dv * 10.0);
    HIPCHK(E, launch_grad_scale(dfeat, (size_t)B * e, 64.0f, st.scale_dev, s));
    HIPCHK(E, launch_sgemm_bt(dfeat, E->vproj, E->dcls32, B, dv, e, st.scale_dev, s)); }
  { ProfScope ps(E, s, PC_GLUE, 0, (double)T * dv * 4.0);
    HIPCHK(E, launch_zero(st.dx32, T * dv * 4, s)); }
  int l_top = st.layers - 1;
  const int xs = st.xs;
  if (E->v_cls_last) {
    // last block, CLS rows only (see mvlpt_image_fwd): ln_post, MLP, ln_2, out-proj on B compact rows; the attention
    // backward of a single query per head; then the full-width QKV^T GEMM and ln_1 (keys / values of every token)
    const int l = st.layers - 1;
    const Block& Bk = E->vis.blocks[l];
    const int Ti = (int)T;
    HIPCHK(E, ln_bwd(E, E->dcls32, DT_F32, E->xco32, nullptr, 1, E->ln_post, nullptr, E->dxc32, E->dxc16, B, dv, s, -1, xs));
    HIPCHK(E, gemm(E, xs ? EPI_GELUBWD_SPLIT : EPI_GELUBWD, E->dxc16, Bk.pr.bw(), B, 4 * dv, dv, nullptr, E->uc16, nullptr, E->duc16, nullptr, s, -1, xs));
    HIPCHK(E, gemm(E, EPI_STORE32, E->duc16, Bk.fc.bw(), B, dv, 4 * dv, nullptr, nullptr, nullptr, E->dhc32, nullptr, s, -1, xs));
    HIPCHK(E, ln_bwd(E, E->dhc32, DT_F32, E->xcm32, nullptr, 1, Bk.ln2, E->dxc32, E->dxc32, E->dxc16, B, dv, s, -1, xs));
    HIPCHK(E, gemm(E, xs ? EPI_STORE_SPLIT : EPI_STORE16, E->dxc16, Bk.o.bw(), B, dv, dv, nullptr, nullptr, nullptr, E->dOc16, nullptr, s, -1, xs));
    if (xs) {
      // split-precision attention backward at full width on a dO that is zero except for the CLS rows
      { ProfScope ps(E, s, PC_GLUE, 0, (double)T * dv * 4.0);
        HIPCHK(E, launch_zero(st.dO16, T * dv * 4, s));
        HIPCHK(E, launch_copy_rows_strided(E->dOc16, st.dO16, B, (size_t)dv * 4, (size_t)Lv * dv * 4, dv * 4, s)); }
      if (int rc = attn32_bwd(E, st, l, s)) return rc;
    } else {
      ProfScope ps(E, s, PC_ATTN_BWD, 10.0 * st.L * 64.0 * st.N * st.H, (double)T * dv * 2.0 * 5.0);
      HIPCHK(E, launch_attn_bwd_cls(E->dt, st.qkv[l], E->ac16, E->dOc16, st.lse[l], st.dqkv16, st.N, st.L, st.H, s)); }
    { ProfScope ps(E, s, PC_GLUE, 0, (double)B * dv * 8.0);      // residual path: d(block input) of the CLS rows
      HIPCHK(E, launch_copy_rows_strided(E->dxc32, st.dx32, B, (size_t)dv * 4, (size_t)Lv * dv * 4, dv * 4, s)); }
    HIPCHK(E, gemm(E, EPI_STORE32, st.dqkv16, Bk.qkv.bw(), Ti, dv, 3 * dv, nullptr, nullptr, nullptr, st.dh32, nullptr, s, -1, xs));
    HIPCHK(E, ln_bwd(E, st.dh32, DT_F32, st.x[2 * l], nullptr, 1, Bk.ln1, st.dx32, st.dx32, st.dx16, Ti, dv, s, -1, xs));
    if (l > 0 && E->v_ndeep > 0 && l <= E->v_ndeep) {
      ProfScope ps(E, s, PC_GLUE, 0, (double)B * n * dv * 10.0);
      HIPCHK(E, launch_reduce_prompt_rows(E->dt, st.dx32, st.dx16, B, Lv, dv, 1, n, dvpt_deep + (size_t)(l - 1) * n * dv,
                                          st.scale_dev, 1, s, xs, E->v_mask ? E->v_mask + (size_t)l * B * n * dv : nullptr));
    }
    l_top = l - 1;
  } else {
    HIPCHK(E, ln_bwd(E, E->dcls32, DT_F32, st.x[2 * st.layers], nullptr, Lv, E->ln_post, nullptr, st.dx32, nullptr, B, dv, s));
    { ProfScope ps(E, s, PC_GLUE, 0, (double)T * dv * 6.0);
      if (xs) HIPCHK(E, launch_cast_f32_split(E->dt, st.dx32, st.dx16, T, dv, nullptr, s, xs == 2));
      else HIPCHK(E, launch_cast_f32_to16(E->dt, st.dx32, st.dx16, T * dv, nullptr, s)); }
  }
  for (int l = l_top; l >= 0; --l) {
    if (st.skip[l]) continue;
    if (int rc = block_bwd(E, E->vis, st, l, s)) return rc;
    if (l > 0 && E->v_ndeep > 0 && l <= E->v_ndeep) {
      ProfScope ps(E, s, PC_GLUE, 0, (double)B * n * dv * 10.0);
      HIPCHK(E, launch_reduce_prompt_rows(E->dt, st.dx32, st.dx16, B, Lv, dv, 1, n, dvpt_deep + (size_t)(l - 1) * n * dv,
                                          st.scale_dev, 1, s, xs, E->v_mask ? E->v_mask + (size_t)l * B * n * dv : nullptr));
    }
  }
  if (n > 0) {
    ProfScope ps(E, s, PC_GLUE, 0, (double)B * n * dv * 4.0);
    HIPCHK(E, launch_reduce_prompt_rows(E->dt, st.dx32, st.dx16, B, Lv, dv, 1, n, dvpt, st.scale_dev, 0, s, 0, E->v_mask));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ text tower
int mvlpt_text_fwd(void* h, const float* prefix, const float* suffix, const float* ctx, int ctx_per_class, int n_ctx,
                   const int32_t* layout, const int32_t* eot, int C, int L, float* feat_out, int save_for_bwd,
                   mvlpt_stream_t stream) {
  Engine* E = (Engine*)h;
  if (!E || !prefix || !suffix || !layout || !eot || !feat_out || C <= 0 || L <= 0)
    return fail(E, MVLPT_ERR_ARG, "text_fwd: null/invalid argument");
  if (int rc = mvlpt_frozen_ready(h)) return rc;
  if ((n_ctx > 0) != (ctx != nullptr) || n_ctx < 0 || n_ctx > L - 2) return fail(E, MVLPT_ERR_ARG, "text_fwd: ctx pointer and n_ctx disagree");
  const MvlptArch& A = E->arch;
  if (L > A.context_length) return fail(E, MVLPT_ERR_ARG, "text_fwd: L exceeds context_length");
  if (L > attn_max_len()) return fail(E, MVLPT_ERR_UNSUPPORTED, "text_fwd: L > 256");
  hipStream_t s = (hipStream_t)stream;
  const int dtw = A.text_width, e = A.embed_dim;
  const bool save = save_for_bwd != 0;
  // The text tower runs with split operands in every mode but MVLPT_PREC_FAST, also when nothing is saved for a backward:
  // n_ctx == 0: the features are constants of the run (computed once and cached by the caller) that enter every
  // image-side gradient through the logits; inference (model_inference, trainers/mvlpt.py:986-987): they are computed once
  // per parameter version, not per batch, so the extra matrix time does not recur — and single operands leave 5-9e-4 on the text
  // features, which put the inference logits of 5 of the 18 reference fixtures outside 1e-3 (profiles/r04_inference_parity.txt)
  const bool exact = E->prec_mode == MVLPT_PREC_SPLIT_ALL || E->prec_mode == MVLPT_PREC_SPLIT_GRAD;
  const size_t X = exact ? 2 : 1;
  size_t need = tower_bytes(E->txt, C, L, save, exact) + 7 * align256((size_t)C * dtw * 4) + 4 * align256((size_t)C * dtw * 2 * X) +
                3 * align256((size_t)C * dtw * 8 * X) + align256((size_t)C * 4) + align256((size_t)C * (n_ctx > 0 ? n_ctx : 1) * 4) + 4096;
  E->ts.valid = false;
  HIPCHK(E, E->txt_ws.reserve(need));
  Bump bp; bp.base = (char*)E->txt_ws.p; bp.cap = E->txt_ws.cap;
  E->eot32 = bp.take<float>((size_t)C * dtw);
  E->deot32 = bp.take<float>((size_t)C * dtw);
  E->eot_rows = bp.take<int32_t>(C);
  E->ctx_pos = bp.take<int32_t>((size_t)C * (n_ctx > 0 ? n_ctx : 1));
  E->txc32 = bp.take<float>((size_t)C * dtw); E->txm32 = bp.take<float>((size_t)C * dtw); E->txo32 = bp.take<float>((size_t)C * dtw);
  E->tdxc32 = bp.take<float>((size_t)C * dtw); E->tdhc32 = bp.take<float>((size_t)C * dtw);
  E->tac16 = bp.take_bytes((size_t)C * dtw * 2 * X); E->thc16 = bp.take_bytes((size_t)C * dtw * 2 * X);
  E->tdxc16 = bp.take_bytes((size_t)C * dtw * 2 * X); E->tdOc16 = bp.take_bytes((size_t)C * dtw * 2 * X);
  E->tgc16 = bp.take_bytes((size_t)C * dtw * 8 * X); E->tuc16 = bp.take_bytes((size_t)C * dtw * 8); E->tduc16 = bp.take_bytes((size_t)C * dtw * 8 * X);
  E->t_eot_last = false;
  carve_tower(bp, E->txt, E->ts, C, L, save, true, exact, split_kind(E));
  TowerState& st = E->ts;
  E->tC = C; E->tL = L; E->t_nctx = n_ctx; E->t_per_class = ctx_per_class;
  st.fold = E->fold_mode >= 2 && (size_t)C * L >= (size_t)E->fold_min_rows && dtw >= 256;
  if (st.fold) if (int rc = prepare_fold(E, s)) return rc;
  { ProfScope ps(E, s, PC_GLUE, 0, (double)C * L * dtw * 12.0);
    HIPCHK(E, launch_assemble_prompts(prefix, suffix, ctx, ctx_per_class, n_ctx, layout, E->tpos, st.x[0], C, L, dtw, s));
    HIPCHK(E, launch_eot_rows(eot, E->eot_rows, C, L, s));
    if (save && n_ctx > 0) HIPCHK(E, launch_build_ctx_pos(layout, E->ctx_pos, C, L, n_ctx, s)); }
  bool ln1_ready = false;
  for (int l = 0; l + 1 < E->txt.layers; ++l) {
    bool produced = false;
    if (int rc = block_fwd(E, E->txt, st, l, s, ln1_ready, st.fold ? &E->txt.blocks[l + 1].ln1 : nullptr, &produced)) return rc;
    ln1_ready = produced;
  }
  {
    // Only x[c, eot_c] of the last block is consumed (trainers/mvlpt.py:126-128): LN1, QKV and the attention run for every
    // position (keys / values), then out-proj, ln_2, the MLP and ln_final run on the C gathered EOT rows — forward and,
    // in mvlpt_text_bwd, backward.
    const int l = E->txt.layers - 1;
    const Block& Bk = E->txt.blocks[l];
    const int T = C * L;
    float* xin = st.x[2 * l];
    E->t_eot_last = save;
    const int xs = st.xs;
    Fold fq;
    if (ln1_ready) fq = fold_consumer(st, 0, Bk.qkv);
    else HIPCHK(E, ln_fwd(E, E->dt, xin, nullptr, 1, Bk.ln1, st.h16, T, dtw, s, xs));
    HIPCHK(E, gemm(E, xs ? EPI_STORE_SPLIT : EPI_STORE16, st.h16, Bk.qkv.fw(), T, 3 * dtw, dtw, ln1_ready ? Bk.qkv.fold_b : Bk.qkv.b, nullptr, nullptr, st.qkv[l], nullptr, s, -1, xs,
                   ln1_ready ? &fq : nullptr));
    if (xs) {
      if (int rc = attn32_fwd(E, st, l, 0, s)) return rc;
    } else {
      AttnArgs a{st.qkv[l], st.attn[l], save ? st.lse[l] : nullptr, st.N, st.L, st.H, 1};
      ProfScope ps(E, s, PC_ATTN_FWD, 2.0 * st.L * st.L * 64.0 * st.N * st.H, (double)T * dtw * 2.0 * 4.0);
      HIPCHK(E, launch_attn_fwd(E->dt, a, s));
    }
    { ProfScope ps(E, s, PC_GLUE, 0, (double)C * dtw * 12.0);
      HIPCHK(E, launch_copy_rows(st.attn[l], E->tac16, E->eot_rows, C, (int)(dtw * 2 * X), 0, s));
      HIPCHK(E, launch_copy_rows(xin, E->txc32, E->eot_rows, C, dtw * 4, 0, s)); }
    HIPCHK(E, gemm(E, EPI_RESID32, E->tac16, Bk.o.fw(), C, dtw, dtw, Bk.o.b, nullptr, E->txc32, E->txm32, nullptr, s, -1, xs));
    HIPCHK(E, ln_fwd(E, E->dt, E->txm32, nullptr, 1, Bk.ln2, E->thc16, C, dtw, s, xs));
    HIPCHK(E, gemm(E, xs ? EPI_GELU_SPLIT : EPI_GELU, E->thc16, Bk.fc.fw(), C, 4 * dtw, dtw, Bk.fc.b, nullptr, nullptr, E->tgc16, save ? E->tuc16 : nullptr, s, -1, xs));
    HIPCHK(E, gemm(E, EPI_RESID32, E->tgc16, Bk.pr.fw(), C, dtw, 4 * dtw, Bk.pr.b, nullptr, E->txm32, E->txo32, nullptr, s, -1, xs));
    HIPCHK(E, ln_fwd(E, DT_F32, E->txo32, nullptr, 1, E->ln_final, E->eot32, C, dtw, s));
  }
  { ProfScope ps(E, s, PC_HEAD, 2.0 * C * e * dtw, 4.0 * ((double)C * dtw + (double)e * dtw + (double)C * e));
    HIPCHK(E, launch_sgemm_bt(E->eot32, E->tproj_t, feat_out, C, e, dtw, nullptr, s)); }
  return 0;
}

int mvlpt_text_bwd(void* h, const float* dfeat, float* dctx, mvlpt_stream_t stream) {
  Engine* E = (Engine*)h;
  if (!E || !dfeat || !dctx) return fail(E, MVLPT_ERR_ARG, "text_bwd: null argument");
  TowerState& st = E->ts;
  if (!st.valid || !st.saved) return fail(E, MVLPT_ERR_STATE, "text_bwd: call text_fwd(save_for_bwd=1) first");
  if (E->t_nctx <= 0) return fail(E, MVLPT_ERR_STATE, "text_bwd: the forward had no context tokens");
  hipStream_t s = (hipStream_t)stream;
  const MvlptArch& A = E->arch;
  const int C = E->tC, L = E->tL, dtw = A.text_width, e = A.embed_dim;
  const size_t T = (size_t)C * L;
  { ProfScope ps(E, s, PC_GLUE, 0, (double)T * dtw * 10.0);
    HIPCHK(E, launch_grad_scale(dfeat, (size_t)C * e, 64.0f, st.scale_dev, s));
    HIPCHK(E, launch_sgemm_bt(dfeat, E->tproj, E->deot32, C, dtw, e, st.scale_dev, s)); }
  { ProfScope ps(E, s, PC_GLUE, 0, (double)T * dtw * 4.0);
    HIPCHK(E, launch_zero(st.dx32, T * dtw * 4, s)); }
  if (!E->t_eot_last) return fail(E, MVLPT_ERR_STATE, "text_bwd: the forward did not keep the last block's activations");
  {
    // last block on the C compact EOT rows (see mvlpt_text_fwd); the attention backward runs at full width on a dO that
    // is zero except for the EOT rows, then the QKV^T GEMM and ln_1 as usual
    const int l = st.layers - 1;
    const Block& Bk = E->txt.blocks[l];
    const int Ti = (int)T;
    const int xs = st.xs;
    const size_t X = xs ? 2 : 1;
    HIPCHK(E, ln_bwd(E, E->deot32, DT_F32, E->txo32, nullptr, 1, E->ln_final, nullptr, E->tdxc32, E->tdxc16, C, dtw, s, -1, xs));
    HIPCHK(E, gemm(E, xs ? EPI_GELUBWD_SPLIT : EPI_GELUBWD, E->tdxc16, Bk.pr.bw(), C, 4 * dtw, dtw, nullptr, E->tuc16, nullptr, E->tduc16, nullptr, s, -1, xs));
    HIPCHK(E, gemm(E, EPI_STORE32, E->tduc16, Bk.fc.bw(), C, dtw, 4 * dtw, nullptr, nullptr, nullptr, E->tdhc32, nullptr, s, -1, xs));
    HIPCHK(E, ln_bwd(E, E->tdhc32, DT_F32, E->txm32, nullptr, 1, Bk.ln2, E->tdxc32, E->tdxc32, E->tdxc16, C, dtw, s, -1, xs));
    HIPCHK(E, gemm(E, xs ? EPI_STORE_SPLIT : EPI_STORE16, E->tdxc16, Bk.o.bw(), C, dtw, dtw, nullptr, nullptr, nullptr, E->tdOc16, nullptr, s, -1, xs));
    { ProfScope ps(E, s, PC_GLUE, 0, (double)T * dtw * 2.0);
      HIPCHK(E, launch_zero(st.dO16, T * dtw * 2 * X, s));      // (exact: dO is a hi|lo pair)
      HIPCHK(E, launch_copy_rows(E->tdOc16, st.dO16, E->eot_rows, C, (int)(dtw * 2 * X), 1, s));
      HIPCHK(E, launch_copy_rows(E->tdxc32, st.dx32, E->eot_rows, C, dtw * 4, 1, s)); }     // residual path (dx32 was zeroed)
    if (xs) {
      if (int rc = attn32_bwd(E, st, l, s)) return rc;
    } else {
      AttnBwdArgs a{st.qkv[l], st.attn[l], st.dO16, st.lse[l], st.delta, st.dqkv16, st.N, st.L, st.H, 1};
      ProfScope ps(E, s, PC_ATTN_BWD, 7.0 * st.L * st.L * 64.0 * st.N * st.H, (double)T * dtw * 2.0 * 8.0);
      HIPCHK(E, launch_attn_bwd(E->dt, a, s));
    }
    HIPCHK(E, gemm(E, EPI_STORE32, st.dqkv16, Bk.qkv.bw(), Ti, dtw, 3 * dtw, nullptr, nullptr, nullptr, st.dh32, nullptr, s, -1, xs));
    HIPCHK(E, ln_bwd(E, st.dh32, DT_F32, st.x[2 * l], nullptr, 1, Bk.ln1, st.dx32, st.dx32, st.dx16, Ti, dtw, s, -1, xs));
  }
  for (int l = st.layers - 2; l >= 0; --l)
    if (int rc = block_bwd(E, E->txt, st, l, s)) return rc;
  { ProfScope ps(E, s, PC_GLUE, 0, (double)C * E->t_nctx * dtw * 4.0);
    HIPCHK(E, launch_gather_ctx_grad(st.dx32, E->ctx_pos, C, L, dtw, E->t_nctx, E->t_per_class, dctx, st.scale_dev, s)); }
  return 0;
}

// ------------------------------------------------------------------------------------------------ head
static int head_reserve(Engine* E, int B, int C) {
  const int e = E->arch.embed_dim;
  const size_t need = align256((size_t)B * e * 4) + align256((size_t)C * e * 4) + align256((size_t)B * 4) +
                      align256((size_t)C * 4) + 4096;
  HIPCHK(E, E->head_ws.reserve(need));
  Bump bp; bp.base = (char*)E->head_ws.p;
  E->imn = bp.take<float>((size_t)B * e); E->txn = bp.take<float>((size_t)C * e);
  E->inorm = bp.take<float>(B); E->tnorm = bp.take<float>(C);
  return 0;
}

int mvlpt_logits_fwd(void* h, const float* img, const float* txt, float scale, const int32_t* lo, const int32_t* hi, int B, int C,
                     float* logits, mvlpt_stream_t stream) {
  Engine* E = (Engine*)h;
  if (!E || !img || !txt || !logits || B <= 0 || C <= 0 || ((lo == nullptr) != (hi == nullptr)))
    return fail(E, MVLPT_ERR_ARG, "logits_fwd: null/invalid argument");
  hipStream_t s = (hipStream_t)stream;
  const int e = E->arch.embed_dim;
  E->hB = 0;
  if (int rc = head_reserve(E, B, C)) return rc;
  ProfScope ps(E, s, PC_HEAD, 2.0 * B * C * e, 4.0 * ((double)B * e + (double)C * e + (double)B * C));
  HIPCHK(E, launch_normalize_rows(img, E->imn, E->inorm, B, e, s));
  HIPCHK(E, launch_normalize_rows(txt, E->txn, E->tnorm, C, e, s));
  HIPCHK(E, launch_logits(E->imn, E->txn, scale, lo, hi, logits, B, C, e, s));
  E->hB = B; E->hC = C; E->h_scale = scale; E->h_lo = lo; E->h_hi = hi;
  return 0;
}

int mvlpt_logits_bwd(void* h, const float* dlogits, float* dimg, float* dtxt, mvlpt_stream_t stream) {
  Engine* E = (Engine*)h;
  if (!E || !dlogits) return fail(E, MVLPT_ERR_ARG, "logits_bwd: null argument");
  if (E->hB <= 0 || !E->imn) return fail(E, MVLPT_ERR_STATE, "logits_bwd: call logits_fwd first");
  hipStream_t s = (hipStream_t)stream;
  const int e = E->arch.embed_dim;
  ProfScope ps(E, s, PC_HEAD, 4.0 * E->hB * E->hC * e, 4.0 * ((double)E->hB * e + (double)E->hC * e + (double)E->hB * E->hC) * 2);
  HIPCHK(E, launch_logits_bwd(dlogits, E->imn, E->txn, E->inorm, E->tnorm, E->h_scale, E->h_lo, E->h_hi, dimg, dtxt, E->hB, E->hC, e, s));
  return 0;
}

int mvlpt_cross_entropy(void* h, const float* logits, const void* labels, int kind, int B, int C, float* loss, float* dlogits,
                        float* ncorrect, mvlpt_stream_t stream) {
  Engine* E = (Engine*)h;
  if (!E || !logits || !labels || !loss || B <= 0 || C <= 0 || (kind != 0 && kind != 1))
    return fail(E, MVLPT_ERR_ARG, "cross_entropy: null/invalid argument");
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(E, E->ce_ws.reserve((size_t)2 * B * 4 + 256));
  ProfScope ps(E, s, PC_HEAD, 8.0 * B * C, 4.0 * 3.0 * B * C);
  HIPCHK(E, launch_cross_entropy(logits, labels, kind, B, C, (float*)E->ce_ws.p, loss, dlogits, ncorrect, s));
  return 0;
}

// ------------------------------------------------------------------------------------------------ kernel-level ops
#define OPCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { g_create_err = std::string(#call) + ": " + hipGetErrorString(_e); return MVLPT_ERR_HIP; } } while (0)

int mvlpt_op_gemm(int dtype, int epi, const void* A, const void* Bt, int M, int N, int K, const float* bias, const void* aux,
                  const float* resid, void* out, void* out2, mvlpt_stream_t stream) {
  GemmArgs g{A, Bt, M, N, K, bias, aux, resid, out, out2};
#ifdef MVLPT_GEMM_TRACE
  // debug builds: row pitches of A / Bt from the environment (tools/pitch_probe.py allocates the operands that wide)
  if (getenv("MVLPT_DBG_LDA")) g.lda = atoi(getenv("MVLPT_DBG_LDA"));
  if (getenv("MVLPT_DBG_LDB")) g.ldb = atoi(getenv("MVLPT_DBG_LDB"));
  // debug builds: timeline of workgroup 0 -> $MVLPT_GEMM_TRACE_FILE (8 waves x 2048 int64 records)
  const char* path = getenv("MVLPT_GEMM_TRACE_FILE");
  long long* tr = nullptr;
  const size_t tr_bytes = 16 * 2048 * sizeof(long long);
  if (path) { OPCHK(hipMalloc(&tr, tr_bytes)); OPCHK(hipMemsetAsync(tr, 0, tr_bytes, (hipStream_t)stream)); g.trace = tr; }
#endif
  OPCHK(launch_gemm(dtype, epi, g, (hipStream_t)stream));
#ifdef MVLPT_GEMM_TRACE
  if (path) {
    std::vector<long long> h(16 * 2048);
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    OPCHK(hipMemcpy(h.data(), tr, tr_bytes, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 1, tr_bytes, f); fclose(f); }
    (void)hipFree(tr);
  }
#endif
  return 0;
}
int mvlpt_op_gemm_split(int dtype, int epi, const void* A, const void* Bt, int M, int N, int K, const float* bias, const void* aux,
                        const float* resid, void* out, void* out2, mvlpt_stream_t stream) {
  GemmArgs g{A, Bt, M, N, K, bias, aux, resid, out, out2};
  g.a_split = 1;
  OPCHK(launch_gemm(dtype, epi, g, (hipStream_t)stream));
  return 0;
}
// ---- mixed pair (GemmArgs::a_split == 2): [hi (cols x 16 bit) | residual bytes (cols, e5m2) | unused], pitch 2*cols elements
int mvlpt_op_pack_weight_mixed(int dtype, const float* w32, int rows, int cols, int transposed, void* out, int* w8_exp,
                               mvlpt_stream_t stream) {
  if (!w32 || !out || !w8_exp || rows <= 0 || cols <= 0) { g_create_err = "pack_weight_mixed: null/invalid argument"; return MVLPT_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const int R = transposed ? cols : rows, K = transposed ? rows : cols, ld = K + K / 2;
  float* sc = nullptr;
  OPCHK(hipMalloc(&sc, 16));
  if (transposed) OPCHK(launch_pack_weight_t(dtype, w32, out, rows, cols, s, ld));
  else OPCHK(launch_pack_weight(dtype, w32, out, rows, cols, ld, s));
  OPCHK(launch_grad_scale(w32, (size_t)rows * cols, 128.0f, sc, s));
  OPCHK(launch_pack_weight8(w32, (uint8_t*)out + (size_t)K * 2, rows, cols, transposed, K, (size_t)ld * 2, sc, s));
  float h = 1.0f;
  OPCHK(hipMemcpyAsync(&h, sc, 4, hipMemcpyDeviceToHost, s));
  OPCHK(hipStreamSynchronize(s));
  (void)hipFree(sc);
  int e = 0; (void)frexpf(h, &e);
  *w8_exp = e - 1; (void)R;
  return 0;
}
int mvlpt_op_gemm_mixed(int dtype, int epi, const void* A, const void* Bt, int ldb, int w8_exp, int M, int N, int K, const float* bias,
                        const void* aux, const float* resid, void* out, void* out2, mvlpt_stream_t stream) {
  GemmArgs g{A, Bt, M, N, K, bias, aux, resid, out, out2};
  g.a_split = 2; g.ldb = ldb; g.w8_exp = w8_exp;
  g.out_lo8 = (epi == EPI_GELU_SPLIT || epi == EPI_GELUBWD_SPLIT) ? 1 : 0;
  OPCHK(launch_gemm(dtype, epi, g, (hipStream_t)stream));
  return 0;
}
// ---- LayerNorm folding, kernel level (the same launches block_fwd makes)
int mvlpt_op_fold_vectors(int dtype, const void* W16, int ld, const float* gamma, const float* beta, const float* b, float* colsum,
                          float* bias2, int N, int K, mvlpt_stream_t stream) {
  OPCHK(launch_fold_vectors(dtype, W16, ld, gamma, beta, b, colsum, bias2, N, K, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_gemm_ln_producer(int dtype, const void* A, int a_split, const void* Bt, int ldb, int w8_exp, int M, int N, int K,
                              const float* bias, const float* resid, const float* gamma, int x16_split, float* out32, void* x16,
                              float* part, int ntp, int* nt, mvlpt_stream_t stream) {
  GemmArgs g{A, Bt, M, N, K, bias, nullptr, resid, out32, nullptr};
  g.a_split = a_split; g.ldb = ldb; g.w8_exp = w8_exp;
  g.ln_gamma = gamma; g.ln_x16 = x16; g.ln_split = x16_split; g.ln_part = part; g.ln_ntp = ntp;
  const int bn = gemm_tile_n(dtype, EPI_RESID32_LN, g, (hipStream_t)stream);
  if (bn <= 0 || N % bn || N / 128 > ntp) { g_create_err = "gemm_ln_producer: ntp smaller than N / 128"; return MVLPT_ERR_ARG; }
  if (nt) *nt = N / 128;
  OPCHK(launch_gemm(dtype, EPI_RESID32_LN, g, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_gemm_folded(int dtype, int epi, const void* A16, int a_split, const void* Bt, int ldb, int w8_exp, int M, int N, int K,
                         const float* colsum, const float* bias2, const float* part, int ntp, int nt, void* out, void* out2,
                         mvlpt_stream_t stream) {
  GemmArgs g{A16, Bt, M, N, K, bias2, nullptr, nullptr, out, out2};
  g.a_split = a_split; g.ldb = ldb; g.w8_exp = w8_exp;
  g.out_lo8 = (a_split == 2 && epi == EPI_GELU_SPLIT) ? 1 : 0;
  g.fold_part = part; g.fold_colsum = colsum; g.fold_ntp = ntp; g.fold_nt = nt;
  OPCHK(launch_gemm(dtype, epi, g, (hipStream_t)stream));
  return 0;
}
// ---- packed residual stream, kernel level (the same launches block_fwd_packed makes)
int mvlpt_op_fold_weight(const void* W16, int ld, const float* gamma, void* Wg16, int ldg, float* colsum, int N, int K, mvlpt_stream_t stream) {
  OPCHK(launch_fold_weight(W16, ld, gamma, Wg16, ldg, colsum, N, K, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_respk_pack(const float* x, void* hi, uint8_t* lo, float* part, int ntp, int rows, int d, mvlpt_stream_t stream) {
  OPCHK(launch_respk_pack_rows(x, hi, lo, part, ntp, rows, d, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_assemble_packed(const float* patch_emb, const float* cls, const float* pos, const float* ln_g, const float* ln_b, void* hi,
                             uint8_t* lo, float* part, int ntp, int batch, int grid2, int d, mvlpt_stream_t stream) {
  OPCHK(launch_assemble_tokens_packed(patch_emb, cls, pos, ln_g, ln_b, hi, lo, part, ntp, batch, grid2, d, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_respk_unpack(const void* hi, const uint8_t* lo, int row_mul, float* out, int rows, int d, mvlpt_stream_t stream) {
  OPCHK(launch_respk_unpack_rows(hi, lo, row_mul, out, rows, d, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_gemm_residp(const void* A, const void* Bt, int ldb, int M, int N, int K, const float* bias, const void* hi_in,
                         const uint8_t* lo_in, void* hi_out, uint8_t* lo_out, float* part, int ntp, int* nt, mvlpt_stream_t stream) {
  GemmArgs g{A, Bt, M, N, K, bias, nullptr, nullptr, hi_out, nullptr};
  g.ldb = ldb;
  g.rp_hi_in = hi_in; g.rp_lo_in = lo_in; g.rp_lo_out = lo_out; g.ln_part = part; g.ln_ntp = ntp;
  const int bn = gemm_tile_n(DT_F16, EPI_RESIDP_LN, g, (hipStream_t)stream);
  if (bn <= 0 || N % bn || N / 128 > ntp) { g_create_err = "gemm_residp: ntp smaller than N / 128"; return MVLPT_ERR_ARG; }
  if (nt) *nt = N / 128;
  OPCHK(launch_gemm(DT_F16, EPI_RESIDP_LN, g, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_cast_mixed(int dtype, const float* in, void* out, int64_t rows, int d, mvlpt_stream_t stream) {
  OPCHK(launch_cast_f32_split(dtype, in, out, (size_t)rows, d, nullptr, (hipStream_t)stream, 1));
  return 0;
}
int mvlpt_op_layernorm_fwd_mixed(int out_dtype, const float* x, const float* gamma, const float* beta, void* y, int rows, int d,
                                 mvlpt_stream_t stream) {
  if (out_dtype == DT_F32) { g_create_err = "layernorm_fwd_mixed: 16-bit output only"; return MVLPT_ERR_ARG; }
  LnFwdArgs a{x, nullptr, 1, gamma, beta, y, rows, d};
  a.split = 2;
  OPCHK(launch_ln_fwd(out_dtype, a, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_layernorm_bwd_mixed(int dtype, const void* dy, const float* x, const float* gamma, const float* resid, float* out32,
                                 void* out16, int rows, int d, mvlpt_stream_t stream) {
  LnBwdArgs a{dy, DT_F32, x, nullptr, 1, gamma, resid, out32, out16, rows, d};
  a.split = 2;
  OPCHK(launch_ln_bwd(dtype, a, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_attention32_fwd_mixed(int dtype, const void* qkv, void* out, float* lse, int N, int L, int H, int causal, int q_rows,
                                   mvlpt_stream_t stream) {
  Attn32Args a{qkv, out, lse, N, L, H, causal, q_rows};
  a.out_lo8 = 1;
  OPCHK(launch_attn32_fwd(dtype, a, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_attention32_bwd_mixed(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                                   void* dqkv, int N, int L, int H, int causal, mvlpt_stream_t stream) {
  Attn32BwdArgs a{qkv, out, dout, lse, delta, dqkv, N, L, H, causal};
  a.lo8 = 1;
  OPCHK(launch_attn32_bwd(dtype, a, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_layernorm_fwd_split(int out_dtype, const float* x, const float* gamma, const float* beta, void* y, int rows, int d,
                                 mvlpt_stream_t stream) {
  if (out_dtype == DT_F32) { g_create_err = "layernorm_fwd_split: 16-bit output only"; return MVLPT_ERR_ARG; }
  LnFwdArgs a{x, nullptr, 1, gamma, beta, y, rows, d};
  a.split = 1;
  OPCHK(launch_ln_fwd(out_dtype, a, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_layernorm_bwd_split(int dtype, const void* dy, const float* x, const float* gamma, const float* resid, float* out32,
                                 void* out16, int rows, int d, mvlpt_stream_t stream) {
  LnBwdArgs a{dy, DT_F32, x, nullptr, 1, gamma, resid, out32, out16, rows, d};
  a.split = 1;
  OPCHK(launch_ln_bwd(dtype, a, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_attention32_fwd(int dtype, const void* qkv, void* out, float* lse, int N, int L, int H, int causal, int q_rows,
                             mvlpt_stream_t stream) {
  Attn32Args a{qkv, out, lse, N, L, H, causal, q_rows};
#ifdef MVLPT_ATTN_TRACE
  // debug builds: timeline of one workgroup -> $MVLPT_ATTN_TRACE_FILE (8 waves x 256 int64 records)
  const char* path = getenv("MVLPT_ATTN_TRACE_FILE");
  long long* tr = nullptr;
  const size_t tr_bytes = 8 * 256 * sizeof(long long);
  if (path) { OPCHK(hipMalloc(&tr, tr_bytes)); OPCHK(hipMemsetAsync(tr, 0, tr_bytes, (hipStream_t)stream)); a.trace = tr; }
#endif
  OPCHK(launch_attn32_fwd(dtype, a, (hipStream_t)stream));
#ifdef MVLPT_ATTN_TRACE
  if (path) {
    std::vector<long long> hbuf(8 * 256);
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    OPCHK(hipMemcpy(hbuf.data(), tr, tr_bytes, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(path, "wb")) { fwrite(hbuf.data(), 1, tr_bytes, f); fclose(f); }
    (void)hipFree(tr);
  }
#endif
  return 0;
}
int mvlpt_op_attention32_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                             void* dqkv, int N, int L, int H, int causal, mvlpt_stream_t stream) {
  Attn32BwdArgs a{qkv, out, dout, lse, delta, dqkv, N, L, H, causal};
#ifdef MVLPT_ATTN_TRACE
  const char* path = getenv("MVLPT_ATTN_TRACE_FILE");
  long long* tr = nullptr;
  const size_t tr_bytes = 8 * 256 * sizeof(long long);
  if (path) { OPCHK(hipMalloc(&tr, tr_bytes)); OPCHK(hipMemsetAsync(tr, 0, tr_bytes, (hipStream_t)stream)); a.trace = tr; }
#endif
  OPCHK(launch_attn32_bwd(dtype, a, (hipStream_t)stream));
#ifdef MVLPT_ATTN_TRACE
  if (path) {
    std::vector<long long> hbuf(8 * 256);
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    OPCHK(hipMemcpy(hbuf.data(), tr, tr_bytes, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(path, "wb")) { fwrite(hbuf.data(), 1, tr_bytes, f); fclose(f); }
    (void)hipFree(tr);
  }
#endif
  return 0;
}
int mvlpt_op_layernorm_fwd(int out_dtype, const float* x, const float* gamma, const float* beta, void* y, int rows, int d,
                           mvlpt_stream_t stream) {
  LnFwdArgs a{x, nullptr, 1, gamma, beta, y, rows, d};
  OPCHK(launch_ln_fwd(out_dtype, a, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_layernorm_bwd(int dtype, const void* dy, const float* x, const float* gamma, const float* resid, float* out32,
                           void* out16, int rows, int d, mvlpt_stream_t stream) {
  LnBwdArgs a{dy, dtype, x, nullptr, 1, gamma, resid, out32, out16, rows, d};
  OPCHK(launch_ln_bwd(dtype, a, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_attention_fwd(int dtype, const void* qkv, void* out, float* lse, int N, int L, int H, int causal,
                           mvlpt_stream_t stream) {
  AttnArgs a{qkv, out, lse, N, L, H, causal};
  OPCHK(launch_attn_fwd(dtype, a, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_attention_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                           void* dqkv, int N, int L, int H, int causal, mvlpt_stream_t stream) {
  AttnBwdArgs a{qkv, out, dout, lse, delta, dqkv, N, L, H, causal};
  OPCHK(launch_attn_bwd(dtype, a, (hipStream_t)stream));
  return 0;
}
int mvlpt_op_cast(int dtype, const float* in, void* out, int64_t n, mvlpt_stream_t stream) {
  OPCHK(launch_cast_f32_to16(dtype, in, out, (size_t)n, nullptr, (hipStream_t)stream));
  return 0;
}

// ------------------------------------------------------------------------------------------------ input pipeline
int mvlpt_preprocess(void* h, const uint8_t* src, int64_t src_bytes, const MvlptImageDesc* descs, int B, int out_h, int out_w,
                     const float* mean, const float* stdv, void* out, int out_dtype, uint8_t* out_u8, mvlpt_stream_t stream) {
  static_assert(sizeof(MvlptImageDesc) == sizeof(PpDesc), "descriptor layouts must match");
  Engine* E = (Engine*)h;
  if (!E) return MVLPT_ERR_ARG;
  if (!src || !descs || B <= 0 || out_h <= 0 || out_w <= 0 || !mean || !stdv || (!out && !out_u8))
    return fail(E, MVLPT_ERR_ARG, "preprocess: null pointer or empty batch / output");
  if (out && out_dtype != DT_F32 && out_dtype != DT_F16 && out_dtype != DT_BF16) return fail(E, MVLPT_ERR_ARG, "preprocess: bad out_dtype");
  int max_crop_h = 0, ks_max = 1;
  for (int b = 0; b < B; ++b) {
    const MvlptImageDesc& d = descs[b];
    const bool ok = d.height > 0 && d.width > 0 && d.offset >= 0 && d.offset + (int64_t)d.height * d.width * 3 <= src_bytes &&
                    d.crop_top >= 0 && d.crop_left >= 0 && d.crop_height > 0 && d.crop_width > 0 &&
                    d.crop_top + d.crop_height <= d.height && d.crop_left + d.crop_width <= d.width && d.resize_height > 0 &&
                    d.resize_width > 0 && d.out_top >= 0 && d.out_left >= 0 && d.out_top + out_h <= d.resize_height &&
                    d.out_left + out_w <= d.resize_width && (d.flip == 0 || d.flip == 1);
    if (!ok) return fail(E, MVLPT_ERR_ARG, "preprocess: descriptor " + std::to_string(b) + " is inconsistent (box / window outside the image, or image outside src)");
    max_crop_h = std::max(max_crop_h, (int)d.crop_height);
    const double sh = std::max(1.0, (double)d.crop_height / d.resize_height), sw = std::max(1.0, (double)d.crop_width / d.resize_width);
    ks_max = std::max(ks_max, std::max((int)std::ceil(2.0 * sh), (int)std::ceil(2.0 * sw)) * 2 + 1);
  }
  hipStream_t s = (hipStream_t)stream;
  const int n_max = std::max(out_h, out_w);
  const size_t table_stride = (size_t)(2 + ks_max) * n_max;                          // int32 per (image, pass)
  const size_t tmp_stride = align256((size_t)max_crop_h * out_w * 3);
  const size_t desc_bytes = align256(sizeof(PpDesc) * (size_t)B), table_bytes = align256(table_stride * 4 * 2 * (size_t)B);
  HIPCHK(E, E->pp_ws.reserve(desc_bytes + table_bytes + tmp_stride * (size_t)B));
  char* base = (char*)E->pp_ws.p;
  PpDesc* descs_dev = (PpDesc*)base;
  int32_t* tables = (int32_t*)(base + desc_bytes);
  uint8_t* tmp = (uint8_t*)(base + desc_bytes + table_bytes);
  HIPCHK(E, hipMemcpyAsync(descs_dev, descs, sizeof(PpDesc) * (size_t)B, hipMemcpyHostToDevice, s));
  HIPCHK(E, launch_preprocess(src, descs_dev, B, max_crop_h, ks_max, out_h, out_w, tables, table_stride, tmp, tmp_stride, mean, stdv,
                              out, out_dtype, out_u8, s));
  return 0;
}

// ------------------------------------------------------------------------------------------------ profiling
int mvlpt_profile_begin(void* h, int all_kernels) {
  Engine* E = (Engine*)h;
  if (!E) return MVLPT_ERR_ARG;
  E->prof.clear(); E->ev_used = 0; E->prof_on = true; E->prof_all = all_kernels != 0;
  return 0;
}
int mvlpt_profile_pause(void* h, int paused) {
  Engine* E = (Engine*)h;
  if (!E) return MVLPT_ERR_ARG;
  E->prof_on = !paused;
  return 0;
}
int mvlpt_profile_end(void* h, MvlptKernelStat* stats, int max_stats) {
  Engine* E = (Engine*)h;
  if (!E || !stats || max_stats <= 0) return MVLPT_ERR_ARG;
  E->prof_on = false;
  MvlptKernelStat acc[PC_COUNT];
  for (int i = 0; i < PC_COUNT; ++i) { memset(&acc[i], 0, sizeof(acc[i])); snprintf(acc[i].name, sizeof(acc[i].name), "%s", kProfNames[i]); }
  // ... and the GEMM launches once more per problem (M, N, K, epilogue, operand format, folded consumer): name "g<M>x<N>x<K> e<epi> s<split> f<fold>"
  std::map<std::array<int, 6>, MvlptKernelStat> shapes;
  std::vector<std::pair<float, float>> iv[PC_COUNT];     // [start, end] in ms relative to the first recorded event
  hipEvent_t ref = E->prof.empty() ? nullptr : E->prof.front().a;
  for (const ProfRec& r : E->prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f, t0 = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    acc[r.cls].launches += 1; acc[r.cls].ms += ms; acc[r.cls].flops += r.flops; acc[r.cls].bytes += r.bytes;
    acc[r.cls].flops_executed += r.flops_exec;
    if (r.cls == PC_GEMM && r.M > 0) {
      MvlptKernelStat& q = shapes[{r.M, r.N, r.K, r.epi, r.split, r.fold}];
      if (!q.launches) { memset(&q, 0, sizeof(q)); snprintf(q.name, sizeof(q.name), "g%dx%dx%d e%d s%d f%d", r.M, r.N, r.K, r.epi, r.split, r.fold); }
      q.launches += 1; q.ms += ms; q.busy_ms += ms; q.flops += r.flops; q.bytes += r.bytes; q.flops_executed += r.flops_exec;
    }
    if (r.a == ref || hipEventElapsedTime(&t0, ref, r.a) == hipSuccess) iv[r.cls].push_back({t0, t0 + ms});
  }
  for (int c = 0; c < PC_COUNT; ++c) {                    // union of the intervals per class
    std::sort(iv[c].begin(), iv[c].end());
    double busy = 0.0; float cs = 0.f, ce = -1.f;
    for (auto& p : iv[c]) {
      if (ce < 0.f) { cs = p.first; ce = p.second; }
      else if (p.first > ce) { busy += ce - cs; cs = p.first; ce = p.second; }
      else if (p.second > ce) ce = p.second;
    }
    if (ce >= 0.f) busy += ce - cs;
    acc[c].busy_ms = busy;
  }
  int n = 0;
  for (int i = 0; i < PC_COUNT && n < max_stats; ++i) if (acc[i].launches) stats[n++] = acc[i];
  std::vector<MvlptKernelStat> by_time;
  for (auto& kv : shapes) by_time.push_back(kv.second);
  std::sort(by_time.begin(), by_time.end(), [](const MvlptKernelStat& x, const MvlptKernelStat& y) { return x.ms > y.ms; });
  for (size_t i = 0; i < by_time.size() && n < max_stats; ++i) stats[n++] = by_time[i];
  E->prof.clear(); E->ev_used = 0;
  return n;
}

}  // extern "C"
