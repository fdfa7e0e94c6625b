// Attention core of the SPLIT-PRECISION mode (head_dim 64) for the towers that carry a gradient (DESIGN.md "Precision
// modes"): nn.MultiheadAttention core of clip/model.py:181-183 and its backward at ~fp32 accuracy.
//
// Why it exists: with 16-bit Q, K, V, P, dO, dS the prompt gradients of a 12-layer tower differ from the reference's
// fp32 CPU path by 2-4e-3 (every operand rounding contributes ~3e-4); the GEMMs get ~22-bit activations from hi+lo
// operand pairs (GemmArgs::a_split) and so does the attention core.
//
// Data flow: every activation is a PAIR of 16-bit values, hi = round16(x), lo = round16(x - hi), stored [rows, 2*cols] =
// [hi(cols) | lo(cols)].  qkv pair [N*L, 6d] (QKV GEMM, EPI_STORE_SPLIT) -> O pair [N*L, 2d] (the split A operand of the
// out-projection GEMM) + lse;  backward: dO pair [N*L, 2d] -> dqkv pair [N*L, 6d].
//
// Arithmetic: a product of two pairs keeps the three terms  hi*hi + hi*lo + lo*hi  on the 16-bit MFMA with fp32
// accumulation (the dropped lo*lo is 2^-22 of the product); softmax, P, dS are fp32 registers that are split in place
// right where the fast-mode kernels round them.
//
// Structure of the streamed kernels (fwd, dQ, dK/dV): a workgroup of 4 waves owns 64 rows (queries, or keys in the
// dK/dV kernel), 16 per wave, whose operands stay in registers; the other side is streamed in 64-row chunks: four
// swizzled 16-bit LDS images per chunk (hi and lo of two matrices; the layout of attention.hip: 128-byte rows, 16-byte
// chunk c at c ^ (row & 7)) filled by LDS-DMA straight from the pair tensors — staging costs no VALU work.  Products
// use swapped operands so that a lane holds S[own row = lane&15][streamed row = 16*ct + 4*(lane>>4) + r]: softmax
// statistics are lane-local plus two shuffles, and those registers are directly the B operand of the second product
// (P.V, dS.K, P^T.dO, dS^T.Q) — no LDS round trip.  Transposed operands come from the hardware transpose read (frag_vt).
// Workgroups of one head run on the same XCD (its K/V stay in that L2).
#include <cstdlib>
#include "attn_common.h"

namespace mvlpt {

// Debug timeline of one workgroup of the streamed forward (tools/attn_trace.py; builds with -DMVLPT_ATTN_TRACE only)
#ifdef MVLPT_ATTN_TRACE
constexpr int ATR_MAX = 256;
#define MVLPT_ATR(p) do { if (a.trace && blockIdx.x == 1024 && lane == 0 && atr_n < ATR_MAX) \
    a.trace[wave * ATR_MAX + atr_n++] = ((long long)(p) << 56) | ((long long)__builtin_amdgcn_s_memtime() & 0xffffffffffffffLL); } while (0)
#define MVLPT_ATRB(p) do { if (a.trace && blockIdx.x == 5 && blockIdx.y == (gridDim.y >> 1) && lane == 0 && atr_n < ATR_MAX) \
    a.trace[wave * ATR_MAX + atr_n++] = ((long long)(p) << 56) | ((long long)__builtin_amdgcn_s_memtime() & 0xffffffffffffffLL); } while (0)
#else
#define MVLPT_ATR(p) do { } while (0)
#define MVLPT_ATRB(p) do { } while (0)
#endif

namespace {
constexpr int CH = 64;                 // streamed chunk / rows per workgroup
constexpr int XIMG = CH * 128;         // one 64 x 64 16-bit image
constexpr float SCALE = 0.125f;        // 1/sqrt(64)

// 16-bit pair store of 4 consecutive values: hi at p, lo at p + lo_off
template <typename T>
__device__ __forceinline__ void store_pair4(T* p, size_t lo_off, f32x4 v) {
  typename Vec<T>::v4 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) { T h, l; split16<T>(v[e], h, l); hi[e] = h; lo[e] = l; }
  *(typename Vec<T>::v4*)p = hi;
  *(typename Vec<T>::v4*)(p + lo_off) = lo;
}
// ... or, for a tensor whose only other reader is a GEMM (attention output, dQ/dK/dV), the mixed pair of
// GemmArgs::a_split == 2: hi at p, the four residual bytes (e5m2, common.h) at byte 2*plane + col of the row, where
// `col` is the column of p inside its `plane`-wide hi block
template <typename T>
__device__ __forceinline__ void store_pair4_m(T* p, size_t plane, int col, f32x4 v, int lo8) {
  if (lo8) {
    typename Vec<T>::v4 hi;
    const uint32_t w = split_lo8x4<T>(v, hi);
    *(typename Vec<T>::v4*)p = hi;
    *(uint32_t*)((char*)(p - col) + 2 * plane + col) = w;
  } else store_pair4<T>(p, plane, v);
}
template <typename T>
__device__ __forceinline__ f32x4 load_pair4_m(const T* p, size_t plane, int col, int lo8) {
  if (!lo8) {
    const typename Vec<T>::v4 hi = *(const typename Vec<T>::v4*)p, lo = *(const typename Vec<T>::v4*)(p + plane);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = to_f32<T>(hi[e]) + to_f32<T>(lo[e]);
    return r;
  }
  const typename Vec<T>::v4 hi = *(const typename Vec<T>::v4*)p;
  const uint32_t w = *(const uint32_t*)((const char*)(p - col) + 2 * plane + col);
  return f32x4{join_lo8<T>(hi[0], w, 0), join_lo8<T>(hi[1], w, 1), join_lo8<T>(hi[2], w, 2), join_lo8<T>(hi[3], w, 3)};
}
template <typename T>
__device__ __forceinline__ f32x4 load_pair4(const T* p, size_t lo_off) {
  const typename Vec<T>::v4 hi = *(const typename Vec<T>::v4*)p, lo = *(const typename Vec<T>::v4*)(p + lo_off);
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = to_f32<T>(hi[e]) + to_f32<T>(lo[e]);      // exact: <= 22 significant bits
  return r;
}
// This lane's part of delta = rowsum(dO * O) of one (row, head): dO as the own-row fragments (gh, gl: elements 32 ks + 8 fg + e, the
// registers phase A multiplies with), O read in the same layout (16-bit plane + 16-bit or e5m2 residual plane) — half the load
// instructions of a 4-column walk, and no second read of dO.  quad_sum() of the result is the row's delta.
template <typename T>
__device__ __forceinline__ float delta_part(const T* orow_head, size_t d, int col_head, int fg, int lo8,
                                            const typename Vec<T>::v8 (&gh)[2], const typename Vec<T>::v8 (&gl)[2]) {
  using v8 = typename Vec<T>::v8;
  float acc = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = ks * 32 + fg * 8;
    const v8 oh = *(const v8*)(orow_head + c);
    float o[8];
    if (!lo8) {
      const v8 ol = *(const v8*)(orow_head + d + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = to_f32<T>(oh[e]) + to_f32<T>(ol[e]);
    } else {
      const uint2 w = *(const uint2*)((const char*)(orow_head - col_head) + 2 * d + col_head + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = join_lo8<T>(oh[e], w.x, e); o[e + 4] = join_lo8<T>(oh[e + 4], w.y, e); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += o[e] * (to_f32<T>(gh[ks][e]) + to_f32<T>(gl[ks][e]));
  }
  return acc;
}

// head-chunk workgroup order: consecutive block ids go round-robin over the 8 XCDs, so the chunks of one head take ids
// that are 8 apart (same XCD, same L2)
__device__ __forceinline__ bool map_block(int nheads, int nchunks, int& nh, int& ch) {
  const int b = blockIdx.x, x = b & 7, r = b >> 3;
  nh = (r / nchunks) * 8 + x;
  ch = r % nchunks;
  return nh < nheads;
}
static inline int stream_grid(int nheads, int nchunks) { return ((nheads + 7) / 8) * 8 * nchunks; }

// DMA rows [r0, r0+64) of a pair matrix into the hi and lo LDS images of a ring slot (8 slabs of 8 rows per image, dealt
// round-robin to the NW waves; rows past the end re-read the last row: finite, always masked)
template <typename T, int NW>
__device__ __forceinline__ void stage_pair(char* hi, char* lo, const T* src, size_t lo_off, size_t ld, int r0, int rows_total,
                                           int wave, int lane) {
  const int srow = lane >> 3, chunk = (lane & 7) ^ srow;
#pragma unroll
  for (int i = 0; i < 8 / NW + (8 % NW ? 1 : 0); ++i) {
    const int sl = wave + NW * i;
    if (8 % NW == 0 || sl < 8) {
      int row = r0 + sl * 8 + srow;
      row = row < rows_total ? row : rows_total - 1;
      const T* g = src + (size_t)row * ld + chunk * 8;
      dma_raw<16>(g, hi + sl * 1024);
      dma_raw<16>(g + lo_off, lo + sl * 1024);
    }
  }
}
template <int NW>
__device__ __forceinline__ void wait_prev_chunk() {       // all but the newest chunk's DMA (4 images x 8 slabs / NW waves) landed
  if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}
// own rows (two 16-row tiles per wave) as B operands of the 16x16x32 MFMA: dims ks*32 + 8*fg + 0..7, ks = 0, 1
template <typename T>
__device__ __forceinline__ void load_own_pair(typename Vec<T>::v8 (&h)[2], typename Vec<T>::v8 (&l)[2], const T* row, size_t lo_off, int fg) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    h[ks] = *(const typename Vec<T>::v8*)(row + ks * 32 + fg * 8);
    l[ks] = *(const typename Vec<T>::v8*)(row + lo_off + ks * 32 + fg * 8);
  }
}
// acc[t][ct]: lane holds [own row = tile t, fr][streamed = 16ct + 4fg + r]; every LDS fragment feeds both own tiles
template <typename T>
__device__ __forceinline__ void mm_rows3(f32x4 (&acc)[2][4], const char* hi, const char* lo, const typename Vec<T>::v8 (&oh)[2][2],
                                         const typename Vec<T>::v8 (&ol)[2][2], int fr, int fg) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const typename Vec<T>::v8 ah = frag_rows<T>(hi, ct, ks, fr, fg), al = frag_rows<T>(lo, ct, ks, fr, fg);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t][ct] = mfma16<T>(ah, oh[t][ks], acc[t][ct]);
        acc[t][ct] = mfma16<T>(ah, ol[t][ks], acc[t][ct]);
        acc[t][ct] = mfma16<T>(al, oh[t][ks], acc[t][ct]);
      }
    }
}
// out[t][dt][r'] (own = tile t, fr; dim = 16dt + 4fg + r') += sum_streamed p[t][own][streamed] * C[streamed][dim]
template <typename T>
__device__ __forceinline__ void mm_accum3(f32x4 (&out)[2][4], const char* hi, const char* lo, const f32x4 (&p)[2][4], int fr, int fg) {
  using v8 = typename Vec<T>::v8;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    v8 ph[2], pl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        T x, y;
        split16<T>(p[t][2 * kb][e], x, y); ph[t][e] = x; pl[t][e] = y;
        split16<T>(p[t][2 * kb + 1][e], x, y); ph[t][e + 4] = x; pl[t][e + 4] = y;
      }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const v8 vh = frag_vt<T>(hi, kb, dt, fr, fg), vl = frag_vt<T>(lo, kb, dt, fr, fg);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        out[t][dt] = mfma16<T>(vh, ph[t], out[t][dt]);
        out[t][dt] = mfma16<T>(vh, pl[t], out[t][dt]);
        out[t][dt] = mfma16<T>(vl, ph[t], out[t][dt]);
      }
    }
  }
}
constexpr float LOG2E = 1.4426950408889634f;
constexpr float SC2 = SCALE * LOG2E;      // exp(s/8 - m/8) = exp2(s*SC2 - m*SC2)
}  // namespace

// ------------------------------------------------------------------------------------------------ streamed kernels
// NW waves x 32 own rows; ring of two slots x four 8 KiB images (64 KiB): chunk c+1 is in flight while chunk c is multiplied
template <typename T, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64) void attn32x_fwd_kernel(Attn32Args a, int nqc) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  using v8 = typename Vec<T>::v8;
  constexpr int WR = NW * 32;
  int nh, qcx;
  if (!map_block(a.N * a.H, nqc, nh, qcx)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
#ifdef MVLPT_ATTN_TRACE
  int atr_n = 0;
#endif
  MVLPT_ATR(10);
  const int n = nh / a.H, h = nh % a.H, L = a.L, d = a.H * 64;
  const size_t ld = 6 * (size_t)d, lo = 3 * (size_t)d;
  const T* base = (const T*)a.qkv_split + (size_t)n * L * ld + h * 64;
  const int qb = qcx * WR;
  const int kend = CAUSAL ? (qb + WR < L ? qb + WR : L) : L;
  const int nch = (kend + CH - 1) / CH;
  auto issue = [&](int c) {
    char* buf = sm + (c & 1) * 4 * XIMG;
    stage_pair<T, NW>(buf, buf + XIMG, base + d, lo, ld, c * CH, L, wave, lane);
    stage_pair<T, NW>(buf + 2 * XIMG, buf + 3 * XIMG, base + 2 * d, lo, ld, c * CH, L, wave, lane);
  };
  issue(0);
  int q[2];
  v8 Qh[2][2], Ql[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    q[t] = qb + wave * 32 + t * 16 + fr;
    load_own_pair<T>(Qh[t], Ql[t], base + (size_t)(q[t] < L ? q[t] : L - 1) * ld, lo, fg);
  }
  // the compiler's own wait for these loads sits here, not inside the loop (it does not see the asm-issued DMA)
  asm volatile("" ::"v"(Qh[0][0]), "v"(Qh[0][1]), "v"(Qh[1][0]), "v"(Qh[1][1]), "v"(Ql[0][0]), "v"(Ql[0][1]), "v"(Ql[1][0]), "v"(Ql[1][1]));
  MVLPT_ATR(11);
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};      // m in raw-score units; l is this lane's share of the row sum
  f32x4 O[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) O[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nch; ++c) {
    MVLPT_ATR(1);
    if (c > 0) __builtin_amdgcn_s_barrier();                    // every wave is done with the slot chunk c+1 overwrites
    MVLPT_ATR(2);
    if (c + 1 < nch) { issue(c + 1); MVLPT_ATR(3); wait_prev_chunk<NW>(); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MVLPT_ATR(4);
    __builtin_amdgcn_s_barrier();                               // all pieces of chunk c have landed
    MVLPT_ATR(5);
    const char* Kh = sm + (c & 1) * 4 * XIMG;
    const int k0 = c * CH;
    f32x4 S[2][4];
    mm_rows3<T>(S, Kh, Kh + XIMG, Qh, Ql, fr, fg);
#ifdef MVLPT_ATTN_TRACE
    asm volatile("" ::"v"(S[0][0]), "v"(S[1][3]));
#endif
    MVLPT_ATR(6);
    const bool edge = CAUSAL || k0 + CH > L;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float mx = -INFINITY;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (edge) {
            const int kk = k0 + 16 * ct + 4 * fg + r;
            if (!(kk < L && (!CAUSAL || kk <= q[t]))) S[t][ct][r] = -INFINITY;
          }
          mx = fmaxf(mx, S[t][ct][r]);
        }
      const float m_new = fmaxf(m[t], quad_max(mx));            // finite from chunk 0 on (key 0 is never masked)
      const float alpha = __builtin_amdgcn_exp2f((m[t] - m_new) * SC2), msc = m_new * SC2;
      m[t] = m_new;
      float sum = 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          S[t][ct][r] = __builtin_amdgcn_exp2f(fmaf(S[t][ct][r], SC2, -msc));
          sum += S[t][ct][r];
        }
      l[t] = l[t] * alpha + sum;
#pragma unroll
      for (int i = 0; i < 4; ++i) O[t][i] *= alpha;
    }
#ifdef MVLPT_ATTN_TRACE
    asm volatile("" ::"v"(S[0][0]), "v"(S[1][3]));
#endif
    MVLPT_ATR(7);
    mm_accum3<T>(O, Kh + 2 * XIMG, Kh + 3 * XIMG, S, fr, fg);
#ifdef MVLPT_ATTN_TRACE
    asm volatile("" ::"v"(O[0][0]), "v"(O[1][3]));
#endif
    MVLPT_ATR(8);
  }
  MVLPT_ATR(12);
  const int qlim = a.q_rows > 0 ? (a.q_rows < L ? a.q_rows : L) : L;
  // rows that are not computed get lse = +huge: a backward over the full sequence then sees P = 0 there (finite), not garbage
  if (a.lse && qlim < L && qcx == 0)
    for (int j = qlim + tid; j < L; j += NW * 64) a.lse[((size_t)n * a.H + h) * L + j] = 3.0e38f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float lt = quad_sum(l[t]);
    if (q[t] < qlim) {
      const float inv = 1.f / lt;
      T* orow = (T*)a.out_split + ((size_t)n * L + q[t]) * (2 * (size_t)d) + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store_pair4_m<T>(orow + 16 * dt + 4 * fg, d, h * 64 + 16 * dt + 4 * fg, O[t][dt] * inv, a.out_lo8);
      if (a.lse && fg == 0) a.lse[((size_t)n * a.H + h) * L + q[t]] = m[t] * SCALE + logf(lt);
    }
  }
  MVLPT_ATR(13);
}

// dQ (+ delta) of the NW*32 queries starting at qcx*NW*32 of head nh.  `stat_lds` non-null: -lse*log2(e) and delta of the
// own rows are published in LDS ([0] .. [lpad) and [lpad] .. [2*lpad)) for a dK/dV phase in the same workgroup instead of
// delta going through global memory.
template <typename T, bool CAUSAL, int NW>
__device__ __forceinline__ void attn32x_dq_phase(const Attn32BwdArgs& a, int nh, int qcx, char* sm, float* stat_lds, int lpad) {
  using v8 = typename Vec<T>::v8;
  constexpr int WR = NW * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
  const int n = nh / a.H, h = nh % a.H, L = a.L, d = a.H * 64;
  const size_t ld = 6 * (size_t)d, lo = 3 * (size_t)d;
  const T* base = (const T*)a.qkv_split + (size_t)n * L * ld + h * 64;
  const int qb = qcx * WR;
  const int kend = CAUSAL ? (qb + WR < L ? qb + WR : L) : L;
  const int nch = (kend + CH - 1) / CH;
  auto issue = [&](int c) {
    char* buf = sm + (c & 1) * 4 * XIMG;
    stage_pair<T, NW>(buf, buf + XIMG, base + d, lo, ld, c * CH, L, wave, lane);
    stage_pair<T, NW>(buf + 2 * XIMG, buf + 3 * XIMG, base + 2 * d, lo, ld, c * CH, L, wave, lane);
  };
  issue(0);
  int q[2];
  v8 Qh[2][2], Ql[2][2], Gh[2][2], Gl[2][2];
  float nlse[2], dl[2];       // -lse * log2(e);  delta = rowsum(dO * O)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    q[t] = qb + wave * 32 + t * 16 + fr;
    const int qc = q[t] < L ? q[t] : L - 1;
    const T* grow = (const T*)a.dout_split + ((size_t)n * L + qc) * (2 * (size_t)d) + h * 64;
    const T* orow = (const T*)a.out_split + ((size_t)n * L + qc) * (2 * (size_t)d) + h * 64;
    load_own_pair<T>(Qh[t], Ql[t], base + (size_t)qc * ld, lo, fg);
    load_own_pair<T>(Gh[t], Gl[t], grow, d, fg);
    const size_t stat = ((size_t)n * a.H + h) * L + qc;
    nlse[t] = -a.lse[stat] * LOG2E;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 o = load_pair4_m<T>(orow + 16 * k + 4 * fg, d, h * 64 + 16 * k + 4 * fg, a.lo8);
      const f32x4 g = load_pair4<T>(grow + 16 * k + 4 * fg, d);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += o[e] * g[e];
    }
    dl[t] = quad_sum(acc);
    if (stat_lds) { if (fg == 0 && q[t] < lpad) { stat_lds[q[t]] = nlse[t]; stat_lds[lpad + q[t]] = dl[t]; } }
    else if (fg == 0 && q[t] < L) a.delta[stat] = dl[t];
  }
  asm volatile("" ::"v"(Qh[0][0]), "v"(Qh[0][1]), "v"(Qh[1][0]), "v"(Qh[1][1]), "v"(Ql[0][0]), "v"(Ql[0][1]), "v"(Ql[1][0]), "v"(Ql[1][1]));
  asm volatile("" ::"v"(Gh[0][0]), "v"(Gh[0][1]), "v"(Gh[1][0]), "v"(Gh[1][1]), "v"(Gl[0][0]), "v"(Gl[0][1]), "v"(Gl[1][0]), "v"(Gl[1][1]));
  f32x4 dQ[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) dQ[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nch; ++c) {
    if (c > 0) __builtin_amdgcn_s_barrier();
    if (c + 1 < nch) { issue(c + 1); wait_prev_chunk<NW>(); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* Kh = sm + (c & 1) * 4 * XIMG;
    const int k0 = c * CH;
    f32x4 S[2][4], dP[2][4];
    mm_rows3<T>(S, Kh, Kh + XIMG, Qh, Ql, fr, fg);
    mm_rows3<T>(dP, Kh + 2 * XIMG, Kh + 3 * XIMG, Gh, Gl, fr, fg);
    const bool edge = CAUSAL || k0 + CH > L;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(S[t][ct][r], SC2, nlse[t]));
          if (edge) {
            const int kk = k0 + 16 * ct + 4 * fg + r;
            if (!(kk < L && (!CAUSAL || kk <= q[t]))) p = 0.f;
          }
          S[t][ct][r] = p * (dP[t][ct][r] - dl[t]);        // dS
        }
    mm_accum3<T>(dQ, Kh, Kh + XIMG, S, fr, fg);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (q[t] < L) {
      T* row = (T*)a.dqkv_split + ((size_t)n * L + q[t]) * ld + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store_pair4_m<T>(row + 16 * dt + 4 * fg, lo, h * 64 + 16 * dt + 4 * fg, dQ[t][dt] * SCALE, a.lo8);
    }
}

// dK, dV of the NW*32 keys starting at kcx*NW*32 of head nh.  `preload`: -lse*log2(e) and delta of the whole sequence are
// read from global into the LDS arrays behind the ring; otherwise a dQ phase of this workgroup has left them there.
template <typename T, bool CAUSAL, int NW>
__device__ __forceinline__ void attn32x_dkv_phase(const Attn32BwdArgs& a, int nh, int kcx, char* sm, int lpad, bool preload) {
  using v8 = typename Vec<T>::v8;
  constexpr int WR = NW * 32;
  float* nlse_s = (float*)(sm + 8 * XIMG);      // [lpad] -lse * log2(e)
  float* del_s = nlse_s + lpad;                 // [lpad]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
  const int n = nh / a.H, h = nh % a.H, L = a.L, d = a.H * 64;
  const size_t ld = 6 * (size_t)d, lo = 3 * (size_t)d, gld = 2 * (size_t)d;
  const T* base = (const T*)a.qkv_split + (size_t)n * L * ld + h * 64;
  const T* gbase = (const T*)a.dout_split + (size_t)n * L * gld + h * 64;
  const int kb = kcx * WR;
  const int c0 = CAUSAL ? kb / CH : 0, nch = (L + CH - 1) / CH;     // causal: queries before the first own key see none of them
  auto issue = [&](int c) {
    char* buf = sm + (c & 1) * 4 * XIMG;
    stage_pair<T, NW>(buf, buf + XIMG, base, lo, ld, c * CH, L, wave, lane);
    stage_pair<T, NW>(buf + 2 * XIMG, buf + 3 * XIMG, gbase, d, gld, c * CH, L, wave, lane);
  };
  issue(c0);
  const size_t stat0 = ((size_t)n * a.H + h) * L;
  if (preload)
    for (int j = tid; j < lpad; j += NW * 64) {
      const int jj = j < L ? j : L - 1;
      nlse_s[j] = -a.lse[stat0 + jj] * LOG2E;
      del_s[j] = a.delta[stat0 + jj];
    }
  int kk[2];
  v8 Kh[2][2], Kl[2][2], Vh[2][2], Vl[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    kk[t] = kb + wave * 32 + t * 16 + fr;
    const int kc = kk[t] < L ? kk[t] : L - 1;
    load_own_pair<T>(Kh[t], Kl[t], base + d + (size_t)kc * ld, lo, fg);
    load_own_pair<T>(Vh[t], Vl[t], base + 2 * d + (size_t)kc * ld, lo, fg);
  }
  asm volatile("" ::"v"(Kh[0][0]), "v"(Kh[0][1]), "v"(Kh[1][0]), "v"(Kh[1][1]), "v"(Kl[0][0]), "v"(Kl[0][1]), "v"(Kl[1][0]), "v"(Kl[1][1]));
  asm volatile("" ::"v"(Vh[0][0]), "v"(Vh[0][1]), "v"(Vh[1][0]), "v"(Vh[1][1]), "v"(Vl[0][0]), "v"(Vl[0][1]), "v"(Vl[1][0]), "v"(Vl[1][1]));
  f32x4 dK[2][4], dV[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) { dK[t][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[t][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int c = c0; c < nch; ++c) {
    if (c > c0) __builtin_amdgcn_s_barrier();
    if (c + 1 < nch) { issue(c + 1); wait_prev_chunk<NW>(); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // (also publishes nlse_s / del_s the first time round)
    const char* Qh = sm + (c & 1) * 4 * XIMG;
    const int q0 = c * CH;
    f32x4 S[2][4], dP[2][4];
    mm_rows3<T>(S, Qh, Qh + XIMG, Kh, Kl, fr, fg);          // lane: [key = tile t, fr][query = q0 + 16ct + 4fg + r]
    mm_rows3<T>(dP, Qh + 2 * XIMG, Qh + 3 * XIMG, Vh, Vl, fr, fg);
    const bool edge = CAUSAL || q0 + CH > L;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const f32x4 nl = *(const f32x4*)(nlse_s + q0 + 16 * ct + 4 * fg), de = *(const f32x4*)(del_s + q0 + 16 * ct + 4 * fg);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(S[t][ct][r], SC2, nl[r]));
          if (edge) {
            const int qq = q0 + 16 * ct + 4 * fg + r;
            if (!(qq < L && (!CAUSAL || kk[t] <= qq))) p = 0.f;
          }
          S[t][ct][r] = p;
          dP[t][ct][r] = p * (dP[t][ct][r] - de[r]);   // dS
        }
    }
    mm_accum3<T>(dV, Qh + 2 * XIMG, Qh + 3 * XIMG, S, fr, fg);
    mm_accum3<T>(dK, Qh, Qh + XIMG, dP, fr, fg);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (kk[t] < L) {
      T* row = (T*)a.dqkv_split + ((size_t)n * L + kk[t]) * ld + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store_pair4_m<T>(row + d + 16 * dt + 4 * fg, lo, d + h * 64 + 16 * dt + 4 * fg, dK[t][dt] * SCALE, a.lo8);
        store_pair4_m<T>(row + 2 * d + 16 * dt + 4 * fg, lo, 2 * d + h * 64 + 16 * dt + 4 * fg, dV[t][dt], a.lo8);
      }
    }
}

template <typename T, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64) void attn32x_dq_kernel(Attn32BwdArgs a, int nqc) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  int nh, qcx;
  if (!map_block(a.N * a.H, nqc, nh, qcx)) return;
  attn32x_dq_phase<T, CAUSAL, NW>(a, nh, qcx, sm, nullptr, 0);
}
template <typename T, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64) void attn32x_dkv_kernel(Attn32BwdArgs a, int nkc, int lpad) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  int nh, kcx;
  if (!map_block(a.N * a.H, nkc, nh, kcx)) return;
  attn32x_dkv_phase<T, CAUSAL, NW>(a, nh, kcx, sm, lpad, true);
}
// Sequences that fit ONE workgroup (L <= NW*32): dQ and dK/dV in one launch.  The second phase re-reads Q / dO (own rows of
// the first phase) and K / V (streamed by it) while they are still in this XCD's L2, lse comes from the first phase's
// registers and delta never goes through global memory: 1.29 GB instead of 1.93 GB of HBM traffic per ViT-B/16 layer.
template <typename T, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64) void attn32x_bwd_fused_kernel(Attn32BwdArgs a, int lpad) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  int nh, c;
  if (!map_block(a.N * a.H, 1, nh, c)) return;
  attn32x_dq_phase<T, CAUSAL, NW>(a, nh, 0, sm, (float*)(sm + 8 * XIMG), lpad);
  __syncthreads();           // every wave is done with the ring; the statistics of all rows are in LDS
  attn32x_dkv_phase<T, CAUSAL, NW>(a, nh, 0, sm, lpad, false);
}

// ------------------------------------------------------------------------------------------------ short sequences
// L <= 80 (every text sequence: context_length 77; the tiny test towers): ONE workgroup of 5 waves per (sequence, head),
// wave t owns the 16-row tile t; all of K, V (forward) or K, V then Q, dO (backward) are staged in LDS once by LDS-DMA, so
// the whole head costs two barriers instead of a staged chunk per 64 rows per pass, and the backward runs dQ and dK/dV in
// one launch.  100 classes x 8 heads, L = 77: forward 15 us, backward 49 us (the same structure on the f32-input MFMA,
// v_mfma_f32_16x16x4_f32 at 1/16 of the 16-bit rate, took 24 / 72 us; it lives in the history before this commit).
namespace {
constexpr int SNT = 5, SROWS = SNT * 16;           // tiles / padded rows
constexpr int TIMG = SROWS * 128;                  // one 80 x 64 16-bit image

// all SROWS rows of a pair matrix -> hi and lo images (10 slabs of 8 rows each, two per wave)
template <typename T>
__device__ __forceinline__ void stage_short(char* hi, char* lo, const T* src, size_t lo_off, size_t ld, int L, int wave, int lane) {
  const int srow = lane >> 3, chunk = (lane & 7) ^ srow;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int sl = wave + SNT * i;
    int row = sl * 8 + srow;
    row = row < L ? row : L - 1;
    const T* g = src + (size_t)row * ld + chunk * 8;
    dma_raw<16>(g, hi + sl * 1024);
    dma_raw<16>(g + lo_off, lo + sl * 1024);
  }
}
// one streamed 16-row tile against the own rows: lane holds [own = fr][streamed = 16*kt + 4fg + r]
template <typename T>
__device__ __forceinline__ f32x4 tile_rows3(const char* hi, const char* lo, int kt, const typename Vec<T>::v8 (&oh)[2],
                                            const typename Vec<T>::v8 (&ol)[2], int fr, int fg) {
  f32x4 acc[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const typename Vec<T>::v8 ah = frag_rows<T>(hi, kt, ks, fr, fg), al = frag_rows<T>(lo, kt, ks, fr, fg);
    acc[ks] = mfma16<T>(ah, oh[ks], f32x4{0.f, 0.f, 0.f, 0.f});
    acc[ks] = mfma16<T>(ah, ol[ks], acc[ks]);
    acc[ks] = mfma16<T>(al, oh[ks], acc[ks]);
  }
  return acc[0] + acc[1];
}
// out[dt] += sum over the 32 streamed rows of block kb of p * C; p0 / p1 are the block's two 16-row tiles (p1 must be 0
// where the tile does not exist); HALF: the block has only its first tile (kb = 2 of a 5-tile sequence)
template <typename T, bool HALF>
__device__ __forceinline__ void block_accum3(f32x4 (&out)[4], const char* hi, const char* lo, int kb, const f32x4& p0, const f32x4& p1,
                                             int fr, int fg) {
  using v8 = typename Vec<T>::v8;
  v8 ph, pl;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    T x, y;
    split16<T>(p0[e], x, y); ph[e] = x; pl[e] = y;
    split16<T>(p1[e], x, y); ph[e + 4] = x; pl[e + 4] = y;
  }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const v8 vh = HALF ? frag_vt_half<T>(hi, kb, dt, fr, fg) : frag_vt<T>(hi, kb, dt, fr, fg);
    const v8 vl = HALF ? frag_vt_half<T>(lo, kb, dt, fr, fg) : frag_vt<T>(lo, kb, dt, fr, fg);
    out[dt] = mfma16<T>(vh, ph, out[dt]);
    out[dt] = mfma16<T>(vh, pl, out[dt]);
    out[dt] = mfma16<T>(vl, ph, out[dt]);
  }
}
// p[0..4] (tile kt zero where unused) times the image: blocks 0, 1 full, block 2 = tile 4 alone
template <typename T>
__device__ __forceinline__ void accum_all3(f32x4 (&out)[4], const char* hi, const char* lo, const f32x4 (&p)[SNT], int t_lo, int t_hi,
                                           int fr, int fg) {
  // tiles t_lo .. t_hi-1 are non-zero (wave-uniform bounds)
  if (t_lo < 2 && t_hi > 0) block_accum3<T, false>(out, hi, lo, 0, p[0], p[1], fr, fg);
  if (t_lo < 4 && t_hi > 2) block_accum3<T, false>(out, hi, lo, 1, p[2], p[3], fr, fg);
  if (t_hi > 4) block_accum3<T, true>(out, hi, lo, 2, p[4], f32x4{0.f, 0.f, 0.f, 0.f}, fr, fg);
}
}  // namespace

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(SNT * 64) void attn32t_fwd_kernel(Attn32Args a) {
  __shared__ __attribute__((aligned(16))) char sm[4 * TIMG];
  char *Kh = sm, *Kl = sm + TIMG, *Vh = sm + 2 * TIMG, *Vl = sm + 3 * TIMG;
  using v8 = typename Vec<T>::v8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.y, h = blockIdx.x, L = a.L, d = a.H * 64;
  const size_t ld = 6 * (size_t)d, lo = 3 * (size_t)d;
  const T* base = (const T*)a.qkv_split + (size_t)n * L * ld + h * 64;
  stage_short<T>(Kh, Kl, base + d, lo, ld, L, wave, lane);
  stage_short<T>(Vh, Vl, base + 2 * d, lo, ld, L, wave, lane);
  const int nt = (L + 15) >> 4;
  const int qlim = a.q_rows > 0 ? (a.q_rows < L ? a.q_rows : L) : L;
  const int q = wave * 16 + fr, qc = q < L ? q : L - 1;
  v8 Qh[2], Ql[2];
  load_own_pair<T>(Qh, Ql, base + (size_t)qc * ld, lo, fg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (a.lse && qlim < L)          // rows that are not computed: lse = +huge (P = 0 in a backward over the full sequence)
    for (int j = qlim + tid; j < L; j += SNT * 64) a.lse[((size_t)n * a.H + h) * L + j] = 3.0e38f;
  if (wave * 16 >= qlim || wave >= nt) return;
  const int kt_end = CAUSAL ? wave + 1 : nt;
  f32x4 S[SNT];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < SNT; ++kt) {
    S[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kt < kt_end) {
      S[kt] = tile_rows3<T>(Kh, Kl, kt, Qh, Ql, fr, fg);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = 16 * kt + 4 * fg + r;
        const bool ok = kk < L && (!CAUSAL || kk <= q);
        S[kt][r] = ok ? S[kt][r] : -INFINITY;
        mx = fmaxf(mx, S[kt][r]);
      }
    }
  }
  mx = quad_max(mx);
  const float msc = mx * SC2;
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < SNT; ++kt)
    if (kt < kt_end) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { S[kt][r] = __builtin_amdgcn_exp2f(fmaf(S[kt][r], SC2, -msc)); sum += S[kt][r]; }
    }
  sum = quad_sum(sum);
  f32x4 O[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) O[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  accum_all3<T>(O, Vh, Vl, S, 0, kt_end, fr, fg);
  if (q < qlim) {
    const float inv = 1.f / sum;
    T* orow = (T*)a.out_split + ((size_t)n * L + q) * (2 * (size_t)d) + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store_pair4_m<T>(orow + 16 * dt + 4 * fg, d, h * 64 + 16 * dt + 4 * fg, O[dt] * inv, a.out_lo8);
    if (a.lse && fg == 0) a.lse[((size_t)n * a.H + h) * L + q] = mx * SCALE + logf(sum);
  }
}

// (4 waves per SIMD = 128 VGPRs, three workgroups per CU instead of the two that the unconstrained 136 allowed: 100 sequences
// 55.9 -> 51.2 us, 2191 sequences 1 192 -> 1 010 us per layer; the forward at 5 waves per SIMD (96 VGPRs, 3 spilled) got slower)
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(SNT * 64, 4) void attn32t_bwd_kernel(Attn32BwdArgs a) {
  // one set of four images (40 KiB, three workgroups per CU): K, V while the waves own query tiles (phase A: dQ), then Q, dO
  // while they own key tiles (phase B: dK, dV); the own rows live in registers in both phases.  (Eight images in flight
  // at once, 80 KiB, no restaging barrier: measured 20 % slower at 100 sequences, 4 % at 2191.)
  __shared__ __attribute__((aligned(16))) char sm[4 * TIMG];
  __shared__ float nlse_s[SROWS], del_s[SROWS];
  char *I0h = sm, *I0l = sm + TIMG, *I1h = sm + 2 * TIMG, *I1l = sm + 3 * TIMG;
  using v8 = typename Vec<T>::v8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.y, h = blockIdx.x, L = a.L, d = a.H * 64;
  const size_t ld = 6 * (size_t)d, lo = 3 * (size_t)d, gld = 2 * (size_t)d;
  const T* base = (const T*)a.qkv_split + (size_t)n * L * ld + h * 64;
  const T* gbase = (const T*)a.dout_split + (size_t)n * L * gld + h * 64;
#ifdef MVLPT_ATTN_TRACE
  int atr_n = 0;
#endif
  MVLPT_ATRB(40);
  stage_short<T>(I0h, I0l, base + d, lo, ld, L, wave, lane);          // K
  stage_short<T>(I1h, I1l, base + 2 * d, lo, ld, L, wave, lane);      // V
  const size_t stat0 = ((size_t)n * a.H + h) * L;
  const int nt = (L + 15) >> 4;
  const int row = wave * 16 + fr, rc = row < L ? row : L - 1;     // own row: query in phase A, key in phase B
  v8 Qh[2], Ql[2], Gh[2], Gl[2], Kh[2], Kl[2], Vh[2], Vl[2];
  load_own_pair<T>(Qh, Ql, base + (size_t)rc * ld, lo, fg);
  load_own_pair<T>(Gh, Gl, gbase + (size_t)rc * gld, d, fg);
  float dl = 0.f;       // delta = rowsum(dO * O) of the own query row
  {
    // dO of the own row is in registers already (Gh, Gl): O in the same fragment layout, no second read of dO
    const T* orow = (const T*)a.out_split + ((size_t)n * L + rc) * gld + h * 64;
    dl = delta_part<T>(orow, d, h * 64, fg, a.lo8, Gh, Gl);
    dl = quad_sum(dl);
  }
  const float nlse = -a.lse[stat0 + rc] * LOG2E;
  if (fg == 0) { nlse_s[row] = nlse; del_s[row] = dl; }
  MVLPT_ATRB(41);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MVLPT_ATRB(42);
  // the own key / value rows (phase B) out of the K / V images that have just landed — not a second time from memory (rows >= L of
  // an image hold row L-1, like the clamped global row); read before phase A, the images are restaged behind it
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    Kh[ks] = frag_rows<T>(I0h, wave, ks, fr, fg); Kl[ks] = frag_rows<T>(I0l, wave, ks, fr, fg);
    Vh[ks] = frag_rows<T>(I1h, wave, ks, fr, fg); Vl[ks] = frag_rows<T>(I1l, wave, ks, fr, fg);
  }
  asm volatile("" ::: "memory");
  const bool active = wave < nt;
  T* orow = (T*)a.dqkv_split + ((size_t)n * L + rc) * ld + h * 64;
  // ---- phase A: own query tile -> dQ
  if (active) {
    const int kt_end = CAUSAL ? wave + 1 : nt;
    f32x4 dS[SNT];
#pragma unroll
    for (int kt = 0; kt < SNT; ++kt) {
      dS[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < kt_end) {
        const f32x4 S = tile_rows3<T>(I0h, I0l, kt, Qh, Ql, fr, fg);
        const f32x4 dP = tile_rows3<T>(I1h, I1l, kt, Gh, Gl, fr, fg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = 16 * kt + 4 * fg + r;
          const bool ok = kk < L && (!CAUSAL || kk <= row);
          const float p = ok ? __builtin_amdgcn_exp2f(fmaf(S[r], SC2, nlse)) : 0.f;
          dS[kt][r] = p * (dP[r] - dl);
        }
      }
    }
    f32x4 dQ[4];
    MVLPT_ATRB(43);
#pragma unroll
    for (int i = 0; i < 4; ++i) dQ[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    accum_all3<T>(dQ, I0h, I0l, dS, 0, kt_end, fr, fg);
    MVLPT_ATRB(44);
    if (row < L) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store_pair4_m<T>(orow + 16 * dt + 4 * fg, lo, h * 64 + 16 * dt + 4 * fg, dQ[dt] * SCALE, a.lo8);
    }
  }
  MVLPT_ATRB(45);
  __syncthreads();
  MVLPT_ATRB(46);
  stage_short<T>(I0h, I0l, base, lo, ld, L, wave, lane);              // Q
  stage_short<T>(I1h, I1l, gbase, d, gld, L, wave, lane);             // dO
  MVLPT_ATRB(47);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MVLPT_ATRB(48);
  // ---- phase B: own key tile -> dK, dV
  if (active) {
    const int qt_lo = CAUSAL ? wave : 0;
    f32x4 P[SNT], dS[SNT];
#pragma unroll
    for (int qt = 0; qt < SNT; ++qt) {
      P[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dS[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (qt < nt && qt >= qt_lo) {
        const f32x4 S = tile_rows3<T>(I0h, I0l, qt, Kh, Kl, fr, fg);      // lane: [key = fr][query = 16qt + 4fg + r]
        const f32x4 dP = tile_rows3<T>(I1h, I1l, qt, Vh, Vl, fr, fg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = 16 * qt + 4 * fg + r;
          const bool ok = qq < L && (!CAUSAL || row <= qq);
          const float p = ok ? __builtin_amdgcn_exp2f(fmaf(S[r], SC2, nlse_s[qq])) : 0.f;
          P[qt][r] = p;
          dS[qt][r] = p * (dP[r] - del_s[qq]);
        }
      }
    }
    f32x4 dK[4], dV[4];
    MVLPT_ATRB(49);
#pragma unroll
    for (int i = 0; i < 4; ++i) { dK[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    accum_all3<T>(dV, I1h, I1l, P, qt_lo, nt, fr, fg);
    accum_all3<T>(dK, I0h, I0l, dS, qt_lo, nt, fr, fg);
    MVLPT_ATRB(50);
    if (row < L) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store_pair4_m<T>(orow + d + 16 * dt + 4 * fg, lo, d + h * 64 + 16 * dt + 4 * fg, dK[dt] * SCALE, a.lo8);
        store_pair4_m<T>(orow + 2 * d + 16 * dt + 4 * fg, lo, 2 * d + h * 64 + 16 * dt + 4 * fg, dV[dt], a.lo8);
      }
    }
  }
  MVLPT_ATRB(51);
}

// ------------------------------------------------------------------------------------------------ resident, medium sequences
// 80 < L <= 208, not causal (ViT-B/16 and ViT-L/14 at 224 px with their prompt tokens: 197 .. 205 and 257 is too long): the structure
// of the short kernels above at RNT = 13 tiles — ALL of K and V (forward) or K, V and then Q, dO (backward) of one (sequence, head)
// resident in LDS as four 208-row images (104 KiB), RNW = 7 waves that own two query (key) tiles each (tile w and tile w + 7), every
// product straight out of LDS with no barrier inside a phase: a head costs two barriers (forward) or four (backward) instead of
// eight / sixteen.  Measured at 256 x 12 heads, L = 205: forward 276 vs 292 us, backward 749 vs 785 us (same box, same run) — the
// chunk barriers and DMA waits that make up 44 % of a streamed head (tools/attn_trace.py) were a symptom: what remains is staging that
// nothing overlaps (one workgroup per CU) plus ~28 000 cycles of MFMA + VALU + LDS work per head for ~10 000 of MFMA issue.  Starting
// the second wave of each SIMD out of phase (s_sleep 640 .. 2 560 cycles) changes nothing either.
namespace {
constexpr int RNT = 13, RNW = 7, RROWS = RNT * 16, RIMG = RROWS * 128;
constexpr int RLDS = 4 * RIMG, RLDS_BWD = 4 * RIMG + 2 * RROWS * 4;

template <typename T>
__device__ __forceinline__ void stage_res(char* hi, char* lo, const T* src, size_t lo_off, size_t ld, int L, int wave, int lane) {
  const int srow = lane >> 3, chunk = (lane & 7) ^ srow;
  for (int sl = wave; sl < RROWS / 8; sl += RNW) {
    int row = sl * 8 + srow;
    row = row < L ? row : L - 1;
    const T* g = src + (size_t)row * ld + chunk * 8;
    dma_raw<16>(g, hi + sl * 1024);
    dma_raw<16>(g + lo_off, lo + sl * 1024);
  }
}
// p[0 .. RNT) (tiles outside [t_lo, t_hi) are zero) times the image: six full 32-row blocks and the 13th tile alone
template <typename T>
__device__ __forceinline__ void accum_res3(f32x4 (&out)[4], const char* hi, const char* lo, const f32x4 (&p)[RNT], int t_lo, int t_hi, int fr, int fg) {
#pragma unroll
  for (int kb = 0; kb < RNT / 2; ++kb)
    if (t_lo < 2 * kb + 2 && t_hi > 2 * kb) block_accum3<T, false>(out, hi, lo, kb, p[2 * kb], p[2 * kb + 1], fr, fg);
  if (t_hi > RNT - 1) block_accum3<T, true>(out, hi, lo, RNT / 2, p[RNT - 1], f32x4{0.f, 0.f, 0.f, 0.f}, fr, fg);
}
}  // namespace

template <typename T>
__global__ __launch_bounds__(RNW * 64) void attn32r_fwd_kernel(Attn32Args a) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char *Kh = sm, *Kl = sm + RIMG, *Vh = sm + 2 * RIMG, *Vl = sm + 3 * RIMG;
  using v8 = typename Vec<T>::v8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.y, h = blockIdx.x, L = a.L, d = a.H * 64;
  const size_t ld = 6 * (size_t)d, lo = 3 * (size_t)d;
  const T* base = (const T*)a.qkv_split + (size_t)n * L * ld + h * 64;
#if !defined(MVLPT_ABL) || MVLPT_ABL != 2      // ablation 2: no staging (compute on whatever LDS holds)
  stage_res<T>(Kh, Kl, base + d, lo, ld, L, wave, lane);
  stage_res<T>(Vh, Vl, base + 2 * d, lo, ld, L, wave, lane);
#endif
  const int nt = (L + 15) >> 4;
  const int qlim = a.q_rows > 0 ? (a.q_rows < L ? a.q_rows : L) : L;
  v8 Qh[2][2], Ql[2][2];
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int q = (wave + o * RNW) * 16 + fr;
    load_own_pair<T>(Qh[o], Ql[o], base + (size_t)(q < L ? q : L - 1) * ld, lo, fg);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (a.lse && qlim < L)          // rows that are not computed: lse = +huge (P = 0 in a backward over the full sequence)
    for (int j = qlim + tid; j < L; j += RNW * 64) a.lse[((size_t)n * a.H + h) * L + j] = 3.0e38f;
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int tile = wave + o * RNW;
    if (tile >= nt || tile * 16 >= qlim) continue;
    const int q = tile * 16 + fr;
    f32x4 S[RNT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < RNT; ++kt) {
      S[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#if defined(MVLPT_ABL) && MVLPT_ABL == 1       // ablation 1: no arithmetic (staging, own rows, stores only)
      if (false) {
#else
      if (kt < nt) {
#endif
        S[kt] = tile_rows3<T>(Kh, Kl, kt, Qh[o], Ql[o], fr, fg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = 16 * kt + 4 * fg + r;
          S[kt][r] = kk < L ? S[kt][r] : -INFINITY;
          mx = fmaxf(mx, S[kt][r]);
        }
      }
    }
    mx = quad_max(mx);
    const float msc = mx * SC2;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < RNT; ++kt)
      if (kt < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { S[kt][r] = __builtin_amdgcn_exp2f(fmaf(S[kt][r], SC2, -msc)); sum += S[kt][r]; }
      }
    sum = quad_sum(sum);
    f32x4 O[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) O[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#if !defined(MVLPT_ABL) || MVLPT_ABL != 1
    accum_res3<T>(O, Vh, Vl, S, 0, nt, fr, fg);
#endif
    if (q < qlim) {
      const float inv = 1.f / sum;
      T* orow = (T*)a.out_split + ((size_t)n * L + q) * (2 * (size_t)d) + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store_pair4_m<T>(orow + 16 * dt + 4 * fg, d, h * 64 + 16 * dt + 4 * fg, O[dt] * inv, a.out_lo8);
      if (a.lse && fg == 0) a.lse[((size_t)n * a.H + h) * L + q] = mx * SCALE + logf(sum);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(RNW * 64) void attn32r_bwd_kernel(Attn32BwdArgs a) {
  // phase A: K, V resident, the waves own query tiles -> dQ; phase B: Q, dO resident, the waves own key tiles -> dK, dV
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char *I0h = sm, *I0l = sm + RIMG, *I1h = sm + 2 * RIMG, *I1l = sm + 3 * RIMG;
  float* nlse_s = (float*)(sm + 4 * RIMG);
  float* del_s = nlse_s + RROWS;
  using v8 = typename Vec<T>::v8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.y, h = blockIdx.x, L = a.L, d = a.H * 64;
  const size_t ld = 6 * (size_t)d, lo = 3 * (size_t)d, gld = 2 * (size_t)d;
  const T* base = (const T*)a.qkv_split + (size_t)n * L * ld + h * 64;
  const T* gbase = (const T*)a.dout_split + (size_t)n * L * gld + h * 64;
#ifdef MVLPT_ATTN_TRACE
  int atr_n = 0;
#endif
  MVLPT_ATRB(20);
  stage_res<T>(I0h, I0l, base + d, lo, ld, L, wave, lane);          // K
  stage_res<T>(I1h, I1l, base + 2 * d, lo, ld, L, wave, lane);      // V
  const size_t stat0 = ((size_t)n * a.H + h) * L;
  const int nt = (L + 15) >> 4;
  float nlse[2], dl[2];       // of the own query rows: -lse * log2(e), delta = rowsum(dO * O)
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int row = (wave + o * RNW) * 16 + fr, rc = row < L ? row : L - 1;
    const T* orow = (const T*)a.out_split + ((size_t)n * L + rc) * gld + h * 64;
    v8 gh[2], gl[2];
    load_own_pair<T>(gh, gl, gbase + (size_t)rc * gld, d, fg);
    dl[o] = quad_sum(delta_part<T>(orow, d, h * 64, fg, a.lo8, gh, gl));
    nlse[o] = -a.lse[stat0 + rc] * LOG2E;
    if (fg == 0 && row < RROWS) { nlse_s[row] = nlse[o]; del_s[row] = dl[o]; }
  }
  MVLPT_ATRB(21);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MVLPT_ATRB(22);
  // ---- phase A: own query tiles -> dQ
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int tile = wave + o * RNW;
    if (tile >= nt) continue;
    const int row = tile * 16 + fr, rc = row < L ? row : L - 1;
    v8 Qh[2], Ql[2], Gh[2], Gl[2];
    load_own_pair<T>(Qh, Ql, base + (size_t)rc * ld, lo, fg);
    load_own_pair<T>(Gh, Gl, gbase + (size_t)rc * gld, d, fg);
    MVLPT_ATRB(23);
    f32x4 dS[RNT];
#pragma unroll
    for (int kt = 0; kt < RNT; ++kt) {
      dS[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < nt) {
        const f32x4 S = tile_rows3<T>(I0h, I0l, kt, Qh, Ql, fr, fg);
        const f32x4 dP = tile_rows3<T>(I1h, I1l, kt, Gh, Gl, fr, fg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = 16 * kt + 4 * fg + r;
          const float p = kk < L ? __builtin_amdgcn_exp2f(fmaf(S[r], SC2, nlse[o])) : 0.f;
          dS[kt][r] = p * (dP[r] - dl[o]);
        }
      }
    }
    MVLPT_ATRB(24);
    f32x4 dQ[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dQ[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    accum_res3<T>(dQ, I0h, I0l, dS, 0, nt, fr, fg);
    MVLPT_ATRB(25);
    if (row < L) {
      T* orow = (T*)a.dqkv_split + ((size_t)n * L + row) * ld + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store_pair4_m<T>(orow + 16 * dt + 4 * fg, lo, h * 64 + 16 * dt + 4 * fg, dQ[dt] * SCALE, a.lo8);
    }
    MVLPT_ATRB(26);
  }
  __syncthreads();
  MVLPT_ATRB(27);
  stage_res<T>(I0h, I0l, base, lo, ld, L, wave, lane);              // Q
  stage_res<T>(I1h, I1l, gbase, d, gld, L, wave, lane);             // dO
  MVLPT_ATRB(28);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MVLPT_ATRB(29);
  // ---- phase B: own key tiles -> dK, dV
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int tile = wave + o * RNW;
    if (tile >= nt) continue;
    const int row = tile * 16 + fr, rc = row < L ? row : L - 1;
    v8 Kh[2], Kl[2], Vh[2], Vl[2];
    load_own_pair<T>(Kh, Kl, base + d + (size_t)rc * ld, lo, fg);
    load_own_pair<T>(Vh, Vl, base + 2 * d + (size_t)rc * ld, lo, fg);
    MVLPT_ATRB(30);
    f32x4 P[RNT], dS[RNT];
#pragma unroll
    for (int qt = 0; qt < RNT; ++qt) {
      P[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dS[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (qt < nt) {
        const f32x4 S = tile_rows3<T>(I0h, I0l, qt, Kh, Kl, fr, fg);      // lane: [key = fr][query = 16qt + 4fg + r]
        const f32x4 dP = tile_rows3<T>(I1h, I1l, qt, Vh, Vl, fr, fg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = 16 * qt + 4 * fg + r;
          const float p = qq < L ? __builtin_amdgcn_exp2f(fmaf(S[r], SC2, nlse_s[qq])) : 0.f;
          P[qt][r] = p;
          dS[qt][r] = p * (dP[r] - del_s[qq]);
        }
      }
    }
    MVLPT_ATRB(31);
    f32x4 dK[4], dV[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dK[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    accum_res3<T>(dV, I1h, I1l, P, 0, nt, fr, fg);
    MVLPT_ATRB(32);
    accum_res3<T>(dK, I0h, I0l, dS, 0, nt, fr, fg);
    MVLPT_ATRB(33);
    if (row < L) {
      T* orow = (T*)a.dqkv_split + ((size_t)n * L + row) * ld + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store_pair4_m<T>(orow + d + 16 * dt + 4 * fg, lo, d + h * 64 + 16 * dt + 4 * fg, dK[dt] * SCALE, a.lo8);
        store_pair4_m<T>(orow + 2 * d + 16 * dt + 4 * fg, lo, 2 * d + h * 64 + 16 * dt + 4 * fg, dV[dt], a.lo8);
      }
    }
    MVLPT_ATRB(34);
  }
  MVLPT_ATRB(35);
}
// ------------------------------------------------------------------------------------------------ resident + persistent
// The resident forward above leaves its staging exposed: one workgroup per CU (104 KiB), so nothing overlaps a head's 104 KiB
// of LDS-DMA with another head's arithmetic (ablation: staging alone 112 of 264 us).  Here a workgroup WALKS the (sequence, head)
// list and the next head's operands land while the current one is multiplied: THREE pair buffers (hi + lo image, 52 KiB each,
// 156 KiB) in rotation — head i has K in buffer 2i mod 3 and V in 2i+1 mod 3; K of head i+1 is requested at the top of head i
// into the buffer V of head i-1 has just left, V of head i+1 behind the S phase of head i into K_i's buffer.  For that the two
// own tiles of a wave go through the S phase together (every K fragment feeds both: half the fragment reads), then — one barrier,
// K_i is dead — through softmax and P.V together.  Every wave issues the same number of DMA instructions per matrix (8: slabs
// past the 26th re-copy the last one) so the counted waits are compile-time constants: at the top of a head everything but the
// wave's own output stores of the previous head (vmcnt retires in order) has landed.
namespace {
constexpr int RBUF = 2 * RIMG;                 // one pair buffer: hi image, lo image
constexpr int RLDS_P = 3 * RBUF;
template <typename T>
__device__ __forceinline__ void stage_res8(char* buf, const T* src, size_t lo_off, size_t ld, int L, int wave, int lane) {
  const int srow = lane >> 3, chunk = (lane & 7) ^ srow;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int sl = wave + RNW * i;
    sl = sl < RROWS / 8 ? sl : RROWS / 8 - 1;
    int row = sl * 8 + srow;
    row = row < L ? row : L - 1;
    const T* g = src + (size_t)row * ld + chunk * 8;
    dma_raw<16>(g, buf + sl * 1024);
    dma_raw<16>(g + lo_off, buf + RIMG + sl * 1024);
  }
}
}  // namespace

// The K fragments of key tile kt + 1 are requested UNCONDITIONALLY in front of tile kt's MFMAs (tiles past the sequence still lie inside
// the image): with the request inside `if (kt < nt)` hipcc's wait-count pass merges the "issued / not issued" paths pessimistically and
// waits for the prefetched fragments in front of the current tile's MFMAs (lgkmcnt(3) .. (0) instead of (7) .. (4)): no overlap at all.
// (A compile-time tile count makes the phases straight-line code, which hipcc schedules into 60-80 spilled registers — spill traffic
// counts in vmcnt, its waits drain the prefetch DMA; leaving the tile loop with `break` turns S[][] into a dynamically indexed stack array.)
template <typename T>
__global__ __launch_bounds__(RNW * 64) void attn32p_fwd_kernel(Attn32Args a, int total) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  using v8 = typename Vec<T>::v8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
  const int L = a.L, d = a.H * 64, G = gridDim.x;
  const size_t ld = 6 * (size_t)d, lo = 3 * (size_t)d;
  const int nt = (L + 15) >> 4;
  const bool two = wave + RNW < nt;          // the wave's second tile exists
  auto head_base = [&](int item) -> const T* {
    const int n = item / a.H, h = item - n * a.H;
    return (const T*)a.qkv_split + (size_t)n * L * ld + h * 64;
  };
  auto load_q = [&](const T* base, v8 (&Qh)[2][2], v8 (&Ql)[2][2]) {
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int q = (wave + o * RNW) * 16 + fr;
      load_own_pair<T>(Qh[o], Ql[o], base + (size_t)(q < L ? q : L - 1) * ld, lo, fg);
    }
  };
  int item = blockIdx.x;
  if (item >= total) return;
  int kbuf = 0;
  v8 Qh[2][2], Ql[2][2];
  {
    const T* base = head_base(item);
    stage_res8<T>(sm, base + d, lo, ld, L, wave, lane);                  // K_0
    stage_res8<T>(sm + RBUF, base + 2 * d, lo, ld, L, wave, lane);       // V_0
    load_q(base, Qh, Ql);
  }
  for (bool first = true; item < total; item += G, first = false) {
    const int n = item / a.H, h = item - n * a.H;
    char* const Kb = sm + kbuf * RBUF;
    char* const Vb = sm + (kbuf + 1 >= 3 ? kbuf - 2 : kbuf + 1) * RBUF;
    char* const Nb = sm + (kbuf + 2 >= 3 ? kbuf - 1 : kbuf + 2) * RBUF;
    const int next = item + G;
    // everything this wave requested has landed, except (vmcnt retires in order) its output stores of the previous head
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (two) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // everybody's pieces of K_i, V_i are in; everybody is done with V_(i-1)
    // the compiler's own vmcnt waits for the Q fragments (ordinary loads it tracks) must sit HERE, in front of the DMA it does not see:
    // left to themselves they land inside the S phase (vmcnt(7) .. vmcnt(0) in front of every MFMA group) and drain the K_(i+1)
    // requests just issued — the whole prefetch would be waited for on the spot
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) asm volatile("" ::"v"(Qh[o][ks]), "v"(Ql[o][ks]));
    if (next < total) stage_res8<T>(Nb, head_base(next) + d, lo, ld, L, wave, lane);          // K_(i+1)
    // ---- S phase: both own tiles against every key tile; the K fragments of tile kt + 1 are requested behind the first MFMA of tile
    // kt (one fragment set in flight: a second buffer of 16 registers does not fit next to S[2][13] without spilling, and spill
    // traffic counts in vmcnt: its waits would drain the prefetch DMA)
    f32x4 S[2][RNT];
    v8 kf[4], kn[4];                           // ks * 2 + {hi, lo}: current / next tile
    auto load_k = [&](int kt, v8 (&f)[4]) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { f[ks * 2] = frag_rows<T>(Kb, kt, ks, fr, fg); f[ks * 2 + 1] = frag_rows<T>(Kb + RIMG, kt, ks, fr, fg); }
    };
    load_k(0, kf);
#pragma unroll
    for (int kt = 0; kt < RNT; ++kt) {
      S[0][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; S[1][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < RNT) load_k(kt + 1, kn);    // UNCONDITIONAL (a compile-time test): inside the image whatever L is
      __builtin_amdgcn_sched_barrier(0);
      if (kt < nt) {
        f32x4 acc[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int o = 0; o < 2; ++o) {        // (a wave without a second tile multiplies clamped rows: it would wait at the barrier otherwise)
            acc[o][ks] = mfma16<T>(kf[ks * 2], Qh[o][ks], f32x4{0.f, 0.f, 0.f, 0.f});
            acc[o][ks] = mfma16<T>(kf[ks * 2], Ql[o][ks], acc[o][ks]);
            acc[o][ks] = mfma16<T>(kf[ks * 2 + 1], Qh[o][ks], acc[o][ks]);
          }
        S[0][kt] = acc[0][0] + acc[0][1];
        S[1][kt] = acc[1][0] + acc[1][1];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) kf[q] = kn[q];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // K_i is dead
    if (next < total) {
      const T* nb = head_base(next);
      stage_res8<T>(Kb, nb + 2 * d, lo, ld, L, wave, lane);                                   // V_(i+1)
      load_q(nb, Qh, Ql);                                                                     // Q_(i+1): Q_i is dead too
    }
    const int qlim = a.q_rows > 0 ? (a.q_rows < L ? a.q_rows : L) : L;
    // ---- softmax of both tiles
    float mx[2], sum[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      mx[o] = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < RNT; ++kt)
        if (kt < nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kk = 16 * kt + 4 * fg + r;
            S[o][kt][r] = kk < L ? S[o][kt][r] : -INFINITY;
            mx[o] = fmaxf(mx[o], S[o][kt][r]);
          }
        }
      mx[o] = quad_max(mx[o]);
      const float msc = mx[o] * SC2;
      sum[o] = 0.f;
#pragma unroll
      for (int kt = 0; kt < RNT; ++kt)
        if (kt < nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { S[o][kt][r] = __builtin_amdgcn_exp2f(fmaf(S[o][kt][r], SC2, -msc)); sum[o] += S[o][kt][r]; }
        }
      sum[o] = quad_sum(sum[o]);
    }
    // ---- O = P V: every V^T fragment feeds both tiles
    f32x4 O[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int i = 0; i < 4; ++i) O[o][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (V^T fragments of the next column block requested unconditionally in front of this block's MFMAs, as in the S phase; a key
    // block whose second tile lies past the sequence meets P = 0 there and finite, clamped rows in the image)
    v8 vf[2], vn[2];
    auto load_v = [&](int kb, int dt, v8 (&f)[2]) {
      if (2 * kb + 1 >= RNT) { f[0] = frag_vt_half<T>(Vb, kb, dt, fr, fg); f[1] = frag_vt_half<T>(Vb + RIMG, kb, dt, fr, fg); }
      else { f[0] = frag_vt<T>(Vb, kb, dt, fr, fg); f[1] = frag_vt<T>(Vb + RIMG, kb, dt, fr, fg); }
    };
    load_v(0, 0, vf);
#pragma unroll
    for (int kb = 0; kb < (RNT + 1) / 2; ++kb) {
      v8 ph[2], pl[2];
      if (2 * kb < nt) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            T x, y;
            split16<T>(S[o][2 * kb][e], x, y); ph[o][e] = x; pl[o][e] = y;
            split16<T>(2 * kb + 1 < RNT ? S[o][2 * kb + 1 < RNT ? 2 * kb + 1 : 0][e] : 0.f, x, y); ph[o][e + 4] = x; pl[o][e + 4] = y;
          }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        __builtin_amdgcn_sched_barrier(0);
        if (dt + 1 < 4) load_v(kb, dt + 1, vn);
        else if (kb + 1 < (RNT + 1) / 2) load_v(kb + 1, 0, vn);
        __builtin_amdgcn_sched_barrier(0);
        if (2 * kb < nt) {
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            O[o][dt] = mfma16<T>(vf[0], ph[o], O[o][dt]);
            O[o][dt] = mfma16<T>(vf[0], pl[o], O[o][dt]);
            O[o][dt] = mfma16<T>(vf[1], ph[o], O[o][dt]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        vf[0] = vn[0]; vf[1] = vn[1];
      }
    }
    // ---- outputs (8 stores per existing tile + lse)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      if (o == 1 && !two) continue;
      const int q = (wave + o * RNW) * 16 + fr;
      const bool ok = q < qlim;
      const int qc = ok ? q : 0;
      const float inv = 1.f / sum[o];
      T* orow = (T*)a.out_split + ((size_t)n * L + qc) * (2 * (size_t)d) + h * 64;
      if (ok) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store_pair4_m<T>(orow + 16 * dt + 4 * fg, d, h * 64 + 16 * dt + 4 * fg, O[o][dt] * inv, a.out_lo8);
        if (a.lse && fg == 0) a.lse[((size_t)n * a.H + h) * L + q] = mx[o] * SCALE + logf(sum[o]);
      }
    }
    kbuf = kbuf + 2 >= 3 ? kbuf - 1 : kbuf + 2;
  }
}

static bool attn32_resident(int L, int causal) {
  static const bool on = !(getenv("MVLPT_ATTN32_RESIDENT") && atoi(getenv("MVLPT_ATTN32_RESIDENT")) == 0);
  return on && !causal && L <= RROWS;
}

// workgroup height: 8 waves (256 own rows) stream the other side fewer times; 4 waves (128 rows) pad less.
// MVLPT_ATTN32_NW = 4 / 8 forces one (experiments)
static int attn32_nw(int rows) {
  static const int forced = getenv("MVLPT_ATTN32_NW") ? atoi(getenv("MVLPT_ATTN32_NW")) : 0;
  if (forced == 4 || forced == 8) return forced;
  const int pad8 = (rows + 255) / 256 * 256, pad4 = (rows + 127) / 128 * 128;
  return pad8 * 8 <= pad4 * 9 ? 8 : 4;          // take 256-row workgroups unless they add more than 1/8 of padding
}
template <typename K>
static void set_lds(K kernel, int bytes) { (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }

template <typename T, bool CAUSAL, int NW>
static hipError_t fwd_x(const Attn32Args& a, hipStream_t s) {
  const int lq = a.q_rows > 0 ? (a.q_rows < a.L ? a.q_rows : a.L) : a.L;
  const int nqc = (lq + NW * 32 - 1) / (NW * 32);
  constexpr int lds = 8 * XIMG;
  static bool set = false;
  if (!set) { set_lds(attn32x_fwd_kernel<T, CAUSAL, NW>, lds); set = true; }
  hipLaunchKernelGGL((attn32x_fwd_kernel<T, CAUSAL, NW>), dim3(stream_grid(a.N * a.H, nqc)), dim3(NW * 64), lds, s, a, nqc);
  return hipGetLastError();
}
template <typename T, bool CAUSAL, int NW>
static hipError_t bwd_x(const Attn32BwdArgs& a, hipStream_t s) {
  const int nc = (a.L + NW * 32 - 1) / (NW * 32), lpad = (a.L + CH - 1) / CH * CH;
  constexpr int lds = 8 * XIMG;
  const int lds_kv = lds + 8 * lpad;
  static int set = 0;
  if (set < lds_kv) {
    set_lds(attn32x_dq_kernel<T, CAUSAL, NW>, lds); set_lds(attn32x_dkv_kernel<T, CAUSAL, NW>, lds_kv);
    set_lds(attn32x_bwd_fused_kernel<T, CAUSAL, NW>, lds_kv); set = lds_kv;
  }
  const dim3 grid(stream_grid(a.N * a.H, nc)), block(NW * 64);
  static const bool fuse = !(getenv("MVLPT_ATTN32_FUSED_BWD") && atoi(getenv("MVLPT_ATTN32_FUSED_BWD")) == 0);
  if (nc == 1 && fuse) {
    hipLaunchKernelGGL((attn32x_bwd_fused_kernel<T, CAUSAL, NW>), grid, block, lds_kv, s, a, lpad);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((attn32x_dq_kernel<T, CAUSAL, NW>), grid, block, lds, s, a, nc);
  hipLaunchKernelGGL((attn32x_dkv_kernel<T, CAUSAL, NW>), grid, block, lds_kv, s, a, nc, lpad);
  return hipGetLastError();
}

template <typename T>
static hipError_t fwd_t(const Attn32Args& a, hipStream_t s) {
  if (a.L <= SROWS) {
    dim3 grid(a.H, a.N), block(SNT * 64);
    if (a.causal) hipLaunchKernelGGL((attn32t_fwd_kernel<T, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn32t_fwd_kernel<T, false>), grid, block, 0, s, a);
    return hipGetLastError();
  }
  if (attn32_resident(a.L, a.causal)) {
    // persistent variant (next head's K / V land under this head's arithmetic): full-sequence launches with more heads than CUs
    static const bool persist = !(getenv("MVLPT_ATTN32_PERSIST") && atoi(getenv("MVLPT_ATTN32_PERSIST")) == 0);
    const int total = a.N * a.H, cus = stream_cus(s);
    if (persist && a.q_rows <= 0 && total >= 2 * cus) {
      static bool setp = false;
      if (!setp) { set_lds(attn32p_fwd_kernel<T>, RLDS_P); setp = true; }
      hipLaunchKernelGGL((attn32p_fwd_kernel<T>), dim3(cus), dim3(RNW * 64), RLDS_P, s, a, total);
      return hipGetLastError();
    }
    static bool set = false;
    if (!set) { set_lds(attn32r_fwd_kernel<T>, RLDS); set = true; }
    hipLaunchKernelGGL((attn32r_fwd_kernel<T>), dim3(a.H, a.N), dim3(RNW * 64), RLDS, s, a);
    return hipGetLastError();
  }
  const int lq = a.q_rows > 0 ? (a.q_rows < a.L ? a.q_rows : a.L) : a.L;
  if (attn32_nw(lq) == 8) return a.causal ? fwd_x<T, true, 8>(a, s) : fwd_x<T, false, 8>(a, s);
  return a.causal ? fwd_x<T, true, 4>(a, s) : fwd_x<T, false, 4>(a, s);
}
template <typename T>
static hipError_t bwd_t(const Attn32BwdArgs& a, hipStream_t s) {
  if (a.L <= SROWS) {
    dim3 grid(a.H, a.N), block(SNT * 64);
    if (a.causal) hipLaunchKernelGGL((attn32t_bwd_kernel<T, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn32t_bwd_kernel<T, false>), grid, block, 0, s, a);
    return hipGetLastError();
  }
  if (attn32_resident(a.L, a.causal)) {
    static bool set = false;
    if (!set) { set_lds(attn32r_bwd_kernel<T>, RLDS_BWD); set = true; }
    hipLaunchKernelGGL((attn32r_bwd_kernel<T>), dim3(a.H, a.N), dim3(RNW * 64), RLDS_BWD, s, a);
    return hipGetLastError();
  }
  if (a.L > 8192) return hipErrorInvalidValue;       // the dK/dV kernel keeps lse and delta of a whole sequence in LDS
  if (attn32_nw(a.L) == 8) return a.causal ? bwd_x<T, true, 8>(a, s) : bwd_x<T, false, 8>(a, s);
  return a.causal ? bwd_x<T, true, 4>(a, s) : bwd_x<T, false, 4>(a, s);
}

hipError_t launch_attn32_fwd(int dtype, const Attn32Args& a, hipStream_t s) {
  if (a.L <= 0 || a.N <= 0 || a.H <= 0 || !a.qkv_split || !a.out_split) return hipErrorInvalidValue;
  if (dtype == DT_F16) return fwd_t<f16>(a, s);
  if (dtype == DT_BF16) return fwd_t<bf16>(a, s);
  return hipErrorInvalidValue;
}
hipError_t launch_attn32_bwd(int dtype, const Attn32BwdArgs& a, hipStream_t s) {
  if (a.L <= 0 || a.N <= 0 || a.H <= 0 || !a.qkv_split || !a.out_split || !a.dout_split || !a.lse || !a.delta || !a.dqkv_split)
    return hipErrorInvalidValue;
  if (dtype == DT_F16) return bwd_t<f16>(a, s);
  if (dtype == DT_BF16) return bwd_t<bf16>(a, s);
  return hipErrorInvalidValue;
}

}  // namespace mvlpt
