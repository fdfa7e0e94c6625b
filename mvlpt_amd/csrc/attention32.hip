// fp32 multi-head attention (head_dim 64) on the f32-input MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32
// accumulation, 1/16 of the 16-bit MFMA rate) for the SPLIT-PRECISION mode of the towers that carry a gradient
// (DESIGN.md "Precision modes").  nn.MultiheadAttention core of clip/model.py:181-183 and its backward.
//
// Why it exists: with 16-bit Q, K, V, P, dO, dS the prompt gradients of a 12-layer tower differ from the reference's
// fp32 CPU path by 2-4e-3 (every operand rounding contributes ~3e-4); the GEMMs get ~22-bit activations from hi+lo
// operand pairs (GemmArgs::a_split), and the attention core — 4 % of the FLOPs — simply runs in fp32.
//
// Data flow: qkv32 [N*L, 3d] fp32 (QKV GEMM with the fp32-store epilogue) -> O as a 16-bit hi|lo pair [N*L, 2d] (the
// split A operand of the out-projection GEMM) + lse;  backward: dO32 [N*L, d] fp32 -> dqkv as a pair [N*L, 6d].
//
// Structure (all three kernels): a workgroup of 4 waves owns 64 rows (queries, or keys in the dK/dV kernel), 16 per
// wave, whose two operands stay in registers; the other side is streamed in 64-row chunks through LDS (fp32 rows padded
// to 68 floats: conflict-free for both access patterns).  Products use swapped operands so that a lane holds
// S[own row = lane&15][streamed row = 16*ct + 4*(lane>>4) + r]: softmax statistics are lane-local plus two shuffles,
// and those registers are directly the B operand of the second product (P.V, dS.K, P^T.dO, dS^T.Q) — no LDS round trip.
#include <cstdlib>
#include "attn_common.h"

namespace mvlpt {

namespace {
constexpr int RS = 68;                 // padded LDS row (floats)
constexpr int CH = 64;                 // streamed chunk / rows per workgroup
constexpr float SCALE = 0.125f;        // 1/sqrt(64)

__device__ __forceinline__ f32x4 mfma32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float quad_sum32(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float quad_max32(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

// rows [r0, r0+64) x 64 floats of a [*, ld] matrix -> LDS (rows >= rows_total re-read the last row: finite, always masked)
__device__ __forceinline__ void stage64(float* dst, const float* src, size_t ld, int r0, int rows_total, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 256 + tid, r = idx >> 4, c = (idx & 15) * 4;
    int row = r0 + r; row = row < rows_total ? row : rows_total - 1;
    *(f32x4*)(dst + r * RS + c) = *(const f32x4*)(src + (size_t)row * ld + c);
  }
}
// own-row operand fragment: X[row][16t + 4fg + s], t = 0..3 (register-resident for the whole kernel)
__device__ __forceinline__ void load_own(f32x4 (&reg)[4], const float* row_ptr, int fg) {
#pragma unroll
  for (int t = 0; t < 4; ++t) reg[t] = *(const f32x4*)(row_ptr + 16 * t + 4 * fg);
}
// acc[ct][r] = sum_dim C[16ct + (lane&15)][dim] * own[lane&15][dim]   ->  lane holds [own = fr][streamed = 16ct + 4fg + r]
// (the f32 MFMA issues every 32 cycles but a dependent one only after 40: the four column tiles are four INDEPENDENT
// accumulator chains, interleaved)
__device__ __forceinline__ void mm_rows(f32x4 (&acc)[4], const float* lds, const f32x4 (&own)[4], int fr, int fg) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f32x4 c[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) c[ct] = *(const f32x4*)(lds + (16 * ct + fr) * RS + 16 * t + 4 * fg);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma32(c[ct][s], own[t][s], acc[ct]);
  }
}
// out[dt][r'] (own = fr, dim = 16dt + 4fg + r') += sum_streamed p[own][streamed] * C[streamed][dim]
__device__ __forceinline__ void mm_accum(f32x4 (&out)[4], const float* lds, const f32x4 (&p)[4], int fr, int fg) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) out[dt] = mfma32(lds[(16 * ct + 4 * fg + r) * RS + 16 * dt + fr], p[ct][r], out[dt]);
}
// 16-bit pair store of 4 consecutive values: hi at p, lo at p + lo_off
template <typename T>
__device__ __forceinline__ void store_pair4(T* p, size_t lo_off, f32x4 v) {
  typename Vec<T>::v4 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) { T h, l; split16<T>(v[e], h, l); hi[e] = h; lo[e] = l; }
  *(typename Vec<T>::v4*)p = hi;
  *(typename Vec<T>::v4*)(p + lo_off) = lo;
}
template <typename T>
__device__ __forceinline__ f32x4 load_pair4(const T* p, size_t lo_off) {
  const typename Vec<T>::v4 hi = *(const typename Vec<T>::v4*)p, lo = *(const typename Vec<T>::v4*)(p + lo_off);
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = to_f32<T>(hi[e]) + to_f32<T>(lo[e]);
  return r;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ forward
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256) void attn32_fwd_kernel(Attn32Args a) {
  __shared__ __attribute__((aligned(16))) float Ks[CH * RS];
  __shared__ __attribute__((aligned(16))) float Vs[CH * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.z, h = blockIdx.y, L = a.L, d = a.H * 64;
  const size_t ld = 3 * (size_t)d;
  const float* base = a.qkv + (size_t)n * L * ld + h * 64;
  const int qb = blockIdx.x * CH, q = qb + wave * 16 + fr;
  const int qc = q < L ? q : L - 1;
  f32x4 Q[4];
  load_own(Q, base + (size_t)qc * ld, fg);
  float m = -INFINITY, l = 0.f;
  f32x4 O[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) O[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kend = CAUSAL ? (qb + CH < L ? qb + CH : L) : L;
  for (int k0 = 0; k0 < kend; k0 += CH) {
    __syncthreads();
    stage64(Ks, base + d, ld, k0, L, tid);
    stage64(Vs, base + 2 * d, ld, k0, L, tid);
    __syncthreads();
    f32x4 S[4];
    mm_rows(S, Ks, Q, fr, fg);
    float mx = -INFINITY;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = k0 + 16 * ct + 4 * fg + r;
        const bool ok = kk < L && (!CAUSAL || kk <= q);
        S[ct][r] = ok ? S[ct][r] * SCALE : -INFINITY;
        mx = fmaxf(mx, S[ct][r]);
      }
    const float m_new = fmaxf(m, quad_max32(mx));
    const float alpha = (m == -INFINITY) ? 0.f : __expf(m - m_new);
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        S[ct][r] = (m_new == -INFINITY) ? 0.f : __expf(S[ct][r] - m_new);
        sum += S[ct][r];
      }
    l = l * alpha + quad_sum32(sum);
    m = m_new;
#pragma unroll
    for (int i = 0; i < 4; ++i) O[i] *= alpha;
    mm_accum(O, Vs, S, fr, fg);
  }
  const int qlim = a.q_rows > 0 ? (a.q_rows < L ? a.q_rows : L) : L;
  if (q < qlim) {
    const float inv = 1.f / l;
    T* orow = (T*)a.out_split + ((size_t)n * L + q) * (2 * (size_t)d) + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store_pair4<T>(orow + 16 * dt + 4 * fg, d, O[dt] * inv);
    if (a.lse && fg == 0) a.lse[((size_t)n * a.H + h) * L + q] = m + logf(l);
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256) void attn32_dq_kernel(Attn32BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[CH * RS];
  __shared__ __attribute__((aligned(16))) float Vs[CH * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.z, h = blockIdx.y, L = a.L, d = a.H * 64;
  const size_t ld = 3 * (size_t)d;
  const float* base = a.qkv + (size_t)n * L * ld + h * 64;
  const int qb = blockIdx.x * CH, q = qb + wave * 16 + fr;
  const int qc = q < L ? q : L - 1;
  f32x4 Q[4], dO[4];
  load_own(Q, base + (size_t)qc * ld, fg);
  load_own(dO, a.dout32 + ((size_t)n * L + qc) * d + h * 64, fg);
  const size_t stat = ((size_t)n * a.H + h) * L + qc;
  const float lse = a.lse[stat];
  // delta = rowsum(dO * O)
  float dl = 0.f;
  {
    const T* orow = (const T*)a.out_split + ((size_t)n * L + qc) * (2 * (size_t)d) + h * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 o = load_pair4<T>(orow + 16 * t + 4 * fg, d);
#pragma unroll
      for (int e = 0; e < 4; ++e) dl += o[e] * dO[t][e];
    }
    dl = quad_sum32(dl);
    if (fg == 0 && q < L) a.delta[stat] = dl;
  }
  f32x4 dQ[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dQ[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kend = CAUSAL ? (qb + CH < L ? qb + CH : L) : L;
  for (int k0 = 0; k0 < kend; k0 += CH) {
    __syncthreads();
    stage64(Ks, base + d, ld, k0, L, tid);
    stage64(Vs, base + 2 * d, ld, k0, L, tid);
    __syncthreads();
    f32x4 S[4], dP[4];
    mm_rows(S, Ks, Q, fr, fg);
    mm_rows(dP, Vs, dO, fr, fg);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = k0 + 16 * ct + 4 * fg + r;
        const bool ok = kk < L && (!CAUSAL || kk <= q);
        const float p = ok ? __expf(S[ct][r] * SCALE - lse) : 0.f;
        S[ct][r] = p * (dP[ct][r] - dl);        // dS
      }
    mm_accum(dQ, Ks, S, fr, fg);
  }
  if (q < L) {
    T* row = (T*)a.dqkv_split + ((size_t)n * L + q) * (6 * (size_t)d) + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store_pair4<T>(row + 16 * dt + 4 * fg, 3 * (size_t)d, dQ[dt] * SCALE);
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256) void attn32_dkv_kernel(Attn32BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float Qs[CH * RS];
  __shared__ __attribute__((aligned(16))) float Gs[CH * RS];      // dO chunk
  __shared__ float lse_s[CH], del_s[CH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.z, h = blockIdx.y, L = a.L, d = a.H * 64;
  const size_t ld = 3 * (size_t)d;
  const float* base = a.qkv + (size_t)n * L * ld + h * 64;
  const float* dobase = a.dout32 + (size_t)n * L * d + h * 64;
  const int kb = blockIdx.x * CH, kk = kb + wave * 16 + fr;
  const int kc = kk < L ? kk : L - 1;
  f32x4 K[4], V[4];
  load_own(K, base + d + (size_t)kc * ld, fg);
  load_own(V, base + 2 * d + (size_t)kc * ld, fg);
  f32x4 dK[4], dV[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { dK[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const size_t stat0 = ((size_t)n * a.H + h) * L;
  for (int q0 = CAUSAL ? kb : 0; q0 < L; q0 += CH) {
    __syncthreads();
    stage64(Qs, base, ld, q0, L, tid);
    stage64(Gs, dobase, d, q0, L, tid);
    if (tid < CH) {
      int qq = q0 + tid; qq = qq < L ? qq : L - 1;
      lse_s[tid] = a.lse[stat0 + qq];
      del_s[tid] = a.delta[stat0 + qq];
    }
    __syncthreads();
    f32x4 S[4], dP[4];
    mm_rows(S, Qs, K, fr, fg);          // lane: [key = fr][query = q0 + 16ct + 4fg + r]
    mm_rows(dP, Gs, V, fr, fg);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * ct + 4 * fg + r, qq = q0 + j;
        const bool ok = qq < L && kk < L && (!CAUSAL || kk <= qq);
        const float p = ok ? __expf(S[ct][r] * SCALE - lse_s[j]) : 0.f;
        S[ct][r] = p;
        dP[ct][r] = p * (dP[ct][r] - del_s[j]);   // dS
      }
    mm_accum(dV, Gs, S, fr, fg);
    mm_accum(dK, Qs, dP, fr, fg);
  }
  if (kk < L) {
    T* row = (T*)a.dqkv_split + ((size_t)n * L + kk) * (6 * (size_t)d) + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      store_pair4<T>(row + d + 16 * dt + 4 * fg, 3 * (size_t)d, dK[dt] * SCALE);
      store_pair4<T>(row + 2 * d + 16 * dt + 4 * fg, 3 * (size_t)d, dV[dt]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ 16-bit MFMA, three terms
// Same three kernels with the products on the 16-bit MFMA (16x the f32-MFMA rate) at ~fp32 accuracy: every fp32 operand is a
// pair (hi, lo) of 16-bit values and a product keeps the three terms  hi*hi + hi*lo + lo*hi  (the dropped lo*lo is 2^-22
// of the product), accumulated in fp32.  The streamed side is split ONCE while it is staged (two swizzled 16-bit LDS images,
// the layout of attention.hip: 128-byte rows, 16-byte chunk c at c ^ (row & 7)); the own rows are split once into
// registers; P / dS are split in registers right where the fp16 kernels round them.  Transposed operands come from the
// hardware transpose read (frag_vt), exactly as in the 16-bit kernels.  L = 205: fwd 554 -> see DESIGN.md.
namespace {
constexpr int XIMG = CH * 128;          // one 64 x 64 16-bit image

// Staging is split in two so that the NEXT chunk's global loads fly while the current chunk is multiplied:
// fetch64 (global -> registers: 2 x 8 floats per thread and array) ... compute ... commit64_pair (split + LDS store).
struct Fetch64 { f32x4 v[4]; };
__device__ __forceinline__ void fetch64(Fetch64& f, const float* src, size_t ld, int r0, int rows_total, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = i * 256 + tid, r = idx >> 3, c = idx & 7;
    int row = r0 + r; row = row < rows_total ? row : rows_total - 1;
    f.v[2 * i] = *(const f32x4*)(src + (size_t)row * ld + c * 8);
    f.v[2 * i + 1] = *(const f32x4*)(src + (size_t)row * ld + c * 8 + 4);
  }
}
template <typename T>
__device__ __forceinline__ void commit64_pair(char* hi, char* lo, const Fetch64& f, int tid) {
  using v8 = typename Vec<T>::v8;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = i * 256 + tid, r = idx >> 3, c = idx & 7;
    v8 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      T x, y;
      split16<T>(f.v[2 * i][e], x, y); h[e] = x; l[e] = y;
      split16<T>(f.v[2 * i + 1][e], x, y); h[e + 4] = x; l[e + 4] = y;
    }
    const int off = r * 128 + ((c ^ (r & 7)) * 16);
    *(v8*)(hi + off) = h;
    *(v8*)(lo + off) = l;
  }
}
// own row as B operands of the 16x16x32 MFMA: dims ks*32 + 8*fg + 0..7, ks = 0, 1
template <typename T>
__device__ __forceinline__ void load_own_pair(typename Vec<T>::v8 (&h)[2], typename Vec<T>::v8 (&l)[2], const float* row_ptr, int fg) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const f32x4 a = *(const f32x4*)(row_ptr + ks * 32 + fg * 8), b = *(const f32x4*)(row_ptr + ks * 32 + fg * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      T x, y;
      split16<T>(a[e], x, y); h[ks][e] = x; l[ks][e] = y;
      split16<T>(b[e], x, y); h[ks][e + 4] = x; l[ks][e + 4] = y;
    }
  }
}
// lane holds [own = fr][streamed = 16ct + 4fg + r]
template <typename T>
__device__ __forceinline__ void mm_rows3(f32x4 (&acc)[4], const char* hi, const char* lo, const typename Vec<T>::v8 (&oh)[2],
                                         const typename Vec<T>::v8 (&ol)[2], int fr, int fg) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const typename Vec<T>::v8 ah = frag_rows<T>(hi, ct, ks, fr, fg), al = frag_rows<T>(lo, ct, ks, fr, fg);
      acc[ct] = mfma16<T>(ah, oh[ks], acc[ct]);
      acc[ct] = mfma16<T>(ah, ol[ks], acc[ct]);
      acc[ct] = mfma16<T>(al, oh[ks], acc[ct]);
    }
}
// out[dt][r'] (own = fr, dim = 16dt + 4fg + r') += sum_streamed p[own][streamed] * C[streamed][dim]
template <typename T>
__device__ __forceinline__ void mm_accum3(f32x4 (&out)[4], const char* hi, const char* lo, const f32x4 (&p)[4], int fr, int fg) {
  using v8 = typename Vec<T>::v8;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    v8 ph, pl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      T x, y;
      split16<T>(p[2 * kb][e], x, y); ph[e] = x; pl[e] = y;
      split16<T>(p[2 * kb + 1][e], x, y); ph[e + 4] = x; pl[e + 4] = y;
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const v8 vh = frag_vt<T>(hi, kb, dt, fr, fg), vl = frag_vt<T>(lo, kb, dt, fr, fg);
      out[dt] = mfma16<T>(vh, ph, out[dt]);
      out[dt] = mfma16<T>(vh, pl, out[dt]);
      out[dt] = mfma16<T>(vl, ph, out[dt]);
    }
  }
}
}  // namespace

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256) void attn32x_fwd_kernel(Attn32Args a) {
  __shared__ __attribute__((aligned(16))) char sm[4 * XIMG];
  char *Kh = sm, *Kl = sm + XIMG, *Vh = sm + 2 * XIMG, *Vl = sm + 3 * XIMG;
  using v8 = typename Vec<T>::v8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.z, h = blockIdx.y, L = a.L, d = a.H * 64;
  const size_t ld = 3 * (size_t)d;
  const float* base = a.qkv + (size_t)n * L * ld + h * 64;
  const int qb = blockIdx.x * CH, q = qb + wave * 16 + fr;
  const int qc = q < L ? q : L - 1;
  v8 Qh[2], Ql[2];
  load_own_pair<T>(Qh, Ql, base + (size_t)qc * ld, fg);
  float m = -INFINITY, l = 0.f;
  f32x4 O[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) O[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kend = CAUSAL ? (qb + CH < L ? qb + CH : L) : L;
  Fetch64 fk, fv;
  fetch64(fk, base + d, ld, 0, L, tid);
  fetch64(fv, base + 2 * d, ld, 0, L, tid);
  for (int k0 = 0; k0 < kend; k0 += CH) {
    __syncthreads();
    commit64_pair<T>(Kh, Kl, fk, tid);
    commit64_pair<T>(Vh, Vl, fv, tid);
    __syncthreads();
    if (k0 + CH < kend) {
      fetch64(fk, base + d, ld, k0 + CH, L, tid);
      fetch64(fv, base + 2 * d, ld, k0 + CH, L, tid);
    }
    f32x4 S[4];
    mm_rows3<T>(S, Kh, Kl, Qh, Ql, fr, fg);
    float mx = -INFINITY;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = k0 + 16 * ct + 4 * fg + r;
        const bool ok = kk < L && (!CAUSAL || kk <= q);
        S[ct][r] = ok ? S[ct][r] * SCALE : -INFINITY;
        mx = fmaxf(mx, S[ct][r]);
      }
    const float m_new = fmaxf(m, quad_max32(mx));
    const float alpha = (m == -INFINITY) ? 0.f : __expf(m - m_new);
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        S[ct][r] = (m_new == -INFINITY) ? 0.f : __expf(S[ct][r] - m_new);
        sum += S[ct][r];
      }
    l = l * alpha + quad_sum32(sum);
    m = m_new;
#pragma unroll
    for (int i = 0; i < 4; ++i) O[i] *= alpha;
    mm_accum3<T>(O, Vh, Vl, S, fr, fg);
  }
  const int qlim = a.q_rows > 0 ? (a.q_rows < L ? a.q_rows : L) : L;
  if (q < qlim) {
    const float inv = 1.f / l;
    T* orow = (T*)a.out_split + ((size_t)n * L + q) * (2 * (size_t)d) + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store_pair4<T>(orow + 16 * dt + 4 * fg, d, O[dt] * inv);
    if (a.lse && fg == 0) a.lse[((size_t)n * a.H + h) * L + q] = m + logf(l);
  }
}

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256) void attn32x_dq_kernel(Attn32BwdArgs a) {
  __shared__ __attribute__((aligned(16))) char sm[4 * XIMG];
  char *Kh = sm, *Kl = sm + XIMG, *Vh = sm + 2 * XIMG, *Vl = sm + 3 * XIMG;
  using v8 = typename Vec<T>::v8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.z, h = blockIdx.y, L = a.L, d = a.H * 64;
  const size_t ld = 3 * (size_t)d;
  const float* base = a.qkv + (size_t)n * L * ld + h * 64;
  const int qb = blockIdx.x * CH, q = qb + wave * 16 + fr;
  const int qc = q < L ? q : L - 1;
  const float* grow = a.dout32 + ((size_t)n * L + qc) * d + h * 64;
  v8 Qh[2], Ql[2], Gh[2], Gl[2];
  load_own_pair<T>(Qh, Ql, base + (size_t)qc * ld, fg);
  load_own_pair<T>(Gh, Gl, grow, fg);
  const size_t stat = ((size_t)n * a.H + h) * L + qc;
  const float lse = a.lse[stat];
  float dl = 0.f;      // delta = rowsum(dO * O), fp32 from global
  {
    const T* orow = (const T*)a.out_split + ((size_t)n * L + qc) * (2 * (size_t)d) + h * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 o = load_pair4<T>(orow + 16 * t + 4 * fg, d);
      const f32x4 g = *(const f32x4*)(grow + 16 * t + 4 * fg);
#pragma unroll
      for (int e = 0; e < 4; ++e) dl += o[e] * g[e];
    }
    dl = quad_sum32(dl);
    if (fg == 0 && q < L) a.delta[stat] = dl;
  }
  f32x4 dQ[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dQ[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kend = CAUSAL ? (qb + CH < L ? qb + CH : L) : L;
  Fetch64 fk, fv;
  fetch64(fk, base + d, ld, 0, L, tid);
  fetch64(fv, base + 2 * d, ld, 0, L, tid);
  for (int k0 = 0; k0 < kend; k0 += CH) {
    __syncthreads();
    commit64_pair<T>(Kh, Kl, fk, tid);
    commit64_pair<T>(Vh, Vl, fv, tid);
    __syncthreads();
    if (k0 + CH < kend) {
      fetch64(fk, base + d, ld, k0 + CH, L, tid);
      fetch64(fv, base + 2 * d, ld, k0 + CH, L, tid);
    }
    f32x4 S[4], dP[4];
    mm_rows3<T>(S, Kh, Kl, Qh, Ql, fr, fg);
    mm_rows3<T>(dP, Vh, Vl, Gh, Gl, fr, fg);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = k0 + 16 * ct + 4 * fg + r;
        const bool ok = kk < L && (!CAUSAL || kk <= q);
        const float p = ok ? __expf(S[ct][r] * SCALE - lse) : 0.f;
        S[ct][r] = p * (dP[ct][r] - dl);        // dS
      }
    mm_accum3<T>(dQ, Kh, Kl, S, fr, fg);
  }
  if (q < L) {
    T* row = (T*)a.dqkv_split + ((size_t)n * L + q) * (6 * (size_t)d) + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store_pair4<T>(row + 16 * dt + 4 * fg, 3 * (size_t)d, dQ[dt] * SCALE);
  }
}

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256) void attn32x_dkv_kernel(Attn32BwdArgs a) {
  __shared__ __attribute__((aligned(16))) char sm[4 * XIMG];
  __shared__ float lse_s[CH], del_s[CH];
  char *Qh = sm, *Ql = sm + XIMG, *Gh = sm + 2 * XIMG, *Gl = sm + 3 * XIMG;
  using v8 = typename Vec<T>::v8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.z, h = blockIdx.y, L = a.L, d = a.H * 64;
  const size_t ld = 3 * (size_t)d;
  const float* base = a.qkv + (size_t)n * L * ld + h * 64;
  const float* dobase = a.dout32 + (size_t)n * L * d + h * 64;
  const int kb = blockIdx.x * CH, kk = kb + wave * 16 + fr;
  const int kc = kk < L ? kk : L - 1;
  v8 Kh[2], Kl[2], Vh[2], Vl[2];
  load_own_pair<T>(Kh, Kl, base + d + (size_t)kc * ld, fg);
  load_own_pair<T>(Vh, Vl, base + 2 * d + (size_t)kc * ld, fg);
  f32x4 dK[4], dV[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { dK[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const size_t stat0 = ((size_t)n * a.H + h) * L;
  Fetch64 fq, fg_;
  fetch64(fq, base, ld, CAUSAL ? kb : 0, L, tid);
  fetch64(fg_, dobase, d, CAUSAL ? kb : 0, L, tid);
  for (int q0 = CAUSAL ? kb : 0; q0 < L; q0 += CH) {
    __syncthreads();
    commit64_pair<T>(Qh, Ql, fq, tid);
    commit64_pair<T>(Gh, Gl, fg_, tid);
    if (tid < CH) {
      int qq = q0 + tid; qq = qq < L ? qq : L - 1;
      lse_s[tid] = a.lse[stat0 + qq];
      del_s[tid] = a.delta[stat0 + qq];
    }
    __syncthreads();
    if (q0 + CH < L) {
      fetch64(fq, base, ld, q0 + CH, L, tid);
      fetch64(fg_, dobase, d, q0 + CH, L, tid);
    }
    f32x4 S[4], dP[4];
    mm_rows3<T>(S, Qh, Ql, Kh, Kl, fr, fg);          // lane: [key = fr][query = q0 + 16ct + 4fg + r]
    mm_rows3<T>(dP, Gh, Gl, Vh, Vl, fr, fg);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * ct + 4 * fg + r, qq = q0 + j;
        const bool ok = qq < L && kk < L && (!CAUSAL || kk <= qq);
        const float p = ok ? __expf(S[ct][r] * SCALE - lse_s[j]) : 0.f;
        S[ct][r] = p;
        dP[ct][r] = p * (dP[ct][r] - del_s[j]);   // dS
      }
    mm_accum3<T>(dV, Gh, Gl, S, fr, fg);
    mm_accum3<T>(dK, Qh, Ql, dP, fr, fg);
  }
  if (kk < L) {
    T* row = (T*)a.dqkv_split + ((size_t)n * L + kk) * (6 * (size_t)d) + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      store_pair4<T>(row + d + 16 * dt + 4 * fg, 3 * (size_t)d, dK[dt] * SCALE);
      store_pair4<T>(row + 2 * d + 16 * dt + 4 * fg, 3 * (size_t)d, dV[dt]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ short sequences
// L <= 80 (every text sequence: context_length 77; the tiny test towers): ONE workgroup of 5 waves per (sequence, head),
// wave t owns the 16-row tile t; all of K, V (forward) or Q, K, V, dO (backward) are staged in LDS once, so the whole
// head costs two barriers instead of a staged chunk per 64 rows per pass, and the backward runs dQ and dK/dV in one
// launch (100 classes x 8 heads x 12 layers of L = 77: 151 -> ~30 us per layer for the backward).
namespace {
constexpr int SNT = 5, SROWS = SNT * 16;           // tiles / padded rows

__device__ __forceinline__ void stage_rows(float* dst, const float* src, size_t ld, int L, int tid, int nthreads) {
  for (int idx = tid; idx < SROWS * 16; idx += nthreads) {
    const int r = idx >> 4, c = (idx & 15) * 4;
    const int row = r < L ? r : L - 1;
    *(f32x4*)(dst + r * RS + c) = *(const f32x4*)(src + (size_t)row * ld + c);
  }
}
// one 16-row tile of the streamed side: lane holds [own = fr][streamed = 16*kt + 4fg + r]
__device__ __forceinline__ f32x4 mm_tile(const float* lds, int kt, const f32x4 (&own)[4], int fr, int fg) {
  f32x4 a[4];                                     // four independent chains (one per 16-wide slice of the head dimension)
#pragma unroll
  for (int t = 0; t < 4; ++t) a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 c[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = *(const f32x4*)(lds + (16 * kt + fr) * RS + 16 * t + 4 * fg);
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = mfma32(c[t][s], own[t][s], a[t]);
  return (a[0] + a[1]) + (a[2] + a[3]);
}
__device__ __forceinline__ void accum_tile(f32x4 (&out)[4], const float* lds, int kt, const f32x4& p, int fr, int fg) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) out[dt] = mfma32(lds[(16 * kt + 4 * fg + r) * RS + 16 * dt + fr], p[r], out[dt]);
}
}  // namespace

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(SNT * 64) void attn32s_fwd_kernel(Attn32Args a) {
  __shared__ __attribute__((aligned(16))) float Ks[SROWS * RS];
  __shared__ __attribute__((aligned(16))) float Vs[SROWS * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.y, h = blockIdx.x, L = a.L, d = a.H * 64;
  const size_t ld = 3 * (size_t)d;
  const float* base = a.qkv + (size_t)n * L * ld + h * 64;
  stage_rows(Ks, base + d, ld, L, tid, SNT * 64);
  stage_rows(Vs, base + 2 * d, ld, L, tid, SNT * 64);
  const int nt = (L + 15) >> 4;
  const int qlim = a.q_rows > 0 ? (a.q_rows < L ? a.q_rows : L) : L;
  const int q = wave * 16 + fr, qc = q < L ? q : L - 1;
  f32x4 Q[4];
  load_own(Q, base + (size_t)qc * ld, fg);
  __syncthreads();
  if (wave * 16 >= qlim) return;
  const int kt_end = CAUSAL ? wave + 1 : nt;
  f32x4 S[SNT];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < SNT; ++kt) {
    if (kt < kt_end) {
      S[kt] = mm_tile(Ks, kt, Q, fr, fg);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = 16 * kt + 4 * fg + r;
        const bool ok = kk < L && (!CAUSAL || kk <= q);
        S[kt][r] = ok ? S[kt][r] * SCALE : -INFINITY;
        mx = fmaxf(mx, S[kt][r]);
      }
    }
  }
  mx = quad_max32(mx);
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < SNT; ++kt)
    if (kt < kt_end) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { S[kt][r] = __expf(S[kt][r] - mx); sum += S[kt][r]; }
    }
  sum = quad_sum32(sum);
  f32x4 O[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) O[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < SNT; ++kt)
    if (kt < kt_end) accum_tile(O, Vs, kt, S[kt], fr, fg);
  if (q < qlim) {
    const float inv = 1.f / sum;
    T* orow = (T*)a.out_split + ((size_t)n * L + q) * (2 * (size_t)d) + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store_pair4<T>(orow + 16 * dt + 4 * fg, d, O[dt] * inv);
    if (a.lse && fg == 0) a.lse[((size_t)n * a.H + h) * L + q] = mx + logf(sum);
  }
}

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(SNT * 64) void attn32s_bwd_kernel(Attn32BwdArgs a) {
  // Two LDS images (43.5 KiB: three workgroups per CU): K, V while the waves own query tiles (phase A: dQ), then Q, dO
  // while they own key tiles (phase B: dK, dV).  The own rows live in registers in both phases.
  __shared__ __attribute__((aligned(16))) float S0[SROWS * RS];
  __shared__ __attribute__((aligned(16))) float S1[SROWS * RS];
  __shared__ float lse_s[SROWS], del_s[SROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int n = blockIdx.y, h = blockIdx.x, L = a.L, d = a.H * 64;
  const size_t ld = 3 * (size_t)d;
  const float* base = a.qkv + (size_t)n * L * ld + h * 64;
  const float* gbase = a.dout32 + (size_t)n * L * d + h * 64;
  stage_rows(S0, base + d, ld, L, tid, SNT * 64);          // K
  stage_rows(S1, base + 2 * d, ld, L, tid, SNT * 64);      // V
  const size_t stat0 = ((size_t)n * a.H + h) * L;
  const int nt = (L + 15) >> 4;
  const int row = wave * 16 + fr, rc = row < L ? row : L - 1;     // own row: query in phase A, key in phase B
  f32x4 Q[4], dO[4];
  load_own(Q, base + (size_t)rc * ld, fg);
  load_own(dO, gbase + (size_t)rc * d, fg);
  // delta = rowsum(dO * O) of the own query row (O as a hi|lo pair)
  float dl = 0.f;
  {
    const T* orow = (const T*)a.out_split + ((size_t)n * L + rc) * (2 * (size_t)d) + h * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 o = load_pair4<T>(orow + 16 * t + 4 * fg, d);
#pragma unroll
      for (int e = 0; e < 4; ++e) dl += o[e] * dO[t][e];
    }
    dl = quad_sum32(dl);
  }
  const float lse = a.lse[stat0 + rc];
  if (fg == 0) { lse_s[row] = lse; del_s[row] = dl; }
  __syncthreads();
  const bool active = wave < nt;
  T* orow = (T*)a.dqkv_split + ((size_t)n * L + rc) * (6 * (size_t)d) + h * 64;
  f32x4 K[4], V[4];
  // ---- phase A: own query tile -> dQ
  if (active) {
    f32x4 dQ[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dQ[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kt_end = CAUSAL ? wave + 1 : nt;
#pragma unroll
    for (int kt = 0; kt < SNT; ++kt)
      if (kt < kt_end) {
        f32x4 S = mm_tile(S0, kt, Q, fr, fg);
        const f32x4 dP = mm_tile(S1, kt, dO, fr, fg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = 16 * kt + 4 * fg + r;
          const bool ok = kk < L && (!CAUSAL || kk <= row);
          const float p = ok ? __expf(S[r] * SCALE - lse) : 0.f;
          S[r] = p * (dP[r] - dl);
        }
        accum_tile(dQ, S0, kt, S, fr, fg);
      }
    if (row < L) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store_pair4<T>(orow + 16 * dt + 4 * fg, 3 * (size_t)d, dQ[dt] * SCALE);
    }
    load_own(K, S0 + row * RS, fg);                        // own key row for phase B, before the images are replaced
    load_own(V, S1 + row * RS, fg);
  }
  __syncthreads();
  stage_rows(S0, base, ld, L, tid, SNT * 64);              // Q
  stage_rows(S1, gbase, d, L, tid, SNT * 64);              // dO
  __syncthreads();
  // ---- phase B: own key tile -> dK, dV
  if (active) {
    f32x4 dK[4], dV[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dK[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int qt = 0; qt < SNT; ++qt)
      if (qt < nt && (!CAUSAL || qt >= wave)) {
        f32x4 S = mm_tile(S0, qt, K, fr, fg);            // lane: [key = fr][query = 16qt + 4fg + r]
        f32x4 dP = mm_tile(S1, qt, V, fr, fg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = 16 * qt + 4 * fg + r;
          const bool ok = qq < L && row < L && (!CAUSAL || row <= qq);
          const float p = ok ? __expf(S[r] * SCALE - lse_s[qq]) : 0.f;
          S[r] = p;
          dP[r] = p * (dP[r] - del_s[qq]);
        }
        accum_tile(dV, S1, qt, S, fr, fg);
        accum_tile(dK, S0, qt, dP, fr, fg);
      }
    if (row < L) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store_pair4<T>(orow + d + 16 * dt + 4 * fg, 3 * (size_t)d, dK[dt] * SCALE);
        store_pair4<T>(orow + 2 * d + 16 * dt + 4 * fg, 3 * (size_t)d, dV[dt]);
      }
    }
  }
}

// MVLPT_ATTN32_MODE: 0 = f32 MFMA (exact fp32 products), 1 (default) = three-term products on the 16-bit MFMA;
// bit 1 (value 2 / 3) additionally routes short sequences (L <= 80) to the generic kernels (experiments)
static int attn32_mode() {
  static const int m = getenv("MVLPT_ATTN32_MODE") ? atoi(getenv("MVLPT_ATTN32_MODE")) : 1;
  return m & 1;
}
static bool attn32_short_ok() {
  static const int m = getenv("MVLPT_ATTN32_MODE") ? atoi(getenv("MVLPT_ATTN32_MODE")) : 1;
  return (m & 2) == 0;
}
template <typename T>
static hipError_t fwd_t(const Attn32Args& a, hipStream_t s) {
  if (a.L <= SROWS && attn32_short_ok()) {
    dim3 grid(a.H, a.N), block(SNT * 64);
    if (a.causal) hipLaunchKernelGGL((attn32s_fwd_kernel<T, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn32s_fwd_kernel<T, false>), grid, block, 0, s, a);
    return hipGetLastError();
  }
  const int lq = a.q_rows > 0 ? (a.q_rows < a.L ? a.q_rows : a.L) : a.L;
  dim3 grid((lq + CH - 1) / CH, a.H, a.N), block(256);
  if (attn32_mode() == 0) {
    if (a.causal) hipLaunchKernelGGL((attn32_fwd_kernel<T, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn32_fwd_kernel<T, false>), grid, block, 0, s, a);
  } else {
    if (a.causal) hipLaunchKernelGGL((attn32x_fwd_kernel<T, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn32x_fwd_kernel<T, false>), grid, block, 0, s, a);
  }
  return hipGetLastError();
}
template <typename T>
static hipError_t bwd_t(const Attn32BwdArgs& a, hipStream_t s) {
  if (a.L <= SROWS && attn32_short_ok()) {
    dim3 grid(a.H, a.N), block(SNT * 64);
    if (a.causal) hipLaunchKernelGGL((attn32s_bwd_kernel<T, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn32s_bwd_kernel<T, false>), grid, block, 0, s, a);
    return hipGetLastError();
  }
  dim3 grid((a.L + CH - 1) / CH, a.H, a.N), block(256);
  if (attn32_mode() == 0) {
    if (a.causal) {
      hipLaunchKernelGGL((attn32_dq_kernel<T, true>), grid, block, 0, s, a);
      hipLaunchKernelGGL((attn32_dkv_kernel<T, true>), grid, block, 0, s, a);
    } else {
      hipLaunchKernelGGL((attn32_dq_kernel<T, false>), grid, block, 0, s, a);
      hipLaunchKernelGGL((attn32_dkv_kernel<T, false>), grid, block, 0, s, a);
    }
  } else if (a.causal) {
    hipLaunchKernelGGL((attn32x_dq_kernel<T, true>), grid, block, 0, s, a);
    hipLaunchKernelGGL((attn32x_dkv_kernel<T, true>), grid, block, 0, s, a);
  } else {
    hipLaunchKernelGGL((attn32x_dq_kernel<T, false>), grid, block, 0, s, a);
    hipLaunchKernelGGL((attn32x_dkv_kernel<T, false>), grid, block, 0, s, a);
  }
  return hipGetLastError();
}

hipError_t launch_attn32_fwd(int dtype, const Attn32Args& a, hipStream_t s) {
  if (a.L <= 0 || a.N <= 0 || a.H <= 0 || !a.qkv || !a.out_split) return hipErrorInvalidValue;
  if (dtype == DT_F16) return fwd_t<f16>(a, s);
  if (dtype == DT_BF16) return fwd_t<bf16>(a, s);
  return hipErrorInvalidValue;
}
hipError_t launch_attn32_bwd(int dtype, const Attn32BwdArgs& a, hipStream_t s) {
  if (a.L <= 0 || a.N <= 0 || a.H <= 0 || !a.qkv || !a.out_split || !a.dout32 || !a.lse || !a.delta || !a.dqkv_split)
    return hipErrorInvalidValue;
  if (dtype == DT_F16) return bwd_t<f16>(a, s);
  if (dtype == DT_BF16) return bwd_t<bf16>(a, s);
  return hipErrorInvalidValue;
}

}  // namespace mvlpt
