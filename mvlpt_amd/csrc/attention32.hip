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
#include "kernels.h"

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
__device__ __forceinline__ void mm_rows(f32x4 (&acc)[4], const float* lds, const f32x4 (&own)[4], int fr, int fg) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 c = *(const f32x4*)(lds + (16 * ct + fr) * RS + 16 * t + 4 * fg);
#pragma unroll
      for (int s = 0; s < 4; ++s) a = mfma32(c[s], own[t][s], a);
    }
    acc[ct] = a;
  }
}
// out[dt][r'] (own = fr, dim = 16dt + 4fg + r') += sum_streamed p[own][streamed] * C[streamed][dim]
__device__ __forceinline__ void mm_accum(f32x4 (&out)[4], const float* lds, const f32x4 (&p)[4], int fr, int fg) {
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[dt] = mfma32(lds[(16 * ct + 4 * fg + r) * RS + 16 * dt + fr], p[ct][r], out[dt]);
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
    const float alpha = (m == -INFINITY) ? 0.f : expf(m - m_new);
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        S[ct][r] = (m_new == -INFINITY) ? 0.f : expf(S[ct][r] - m_new);
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
        const float p = ok ? expf(S[ct][r] * SCALE - lse) : 0.f;
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
        const float p = ok ? expf(S[ct][r] * SCALE - lse_s[j]) : 0.f;
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

template <typename T>
static hipError_t fwd_t(const Attn32Args& a, hipStream_t s) {
  const int lq = a.q_rows > 0 ? (a.q_rows < a.L ? a.q_rows : a.L) : a.L;
  dim3 grid((lq + CH - 1) / CH, a.H, a.N), block(256);
  if (a.causal) hipLaunchKernelGGL((attn32_fwd_kernel<T, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((attn32_fwd_kernel<T, false>), grid, block, 0, s, a);
  return hipGetLastError();
}
template <typename T>
static hipError_t bwd_t(const Attn32BwdArgs& a, hipStream_t s) {
  dim3 grid((a.L + CH - 1) / CH, a.H, a.N), block(256);
  if (a.causal) {
    hipLaunchKernelGGL((attn32_dq_kernel<T, true>), grid, block, 0, s, a);
    hipLaunchKernelGGL((attn32_dkv_kernel<T, true>), grid, block, 0, s, a);
  } else {
    hipLaunchKernelGGL((attn32_dq_kernel<T, false>), grid, block, 0, s, a);
    hipLaunchKernelGGL((attn32_dkv_kernel<T, false>), grid, block, 0, s, a);
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
