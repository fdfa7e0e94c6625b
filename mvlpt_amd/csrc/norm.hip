// LayerNorm forward / backward-dX, one wave64 per row, fp32 statistics (clip/model.py:153-159:
// the reference's LayerNorm subclass always computes in fp32, eps 1e-5, affine).  HBM-bound:
// float4 loads, the row is held in registers, reductions are wavefront shuffles (no LDS).
// gamma/beta are frozen, so the backward produces dX only and fuses the residual add
// (dx_out = dx_resid + LN'(dy)) plus the 16-bit copy that feeds the next dX GEMM.
#include "kernels.h"

namespace mvlpt {

constexpr float LN_EPS = 1e-5f;
constexpr int LN_MAXV = 8;  // float4 per lane -> d <= 2048

template <typename TO>
__device__ __forceinline__ void store4(TO* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<f16>(f16* p, f32x4 v) {
  f16x4 w; for (int e = 0; e < 4; ++e) w[e] = (f16)v[e]; *(f16x4*)p = w;
}
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, f32x4 v) {
  bf16x4 w; for (int e = 0; e < 4; ++e) w[e] = (bf16)v[e]; *(bf16x4*)p = w;
}
template <typename TI>
__device__ __forceinline__ f32x4 load4(const TI* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load4<f16>(const f16* p) {
  f16x4 w = *(const f16x4*)p; return f32x4{(float)w[0], (float)w[1], (float)w[2], (float)w[3]};
}
template <> __device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
  bf16x4 w = *(const bf16x4*)p; return f32x4{(float)w[0], (float)w[1], (float)w[2], (float)w[3]};
}

// mean / rstd of the row held in v[0..nv) (lanes past d hold zeros and are excluded by `cnt`)
__device__ __forceinline__ void row_stats(const f32x4* v, const bool* ok, int d, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) if (ok[i]) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) if (ok[i]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { float c = v[i][e] - mean; q += c * c; }
  }
  rstd = rsqrtf(wave_sum(q) / (float)d + LN_EPS);
}

template <typename TO>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const size_t in_row = a.row_idx ? (size_t)a.row_idx[row] : (size_t)row * a.row_mul;
  const float* x = a.x + in_row * a.d;
  f32x4 v[LN_MAXV]; bool ok[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    ok[i] = c < a.d;
    v[i] = ok[i] ? *(const f32x4*)(x + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float mean, rstd;
  row_stats(v, ok, a.d, mean, rstd);
  TO* y = (TO*)a.y + (size_t)row * a.d;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) if (ok[i]) {
    const int c = (i * 64 + lane) * 4;
    const f32x4 g = *(const f32x4*)(a.gamma + c), b = *(const f32x4*)(a.beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
    store4<TO>(y + c, o);
  }
}

template <typename TDY, typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const size_t in_row = a.row_idx ? (size_t)a.row_idx[row] : (size_t)row * a.row_mul;
  const float* x = a.x + in_row * a.d;
  const TDY* dy = (const TDY*)a.dy + (size_t)row * a.d;
  f32x4 v[LN_MAXV], gg[LN_MAXV]; bool ok[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    ok[i] = c < a.d;
    v[i] = ok[i] ? *(const f32x4*)(x + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float mean, rstd;
  row_stats(v, ok, a.d, mean, rstd);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) if (ok[i]) {
    const int c = (i * 64 + lane) * 4;
    const f32x4 g = *(const f32x4*)(a.gamma + c);
    const f32x4 d = load4<TDY>(dy + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i][e] = (v[i][e] - mean) * rstd;   // xhat
      gg[i][e] = d[e] * g[e];
      s1 += gg[i][e];
      s2 += gg[i][e] * v[i][e];
    }
  }
  s1 = wave_sum(s1) / (float)a.d;
  s2 = wave_sum(s2) / (float)a.d;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) if (ok[i]) {
    const int c = (i * 64 + lane) * 4;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = rstd * (gg[i][e] - s1 - v[i][e] * s2);
    if (a.resid) o += *(const f32x4*)(a.resid + in_row * a.d + c);
    *(f32x4*)(a.out32 + in_row * a.d + c) = o;
    if (a.out16) store4<T>((T*)a.out16 + in_row * a.d + c, o);
  }
}

hipError_t launch_ln_fwd(int out_dtype, const LnFwdArgs& a, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  if (a.d % 4 != 0 || a.d > LN_MAXV * 256) return hipErrorInvalidValue;
  dim3 grid((a.rows + 3) / 4), block(256);
  if (out_dtype == DT_F32) hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, s, a);
  else if (out_dtype == DT_F16) hipLaunchKernelGGL(ln_fwd_kernel<f16>, grid, block, 0, s, a);
  else if (out_dtype == DT_BF16) hipLaunchKernelGGL(ln_fwd_kernel<bf16>, grid, block, 0, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_ln_bwd(int dtype, const LnBwdArgs& a, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  if (a.d % 4 != 0 || a.d > LN_MAXV * 256) return hipErrorInvalidValue;
  dim3 grid((a.rows + 3) / 4), block(256);
  const bool dy32 = a.dy_dtype == DT_F32;
  if (dtype == DT_F16) {
    if (dy32) hipLaunchKernelGGL((ln_bwd_kernel<float, f16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((ln_bwd_kernel<f16, f16>), grid, block, 0, s, a);
  } else if (dtype == DT_BF16) {
    if (dy32) hipLaunchKernelGGL((ln_bwd_kernel<float, bf16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((ln_bwd_kernel<bf16, bf16>), grid, block, 0, s, a);
  } else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace mvlpt
