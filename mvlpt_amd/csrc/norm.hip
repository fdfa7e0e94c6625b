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
// split-precision pair: hi = round16(v) at p, lo = round16(v - hi) at p + lo_off
template <typename TO>
__device__ __forceinline__ void store4_split(TO* p, int lo_off, f32x4 v) {
  typename Vec<TO>::v4 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) { TO h, l; split16<TO>(v[e], h, l); hi[e] = h; lo[e] = l; }
  *(typename Vec<TO>::v4*)p = hi;
  *(typename Vec<TO>::v4*)(p + lo_off) = lo;
}
template <> __device__ __forceinline__ void store4_split<float>(float* p, int, f32x4 v) { *(f32x4*)p = v; }
// mixed pair (GemmArgs::a_split == 2): hi at row + c, the four residual bytes at byte 2d + c of the same row
template <typename TO>
__device__ __forceinline__ void store4_lo8(TO* row, int d, int c, f32x4 v) {
  typename Vec<TO>::v4 hi;
  const uint32_t lo8 = split_lo8x4<TO>(v, hi);
  *(typename Vec<TO>::v4*)(row + c) = hi;
  *(uint32_t*)((char*)row + 2 * d + c) = lo8;
}
template <> __device__ __forceinline__ void store4_lo8<float>(float* row, int, int c, f32x4 v) { *(f32x4*)(row + c) = v; }
// split: 0 plain, 1 [hi | lo] pair, 2 mixed pair
template <typename TO>
__device__ __forceinline__ void store4_mode(TO* row, int d, int c, f32x4 v, int split) {
  if (split == 2) store4_lo8<TO>(row, d, c, v);
  else if (split) store4_split<TO>(row + c, d, v);
  else store4<TO>(row + c, v);
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
  TO* y = (TO*)a.y + (size_t)row * a.d * (a.split ? 2 : 1);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) if (ok[i]) {
    const int c = (i * 64 + lane) * 4;
    const f32x4 g = *(const f32x4*)(a.gamma + c), b = *(const f32x4*)(a.beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
    store4_mode<TO>(y, a.d, c, o, a.split);
  }
}

// Grid-stride variant: a fixed number of resident waves walk the rows, the NEXT row's loads are in flight while the
// current row is reduced and stored, gamma/beta stay in registers.  NV = float4 per lane (d <= NV*256).
template <typename TO, int NV>
__global__ __launch_bounds__(256) void ln_fwd_stream_kernel(LnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  bool ok[NV]; f32x4 g[NV], b[NV], cur[NV], nxt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    ok[i] = c < a.d;
    g[i] = ok[i] ? *(const f32x4*)(a.gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    b[i] = ok[i] ? *(const f32x4*)(a.beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto load_row = [&](int r, f32x4* v) {
    const size_t in_row = a.row_idx ? (size_t)a.row_idx[r] : (size_t)r * a.row_mul;
    const float* x = a.x + in_row * a.d;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      v[i] = ok[i] ? __builtin_nontemporal_load((const f32x4*)(x + (i * 64 + lane) * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  if (row < a.rows) load_row(row, cur);
  for (; row < a.rows; row += nw) {
    if (row + nw < a.rows) load_row(row + nw, nxt);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) sum += cur[i][0] + cur[i][1] + cur[i][2] + cur[i][3];   // padding lanes hold zeros
    const float mean = wave_sum(sum) / (float)a.d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) if (ok[i]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float c = cur[i][e] - mean; q += c * c; }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)a.d + LN_EPS);
    TO* y = (TO*)a.y + (size_t)row * a.d * (a.split ? 2 : 1);
#pragma unroll
    for (int i = 0; i < NV; ++i) if (ok[i]) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (cur[i][e] - mean) * rstd * g[i][e] + b[i][e];
      store4_mode<TO>(y, a.d, (i * 64 + lane) * 4, o, a.split);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) cur[i] = nxt[i];
  }
}

template <typename TO>
static void launch_ln_fwd_stream(const LnFwdArgs& a, hipStream_t s) {
  const int cus = stream_cus(s);
  static const int per_cu = [] { const char* e = getenv("MVLPT_LN_BLOCKS_PER_CU"); return e ? atoi(e) : 8; }();
  const int want = (a.rows + 3) / 4;
  dim3 grid(want < cus * per_cu ? want : cus * per_cu), block(256);
  const int nv = (a.d + 255) / 256;
  if (nv <= 2) hipLaunchKernelGGL((ln_fwd_stream_kernel<TO, 2>), grid, block, 0, s, a);
  else if (nv == 3) hipLaunchKernelGGL((ln_fwd_stream_kernel<TO, 3>), grid, block, 0, s, a);
  else if (nv == 4) hipLaunchKernelGGL((ln_fwd_stream_kernel<TO, 4>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((ln_fwd_stream_kernel<TO, LN_MAXV>), grid, block, 0, s, a);
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
    if (a.out16) store4_mode<T>((T*)a.out16 + in_row * a.d * (a.split ? 2 : 1), a.d, c, o, a.split);
  }
}

// Grid-stride backward (same scheme as ln_fwd_stream_kernel): the next row's x / dy / residual loads are issued before
// the current row's three reductions; streamed operands use non-temporal loads/stores (each is touched once).
template <typename TDY, typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_stream_kernel(LnBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  bool ok[NV]; f32x4 g[NV], xc[NV], dc[NV], rc[NV], xn[NV], dn[NV], rn[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    ok[i] = c < a.d;
    g[i] = ok[i] ? *(const f32x4*)(a.gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  auto in_row_of = [&](int r) { return a.row_idx ? (size_t)a.row_idx[r] : (size_t)r * a.row_mul; };
  auto load_row = [&](int r, f32x4* x, f32x4* d, f32x4* rs) {
    const size_t in_row = in_row_of(r);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      x[i] = ok[i] ? __builtin_nontemporal_load((const f32x4*)(a.x + in_row * a.d + c)) : zero;
      if (ok[i]) {
        if constexpr (sizeof(TDY) == 4) d[i] = __builtin_nontemporal_load((const f32x4*)((const float*)a.dy + (size_t)r * a.d + c));
        else d[i] = load4<TDY>((const TDY*)a.dy + (size_t)r * a.d + c);
      } else d[i] = zero;
      rs[i] = (ok[i] && a.resid) ? __builtin_nontemporal_load((const f32x4*)(a.resid + in_row * a.d + c)) : zero;
    }
  };
  if (row < a.rows) load_row(row, xc, dc, rc);
  for (; row < a.rows; row += nw) {
    if (row + nw < a.rows) load_row(row + nw, xn, dn, rn);
    const size_t in_row = in_row_of(row);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) sum += xc[i][0] + xc[i][1] + xc[i][2] + xc[i][3];
    const float mean = wave_sum(sum) / (float)a.d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) if (ok[i]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float c = xc[i][e] - mean; q += c * c; }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)a.d + LN_EPS);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) if (ok[i]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xc[i][e] = (xc[i][e] - mean) * rstd;   // xhat
        dc[i][e] = dc[i][e] * g[i][e];
        s1 += dc[i][e];
        s2 += dc[i][e] * xc[i][e];
      }
    }
    s1 = wave_sum(s1) / (float)a.d;
    s2 = wave_sum(s2) / (float)a.d;
#pragma unroll
    for (int i = 0; i < NV; ++i) if (ok[i]) {
      const int c = (i * 64 + lane) * 4;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rstd * (dc[i][e] - s1 - xc[i][e] * s2);
      if (a.resid) o += rc[i];
      __builtin_nontemporal_store(o, (f32x4*)(a.out32 + in_row * a.d + c));
      if (a.out16) store4_mode<T>((T*)a.out16 + in_row * a.d * (a.split ? 2 : 1), a.d, c, o, a.split);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) { xc[i] = xn[i]; dc[i] = dn[i]; rc[i] = rn[i]; }
  }
}

template <typename TDY, typename T>
static void launch_ln_bwd_stream(const LnBwdArgs& a, hipStream_t s) {
  const int cus = stream_cus(s);
  static const int per_cu = [] { const char* e = getenv("MVLPT_LNB_BLOCKS_PER_CU"); return e ? atoi(e) : 8; }();
  const int want = (a.rows + 3) / 4;
  dim3 grid(want < cus * per_cu ? want : cus * per_cu), block(256);
  const int nv = (a.d + 255) / 256;
  if (nv <= 2) hipLaunchKernelGGL((ln_bwd_stream_kernel<TDY, T, 2>), grid, block, 0, s, a);
  else if (nv == 3) hipLaunchKernelGGL((ln_bwd_stream_kernel<TDY, T, 3>), grid, block, 0, s, a);
  else if (nv == 4) hipLaunchKernelGGL((ln_bwd_stream_kernel<TDY, T, 4>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((ln_bwd_stream_kernel<TDY, T, LN_MAXV>), grid, block, 0, s, a);
}

hipError_t launch_ln_fwd(int out_dtype, const LnFwdArgs& a, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  if (a.d % 4 != 0 || a.d > LN_MAXV * 256) return hipErrorInvalidValue;
  static const int stream_mode = [] { const char* e = getenv("MVLPT_LN_STREAM"); return e ? atoi(e) : 1; }();
  if (stream_mode && a.rows >= 4096) {
    if (out_dtype == DT_F32) launch_ln_fwd_stream<float>(a, s);
    else if (out_dtype == DT_F16) launch_ln_fwd_stream<f16>(a, s);
    else if (out_dtype == DT_BF16) launch_ln_fwd_stream<bf16>(a, s);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
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
  static const int stream_mode = [] { const char* e = getenv("MVLPT_LNB_STREAM"); return e ? atoi(e) : 1; }();
  if (stream_mode && a.rows >= 4096) {
    if (dtype == DT_F16) { if (dy32) launch_ln_bwd_stream<float, f16>(a, s); else launch_ln_bwd_stream<f16, f16>(a, s); }
    else if (dtype == DT_BF16) { if (dy32) launch_ln_bwd_stream<float, bf16>(a, s); else launch_ln_bwd_stream<bf16, bf16>(a, s); }
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
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
