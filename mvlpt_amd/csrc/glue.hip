// HBM-bound glue kernels around the GEMM / LayerNorm / attention kernels: patch extraction, token and prompt
// assembly with the learnable prompt rows spliced in, prompt-gradient reductions, casts / weight packing, and
// the fp32 head (cosine logits + cross-entropy).  All coalesced along the feature dimension, 16-byte
// accesses where the layout allows; reductions use wavefront shuffles.
#include <type_traits>
#include "kernels.h"

namespace mvlpt {

template <typename T> __device__ __forceinline__ float ld_f32(const void* p, size_t i) { return (float)((const T*)p)[i]; }

// ------------------------------------------------------------------------------------------------ casts / packing
template <typename T>
__global__ void cast_to16_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n4, const float* scale_dev) {
  const float sc = scale_dev ? scale_dev[0] : 1.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = *(const f32x4*)(in + i * 4);
    typename Vec<T>::v4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (T)(v[e] * sc);
    *(typename Vec<T>::v4*)(out + i * 4) = w;
  }
}
hipError_t launch_cast_f32_to16(int dtype, const float* in, void* out, size_t n, const float* scale_dev, hipStream_t s) {
  if (n == 0) return hipSuccess;
  if (n % 4) return hipErrorInvalidValue;
  const size_t n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  if (dtype == DT_F16) hipLaunchKernelGGL(cast_to16_kernel<f16>, dim3(grid), dim3(256), 0, s, in, (f16*)out, n4, scale_dev);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(cast_to16_kernel<bf16>, dim3(grid), dim3(256), 0, s, in, (bf16*)out, n4, scale_dev);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// fp32 [rows, d] -> 16-bit pair [rows, 2d] = [hi | lo]  (split-precision A operand, GemmArgs::a_split)
template <typename T>
__global__ void cast_split_kernel(const float* __restrict__ in, T* __restrict__ out, size_t rows, int d4, const float* scale_dev, int lo8) {
  const float sc = scale_dev ? scale_dev[0] : 1.0f;
  const size_t n4 = rows * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / d4; const int c = (int)(i - r * d4) * 4;
    const f32x4 v = *(const f32x4*)(in + i * 4) * sc;
    typename Vec<T>::v4 hi, lo;
    T* row = out + r * (size_t)(8 * d4);
    if (lo8) {      // mixed pair: [hi | lo8 bytes | unused]
      const uint32_t w = split_lo8x4<T>(v, hi);
      *(typename Vec<T>::v4*)(row + c) = hi;
      *(uint32_t*)((char*)row + 8 * d4 + c) = w;
      continue;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { T h, l; split16<T>(v[e], h, l); hi[e] = h; lo[e] = l; }
    *(typename Vec<T>::v4*)(row + c) = hi;
    *(typename Vec<T>::v4*)(row + 4 * d4 + c) = lo;
  }
}
hipError_t launch_cast_f32_split(int dtype, const float* in, void* out, size_t rows, int d, const float* scale_dev, hipStream_t s, int lo8) {
  if (rows == 0) return hipSuccess;
  if (d % 4) return hipErrorInvalidValue;
  const size_t n4 = rows * (d / 4);
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  if (dtype == DT_F16) hipLaunchKernelGGL(cast_split_kernel<f16>, dim3(grid), dim3(256), 0, s, in, (f16*)out, rows, d / 4, scale_dev, lo8);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(cast_split_kernel<bf16>, dim3(grid), dim3(256), 0, s, in, (bf16*)out, rows, d / 4, scale_dev, lo8);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
hipError_t launch_cast_any_to_f32(int in_dtype, const void* in, float* out, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  if (in_dtype == DT_F32) return hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, s);
  if (in_dtype == DT_F16) hipLaunchKernelGGL(cast_to_f32_kernel<f16>, dim3(grid), dim3(256), 0, s, (const f16*)in, out, n);
  else if (in_dtype == DT_BF16) hipLaunchKernelGGL(cast_to_f32_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)in, out, n);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int rows, int cols, int ld_out) {
  const size_t n = (size_t)rows * ld_out;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld_out), c = (int)(i % ld_out);
    out[i] = c < cols ? (T)w[(size_t)r * cols + c] : (T)0.f;
  }
}
hipError_t launch_pack_weight(int dtype, const float* w, void* out, int rows, int cols, int ld_out, hipStream_t s) {
  const size_t n = (size_t)rows * ld_out;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  if (dtype == DT_F32) hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, s, w, (float*)out, rows, cols, ld_out);
  else if (dtype == DT_F16) hipLaunchKernelGGL(pack_weight_kernel<f16>, dim3(grid), dim3(256), 0, s, w, (f16*)out, rows, cols, ld_out);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(grid), dim3(256), 0, s, w, (bf16*)out, rows, cols, ld_out);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// tiled transpose through LDS: out[c][r] = w[r][c]
template <typename T>
__global__ void pack_weight_t_kernel(const float* __restrict__ w, T* __restrict__ out, int rows, int cols, int ld_out) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < rows && c < cols) ? w[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < cols && r < rows) out[(size_t)c * ld_out + r] = (T)tile[tx][k];
  }
}
hipError_t launch_pack_weight_t(int dtype, const float* w, void* out, int rows, int cols, hipStream_t s, int ld_out) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(256);
  if (ld_out <= 0) ld_out = rows;
  if (dtype == DT_F32) hipLaunchKernelGGL(pack_weight_t_kernel<float>, grid, block, 0, s, w, (float*)out, rows, cols, ld_out);
  else if (dtype == DT_F16) hipLaunchKernelGGL(pack_weight_t_kernel<f16>, grid, block, 0, s, w, (f16*)out, rows, cols, ld_out);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(pack_weight_t_kernel<bf16>, grid, block, 0, s, w, (bf16*)out, rows, cols, ld_out);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// fp8 plane of a packed weight (GemmArgs::a_split == 2): byte (ro, co) of the plane = e4m3(W * scale), W = w[ro][co]
// (transposed: w[co][ro]); columns co >= the source extent are zero.  Load time only.
__global__ void pack_weight8_kernel(const float* __restrict__ w, uint8_t* __restrict__ out8, int rows, int cols, int transposed,
                                    int cols_out, size_t pitch_bytes, const float* scale_dev) {
  const int rows_out = transposed ? cols : rows, cols_src = transposed ? rows : cols;
  const float sc = scale_dev[0];
  const size_t n4 = (size_t)rows_out * (cols_out / 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int ro = (int)(i / (cols_out / 4)), co = (int)(i % (cols_out / 4)) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = co + e;
      const float x = c < cols_src ? (transposed ? w[(size_t)c * cols + ro] : w[(size_t)ro * cols + c]) : 0.f;
      v[e] = fminf(fmaxf(x * sc, -448.0f), 448.0f);
    }
    int word = 0;
    word = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], word, false);
    word = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], word, true);
    *(uint32_t*)(out8 + (size_t)ro * pitch_bytes + co) = (uint32_t)word;
  }
}
hipError_t launch_pack_weight8(const float* w, uint8_t* out8, int rows, int cols, int transposed, int cols_out, size_t pitch_bytes,
                               const float* scale_dev, hipStream_t s) {
  if (rows <= 0 || cols <= 0 || (cols_out % 4) || (pitch_bytes % 4)) return hipErrorInvalidValue;
  const size_t n4 = (size_t)(transposed ? cols : rows) * (cols_out / 4);
  const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weight8_kernel, dim3(grid), dim3(256), 0, s, w, out8, rows, cols, transposed, cols_out, pitch_bytes, scale_dev);
  return hipGetLastError();
}

// Debug: order-independent 64-bit checksum of a buffer (sum of its 32-bit words), added into *out.  mvlpt_debug_checksums uses it to
// fingerprint every intermediate of a tower so that two runs can be compared stage by stage without keeping the tensors.
__global__ __launch_bounds__(256) void checksum_kernel(const uint32_t* __restrict__ p, size_t nwords, unsigned long long* __restrict__ out) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x)
    acc += (unsigned long long)__builtin_nontemporal_load(p + i) * (unsigned long long)((i & 1023) + 1);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
hipError_t launch_checksum(const void* p, size_t bytes, unsigned long long* out, hipStream_t s) {
  const size_t n = bytes / 4;
  if (!n) return hipSuccess;
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(checksum_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, s, (const uint32_t*)p, n, out);
  return hipGetLastError();
}
hipError_t launch_zero(void* p, size_t bytes, hipStream_t s) { return bytes ? hipMemsetAsync(p, 0, bytes, s) : hipSuccess; }

// ------------------------------------------------------------------------------------------------ image side
// patches[b*G2 + gy*G + gx][c*P*P + ky*P + kx] = image[b][c][gy*P+ky][gx*P+kx]     (conv1 as GEMM, clip/model.py:207)
template <typename T, typename TI>
__global__ void patchify_kernel(const TI* __restrict__ img, T* __restrict__ out, int B, int R, int P, int Kp) {
  const int G = R / P, K = 3 * P * P;
  const size_t n8 = (size_t)B * G * G * (Kp / 8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const int cb = (int)(i % (Kp / 8));
    const size_t row = i / (Kp / 8);
    const int gx = (int)(row % G), gy = (int)((row / G) % G), b = (int)(row / ((size_t)G * G));
    typename Vec<T>::v8 w;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = cb * 8 + e;
      float v = 0.f;
      if (col < K) {
        const int c = col / (P * P), ky = (col % (P * P)) / P, kx = col % P;
        v = (float)img[(((size_t)b * 3 + c) * R + gy * P + ky) * R + gx * P + kx];
      }
      w[e] = (T)v;
    }
    *(typename Vec<T>::v8*)(out + row * Kp + cb * 8) = w;
  }
}
// fast path: fp32 image, P % 8 == 0 (ViT-B/16, B/32): 8 consecutive kx are 32 contiguous bytes of one image row
template <typename T>
__global__ void patchify_vec_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int R, int P, int Kp) {
  const int G = R / P, K = 3 * P * P, PP = P * P;
  const size_t n8 = (size_t)B * G * G * (Kp / 8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const int cb = (int)(i % (Kp / 8));
    const size_t row = i / (Kp / 8);
    const int gx = (int)(row % G), gy = (int)((row / G) % G), b = (int)(row / ((size_t)G * G));
    typename Vec<T>::v8 w;
    const int col = cb * 8;
    if (col < K) {
      const int c = col / PP, ky = (col % PP) / P, kx = col % P;
      const float* p = img + (((size_t)b * 3 + c) * R + gy * P + ky) * R + gx * P + kx;
      const f32x4 v0 = *(const f32x4*)p, v1 = *(const f32x4*)(p + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { w[e] = (T)v0[e]; w[e + 4] = (T)v1[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = (T)0.f;
    }
    *(typename Vec<T>::v8*)(out + row * Kp + col) = w;
  }
}
// fast path: 16-bit image of the compute type, P % 8 == 0: a pure copy of 16-byte chunks.  One block per (image, patch row):
// its 3 x P image rows are read as one coalesced stream into LDS and leave as whole patch rows (Kp * 2 contiguous bytes each)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void patchify16_kernel(const u32x4* __restrict__ img, u32x4* __restrict__ out, int R, int P, int Kp) {
  extern __shared__ u32x4 pf_sm[];
  const int G = R / P, RC = R / 8, PC = P / 8;
  const int b = blockIdx.x / G, gy = blockIdx.x - b * G;
  const int n = 3 * P * RC;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i / (P * RC), rem = i - c * (P * RC), ky = rem / RC, xc = rem - ky * RC;
    pf_sm[i] = __builtin_nontemporal_load(img + (((size_t)b * 3 + c) * R + gy * P + ky) * RC + xc);
  }
  __syncthreads();
  const int KC = Kp / 8, KV = 3 * P * PC;
  for (int i = threadIdx.x; i < G * KC; i += 256) {
    const int gx = i / KC, cb = i - gx * KC;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (cb < KV) {
      const int c = cb / (P * PC), rem = cb - c * (P * PC), ky = rem / PC, kc = rem - ky * PC;
      v = pf_sm[(c * P + ky) * RC + gx * PC + kc];
    }
    out[((size_t)(b * G + gy) * G + gx) * KC + cb] = v;
  }
}
template <typename T>
static hipError_t patchify_t(const void* image, int image_dtype, T* out, int B, int R, int P, int Kp, hipStream_t s) {
  if (image_dtype == (sizeof(T) == 2 && std::is_same<T, f16>::value ? DT_F16 : DT_BF16) && P % 8 == 0 && R % 8 == 0 && 3 * P * (R / 8) * 16 <= 65536) {
    hipLaunchKernelGGL(patchify16_kernel, dim3(B * (R / P)), dim3(256), 3 * P * (R / 8) * 16, s, (const u32x4*)image, (u32x4*)out, R, P, Kp);
    return hipGetLastError();
  }
  if (image_dtype == DT_F32 && P % 8 == 0 && R % 4 == 0) {
    const size_t n8v = (size_t)B * (R / P) * (R / P) * (Kp / 8);
    const int gridv = (int)((n8v + 255) / 256 < 16384 ? (n8v + 255) / 256 : 16384);
    hipLaunchKernelGGL((patchify_vec_kernel<T>), dim3(gridv), dim3(256), 0, s, (const float*)image, out, B, R, P, Kp);
    return hipGetLastError();
  }
  const int G = R / P;
  const size_t n8 = (size_t)B * G * G * (Kp / 8);
  const int grid = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
  if (image_dtype == DT_F32) hipLaunchKernelGGL((patchify_kernel<T, float>), dim3(grid), dim3(256), 0, s, (const float*)image, out, B, R, P, Kp);
  else if (image_dtype == DT_F16) hipLaunchKernelGGL((patchify_kernel<T, f16>), dim3(grid), dim3(256), 0, s, (const f16*)image, out, B, R, P, Kp);
  else if (image_dtype == DT_BF16) hipLaunchKernelGGL((patchify_kernel<T, bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)image, out, B, R, P, Kp);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t launch_patchify(int dtype, const void* image, int image_dtype, void* out, int B, int R, int P, int Kp, hipStream_t s) {
  if (Kp % 8 || R % P) return hipErrorInvalidValue;
  if (dtype == DT_F16) return patchify_t<f16>(image, image_dtype, (f16*)out, B, R, P, Kp, s);
  if (dtype == DT_BF16) return patchify_t<bf16>(image, image_dtype, (bf16*)out, B, R, P, Kp, s);
  return hipErrorInvalidValue;
}

// One wave per token row of x [B, 1+n+G2, d].  row 0: ln_pre(cls + pos[0]); rows 1..n: visual prompts, raw
// (no positional embedding, no ln_pre: trainers/mvlpt.py:57-62, 416-437); others: ln_pre(patch + pos[1+i]).
// Grid-stride: resident waves walk the rows, the next row's patch-embedding loads are in flight while the current
// row is normalised; gamma/beta stay in registers; streamed fp32 operands use non-temporal loads/stores.
// PK: the rows go out in the packed residual-stream format (common.h respk_*: x = the hi plane [rows, d] fp16, xlo the byte plane)
// together with the LayerNorm-folding statistics of block 0's ln_1 ({sum, sum of squares} in slot 0 of the row's ntp slots);
// prompt-free towers only.
template <int NV, bool PK>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const float* __restrict__ pe, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, const float* __restrict__ g,
                                                              const float* __restrict__ bt, const float* __restrict__ vpt,
                                                              const float* __restrict__ vmask, int n_vpt, float* __restrict__ x, int B,
                                                              int G2, int d, uint8_t* __restrict__ xlo, float* __restrict__ part, int ntp) {
  const int lane = threadIdx.x & 63;
  const int L = 1 + n_vpt + G2;
  const size_t rows = (size_t)B * L, nw = (size_t)gridDim.x * 4;
  size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  bool ok[NV]; f32x4 gg[NV], bb[NV], cur[NV], nxt[NV];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (k * 64 + lane) * 4;
    ok[k] = c < d;
    gg[k] = ok[k] ? *(const f32x4*)(g + c) : zero;
    bb[k] = ok[k] ? *(const f32x4*)(bt + c) : zero;
  }
  // raw source row (before + pos): prompts rows come from vpt, row 0 from cls, the rest from the patch embedding
  auto load_row = [&](size_t r, f32x4* v) {
    const int b = (int)(r / L), i = (int)(r % L);
    const bool prompt = i >= 1 && i <= n_vpt;
    const float* src = prompt ? vpt + (size_t)(i - 1) * d : (i == 0 ? cls : pe + ((size_t)b * G2 + (i - n_vpt - 1)) * d);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * 64 + lane) * 4;
      v[k] = !ok[k] ? zero : (prompt || i == 0) ? *(const f32x4*)(src + c) : __builtin_nontemporal_load((const f32x4*)(src + c));
      // vpt_dropout (trainers/mvlpt.py:424): the prompt rows are `expand`ed over the batch BEFORE the dropout, so every image has
      // its own mask [B, n_vpt, d] (0 or 1 / (1 - p))
      if (vmask && prompt && ok[k]) v[k] *= *(const f32x4*)(vmask + ((size_t)b * n_vpt + (i - 1)) * d + c);
    }
  };
  if (row < rows) load_row(row, cur);
  for (; row < rows; row += nw) {
    if (row + nw < rows) load_row(row + nw, nxt);
    const int i = (int)(row % L);
    float* xo = x + row * d;
    if (i >= 1 && i <= n_vpt) {
#pragma unroll
      for (int k = 0; k < NV; ++k) if (ok[k]) __builtin_nontemporal_store(cur[k], (f32x4*)(xo + (k * 64 + lane) * 4));
    } else {
      const float* pp = pos + (size_t)(i == 0 ? 0 : i - n_vpt) * d;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) if (ok[k]) {
        cur[k] += *(const f32x4*)(pp + (k * 64 + lane) * 4);
        s += cur[k][0] + cur[k][1] + cur[k][2] + cur[k][3];
      }
      const float mean = wave_sum(s) / (float)d;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) if (ok[k]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = cur[k][e] - mean; q += t * t; }
      }
      const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
      [[maybe_unused]] float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) if (ok[k]) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (cur[k][e] - mean) * rstd * gg[k][e] + bb[k][e];
        if constexpr (PK) {
          s1 += (o[0] + o[1]) + (o[2] + o[3]);
          s2 += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
          f16x4 h;
          const uint32_t l = respk_split4(o, h);
          const size_t oo = row * d + (k * 64 + lane) * 4;
          *(f16x4*)((f16*)x + oo) = h;
          *(uint32_t*)(xlo + oo) = l;
        } else __builtin_nontemporal_store(o, (f32x4*)(xo + (k * 64 + lane) * 4));
      }
      if constexpr (PK) {
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane < ntp) *(float2*)(part + (row * ntp + lane) * 2) = lane == 0 ? float2{s1, s2} : float2{0.f, 0.f};
      }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) cur[k] = nxt[k];
  }
}
hipError_t launch_assemble_tokens(const float* patch_emb, const float* cls, const float* pos, const float* g, const float* b,
                                  const float* vpt, int n_vpt, float* x, int B, int G2, int d, hipStream_t s, const float* vmask) {
  if (d % 4 || d > 2048) return hipErrorInvalidValue;
  const size_t rows = (size_t)B * (1 + n_vpt + G2);
  const size_t want = (rows + 3) / 4;
  const dim3 grid((unsigned)(want < 2048 ? want : 2048)), block(256);
  const int nv = (d + 255) / 256;
#define MVLPT_ASM_TOK(NV) hipLaunchKernelGGL((assemble_tokens_kernel<NV, false>), grid, block, 0, s, patch_emb, cls, pos, g, b, vpt, vmask, n_vpt, x, B, G2, d, nullptr, nullptr, 0)
  if (nv <= 2) MVLPT_ASM_TOK(2);
  else if (nv == 3) MVLPT_ASM_TOK(3);
  else if (nv == 4) MVLPT_ASM_TOK(4);
  else MVLPT_ASM_TOK(8);
#undef MVLPT_ASM_TOK
  return hipGetLastError();
}
hipError_t launch_assemble_tokens_packed(const float* patch_emb, const float* cls, const float* pos, const float* g, const float* b,
                                         void* hi, uint8_t* lo, float* part, int ntp, int B, int G2, int d, hipStream_t s) {
  if (d % 4 || d > 2048 || ntp < 1 || ntp > 64) return hipErrorInvalidValue;
  const size_t rows = (size_t)B * (1 + G2);
  const size_t want = (rows + 3) / 4;
  const dim3 grid((unsigned)(want < 2048 ? want : 2048)), block(256);
  const int nv = (d + 255) / 256;
#define MVLPT_ASM_TOK(NV) hipLaunchKernelGGL((assemble_tokens_kernel<NV, true>), grid, block, 0, s, patch_emb, cls, pos, g, b, nullptr, nullptr, 0, (float*)hi, B, G2, d, lo, part, ntp)
  if (nv <= 2) MVLPT_ASM_TOK(2);
  else if (nv == 3) MVLPT_ASM_TOK(3);
  else if (nv == 4) MVLPT_ASM_TOK(4);
  else MVLPT_ASM_TOK(8);
#undef MVLPT_ASM_TOK
  return hipGetLastError();
}

__global__ void overwrite_rows_kernel(const float* __restrict__ rows, const float* __restrict__ vmask, int n, float* __restrict__ x, int B,
                                      int L, int d) {
  const int d4 = d / 4;
  const size_t total = (size_t)B * n * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % d4), j = (int)((i / d4) % n), b = (int)(i / ((size_t)d4 * n));
    f32x4 v = *(const f32x4*)(rows + (size_t)j * d + c * 4);
    if (vmask) v *= *(const f32x4*)(vmask + ((size_t)b * n + j) * d + c * 4);      // per-image dropout mask of this layer's prompts
    *(f32x4*)(x + ((size_t)b * L + 1 + j) * d + c * 4) = v;
  }
}
hipError_t launch_overwrite_rows(const float* rows, int n, float* x, int B, int L, int d, hipStream_t s, const float* vmask) {
  if (n <= 0) return hipSuccess;
  const size_t total = (size_t)B * n * (d / 4);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(overwrite_rows_kernel, dim3(grid), dim3(256), 0, s, rows, vmask, n, x, B, L, d);
  return hipGetLastError();
}

// dst + r*dst_pitch <- src + r*src_pitch, row_bytes each (multiples of 16 B): the CLS-row gathers / scatters of the
// last image block (one launch instead of a hipMemcpy2DAsync runtime kernel)
__global__ void copy_rows_strided_kernel(const char* __restrict__ src, char* __restrict__ dst, int rows, size_t src_pitch,
                                         size_t dst_pitch, int chunks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * chunks) return;
  const int r = i / chunks, c = i - r * chunks;
  *(f32x4*)(dst + (size_t)r * dst_pitch + (size_t)c * 16) = *(const f32x4*)(src + (size_t)r * src_pitch + (size_t)c * 16);
}
hipError_t launch_copy_rows_strided(const void* src, void* dst, int rows, size_t src_pitch, size_t dst_pitch, int row_bytes, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (row_bytes % 16 || src_pitch % 16 || dst_pitch % 16) return hipErrorInvalidValue;
  const int chunks = row_bytes / 16, n = rows * chunks;
  hipLaunchKernelGGL(copy_rows_strided_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const char*)src, (char*)dst, rows, src_pitch, dst_pitch, chunks);
  return hipGetLastError();
}

// out[j,:] = inv_scale * sum_b dx32[b, row0+j, :]   (prompt rows are `expand`ed over the batch => sum over B);
// optionally zero those rows afterwards (deep prompts overwrite the rows: upstream gradient is 0).
// 16 waves per block: wave w sums images w, w+16, ... (all loads of a wave independent -> one memory latency, not B),
// lane = one float4 column; the 16 partials are added in a fixed order through LDS (deterministic, no atomics).
template <typename T>
__global__ __launch_bounds__(1024) void reduce_prompt_rows_kernel(float* __restrict__ dx32, T* __restrict__ dx16, int B, int L, int d,
                                                                  int row0, int n, float* __restrict__ out, const float* scale_dev,
                                                                  int zero_after, int split16, const float* __restrict__ vmask) {
  __shared__ f32x4 part[16][64];
  const int j = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const bool ok = c < d;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
#pragma unroll 8
    for (int b = wave; b < B; b += 16) {
      const size_t o = ((size_t)b * L + row0 + j) * d + c;
      if (vmask) acc += *(const f32x4*)(dx32 + o) * *(const f32x4*)(vmask + ((size_t)b * n + j) * d + c);   // back through the dropout
      else acc += *(const f32x4*)(dx32 + o);
      if (zero_after) {
        *(f32x4*)(dx32 + o) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (dx16) {
          typename Vec<T>::v4 z; for (int e = 0; e < 4; ++e) z[e] = (T)0.f;
          T* row16 = dx16 + 2 * (o - c);                       // pair rows have a pitch of 2d elements
          if (split16 == 2) {                                  // mixed pair [hi (d x 16 bit) | lo8 (d bytes) | unused]: the 4 residual bytes of columns c..c+3
            *(typename Vec<T>::v4*)(row16 + c) = z;
            *(uint32_t*)((char*)row16 + 2 * d + c) = 0u;
          } else if (split16) { *(typename Vec<T>::v4*)(row16 + c) = z; *(typename Vec<T>::v4*)(row16 + d + c) = z; }
          else *(typename Vec<T>::v4*)(dx16 + o) = z;
        }
      }
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && ok) {
    f32x4 t = part[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += part[w][lane];
    const float inv = scale_dev ? scale_dev[1] : 1.0f;
    *(f32x4*)(out + (size_t)j * d + c) = t * inv;
  }
}
hipError_t launch_reduce_prompt_rows(int dtype, float* dx32, void* dx16, int B, int L, int d, int row0, int n, float* out,
                                     const float* scale_dev, int zero_after, hipStream_t s, int split16, const float* vmask) {
  if (n <= 0) return hipSuccess;
  if (d % 4) return hipErrorInvalidValue;
  dim3 grid((d + 255) / 256, n), block(1024);
  if (dtype == DT_F16) hipLaunchKernelGGL(reduce_prompt_rows_kernel<f16>, grid, block, 0, s, dx32, (f16*)dx16, B, L, d, row0, n, out, scale_dev, zero_after, split16, vmask);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(reduce_prompt_rows_kernel<bf16>, grid, block, 0, s, dx32, (bf16*)dx16, B, L, d, row0, n, out, scale_dev, zero_after, split16, vmask);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ text side
// x[c,i,:] = (layout[c,i] >= 0 ? fixed tokens : ctx row) + pos[i]      (trainers/mvlpt.py:439-515 + :107/:112)
__global__ void assemble_prompts_kernel(const float* __restrict__ prefix, const float* __restrict__ suffix,
                                        const float* __restrict__ ctx, int ctx_per_class, int n_ctx,
                                        const int32_t* __restrict__ layout, const float* __restrict__ pos,
                                        float* __restrict__ x, int C, int L, int d) {
  const int d4 = d / 4;
  const size_t total = (size_t)C * L * d4;
  const int suf_len = L - 1 - n_ctx;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % d4);
    const size_t tok = i / d4;
    const int pos_i = (int)(tok % L), cls = (int)(tok / L);
    const int e = layout[tok];
    const float* src;
    if (e == 0) src = prefix + (size_t)cls * d;
    else if (e > 0) src = suffix + ((size_t)cls * suf_len + (e - 1)) * d;
    else src = ctx + ((size_t)(ctx_per_class ? cls * n_ctx : 0) + (-e - 1)) * d;
    *(f32x4*)(x + tok * d + c4 * 4) = *(const f32x4*)(src + c4 * 4) + *(const f32x4*)(pos + (size_t)pos_i * d + c4 * 4);
  }
}
hipError_t launch_assemble_prompts(const float* prefix, const float* suffix, const float* ctx, int ctx_per_class, int n_ctx,
                                   const int32_t* layout, const float* pos, float* x, int C, int L, int d, hipStream_t s) {
  if (d % 4) return hipErrorInvalidValue;
  const size_t total = (size_t)C * L * (d / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(assemble_prompts_kernel, dim3(grid), dim3(256), 0, s, prefix, suffix, ctx, ctx_per_class, n_ctx, layout,
                     pos, x, C, L, d);
  return hipGetLastError();
}

__global__ void build_ctx_pos_kernel(const int32_t* __restrict__ layout, int32_t* __restrict__ ctx_pos, int C, int L, int n_ctx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * L) return;
  const int e = layout[i];
  if (e < 0) ctx_pos[(i / L) * n_ctx + (-e - 1)] = i % L;
}
hipError_t launch_build_ctx_pos(const int32_t* layout, int32_t* ctx_pos, int C, int L, int n_ctx, hipStream_t s) {
  if (n_ctx <= 0) return hipSuccess;
  hipLaunchKernelGGL(build_ctx_pos_kernel, dim3((C * L + 255) / 256), dim3(256), 0, s, layout, ctx_pos, C, L, n_ctx);
  return hipGetLastError();
}
__global__ void eot_rows_kernel(const int32_t* __restrict__ eot, int32_t* __restrict__ rows, int C, int L) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) rows[c] = c * L + eot[c];
}
// dst[r] = src[idx[r]] (gather) or dst[idx[r]] = src[r] (scatter); rows of row_bytes (a multiple of 16) bytes
__global__ void copy_rows_kernel(const char* __restrict__ src, char* __restrict__ dst, const int32_t* __restrict__ idx, int rows,
                                 int row_bytes, int scatter) {
  const int chunks = row_bytes / 16;
  const size_t total = (size_t)rows * chunks;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / chunks), c = (int)(i % chunks);
    const size_t far = (size_t)idx[r] * row_bytes + (size_t)c * 16, near = (size_t)r * row_bytes + (size_t)c * 16;
    if (scatter) *(f32x4*)(dst + far) = *(const f32x4*)(src + near);
    else *(f32x4*)(dst + near) = *(const f32x4*)(src + far);
  }
}
hipError_t launch_copy_rows(const void* src, void* dst, const int32_t* idx, int rows, int row_bytes, int scatter, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (row_bytes % 16) return hipErrorInvalidValue;
  const size_t total = (size_t)rows * (row_bytes / 16);
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid), dim3(256), 0, s, (const char*)src, (char*)dst, idx, rows, row_bytes, scatter);
  return hipGetLastError();
}

hipError_t launch_eot_rows(const int32_t* eot, int32_t* rows, int C, int L, hipStream_t s) {
  hipLaunchKernelGGL(eot_rows_kernel, dim3((C + 255) / 256), dim3(256), 0, s, eot, rows, C, L);
  return hipGetLastError();
}

// generic ctx: dctx[j,:] = inv * sum_c dx[c, pos(c,j), :]  (ctx is expanded over classes: trainers/mvlpt.py:455-456)
// class-specific (CSC): dctx[c,j,:] = inv * dx[c, pos(c,j), :]
__global__ void gather_ctx_grad_csc_kernel(const float* __restrict__ dx, const int32_t* __restrict__ ctx_pos, int L, int d,
                                           int n_ctx, float* __restrict__ dctx, const float* scale_dev) {
  const int j = blockIdx.y, cls = blockIdx.z;
  const int c0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (c0 >= d) return;
  const float inv = scale_dev ? scale_dev[1] : 1.0f;
  dctx[((size_t)cls * n_ctx + j) * d + c0] = inv * dx[((size_t)cls * L + ctx_pos[cls * n_ctx + j]) * d + c0];
}
// generic context: 16 waves per block, wave w sums classes w, w+16, ... (independent loads), lane = float4 column;
// partials are added in a fixed order through LDS (deterministic)
__global__ __launch_bounds__(1024) void gather_ctx_grad_kernel(const float* __restrict__ dx, const int32_t* __restrict__ ctx_pos,
                                                               int C, int L, int d, int n_ctx, float* __restrict__ dctx,
                                                               const float* scale_dev) {
  __shared__ f32x4 part[16][64];
  const int j = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const bool ok = c < d;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
#pragma unroll 8
    for (int cls = wave; cls < C; cls += 16)
      acc += *(const f32x4*)(dx + ((size_t)cls * L + ctx_pos[cls * n_ctx + j]) * d + c);
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && ok) {
    f32x4 t = part[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += part[w][lane];
    const float inv = scale_dev ? scale_dev[1] : 1.0f;
    *(f32x4*)(dctx + (size_t)j * d + c) = t * inv;
  }
}
hipError_t launch_gather_ctx_grad(const float* dx, const int32_t* ctx_pos, int C, int L, int d, int n_ctx, int per_class,
                                  float* dctx, const float* scale_dev, hipStream_t s) {
  if (n_ctx <= 0) return hipSuccess;
  if (d % 4) return hipErrorInvalidValue;
  if (per_class)
    hipLaunchKernelGGL(gather_ctx_grad_csc_kernel, dim3((d + 255) / 256, n_ctx, C), dim3(256), 0, s, dx, ctx_pos, L, d, n_ctx, dctx, scale_dev);
  else
    hipLaunchKernelGGL(gather_ctx_grad_kernel, dim3((d + 255) / 256, n_ctx), dim3(1024), 0, s, dx, ctx_pos, C, L, d, n_ctx, dctx, scale_dev);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ gradient scaling
// The backward is linear in the incoming gradient, so it is run on  2^k * dfeat  to keep 16-bit activation
// gradients in range (the reference's fp16 mode has no GradScaler, trainers/mvlpt.py:873,927-932, and simply
// underflows); prompt gradients are multiplied by 2^-k when they are reduced.  Single block, deterministic.
// stage 1: amax over many blocks (max is order-independent: atomicMax on the bit pattern of |v| is deterministic);
// stage 2: one thread turns it into the power-of-two scale.  scale_dev = {scale, 1/scale, amax bits (scratch)}.
__global__ __launch_bounds__(256) void grad_amax_kernel(const float* __restrict__ v, size_t n, float* scale_dev) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(v[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned*)(scale_dev + 2), __float_as_uint(m));   // m >= 0 (NaN/inf sort on top)
}
__global__ void grad_scale_finish_kernel(float target, float* scale_dev) {
  const float m = __uint_as_float(*(const unsigned*)(scale_dev + 2));
  float sc = 1.0f;
  if (m > 0.f && isfinite(m)) {
    int e; frexpf(m, &e);                      // m = f * 2^e, f in [0.5, 1)
    int et; frexpf(target, &et);
    int k = et - e; k = k > 60 ? 60 : (k < -60 ? -60 : k);
    sc = ldexpf(1.0f, k);
  }
  scale_dev[0] = sc;
  scale_dev[1] = 1.0f / sc;
}
// the step's own call (dfeat of the text / image tower: C x e or B x e values, a few hundred KB) as ONE launch of one workgroup:
// memset + 50-block amax + finish were three launches and ~25 us on the text tower's critical chain
__global__ __launch_bounds__(1024) void grad_scale_small_kernel(const float* __restrict__ v, int n, float target, float* scale_dev) {
  __shared__ float part[16];
  float m = 0.f;
  for (int i = threadIdx.x * 4; i < n; i += 4096) {
    if (i + 4 <= n) { const f32x4 q = *(const f32x4*)(v + i); m = fmaxf(fmaxf(m, fmaxf(fabsf(q[0]), fabsf(q[1]))), fmaxf(fabsf(q[2]), fabsf(q[3]))); }
    else for (int k = i; k < n; ++k) m = fmaxf(m, fabsf(v[k]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    // NaN / inf must surface as in the two-stage version (atomicMax on the bit pattern): compare bit patterns of the non-negative values
    unsigned mb = 0;
    for (int w = 0; w < 16; ++w) { const unsigned b = __float_as_uint(part[w]); mb = b > mb ? b : mb; }
    const float mm = __uint_as_float(mb);
    float sc = 1.0f;
    if (mm > 0.f && isfinite(mm)) {
      int e; frexpf(mm, &e);
      int et; frexpf(target, &et);
      int k = et - e; k = k > 60 ? 60 : (k < -60 ? -60 : k);
      sc = ldexpf(1.0f, k);
    }
    scale_dev[0] = sc; scale_dev[1] = 1.0f / sc; scale_dev[2] = mm;
  }
}
hipError_t launch_grad_scale(const float* v, size_t n, float target, float* scale_dev, hipStream_t s) {
  if (n <= (1u << 17) && ((size_t)v & 15) == 0) {      // (C x e, B x e of the step; the load-time calls on whole weight matrices take the wide path)
    hipLaunchKernelGGL(grad_scale_small_kernel, dim3(1), dim3(1024), 0, s, v, (int)n, target, scale_dev);
    return hipGetLastError();
  }
  hipError_t e = hipMemsetAsync(scale_dev + 2, 0, sizeof(float), s);
  if (e != hipSuccess) return e;
  const size_t want = (n + 1023) / 1024;
  hipLaunchKernelGGL(grad_amax_kernel, dim3((unsigned)(want < 512 ? (want ? want : 1) : 512)), dim3(256), 0, s, v, n, scale_dev);
  hipLaunchKernelGGL(grad_scale_finish_kernel, dim3(1), dim3(1), 0, s, target, scale_dev);
  return hipGetLastError();
}

// LayerNorm folding (kernels.h, GemmArgs::fold_*): the two per-column vectors of a linear layer that follows a LayerNorm,
// colsum[n] = sum_k W[n,k] gamma[k] and bias2[n] = b[n] + sum_k W[n,k] beta[k], from the packed 16-bit weight (exactly the values
// the MFMA multiplies).  One wave per output column, fp32, fixed summation order.  Load time only.
template <typename T>
__global__ __launch_bounds__(256) void fold_vectors_kernel(const T* __restrict__ W, int ld, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ b,
                                                           float* __restrict__ colsum, float* __restrict__ bias2, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const T* w = W + (size_t)n * ld;
  float sg = 0.f, sb = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float x = to_f32<T>(w[k]);
    sg += x * gamma[k];
    sb += x * beta[k];
  }
  sg = wave_sum(sg); sb = wave_sum(sb);
  if (lane == 0) { colsum[n] = sg; bias2[n] = (b ? b[n] : 0.f) + sb; }
}
hipError_t launch_fold_vectors(int dtype, const void* W16, int ld, const float* gamma, const float* beta, const float* b,
                               float* colsum, float* bias2, int N, int K, hipStream_t s) {
  if (N <= 0 || K <= 0) return hipErrorInvalidValue;
  if (dtype == DT_F16) hipLaunchKernelGGL(fold_vectors_kernel<f16>, dim3((N + 3) / 4), dim3(256), 0, s, (const f16*)W16, ld, gamma, beta, b, colsum, bias2, N, K);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(fold_vectors_kernel<bf16>, dim3((N + 3) / 4), dim3(256), 0, s, (const bf16*)W16, ld, gamma, beta, b, colsum, bias2, N, K);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// Packed residual stream (common.h respk_*, kernels.h EPI_RESIDP_LN): the LayerNorm's gamma moves from the activation into the
// consumer's weight, Wg[n,k] = round16(W[n,k] * gamma[k]) (the values the MFMA multiplies), colsum[n] = sum_k Wg[n,k] in fp32.
// One wave per output column, fixed summation order.  Load time only.
__global__ __launch_bounds__(256) void fold_weight_kernel(const f16* __restrict__ W, int ld, const float* __restrict__ gamma,
                                                          f16* __restrict__ Wg, int ldg, float* __restrict__ colsum, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const f16* w = W + (size_t)n * ld;
  f16* o = Wg + (size_t)n * ldg;
  float sg = 0.f;
  for (int k = lane; k < K; k += 64) {
    float x = (float)w[k] * gamma[k];
    asm volatile("" : "+v"(x));      // the fp32 product, then ONE conversion (not v_fma_mixlo_f16's single rounding of the exact product)
    const f16 v = (f16)x;
    o[k] = v;
    sg += (float)v;
  }
  sg = wave_sum(sg);
  if (lane == 0) colsum[n] = sg;
}
hipError_t launch_fold_weight(const void* W16, int ld, const float* gamma, void* Wg16, int ldg, float* colsum, int N, int K, hipStream_t s) {
  if (N <= 0 || K <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fold_weight_kernel, dim3((N + 3) / 4), dim3(256), 0, s, (const f16*)W16, ld, gamma, (f16*)Wg16, ldg, colsum, N, K);
  return hipGetLastError();
}

// fp32 rows -> packed stream (hi [rows,d] fp16 + lo [rows,d] bytes) and the LayerNorm-folding partials of the rows: slot 0 of
// row r = {sum x, sum x^2} over the whole row, slots 1 .. ntp-1 = 0 (the consumer adds the slots in order).  One wave per row.
__global__ __launch_bounds__(256) void respk_pack_rows_kernel(const float* __restrict__ x, f16* __restrict__ hi, uint8_t* __restrict__ lo,
                                                              float* __restrict__ part, int ntp, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t o = (size_t)row * d;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)(x + o + c));
    s1 += (v[0] + v[1]) + (v[2] + v[3]);
    s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    f16x4 h;
    const uint32_t l = respk_split4(v, h);
    *(f16x4*)(hi + o + c) = h;
    *(uint32_t*)(lo + o + c) = l;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (part && lane < ntp) *(float2*)(part + ((size_t)row * ntp + lane) * 2) = lane == 0 ? float2{s1, s2} : float2{0.f, 0.f};
}
hipError_t launch_respk_pack_rows(const float* x, void* hi, uint8_t* lo, float* part, int ntp, int rows, int d, hipStream_t s) {
  if (rows <= 0 || d <= 0 || (d & 3) || ntp > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(respk_pack_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, (f16*)hi, lo, part, ntp, rows, d);
  return hipGetLastError();
}
// out[r, :] = value of packed row r * row_mul  (fp32; the CLS rows in front of the last block / ln_post)
__global__ __launch_bounds__(256) void respk_unpack_rows_kernel(const f16* __restrict__ hi, const uint8_t* __restrict__ lo, int row_mul,
                                                                float* __restrict__ out, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t o = (size_t)row * row_mul * d;
  for (int c = lane * 4; c < d; c += 256)
    *(f32x4*)(out + (size_t)row * d + c) = respk_join4(*(const f16x4*)(hi + o + c), *(const uint32_t*)(lo + o + c));
}
hipError_t launch_respk_unpack_rows(const void* hi, const uint8_t* lo, int row_mul, float* out, int rows, int d, hipStream_t s) {
  if (rows <= 0 || d <= 0 || (d & 3) || row_mul < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(respk_unpack_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, (const f16*)hi, lo, row_mul, out, rows, d);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ head (fp32)
// xn = x / ||x||  (no epsilon: trainers/mvlpt.py:550-551)
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ xn,
                                                             float* __restrict__ norm, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = x + (size_t)row * d;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += p[c] * p[c];
  const float nm = sqrtf(wave_sum(s));
  for (int c = lane; c < d; c += 64) xn[(size_t)row * d + c] = p[c] / nm;
  if (lane == 0) norm[row] = nm;
}
hipError_t launch_normalize_rows(const float* x, float* xn, float* norm, int rows, int d, hipStream_t s) {
  hipLaunchKernelGGL(normalize_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, xn, norm, rows, d);
  return hipGetLastError();
}

// logits[b,c] = (scale * imn[b,:]) . txn[c,:]  * [lo[b] <= c < hi[b]]      (trainers/mvlpt.py:553-554, 573-581)
__global__ __launch_bounds__(256) void logits_kernel(const float* __restrict__ imn, const float* __restrict__ txn, float scale,
                                                     const int32_t* __restrict__ lo, const int32_t* __restrict__ hi,
                                                     float* __restrict__ logits, int B, int C, int e) {
  extern __shared__ float simg[];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < e; i += 256) simg[i] = scale * imn[(size_t)b * e + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = blockIdx.y * 4 + wave; c < C; c += gridDim.y * 4) {
    const float* t = txn + (size_t)c * e;
    float s = 0.f;
    for (int i = lane; i < e; i += 64) s += simg[i] * t[i];
    s = wave_sum(s);
    if (lane == 0) {
      if (lo && !(c >= lo[b] && c < hi[b])) s *= 0.0f;   // multiplicative 0/1 mask, as in the reference
      logits[(size_t)b * C + c] = s;
    }
  }
}
hipError_t launch_logits(const float* imn, const float* txn, float scale, const int32_t* lo, const int32_t* hi, float* logits,
                         int B, int C, int e, hipStream_t s) {
  int gy = (C + 3) / 4; gy = gy > 64 ? 64 : gy;
  hipLaunchKernelGGL(logits_kernel, dim3(B, gy), dim3(256), e * sizeof(float), s, imn, txn, scale, lo, hi, logits, B, C, e);
  return hipGetLastError();
}

// F.cross_entropy(logits, label), mean over the batch (trainers/mvlpt.py:931); int64 class ids or fp32
// probability rows.  One wave per row; the mean is a second, single-wave deterministic pass.
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const void* __restrict__ labels,
                                                      int label_kind, int B, int C, float* __restrict__ row_loss,
                                                      float* __restrict__ dlogits, float* __restrict__ row_correct) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* z = logits + (size_t)b * C;
  float mx = -INFINITY; int arg = 0;
  for (int c = lane; c < C; c += 64) if (z[c] > mx) { mx = z[c]; arg = c; }
  // arg-max with lowest-index tie break across the wave
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(arg, o, 64);
    if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
  }
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += __expf(z[c] - mx);
  se = wave_sum(se);
  const float lse = mx + __logf(se);
  float loss = 0.f, ysum = 0.f; int target = -1;
  if (label_kind == 0) {
    target = (int)((const int64_t*)labels)[b];
    ysum = 1.0f;
    loss = lse - z[target];
  } else {
    const float* y = (const float*)labels + (size_t)b * C;
    float ymx = -INFINITY; int yarg = 0;
    for (int c = lane; c < C; c += 64) {
      ysum += y[c]; loss += y[c] * (lse - z[c]);
      if (y[c] > ymx) { ymx = y[c]; yarg = c; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(ymx, o, 64); const int oa = __shfl_xor(yarg, o, 64);
      if (om > ymx || (om == ymx && oa < yarg)) { ymx = om; yarg = oa; }
    }
    ysum = wave_sum(ysum); loss = wave_sum(loss);
    target = yarg;                                 // training accuracy uses argmax(label), trainers/mvlpt.py:935-936
  }
  if (lane == 0) { row_loss[b] = loss; if (row_correct) row_correct[b] = arg == target ? 1.f : 0.f; }
  if (dlogits) {
    const float invB = 1.0f / (float)B;
    for (int c = lane; c < C; c += 64) {
      const float p = __expf(z[c] - lse);
      const float y = label_kind == 0 ? (c == target ? 1.f : 0.f) : ((const float*)labels)[(size_t)b * C + c];
      dlogits[(size_t)b * C + c] = (p * ysum - y) * invB;
    }
  }
}
__global__ void ce_mean_kernel(const float* __restrict__ row_loss, const float* __restrict__ row_correct, int B,
                               float* __restrict__ loss, float* __restrict__ ncorrect) {
  const int lane = threadIdx.x;
  float s = 0.f, k = 0.f;
  for (int b = lane; b < B; b += 64) { s += row_loss[b]; if (row_correct) k += row_correct[b]; }
  s = wave_sum(s); k = wave_sum(k);
  if (lane == 0) { loss[0] = s / (float)B; if (ncorrect) ncorrect[0] = k; }
}
hipError_t launch_cross_entropy(const float* logits, const void* labels, int label_kind, int B, int C, float* row_loss,
                                float* loss, float* dlogits, float* ncorrect, hipStream_t s) {
  float* row_correct = ncorrect ? row_loss + B : nullptr;   // caller provides 2*B floats of scratch
  hipLaunchKernelGGL(ce_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, s, logits, labels, label_kind, B, C, row_loss, dlogits, row_correct);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(64), 0, s, row_loss, row_correct, B, loss, ncorrect);
  return hipGetLastError();
}

// d imn = scale * (dlogits*mask) txn ; d txn = scale * (dlogits*mask)^T imn ; then through x/||x||.
// One workgroup per output row (image b / class c); its 8 waves split the reduction axis (classes / images: 32 dependent row loads
// per wave at B = 256 instead of 64 with four waves — the kernel is a latency chain, 36 -> ~20 us stand-alone), every
// lane owns 8 consecutive feature columns (e <= 1024 = 2 x 64 lanes x 8), partial rows are combined through LDS in wave order.
template <bool IMG>
__global__ __launch_bounds__(512) void logits_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ imn,
                                                         const float* __restrict__ txn, const float* __restrict__ norm,
                                                         float scale, const int32_t* __restrict__ lo,
                                                         const int32_t* __restrict__ hi, float* __restrict__ dout,
                                                         int B, int C, int e) {
  __shared__ float part[8][1024];
  __shared__ float red[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x;                       // b (IMG) or c
  const float* other = IMG ? txn : imn;             // rows of the other side, indexed by the reduction index
  const float* self = (IMG ? imn : txn) + (size_t)row * e;
  const int nred = IMG ? C : B;
  f32x4 acc[2][2];
#pragma unroll
  for (int k = 0; k < 2; ++k) { acc[k][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int r = wave; r < nred; r += 8) {
    const int b = IMG ? row : r, c = IMG ? r : row;
    if (lo && !(c >= lo[b] && c < hi[b])) continue;     // multiplicative 0/1 task mask
    const float w = scale * dl[(size_t)b * C + c];
    const float* o = other + (size_t)r * e;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = (k * 64 + lane) * 8;
      if (i < e) {
        acc[k][0] += w * *(const f32x4*)(o + i);
        acc[k][1] += w * *(const f32x4*)(o + i + 4);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = (k * 64 + lane) * 8;
    if (i < e) { *(f32x4*)&part[wave][i] = acc[k][0]; *(f32x4*)&part[wave][i + 4] = acc[k][1]; }
  }
  __syncthreads();
  float g[2], dot = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = threadIdx.x + k * 512;
    g[k] = 0.f;
    if (i < e) {
      g[k] = ((part[0][i] + part[1][i]) + (part[2][i] + part[3][i])) + ((part[4][i] + part[5][i]) + (part[6][i] + part[7][i]));
      dot += g[k] * self[i];
    }
  }
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
  const float inv = 1.0f / norm[row];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = threadIdx.x + k * 512;
    if (i < e) dout[(size_t)row * e + i] = (g[k] - self[i] * dot) * inv;
  }
}
hipError_t launch_logits_bwd(const float* dlogits, const float* imn, const float* txn, const float* inorm, const float* tnorm,
                             float scale, const int32_t* lo, const int32_t* hi, float* dimg, float* dtxt, int B, int C, int e,
                             hipStream_t s) {
  if (e > 1024 || e % 8) return hipErrorInvalidValue;
  if (dimg) hipLaunchKernelGGL(logits_bwd_kernel<true>, dim3(B), dim3(512), 0, s, dlogits, imn, txn, inorm, scale, lo, hi, dimg, B, C, e);
  if (dtxt) hipLaunchKernelGGL(logits_bwd_kernel<false>, dim3(C), dim3(512), 0, s, dlogits, imn, txn, tnorm, scale, lo, hi, dtxt, B, C, e);
  return hipGetLastError();
}

}  // namespace mvlpt
