// Probe for the mixed-precision split product  (hi + lo) * W^T  =  hi*W16^T [fp16 MFMA]  +  lo8*W8^T [fp8 scaled MFMA]:
//   part 1: operand layout / scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 and the error of the mixed product
//           against a double reference (one wave, A [16,128] x W [16,128]^T);
//   part 2: sustained rate of the three K=128 recipes on pseudo-random operands (register-only loops):
//           0 = 4 x f16 16x16x32 (today's hi|lo pair), 1 = 2 x... no: per K=128 of ALGORITHMIC work the pair costs 8 f16
//           MFMAs (4 hi + 4 lo), the mixed recipe 4 f16 + 1 fp8, fp8 alone 1.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_mix_probe tools/mfma_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned pack4_fp8(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}
__device__ inline unsigned pack4_bf8(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
  return (unsigned)w;
}

// A [16,128] fp32, W [16,128] fp32 (fp16-representable).  out[0]: pair product, out[1]: mixed product (lo in e5m2 * 2^LO_EXP,
// W8 in e4m3 * 2^w_exp), out[2]: hi only.   D[m][n] = sum_k A[m][k] W[n][k]
constexpr int LO_EXP = 10;
__global__ void layout_probe(const float* A, const float* W, float* out, int w_exp, int lo_fmt) {
  const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
  f32x4 acc_pair = {0, 0, 0, 0}, acc_mix = {0, 0, 0, 0}, acc_hi = {0, 0, 0, 0};
  // f16 passes: k-step s covers k in [32 s, 32 s + 32), lane holds k = 32 s + 8 fg + e
  for (int s = 0; s < 4; ++s) {
    f16x8 ah, al, w;
    for (int e = 0; e < 8; ++e) {
      const float x = A[fr * 128 + 32 * s + 8 * fg + e];
      const _Float16 h = (_Float16)x;
      ah[e] = h; al[e] = (_Float16)(x - (float)h);
      w[e] = (_Float16)W[fr * 128 + 32 * s + 8 * fg + e];
    }
    // swapped operands as in gemm.hip: D = Wfrag x Afrag -> lane holds D[m = fr][n = 4 fg + e]
    acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, ah, acc_hi, 0, 0, 0);
    acc_pair = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, ah, acc_pair, 0, 0, 0);
    acc_pair = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, al, acc_pair, 0, 0, 0);
    acc_mix = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, ah, acc_mix, 0, 0, 0);
  }
  // fp8 pass: lane holds k = 32 fg + 4 r + b (register r, byte b) of row fr, on both operands
  i32x8 a8, w8;
  for (int r = 0; r < 8; ++r) {
    float lo[4], ww[4];
    for (int b = 0; b < 4; ++b) {
      const float x = A[fr * 128 + 32 * fg + 4 * r + b];
      lo[b] = (x - (float)(_Float16)x) * exp2f((float)LO_EXP);
      ww[b] = W[fr * 128 + 32 * fg + 4 * r + b] * exp2f((float)w_exp);
    }
    a8[r] = lo_fmt == 1 ? pack4_bf8(lo[0], lo[1], lo[2], lo[3]) : pack4_fp8(lo[0], lo[1], lo[2], lo[3]);
    w8[r] = pack4_fp8(ww[0], ww[1], ww[2], ww[3]);
  }
  const int sa = 127 - w_exp, sb = 127 - LO_EXP;      // e8m0 scale bytes (byte 0 selected by opsel 0)
  if (lo_fmt == 1)
    acc_mix = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8, a8, acc_mix, 0, 1, 0, sa, 0, sb);
  else
    acc_mix = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8, a8, acc_mix, 0, 0, 0, sa, 0, sb);
  for (int e = 0; e < 4; ++e) {
    out[0 * 256 + fr * 16 + 4 * fg + e] = acc_pair[e];
    out[1 * 256 + fr * 16 + 4 * fg + e] = acc_mix[e];
    out[2 * 256 + fr * 16 + 4 * fg + e] = acc_hi[e];
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void rate_loop(float* out, int iters) {
  f16x8 a[8], b[8];
  i32x8 a8[4], b8[4];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int r = 0; r < 8; ++r)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u;
      a[r][i] = (_Float16)(((int)(h >> 9) % 2001 - 1000) * 1e-3f);
      h = h * 1664525u + 1013904223u;
      b[r][i] = (_Float16)(((int)(h >> 9) % 2001 - 1000) * 1e-3f);
    }
  for (int r = 0; r < 4; ++r)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; a8[r][i] = (int)(h & 0x7f7f7f7fu) | (int)(h << 3 & 0x80808080u);
      h = h * 1664525u + 1013904223u; b8[r][i] = (int)(h & 0x77777777u);
    }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    // one "unit" = K = 128 of algorithmic work on 8 accumulator tiles
    if constexpr (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) & 7], b[(i + 3 * r) & 7], acc[i], 0, 0, 0);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) & 7], b[(i + 3 * r) & 7], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i & 3], b8[(i + 1) & 3], acc[i], 0, 1, 0, 120, 0, 117);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i & 3], b8[(i + 1) & 3], acc[i], 0, 1, 0, 120, 0, 117);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void rate(const char* name, double seconds, int cus) {
  float* out; hipMalloc(&out, 4);
  const int iters = MODE == 2 ? 40000 : 5000, wpc = 8;
  hipLaunchKernelGGL(rate_loop<MODE>, dim3(cus), dim3(64 * wpc), 0, 0, out, iters);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  long n = 0; double el = 0;
  while (el < seconds) {
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(rate_loop<MODE>, dim3(cus), dim3(64 * wpc), 0, 0, out, iters);
    hipDeviceSynchronize();
    n += 5;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double units = (double)cus * wpc * iters * 8.0 * n;          // (16x16 tile, K = 128) units
  printf("%-34s %.3f ns per (16x16xK128) unit per CU-wave-slot; algorithmic %.1f TFLOP/s (2*16*16*128 per unit)\n", name,
         el / (iters * 8.0 * n) * 1e9, units * 2 * 16 * 16 * 128 / el / 1e12);
  hipFree(out);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  // ---- part 1
  std::vector<float> A(16 * 128), W(16 * 128), out(3 * 256);
  unsigned h = 777;
  auto rnd = [&]() { h = h * 1664525u + 1013904223u; return ((int)(h >> 9) % 20001 - 10000) * 1e-4f; };
  for (auto& v : A) v = rnd() * 3.0f + 0.37f * rnd() * rnd();
  for (auto& v : W) v = (float)(_Float16)(rnd() * 0.11f);
  int w_exp = 11;                                  // max |W| 0.11 * 2^11 = 225 < 448
  float *dA, *dW, *dO;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dO, out.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  for (int fmt = 0; fmt < 2; ++fmt) {
    hipLaunchKernelGGL(layout_probe, dim3(1), dim3(64), 0, 0, dA, dW, dO, w_exp, fmt);
    hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost);
    double e[3] = {0, 0, 0}, ref_max = 0;
    for (int m = 0; m < 16; ++m)
      for (int n = 0; n < 16; ++n) {
        double r = 0;
        for (int k = 0; k < 128; ++k) r += (double)A[m * 128 + k] * (double)W[n * 128 + k];
        ref_max = fmax(ref_max, fabs(r));
        for (int v = 0; v < 3; ++v) e[v] = fmax(e[v], fabs(out[v * 256 + m * 16 + n] - r));
      }
    printf("lo format %s: max|err| / max|ref|:  pair %.3e   mixed %.3e   hi-only %.3e   (mixed should sit near pair, far below hi-only)\n",
           fmt ? "e5m2" : "e4m3", e[0] / ref_max, e[1] / ref_max, e[2] / ref_max);
  }
  // ---- part 2
  rate<0>("pair: 8 x f16 16x16x32", seconds, cus);
  rate<1>("mixed: 4 x f16 + 1 x fp8 K128", seconds, cus);
  rate<2>("fp8 K128 only", seconds, cus);
  return 0;
}
