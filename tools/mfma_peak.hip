// Sustained MFMA-only loop (no LDS, no global traffic inside the loop): what the board delivers on fp16 matrix work at its
// power cap.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_peak tools/mfma_peak.hip ; run under tools/power_probe.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, int vary) {
  // operands: eight different register sets with pseudo-random fp16 values (vary = 1: realistic bit toggling between
  // consecutive MFMAs) or one constant set (vary = 0: the multipliers see the same bits every cycle)
  f16x8 a[8], b[8];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int r = 0; r < 8; ++r)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u;
      const float x = ((int)(h >> 9) % 2001 - 1000) * 1e-3f;
      h = h * 1664525u + 1013904223u;
      const float y = ((int)(h >> 9) % 2001 - 1000) * 1e-3f;
      a[r][i] = (_Float16)(vary ? x : 0.5f);
      b[r][i] = (_Float16)(vary ? y : 0.25f);
    }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) & 7], b[(i + 3 * r) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;      // never true: keeps the loop alive
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
  const int waves_per_cu = argc > 2 ? atoi(argv[2]) : 8;
  const int vary = argc > 3 ? atoi(argv[3]) : 1;
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; hipMalloc(&out, 4);
  const int iters = 20000;                                    // 32 MFMAs per iteration and wave
  const double flop_per_launch = (double)cus * waves_per_cu * iters * 32.0 * 16 * 16 * 32 * 2;
  hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(64 * waves_per_cu), 0, 0, out, iters, vary);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  int n = 0;
  double el = 0;
  while (el < seconds) {
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(64 * waves_per_cu), 0, 0, out, iters, vary);
    hipDeviceSynchronize();
    n += 10;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  printf("MFMA-only loop (%s operands): %d CUs x %d waves, %.2f s: %.1f TFLOP/s (fp16 16x16x32, fp32 accumulate)\n", vary ? "pseudo-random" : "constant", cus, waves_per_cu, el, flop_per_launch * n / el / 1e12);
  return 0;
}
