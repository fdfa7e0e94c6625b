// How fast can a CU pull operand tiles through LDS-DMA when it does nothing else?  The access pattern of the 256x256 GEMM main loop
// (gemm.hip): per "stage" every workgroup of 8 waves requests 64 KiB (8 x 1 KiB pieces per wave, 8 rows x 128 B each, row pitch `pitch`),
// waits for it and passes a barrier; DEPTH stages stay in flight.  Half of the bytes come from a panel the workgroup re-reads every 12
// stages (the weight), half stream through a large array (the activations).  Prints bytes / clock / CU and TB/s.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_bw.hip -o tools/_build/lds_dma_bw && tools/_build/lds_dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int AUX>
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, AUX);
}

// MODE 0: every workgroup streams its own A rows (nothing shared); MODE 1: the tile walk of gemm.hip (tiles N-fastest, 32 consecutive
// tiles per XCD and round: the workgroups that share an A row block run side by side on one XCD), M = 50432, N = tilesN * 256, K = 768
template <int DEPTH, int MODE, int AUX_A, int AUX_W>
__global__ __launch_bounds__(512) void pull_kernel(const char* __restrict__ A, const char* __restrict__ W, size_t a_rows, int pitch, int stages,
                                                   long long* cycles, int tilesN) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int srow = lane >> 3, scol = ((lane & 7) ^ srow) * 16;
  const size_t wg_row0 = ((size_t)blockIdx.x * 256) % (a_rows - 256);
  const int G = gridDim.x, b = blockIdx.x, xcd = b & 7, gq = G >> 3;
  const int b_remap = xcd * gq + (b >> 3);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < stages + DEPTH - 1; ++s) {
    if (s < stages) {
      char* base = smem + (s % DEPTH) * 65536;
      const int k = s % 12;
      // activations: a new 256-row block every 12 stages (streaming), weights: the same 256 rows again and again
      size_t arow = (wg_row0 + (size_t)(s / 12) * 256 * gridDim.x) % (a_rows - 256);
      int wrow = (blockIdx.x % 9) * 256;
      if (MODE == 1) {
        const int t = (s / 12) * G + b_remap, tm = t / tilesN, tn = t - tm * tilesN;
        arow = ((size_t)tm * 256) % (a_rows - 256); wrow = tn * 256;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16<AUX_A>(A + (arow + (i * 8 + wave) * 8 + srow) * pitch + k * 128 + scol, base + (i * 8 + wave) * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16<AUX_W>(W + ((size_t)(wrow + (i * 8 + wave) * 8 + srow)) * pitch + k * 128 + scol, base + 32768 + (i * 8 + wave) * 1024);
    }
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (s >= DEPTH - 1) { if (s < stages) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
  }
  if (threadIdx.x == 0) cycles[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

template <int DEPTH, int MODE, int AUX_A, int AUX_W>
static void run(const char* what, const char* A, const char* W, size_t a_rows, int pitch, int stages, long long* cyc, int tilesN) {
  const int G = 256;
  hipFuncSetAttribute((const void*)pull_kernel<DEPTH, MODE, AUX_A, AUX_W>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((pull_kernel<DEPTH, MODE, AUX_A, AUX_W>), dim3(G), dim3(512), 131072, 0, A, W, a_rows, pitch, stages, cyc, tilesN);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(G); hipMemcpy(h.data(), cyc, G * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= G;
    const double bytes = (double)G * stages * 65536;
    if (rep) printf("%-58s in flight %d: %.3f ms, %5.2f TB/s chip, %5.0f ticks / 64-KiB stage, %4.1f B/tick/CU\n", what, DEPTH, ms, bytes / ms / 1e9, avg / stages,
                    65536.0 * stages / avg);
  }
}

int main() {
  const int pitch = 1536;
  const size_t a_big = 50432 * 4, a_gemm = 50432 + 256;
  char *A, *W; long long* cyc;
  hipMalloc(&A, a_big * pitch); hipMalloc(&W, (size_t)12 * 256 * pitch); hipMalloc(&cyc, 256 * 8);
  hipMemset(A, 1, a_big * pitch); hipMemset(W, 1, (size_t)12 * 256 * pitch);
  run<1, 0, 0, 0>("private A rows (310 MB: HBM), 9 weight panels", A, W, a_big, pitch, 480, cyc, 9);
  run<2, 0, 0, 0>("private A rows (310 MB: HBM), 9 weight panels", A, W, a_big, pitch, 480, cyc, 9);
  run<1, 1, 0, 0>("GEMM tile walk, N = 2304 (QKV), 6.9 rounds", A, W, a_gemm, pitch, 12 * 6, cyc, 9);
  run<2, 1, 0, 0>("GEMM tile walk, N = 2304 (QKV), 6.9 rounds", A, W, a_gemm, pitch, 12 * 6, cyc, 9);
  run<2, 1, 0, 0>("GEMM tile walk, N = 3072 (MLP up), 9 rounds", A, W, a_gemm, pitch, 12 * 9, cyc, 12);
  run<2, 1, 2, 0>("GEMM tile walk, N = 2304, A non-temporal", A, W, a_gemm, pitch, 12 * 6, cyc, 9);
  run<2, 1, 0, 2>("GEMM tile walk, N = 2304, W non-temporal", A, W, a_gemm, pitch, 12 * 6, cyc, 9);
  run<2, 1, 1, 1>("GEMM tile walk, N = 2304, sc0 on both", A, W, a_gemm, pitch, 12 * 6, cyc, 9);
  // everything L2-resident: A rows wrap inside 2 MB
  run<2, 0, 0, 0>("private A rows wrapping in 2 MB (L2-resident)", A, W, 1365 + 256, pitch, 480, cyc, 9);
  return 0;
}
