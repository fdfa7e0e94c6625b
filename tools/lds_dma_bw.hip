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

// VERDICT r5 item 2 at the level of the memory system: "weight fragments straight from L2 into registers; the whole LDS ring for A".
// Per 64-deep K-stage a wave requests its four A pieces by LDS-DMA (32 KiB per workgroup, ADEPTH stages in flight in a ring of 32-KiB
// slots) and its OWN 64 weight columns in the MFMA fragment layout by eight global_load_dwordx4 (lane l: weight row wn*64 + j*16 + (l & 15),
// 16 bytes at k-offset ks*64 + (l >> 4)*16), double-buffered in registers: the fragments of stage s+1 are requested at the top of stage s
// (vmcnt retires in order, so they are requested BEFORE the A pieces of stage s+ADEPTH and waited for with those left in flight).  The two
// waves that share a column block (wm = 0 / 1) request the same weight bytes: 32 KiB of A + 64 KiB of weight requests per stage instead
// of 32 + 32.  No MFMA, no LDS reads: what the operand streams alone cost.
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int ADEPTH>
__global__ __launch_bounds__(512) void pull_bdirect_kernel(const char* __restrict__ A, const char* __restrict__ W, size_t a_rows, int pitch, int stages,
                                                           long long* cycles, int tilesN, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int srow = lane >> 3, scol = ((lane & 7) ^ srow) * 16;
  const int wn = wave & 3;
  const int G = gridDim.x, b = blockIdx.x, xcd = b & 7, gq = G >> 3;
  const int b_remap = xcd * gq + (b >> 3);
  i32x4 bf[2][8];
  int acc = 0;
  auto tile = [&](int s, size_t& arow, int& wrow) {
    const int t = (s / 12) * G + b_remap, tm = t / tilesN, tn = t - tm * tilesN;
    arow = ((size_t)tm * 256) % (a_rows - 256); wrow = tn * 256;
  };
  auto issue_b = [&](int s, i32x4 (&f)[8]) {
    size_t arow; int wrow; tile(s, arow, wrow);
    const int k = s % 12;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        f[j * 2 + ks] = *(const i32x4*)(W + (size_t)(wrow + wn * 64 + j * 16 + (lane & 15)) * pitch + k * 128 + ks * 64 + (lane >> 4) * 16);
  };
  auto issue_a = [&](int s) {
    size_t arow; int wrow; tile(s, arow, wrow);
    char* base = smem + (s % (ADEPTH + 1)) * 32768;
    const int k = s % 12;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16<0>(A + (arow + (i * 8 + wave) * 8 + srow) * pitch + k * 128 + scol, base + (i * 8 + wave) * 1024);
  };
  const long long t0 = __builtin_amdgcn_s_memtime();
  issue_b(0, bf[0]);
  for (int s = 0; s < ADEPTH; ++s) issue_a(s);
  // two stages per trip so that the register buffer a stage consumes / refills is a compile-time choice (a run-time select would make
  // the compiler wait for BOTH buffers, i.e. for the fragments just requested)
  auto stage = [&](int s, i32x4 (&cur)[8], i32x4 (&nxt)[8]) {
    if (s + 1 < stages) issue_b(s + 1, nxt);            // top of stage s: the fragments of stage s+1 ...
    if (s + ADEPTH < stages) issue_a(s + ADEPTH);        // ... then the A pieces of stage s+ADEPTH
#pragma unroll
    for (int q = 0; q < 8; ++q) acc += cur[q][0] ^ cur[q][3];      // "consume" this stage's fragments: they must have landed
    if (s + ADEPTH < stages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (ADEPTH - 1) + 8) : "memory");      // A(s+2..s+ADEPTH) + the fragments just requested stay in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  for (int s = 0; s < stages; s += 2) {
    stage(s, bf[0], bf[1]);
    if (s + 1 < stages) stage(s + 1, bf[1], bf[0]);
  }
  if (threadIdx.x == 0) cycles[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
  if (acc == 0x12345678) sink[0] = acc;
}
template <int ADEPTH>
static void run_bdirect(const char* what, const char* A, const char* W, size_t a_rows, int pitch, int stages, long long* cyc, int tilesN, int* sink) {
  const int G = 256;
  hipFuncSetAttribute((const void*)pull_bdirect_kernel<ADEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 32768);
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((pull_bdirect_kernel<ADEPTH>), dim3(G), dim3(512), 32768 * (ADEPTH + 1), 0, A, W, a_rows, pitch, stages, cyc, tilesN, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(G); hipMemcpy(h.data(), cyc, G * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= G;
    const double bytes = (double)G * stages * 65536;      // the same 64 KiB of DISTINCT operand bytes per stage as the LDS-only scheme
    if (rep) printf("%-58s A in flight %d: %.3f ms, %5.2f TB/s chip (distinct operand bytes), %5.0f ticks / stage, %4.1f B/tick/CU\n", what, ADEPTH, ms,
                    bytes / ms / 1e9, avg / stages, 65536.0 * stages / avg);
  }
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
  // item 2 of VERDICT r5: weight fragments by global_load_dwordx4 into registers, A alone in the LDS ring
  int* sink; hipMalloc(&sink, 64);
  run_bdirect<1>("weights -> registers, N = 2304 (QKV)", A, W, a_gemm, pitch, 12 * 6, cyc, 9, sink);
  run_bdirect<2>("weights -> registers, N = 2304 (QKV)", A, W, a_gemm, pitch, 12 * 6, cyc, 9, sink);
  run_bdirect<3>("weights -> registers, N = 2304 (QKV)", A, W, a_gemm, pitch, 12 * 6, cyc, 9, sink);
  run_bdirect<3>("weights -> registers, N = 3072 (MLP up)", A, W, a_gemm, pitch, 12 * 9, cyc, 12, sink);
  // everything L2-resident: A rows wrap inside 2 MB
  run<2, 0, 0, 0>("private A rows wrapping in 2 MB (L2-resident)", A, W, 1365 + 256, pitch, 480, cyc, 9);
  return 0;
}
