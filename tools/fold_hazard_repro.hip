// Reproducer for the round-2 "LayerNorm folding" hazard: on ONE in-order stream a producer with many workgroups
// stores per-row partial sums, a TINY kernel (31 blocks) turns them into per-row statistics, a consumer with many workgroups
// reads the statistics — while a second stream keeps part of the chip busy.  Every value carries its iteration number, so a
// stale read names the iteration it came from.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/fold_hazard_repro tools/fold_hazard_repro.hip
//   fold_hazard_repro <iters> <busy_wgs (0: no 2nd stream)> <producer store: 0 plain 1 nontemporal> <consumer load: 0 plain 1 nontemporal 2 volatile(sc0 sc1)>
//                     <tiny store: 0 plain 1 nontemporal> <reuse: 1 = the statistics buffer is rewritten every iteration (as the tower did per layer)>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int R = 7700, NT = 4, TINY_BLOCKS = 31;
__device__ __forceinline__ unsigned val(unsigned it, unsigned r, unsigned t) { return it * 65536u + ((r * 7u + t) & 0xfffu); }

__global__ __launch_bounds__(256) void producer(unsigned* partial, float* bulk, unsigned it, int nt_store) {
  // a GEMM epilogue in miniature: each workgroup owns 32 rows of one column tile, streams a block of bulk output and its partials
  const int tile = blockIdx.x % NT, r0 = (blockIdx.x / NT) * 32;
  for (int i = threadIdx.x; i < 32 * 256; i += 256) __builtin_nontemporal_store((float)it, bulk + ((size_t)blockIdx.x * 32 * 256 + i));
  if (threadIdx.x < 32 && r0 + threadIdx.x < R) {
    unsigned* p = partial + (size_t)(r0 + threadIdx.x) * NT + tile;
    const unsigned v = val(it, r0 + threadIdx.x, tile);
    if (nt_store) __builtin_nontemporal_store(v, p); else *p = v;
  }
}
__global__ __launch_bounds__(256) void tiny_stats(const unsigned* partial, unsigned* stats, int nt_store) {
  for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
    unsigned s = 0;
    for (int t = 0; t < NT; ++t) s += partial[(size_t)r * NT + t];
    if (nt_store) __builtin_nontemporal_store(s, stats + r); else stats[r] = s;
  }
}
__global__ __launch_bounds__(256) void consumer(const unsigned* stats, unsigned it, int load_kind, unsigned* bad, unsigned* first) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  unsigned want = 0;
  for (int t = 0; t < NT; ++t) want += val(it, r, t);
  unsigned got;
  if (load_kind == 1) got = __builtin_nontemporal_load(stats + r);
  else if (load_kind == 2) got = *(const volatile unsigned*)(stats + r);
  else got = stats[r];
  if (got != want && atomicAdd(bad, 1u) == 0) { first[0] = it; first[1] = r; first[2] = got; first[3] = want; }
}
__global__ __launch_bounds__(512) void busy(float* buf, size_t n, int rounds) {      // the other tower: streams memory, holds its CUs
  float acc = 0.f;
  for (int k = 0; k < rounds; ++k)
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t)gridDim.x * 512) acc += buf[i];
  if (acc == 12345.678f) buf[0] = acc;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000, busy_wgs = argc > 2 ? atoi(argv[2]) : 96;
  const int p_nt = argc > 3 ? atoi(argv[3]) : 1, c_kind = argc > 4 ? atoi(argv[4]) : 0, t_nt = argc > 5 ? atoi(argv[5]) : 1;
  const int reuse = argc > 6 ? atoi(argv[6]) : 1;
  const int pgrid = ((R + 31) / 32) * NT;
  unsigned *partial, *stats, *bad, *first; float *bulk, *bbuf;
  const size_t bn = (size_t)64 << 20;
  hipMalloc(&partial, (size_t)R * NT * 4 * (reuse ? 1 : 64)); hipMalloc(&stats, (size_t)R * 4 * (reuse ? 1 : 64));
  hipMalloc(&bulk, (size_t)pgrid * 32 * 256 * 4); hipMalloc(&bbuf, bn * 4); hipMalloc(&bad, 4); hipMalloc(&first, 16);
  hipMemset(bad, 0, 4); hipMemset(first, 0, 16); hipMemset(bbuf, 0, bn * 4); hipMemset(stats, 0, (size_t)R * 4 * (reuse ? 1 : 64));
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  for (int it = 1; it <= iters; ++it) {
    const int slot = reuse ? 0 : it % 64;
    if (busy_wgs > 0 && it % 4 == 1) hipLaunchKernelGGL(busy, dim3(busy_wgs), dim3(512), 0, sb, bbuf, bn, 2);
    hipLaunchKernelGGL(producer, dim3(pgrid), dim3(256), 0, sa, partial + (size_t)slot * R * NT, bulk, (unsigned)it, p_nt);
    hipLaunchKernelGGL(tiny_stats, dim3(TINY_BLOCKS), dim3(256), 0, sa, partial + (size_t)slot * R * NT, stats + (size_t)slot * R, t_nt);
    hipLaunchKernelGGL(consumer, dim3((R + 255) / 256), dim3(256), 0, sa, stats + (size_t)slot * R, (unsigned)it, c_kind, bad, first);
  }
  hipDeviceSynchronize();
  unsigned hb = 0, hf[4];
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 16, hipMemcpyDeviceToHost);
  printf("iters %d busy_wgs %d producer_%s tiny_%s consumer_%s reuse %d: %u stale reads", iters, busy_wgs, p_nt ? "nt" : "plain",
         t_nt ? "nt" : "plain", c_kind == 2 ? "volatile" : (c_kind ? "nt" : "plain"), reuse, hb);
  if (hb) printf("; first: iteration %u row %u got 0x%08x (iteration field %u) want 0x%08x", hf[0], hf[1], hf[2], hf[2] >> 18, hf[3]);
  printf("\n");
  return 0;
}
