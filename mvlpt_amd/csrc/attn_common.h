// Device helpers shared by the attention kernels (attention.hip: whole K/V resident in LDS, short sequences;
// attention_stream.hip: K/V (or Q/dO) streamed through a double-buffered LDS ring in 64-row chunks).
#pragma once
#include "kernels.h"

namespace mvlpt {

// ---- staging helpers -------------------------------------------------------------------------------------
// Row-major [LP][64] 16-bit LDS image with 128-B rows whose 16-B chunks are XOR-swizzled by (row & 7), filled by
// LDS-DMA (global_load_lds, 16 B per lane, no VGPR round trip, all slabs in flight at once):
// one wave-instruction writes a lane-linear 1 KiB slab = 8 rows x 128 B, so the chunk swizzle is applied to the
// per-lane SOURCE address.  Rows >= L re-read row L-1 (finite values; they only ever meet P = 0 / masked scores).
// The caller waits with s_waitcnt vmcnt(0) + barrier before the first ds_read.
template <typename T>
__device__ __forceinline__ void stage_rows_dma(char* dst, const T* src, size_t ld, int L, int LP, int wave, int lane, bool nt = false) {
  const int srow = lane >> 3, chunk = (lane & 7) ^ srow;
  for (int sl = wave; sl < LP / 8; sl += 4) {
    int row = sl * 8 + srow;
    row = row < L ? row : L - 1;
    if (nt) glds16_nt(src + (size_t)row * ld + chunk * 8, dst + sl * 1024);
    else glds16(src + (size_t)row * ld + chunk * 8, dst + sl * 1024);
  }
}
// LDS-DMA issued from inline assembly.  With the builtin, the compiler's wait-count pass knows an LDS write is in
// flight and puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 — which would serialise the DMA of chunk
// c+1 behind the arithmetic on chunk c.  Through asm the only waits are the counted ones written in the kernels
// (the memory clobber keeps the compiler from moving LDS reads across them).  M0 carries the wave-uniform LDS base.
template <int BYTES>
__device__ __forceinline__ void dma_raw(const void* gsrc, const void* lds_wave_base) {
  const unsigned m0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_wave_base);
  if constexpr (BYTES == 16)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m0) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(m0) : "memory", "m0");
}

// A-operand fragment of a row-major swizzled image: rows tile*16 + (lane&15), k-step ks (32 wide)
template <typename T>
__device__ __forceinline__ typename Vec<T>::v8 frag_rows(const char* img, int tile, int ks, int fr, int fg) {
  return *(const typename Vec<T>::v8*)(img + (tile * 16 + fr) * 128 + (((ks * 4 + fg) ^ (fr & 7)) * 16));
}
template <typename T>
__device__ __forceinline__ typename Vec<T>::v8 pack8(const f32x4& a, const f32x4& b) {
  typename Vec<T>::v8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = from_f32<T>(a[e]); r[e + 4] = from_f32<T>(b[e]); }
  return r;
}
// Hardware transpose read (gfx950 ds_read_b64_tr_b16).  Each lane passes the address of 4 contiguous 16-bit
// elements; within a 16-lane group, lanes 4r..4r+3 supply row r (16 columns) of a 4x16 block and lane i receives
// column i of that block (4 rows).  It turns a ROW-major [key][d] LDS image of V into the k-slot-major operand
// the MFMA wants (lane = d column, elements = 4 consecutive keys) with no transposed staging pass.
typedef __fp16 hv4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __bf16 bv4_t __attribute__((__vector_size__(4 * sizeof(__bf16))));
template <typename T> __device__ __forceinline__ typename Vec<T>::v4 tr_read4(const char* p);
template <> __device__ __forceinline__ f16x4 tr_read4<f16>(const char* p) {
  hv4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hv4_t*)p);
  return __builtin_bit_cast(f16x4, r);
}
template <> __device__ __forceinline__ bf16x4 tr_read4<bf16>(const char* p) {
  bv4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bv4_t*)p);
  return __builtin_bit_cast(bf16x4, r);
}
// A-operand fragment "V^T rows dt*16 + (lane&15), k-slots of 32-key block kb" out of the row-major swizzled V image
template <typename T>
__device__ __forceinline__ typename Vec<T>::v8 frag_vt(const char* img, int kb, int dt, int fr, int fg) {
  const int koff = fg * 4 + (fr >> 2);                  // key inside the 16-key tile supplied by this lane
  const int chunk = (dt * 2 + ((fr & 3) >> 1)) ^ (koff & 7);
  const char* p = img + (kb * 32 + koff) * 128 + chunk * 16 + (fr & 1) * 8;
  const typename Vec<T>::v4 lo = tr_read4<T>(p);
  const typename Vec<T>::v4 hi = tr_read4<T>(p + 16 * 128);   // second 16-key tile of the block (same key&7)
  typename Vec<T>::v8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = lo[e]; r[e + 4] = hi[e]; }
  return r;
}
// same for a 32-key block whose SECOND 16-key tile does not exist (odd tile count): the upper k-slots re-read the first
// tile (finite values) and must meet P = 0
template <typename T>
__device__ __forceinline__ typename Vec<T>::v8 frag_vt_half(const char* img, int kb, int dt, int fr, int fg) {
  const int koff = fg * 4 + (fr >> 2);
  const int chunk = (dt * 2 + ((fr & 3) >> 1)) ^ (koff & 7);
  const char* p = img + (kb * 32 + koff) * 128 + chunk * 16 + (fr & 1) * 8;
  const typename Vec<T>::v4 lo = tr_read4<T>(p);
  typename Vec<T>::v8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = lo[e]; r[e + 4] = lo[e]; }
  return r;
}
__device__ __forceinline__ float quad_sum(float v) {  // reduce over the four lanes sharing lane&15
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

}  // namespace mvlpt
