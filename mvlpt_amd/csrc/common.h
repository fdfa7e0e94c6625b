// Shared device/host helpers for the MI355X (gfx950, wave64) prompt-tuning kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvlpt {

typedef _Float16 f16;
typedef __bf16 bf16;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Vec;
template <> struct Vec<f16> { using v8 = f16x8; using v4 = f16x4; };
template <> struct Vec<bf16> { using v8 = bf16x8; using v4 = bf16x4; };

// D(16x16, f32) += A(16x32) * B(32x16), 16-bit inputs.  Operand register layout (wave64):
//   A: lane l holds A[row = l&15][k = 8*(l>>4) + 0..7]      B: lane l holds B[k = 8*(l>>4) + 0..7][col = l&15]
//   D: lane l holds D[row = 4*(l>>4) + r][col = l&15], r = 0..3
template <typename T>
__device__ __forceinline__ f32x4 mfma16(typename Vec<T>::v8 a, typename Vec<T>::v8 b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma16<f16>(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<bf16>(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ float to_f32(T v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return static_cast<T>(v); }

// Split-precision pair of a fp32 value: hi = round16(v), lo = round16(v - hi); hi + lo carries ~22 (fp16) / 16 (bf16) bits.
// `v` is materialised first: with fp-contract the compiler otherwise folds a producing multiply into v_fma_mix* for ONE of
// the two uses of hi (single rounding of the exact product) while the stored hi is converted from the rounded fp32 value
// (double rounding) — near a rounding tie the two differ by one 16-bit ulp and the pair is off by that ulp.
template <typename T>
__device__ __forceinline__ void split16(float v, T& hi, T& lo) {
  asm volatile("" : "+v"(v));
  hi = from_f32<T>(v);
  lo = from_f32<T>(v - to_f32<T>(hi));
}

// Mixed split pair (GemmArgs::a_split == 2): hi = round16(v) as above, the residual travels as ONE byte,
// lo8 = e5m2(lo * 2^LO8_EXP).  |lo| <= ulp(hi)/2, i.e. <= 2^-11 |v| (fp16) / 2^-8 |v| (bf16), so with the exponents below
// lo * 2^LO8_EXP stays below the e5m2 maximum (57344) for every finite fp16 value (bf16: clamped), and e5m2's 30 octaves of
// normals cover the residuals of |v| >= 2^-12: the exponent is a constant of the data type, not of the tensor.  The
// consumer multiplies lo8 with the weight's e4m3 copy on v_mfma_scale_f32_16x16x128_f8f6f4 (the scale operand undoes 2^LO8_EXP).
template <typename T> struct Lo8;
template <> struct Lo8<f16> { static constexpr int EXP = 10; };
template <> struct Lo8<bf16> { static constexpr int EXP = 7; };
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// acc += W8 (e4m3, 16 x 128) x A8 (e5m2, 128 x 16) with the two e8m0 scale bytes (byte 0 of sw / sa).  Inline asm with the
// accumulator TIED to the destination: through the builtin hipcc gives this instruction a destination distinct from its
// accumulator input (an early-clobber form), which doubles the live accumulators of a GEMM main loop (+64 VGPRs on a
// 64x64 wave tile, spills on 128x64).  Hazards around it: sources come from ds_read (waitcnt is inserted for asm operands),
// a following MFMA on the same accumulator needs no wait state; a following VALU read of acc needs 11 (8-pass XDL op) —
// callers end a run of these with mfma_lo8_fence().
__device__ __forceinline__ void mfma_lo8(const i32x8& w8, const i32x8& a8, f32x4& acc, int sw, int sa) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] blgp:1"
               : "+v"(acc) : "v"(w8), "v"(a8), "v"(sw), "v"(sa));
}
__device__ __forceinline__ void mfma_lo8_fence() { asm volatile("s_nop 7\n\ts_nop 4" ::: "memory"); }

// hi[0..3] (16-bit) and the four lo8 bytes (little endian: byte e belongs to v[e]) of four fp32 values
template <typename T>
__device__ __forceinline__ uint32_t split_lo8x4(f32x4 v, typename Vec<T>::v4& hi) {
  float l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = v[e];
    asm volatile("" : "+v"(x));
    hi[e] = from_f32<T>(x);
    l[e] = (x - to_f32<T>(hi[e])) * (float)(1 << Lo8<T>::EXP);
    if constexpr (sizeof(T) == 2 && Lo8<T>::EXP != 10) l[e] = fminf(fmaxf(l[e], -57344.0f), 57344.0f);
  }
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_bf8_f32(l[0], l[1], w, false);
  w = __builtin_amdgcn_cvt_pk_bf8_f32(l[2], l[3], w, true);
  return (uint32_t)w;
}
// value of a stored (hi, lo8) element
template <typename T>
__device__ __forceinline__ float join_lo8(T hi, uint32_t lo8_word, int byte) {
  float l;
  switch (byte) {
    case 0: l = __builtin_amdgcn_cvt_f32_bf8((int)lo8_word, 0); break;
    case 1: l = __builtin_amdgcn_cvt_f32_bf8((int)lo8_word, 1); break;
    case 2: l = __builtin_amdgcn_cvt_f32_bf8((int)lo8_word, 2); break;
    default: l = __builtin_amdgcn_cvt_f32_bf8((int)lo8_word, 3); break;
  }
  return to_f32<T>(hi) + l * (1.0f / (float)(1 << Lo8<T>::EXP));
}

// Packed residual stream (fp16 towers without a gradient): a fp32 value x travels as hi = round16(x) — which is at
// the same time the 16-bit A operand of the GEMM behind the next LayerNorm — plus ONE byte that carries the next 8 bits of x:
// for a normal fp16 `hi`, float(hi) has 13 zero mantissa bits and the fp32 bit patterns differ by d = bits(x) - bits(float(hi)),
// |d| <= 2^12 (integer arithmetic on the bit patterns handles the mantissa / exponent carries and both signs; x is saturated to the
// fp16 range first, so hi is finite for every finite x);
// lo = clamp(d >> 5, -128, 127) as int8 and x' = bits(float(hi)) + (lo << 5) is x to 2^-9 of an fp16 ulp (~2^-20 relative; fp16
// subnormals: to 2^-25 absolute).  6 bytes per element and residual update (3 read + 3 written) instead of 10
// (fp32 read + fp32 written + the 16-bit operand copy).
__device__ __forceinline__ uint32_t respk_split4(f32x4 v, f16x4& hi) {
  int q[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = __builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);      // saturate: hi must stay finite (it is the next GEMM's A operand)
    asm volatile("" : "+v"(x));      // (see split16: one rounding of the materialised value)
    hi[e] = (f16)x;
    const int d = __builtin_bit_cast(int, x) - __builtin_bit_cast(int, (float)hi[e]);
    const int t = d >> 5;
    q[e] = t < -128 ? -128 : (t > 127 ? 127 : t);      // (v_med3_i32; only fp16 subnormals and round-to-even ties ever clamp)
  }
  // bytes 0 of q[0..3] -> one word (v_perm_b32: selector bytes 0-3 pick from the second operand, 4-7 from the first, 0x0c = 0)
  const uint32_t a = __builtin_amdgcn_perm((uint32_t)q[1], (uint32_t)q[0], 0x0c0c0400u);
  const uint32_t b = __builtin_amdgcn_perm((uint32_t)q[3], (uint32_t)q[2], 0x04000c0cu);
  return a | b;
}
__device__ __forceinline__ f32x4 respk_join4(f16x4 hi, uint32_t lo) {
  f32x4 r;
  r[0] = __builtin_bit_cast(float, __builtin_bit_cast(int, (float)hi[0]) + (__builtin_amdgcn_sbfe((int)lo, 0, 8) << 5));
  r[1] = __builtin_bit_cast(float, __builtin_bit_cast(int, (float)hi[1]) + (__builtin_amdgcn_sbfe((int)lo, 8, 8) << 5));
  r[2] = __builtin_bit_cast(float, __builtin_bit_cast(int, (float)hi[2]) + (__builtin_amdgcn_sbfe((int)lo, 16, 8) << 5));
  r[3] = __builtin_bit_cast(float, __builtin_bit_cast(int, (float)hi[3]) + (((int)lo >> 24) << 5));
  return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// QuickGELU (clip/model.py:162-164) and its derivative
// (v_exp_f32 + v_rcp_f32, ~1 ulp each: 5 VALU per element instead of the ~20 of an IEEE division; the result is
// rounded to 16 bits right after, which is 2^13 times coarser)
__device__ __forceinline__ float sigmoid_1702(float u) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * u));
}
__device__ __forceinline__ float quick_gelu(float u) { return u * sigmoid_1702(u); }
__device__ __forceinline__ float quick_gelu_grad(float u) {
  const float s = sigmoid_1702(u);
  return s * (1.0f + 1.702f * u * (1.0f - s));
}

// async global -> LDS copy, 16 B per lane; LDS destination = wave-uniform base + lane*16
#ifndef MVLPT_GLDS_ASM
#define MVLPT_GLDS_ASM 0
#endif
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
#if MVLPT_GLDS_ASM
  // from inline asm (M0 saved and restored: it belongs to the compiler): hipcc neither counts the request in vmcnt nor orders
  // it against its own LDS accesses — every wait for it is an explicit s_waitcnt at the call sites
  // (the low 32 bits of a generic pointer into the LDS aperture are the LDS byte address)
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

// same with the non-temporal (streaming) cache policy: for operands that exactly one workgroup reads once
__device__ __forceinline__ void glds16_nt(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}

enum DType { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2 };

}  // namespace mvlpt
