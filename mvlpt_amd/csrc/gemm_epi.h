// Epilogues of the MFMA GEMMs (gemm.hip, gemm_duo.hip): bias / QuickGELU / fp32 residual / split-pair outputs and both sides of the
// LayerNorm folding, staged through a wave-private LDS scratch so that global memory sees full 128-256-byte row segments.
#pragma once
#include "kernels.h"

namespace mvlpt {

constexpr int BK = 64;

// Epilogue: the MFMA result layout gives every lane 4 consecutive columns of 16 different rows, i.e. 8-byte
// pieces scattered over 16 rows per store instruction.  With ~0.8 GFLOP per MB of output (short K) the L2
// request rate of such stores bounds the kernel, so each wave transposes its 64x64 block through a private LDS
// scratch (the ring slot that has just been released) and writes / reads global memory in full 128-256-byte
// row segments (16 B per lane).  fp16 outputs are staged as 16-bit, 32 rows per pass; everything that is
// combined with another global operand in fp32 (residual, GELU' * u) is staged as fp32, 16 rows per pass.
#ifndef MVLPT_RESIDP_WIDE
#define MVLPT_RESIDP_WIDE 1      // 0: the four-columns-per-lane walk of round 5 (A/B builds)
#endif
constexpr int EPI_SCRATCH_PER_WAVE = 4608;   // 32 rows x (128 + 16) B  >=  16 rows x (256 + 16) B

// wave-private scratch rows: either one contiguous region, or (phased kernel) the wave's OWN six 1-KiB LDS-DMA slabs
// of the ring slot that was just released (rows do not straddle slabs)
template <int RS>
struct LinearRows {
  char* base;
  __device__ __forceinline__ char* operator()(int r) const { return base + r * RS; }
};
template <int RS>
struct SlabRows {
  static constexpr int RPS = 1024 / RS;      // rows per slab (7 at 144 B, 3 at 272 B)
  char* slot; int wave; int a_bytes;
  __device__ __forceinline__ char* operator()(int r) const {
    const int sl = r / RPS, k = r - sl * RPS;
    const int off = sl < 4 ? (sl * 8 + wave) * 1024 : a_bytes + ((sl - 4) * 8 + wave) * 1024;
    return slot + off + k * RS;
  }
};

// LayerNorm folding (kernels.h): the 16 KiB LDS region behind the ring.  Consumer: the partial sums of the tile's rows,
// [row][ntp] float2, copied there by LDS-DMA with the tile's last K-stage.  Producer: [row][wave column] float2 of this tile.
constexpr int XLDS_BYTES = 16384, XLDS_BYTES_WIDE = 20480;   // rows of up to 6 / 8 slots (d <= 768 / <= 1024)
// consumer tables behind the partials (at xlds + xlds_tab(ntp)): the tile's colsum / bias slices (<= 256 floats each) and
// {rstd, -rstd * mean} of the tile's rows (256 x 8 B), written once per tile
constexpr int XLDS_COLSUM = 0, XLDS_BIAS = 1024, XLDS_COEF = 2048;
constexpr int FOLD_MAX_NTP = 8;                          // one slot per 128 columns: rows up to 1024 wide
__host__ __device__ constexpr int xlds_tab(int ntp) { return ntp > 6 ? 16384 : 12288; }      // 256 rows x ntp slots x 8 B
__host__ __device__ constexpr int xlds_bytes(int ntp) { return ntp > 6 ? XLDS_BYTES_WIDE : XLDS_BYTES; }
constexpr float FOLD_LN_EPS = 1e-5f;      // = LN_EPS of norm.hip (clip/model.py:153-159)
struct FoldCtx {
  char* xl;      // the region
  char* tab;     // consumer: its tables (colsum, bias, row coefficients)
  int mrel;      // first row of this wave's 64x64 block inside the tile
  int wn, wcn;   // column block of the wave / number of column blocks (producer)
  int xs;        // producer: format of the 16-bit copy (GemmArgs::ln_split; a compile-time 2 in the mixed-pair kernels)
  // gemm_duo.hip: the bias of the workgroup's column panel in LDS (wn * 64 + column) instead of g.bias.  Typed as an LDS pointer:
  // through a generic one hipcc emits flat loads, whose waits (vmcnt(0) lgkmcnt(0)) would drain the LDS-DMA queue
  const __attribute__((address_space(3))) char* bias_lds = nullptr;
};
// {rstd, -rstd * mean} of tile row `row_rel`: the table fold_build_coef left behind the partials
__device__ __forceinline__ void fold_row_coef(const GemmArgs&, const char* tab, int row_rel, float& a, float& cc) {
  const float2 q = *(const float2*)(tab + XLDS_COEF + row_rel * 8);
  a = q.x; cc = q.y;
}
// ... built once per tile (thread r = tile row r) from the row's partial sums, summed in slot order: deterministic
__device__ __forceinline__ void fold_build_coef(const GemmArgs& g, char* xl, char* tab, int row_rel) {
  float a, cc;
  const f32x4* p = (const f32x4*)(xl + (size_t)row_rel * g.fold_ntp * 8);
  float s1 = 0.f, s2 = 0.f;
  for (int t2 = 0; 2 * t2 < g.fold_nt; ++t2) {
    const f32x4 q = p[t2];
    s1 += q[0]; s2 += q[1];
    if (2 * t2 + 1 < g.fold_nt) { s1 += q[2]; s2 += q[3]; }
  }
  const float inv_d = 1.0f / (float)g.K;
  const float mean = s1 * inv_d;
  const float var = fmaxf(s2 * inv_d - mean * mean, 0.f);
  a = rsqrtf(var + FOLD_LN_EPS);
  cc = -a * mean;
  *(float2*)(tab + XLDS_COEF + row_rel * 8) = float2{a, cc};
}
// The consumer side of the folding on one register quad: rstd_r * acc + (-rstd_r * mean_r * colsum + bias2), element by element on
// the SCALAR fma.  Left to itself hipcc packs the four elements into two v_pk_fma_f32 with op_sel (the row coefficients broadcast
// into both halves of the pair), and on gfx950 that form is a HAZARD the compiler does not pad: straight behind the partial
// `s_waitcnt lgkmcnt(1)` that releases the coefficient pair, the LOW lane's read of the pair's HIGH register returns 0 in lanes 48-63
// whenever another wave keeps the matrix pipe busy (round 5: ~40 % of the launches of gemm_bt_kernel<f16, EPI_GELU_SPLIT_FOLD, 128,
// 128, 4, 2, MIXED> under a concurrent text tower; root cause, micro-reproducer and audit: NOTES_experiments.md round 6,
// tools/pkfma_hazard.hip, tools/isa_audit.py).  The whole library is now built without packed fp32 instructions (Makefile NOPK);
// the explicit scalar form stays so that the arithmetic does not depend on that flag.  (Bit-identical results: both forms are fused.)
__device__ __forceinline__ f32x4 fold_apply(float fa, float fcc, const f32x4& acc, const f32x4& colsum, const f32x4& bias2) {
#ifdef MVLPT_FOLD_PK      // the round-5 form that failed (debug builds of the reproducer only): 1 as it was; 2 every input complete and
  {                       // 16 wait states old before the arithmetic; 3 the results 8 wait states old before their first use
    f32x4 a = acc;
#if MVLPT_FOLD_PK == 2
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(a), "+v"(fa), "+v"(fcc));
#endif
    f32x4 r = fa * a + (fcc * colsum + bias2);
#if MVLPT_FOLD_PK == 3
    asm volatile("s_nop 7" : "+v"(r));
#endif
    return r;
  }
#endif
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t = fmaf(fcc, colsum[e], bias2[e]);
    asm volatile("" : "+v"(t));
    float w = fmaf(fa, acc[e], t);
    asm volatile("" : "+v"(w));
    r[e] = w;
  }
  return r;
}
// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: quad butterflies, then the two mirrors.
// VALU only (ds_bpermute-based shuffles would put ~8 LDS round trips into every row segment of the epilogue)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}
// ... over the 8 lanes of a half row (lanes 8k .. 8k+7)
__device__ __forceinline__ float row8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}
// 16-bit copy of four values in the A-operand formats of kernels.h (0 single, 1 hi|lo pair, 2 mixed pair); `base` [M, N or 2N]
template <typename T>
__device__ __forceinline__ void store_a16(void* base, int split, size_t m, int N, int col, f32x4 v) {
  using v4 = typename Vec<T>::v4;
  if (split == 0) {
    v4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(v[e]);
    *(v4*)((T*)base + m * N + col) = w;
  } else if (split == 1) {
    v4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { T h, l; split16<T>(v[e], h, l); hi[e] = h; lo[e] = l; }
    T* row = (T*)base + m * (2 * (size_t)N);
    *(v4*)(row + col) = hi;
    *(v4*)(row + N + col) = lo;
  } else {
    v4 hi;
    const uint32_t lo8 = split_lo8x4<T>(v, hi);
    T* row = (T*)base + m * (2 * (size_t)N);
    *(v4*)(row + col) = hi;
    *(uint32_t*)((char*)row + 2 * (size_t)N + col) = lo8;
  }
}

// H0, H1: the 32-row halves of the wave's 64x64 block this call stores (the K-split kernel gives each of its two wave groups one)
template <typename T, int EPI_, typename Rows16, typename Rows32, int H0 = 0, int H1 = 2>
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, const f32x4 (&acc)[4][4], int mbase, int nbase, int lane,
                                               Rows16 rows16, Rows32 rows32, FoldCtx fc) {
  using v4 = typename Vec<T>::v4;
  using v8 = typename Vec<T>::v8;
  const int M = g.M, N = g.N;      // N % 128 == 0 (checked at launch): no column guard
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int EPI = epi_base(EPI_);
  constexpr bool fold = epi_folds(EPI_);     // consumer side of the LayerNorm folding compiled in
  f32x4 bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (fc.bias_lds && !epi_ln_producer(EPI) && !fold) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *(const __attribute__((address_space(3))) f32x4*)(fc.bias_lds + (fc.wn * 64 + j * 16 + fg * 4) * 4);
  } else if (g.bias && !epi_ln_producer(EPI) && !fold) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(g.bias + nbase + j * 16 + fg * 4);
  }
  if constexpr (EPI == EPI_STORE16 || EPI == EPI_GELU) {
    constexpr int NOUT = (EPI == EPI_GELU) ? 2 : 1;
#pragma unroll
    for (int which = 0; which < NOUT; ++which) {
      // which == 0: the main output (activated for EPI_GELU); which == 1: the saved pre-activation u
      T* outp = (T*)(which == 0 ? g.out : g.out2);
      if (which == 1 && !outp) break;
#pragma unroll
      for (int half = H0; half < H1; ++half) {
        if constexpr (fold) {
          // the tile's colsum / bias slices sit in LDS (ds_read: no vmcnt, nothing to keep in registers across the stores).
          // All LDS reads of the half first (two rows' partials, four column-vector pairs), then the arithmetic: read-then-use
          // per (row, column block) would expose one LDS round trip 32 times per tile
          float fa[2], fcc[2];
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) fold_row_coef(g, fc.tab, fc.mrel + (half * 2 + ii) * 16 + fr, fa[ii], fcc[ii]);
          f32x4 sj[4], bj[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sj[j] = *(const f32x4*)(fc.tab + XLDS_COLSUM + (fc.wn * 64 + j * 16 + fg * 4) * 4);
            bj[j] = *(const f32x4*)(fc.tab + XLDS_BIAS + (fc.wn * 64 + j * 16 + fg * 4) * 4);
          }
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x4 v = fold_apply(fa[ii], fcc[ii], acc[half * 2 + ii][j], sj[j], bj[j]);
              v4 w;
#pragma unroll
              for (int e = 0; e < 4; ++e) w[e] = from_f32<T>((EPI == EPI_GELU && which == 0) ? quick_gelu(v[e]) : v[e]);
              *(v4*)(rows16(ii * 16 + fr) + (j * 16 + fg * 4) * 2) = w;
            }
        } else {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = half * 2 + ii;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x4 v = acc[i][j] + bv[j];
              v4 w;
#pragma unroll
              for (int e = 0; e < 4; ++e) w[e] = from_f32<T>((EPI == EPI_GELU && which == 0) ? quick_gelu(v[e]) : v[e]);
              *(v4*)(rows16(ii * 16 + fr) + (j * 16 + fg * 4) * 2) = w;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // All four row-segment reads of the pass in flight at once, then the stores.  (One read -> wait -> guarded store at a time
        // was four serialised LDS round trips plus four exec-mask branches per pass: ~1 700 cycles per 32 rows, the bulk of the
        // 16-bit epilogue's time — tools/duo_trace.py.)  Whole passes inside the matrix (every tile but the last row of tiles)
        // store without per-lane guards.
        {
          const int c = lane & 7, r0 = lane >> 3;             // 8 lanes x 16 B = one 128-B row segment
          v8 w[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) w[it] = *(const v8*)(rows16(it * 8 + r0) + c * 16);
          T* const o0 = outp + (size_t)(mbase + half * 32 + r0) * N + nbase + c * 8;
          if (mbase + half * 32 + 32 <= M) {
#pragma unroll
            for (int it = 0; it < 4; ++it) __builtin_nontemporal_store(w[it], (v8*)(o0 + (size_t)it * 8 * N));
          } else {
#pragma unroll
            for (int it = 0; it < 4; ++it)
              if (mbase + half * 32 + it * 8 + r0 < M) __builtin_nontemporal_store(w[it], (v8*)(o0 + (size_t)it * 8 * N));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // scratch is rewritten by the next pass
      }
    }
  } else if constexpr (EPI == EPI_RESIDP_LN && MVLPT_RESIDP_WIDE && sizeof(T) == 2 && !__is_same(T, bf16)) {
    // Packed residual stream, EIGHT columns per lane behind the transpose (8 lanes = the 64 columns of a row, 8 rows per pass): the
    // 16-bit plane moves 16 B per lane and the byte plane 8 B — half the load / store instructions of the four-column walk below
    // (32 instead of 64 per 64x64 block; 1 KiB and 512 B per instruction instead of 512 and 256).  Under a saturated memory system an
    // epilogue is paced by its number of requests as much as by its bytes (NOTES round 6).
    int c = lane & 7, rq = lane >> 3;
    asm volatile("" : "+v"(c), "+v"(rq));      // (opaque to loop-invariant code motion, see below)
    v8 uv[4][2];
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 lw[4][2];
#pragma unroll
    for (int i = 2 * H0; i < 2 * H1; ++i)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        int m = mbase + i * 16 + it * 8 + rq;
        m = m < M ? m : M - 1;
        const size_t o = (size_t)m * N + nbase + c * 8;
        uv[i][it] = __builtin_nontemporal_load((const v8*)((const T*)g.rp_hi_in + o));
        lw[i][it] = __builtin_nontemporal_load((const u32x2*)(g.rp_lo_in + o));
      }
    f32x4 cb0 = {0.f, 0.f, 0.f, 0.f}, cb1 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) { cb0 = *(const f32x4*)(g.bias + nbase + c * 8); cb1 = *(const f32x4*)(g.bias + nbase + c * 8 + 4); }
#pragma unroll
    for (int i = 2 * H0; i < 2 * H1; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(rows32(fr) + (j * 16 + fg * 4) * 4) = acc[i][j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int r = it * 8 + rq;
        f32x4 v0 = *(const f32x4*)(rows32(r) + c * 32), v1 = *(const f32x4*)(rows32(r) + c * 32 + 16);
        const int m = mbase + i * 16 + r;
        const size_t o = (size_t)m * N + nbase + c * 8;
        const v8 u = uv[i][it];
        v0 += respk_join4(f16x4{u[0], u[1], u[2], u[3]}, lw[i][it][0]) + cb0;
        v1 += respk_join4(f16x4{u[4], u[5], u[6], u[7]}, lw[i][it][1]) + cb1;
        // partial sums of the row over this wave's 64 columns: the 8 lanes of the row hold them
        float s1 = ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
        float s2 = ((v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3])) + ((v1[0] * v1[0] + v1[1] * v1[1]) + (v1[2] * v1[2] + v1[3] * v1[3]));
        s1 = row8_sum(s1); s2 = row8_sum(s2);
        if (c == 0) *(float2*)(fc.xl + ((size_t)(fc.mrel + i * 16 + r) * fc.wcn + fc.wn) * 8) = float2{s1, s2};
        f16x4 h0, h1;
        u32x2 lo;
        lo[0] = respk_split4(v0, h0);
        lo[1] = respk_split4(v1, h1);
        if (m < M) {
          __builtin_nontemporal_store(v8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]}, (v8*)((T*)g.out + o));
          __builtin_nontemporal_store(lo, (u32x2*)(g.rp_lo_out + o));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  } else {
    int c = lane & 15, rq = lane >> 4;                        // 16 lanes x 16 B = one 256-B row segment (fp32)
    // opaque to loop-invariant code motion: hipcc otherwise hoists the per-(pass, row) address offsets of this epilogue (resid,
    // out, the 16-bit copy: up to ~48 values) out of the persistent tile loop and keeps — or spills — them around the main loop
    // (EPI_RESID32_LN at 256x256: 30 spilled VGPRs without this, 239 VGPRs and none with it)
    asm volatile("" : "+v"(c), "+v"(rq));
    constexpr bool RESID = EPI == EPI_RESID32 || EPI == EPI_RESID32_LN;
    // Every global LOAD of the epilogue is issued before its first store: hipcc waits vmcnt(0) on an ordinary
    // load while LDS-DMA is in flight, and that wait would also drain the stores issued before it.
    f32x4 rv[4][4];
    v4 uv[4][4];
    [[maybe_unused]] uint32_t lw[4][4];      // packed stream: the hi plane arrives in uv, the byte plane here
    if constexpr (EPI == EPI_RESIDP_LN) {
#pragma unroll
      for (int i = 2 * H0; i < 2 * H1; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          int m = mbase + i * 16 + it * 4 + rq;
          m = m < M ? m : M - 1;
          const size_t o = (size_t)m * N + nbase + c * 4;
          uv[i][it] = __builtin_nontemporal_load((const v4*)((const T*)g.rp_hi_in + o));
          lw[i][it] = __builtin_nontemporal_load((const uint32_t*)(g.rp_lo_in + o));
        }
    }
    if constexpr (RESID || EPI == EPI_GELUBWD || EPI == EPI_GELUBWD_SPLIT) {
#pragma unroll
      for (int i = 2 * H0; i < 2 * H1; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          int m = mbase + i * 16 + it * 4 + rq;
          m = m < M ? m : M - 1;
          if constexpr (RESID) rv[i][it] = __builtin_nontemporal_load((const f32x4*)(g.resid + (size_t)m * N + nbase + c * 4));
          else uv[i][it] = __builtin_nontemporal_load((const v4*)((const T*)g.aux + (size_t)m * N + nbase + c * 4));
        }
    }
    // after the transpose a lane owns the SAME four columns nbase + 4c .. + 3 in every pass: per-column vectors cost 4 registers
    [[maybe_unused]] f32x4 colb = {0.f, 0.f, 0.f, 0.f}, cols = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_RESID32_LN) {
      if (g.bias) colb = *(const f32x4*)(g.bias + nbase + c * 4);
      cols = *(const f32x4*)(g.ln_gamma + nbase + c * 4);
    }
    if constexpr (EPI == EPI_RESIDP_LN) { if (g.bias) colb = *(const f32x4*)(g.bias + nbase + c * 4); }
    if constexpr (fold) {      // the bias is added behind the row scale: the raw accumulators are staged
      colb = *(const f32x4*)(fc.tab + XLDS_BIAS + (fc.wn * 64 + c * 4) * 4);
      cols = *(const f32x4*)(fc.tab + XLDS_COLSUM + (fc.wn * 64 + c * 4) * 4);
    }
    constexpr bool stage_raw = fold || epi_ln_producer(EPI);
#pragma unroll
    for (int i = 2 * H0; i < 2 * H1; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(rows32(fr) + (j * 16 + fg * 4) * 4) = stage_raw ? acc[i][j] : acc[i][j] + bv[j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 4 + rq;
        f32x4 v = *(const f32x4*)(rows32(r) + c * 16);
        const int m = mbase + i * 16 + r;
        const size_t o = (size_t)m * N + nbase + c * 4;
        if constexpr (fold) {
          float fa, fcc;
          fold_row_coef(g, fc.tab, fc.mrel + i * 16 + r, fa, fcc);
          v = fold_apply(fa, fcc, v, cols, colb);
        }
        if constexpr (EPI == EPI_RESID32) {
          v += rv[i][it];
          if (m < M) __builtin_nontemporal_store(v, (f32x4*)((float*)g.out + o));
        } else if constexpr (EPI == EPI_RESID32_LN) {
          v += rv[i][it] + colb;
          // partial sums of the row over this wave's 64 columns: the 16 lanes of a row segment hold them
          float s1 = (v[0] + v[1]) + (v[2] + v[3]);
          float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
          s1 = row16_sum(s1); s2 = row16_sum(s2);
          if (c == 0) *(float2*)(fc.xl + ((size_t)(fc.mrel + i * 16 + r) * fc.wcn + fc.wn) * 8) = float2{s1, s2};
          if (m < M) {
            __builtin_nontemporal_store(v, (f32x4*)((float*)g.out + o));
            store_a16<T>(g.ln_x16, fc.xs, (size_t)m, N, nbase + c * 4, v * cols);
          }
          __builtin_amdgcn_sched_barrier(0);     // one row segment at a time: interleaved passes cost registers this kernel does not have
        } else if constexpr (EPI == EPI_RESIDP_LN) {
          if constexpr (sizeof(T) == 2 && !__is_same(T, bf16)) {
            v += respk_join4(uv[i][it], lw[i][it]) + colb;
            float s1 = (v[0] + v[1]) + (v[2] + v[3]);
            float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            s1 = row16_sum(s1); s2 = row16_sum(s2);
            if (c == 0) *(float2*)(fc.xl + ((size_t)(fc.mrel + i * 16 + r) * fc.wcn + fc.wn) * 8) = float2{s1, s2};
            v4 hi;
            const uint32_t lo = respk_split4(v, hi);
            if (m < M) {
              __builtin_nontemporal_store(hi, (v4*)((T*)g.out + o));
              __builtin_nontemporal_store(lo, (uint32_t*)(g.rp_lo_out + o));
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else if constexpr (EPI == EPI_GELUBWD) {
          v4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(v[e] * quick_gelu_grad(to_f32<T>(uv[i][it][e])));
          if (m < M) __builtin_nontemporal_store(w, (v4*)((T*)g.out + o));
        } else if constexpr (EPI == EPI_GELU_SPLIT || EPI == EPI_GELUBWD_SPLIT || EPI == EPI_STORE_SPLIT) {
          v4 hi, lo = {}, u16;
          f32x4 rr;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (EPI == EPI_GELU_SPLIT) { rr[e] = quick_gelu(v[e]); u16[e] = from_f32<T>(v[e]); }
            else if constexpr (EPI == EPI_GELUBWD_SPLIT) rr[e] = v[e] * quick_gelu_grad(to_f32<T>(uv[i][it][e]));
            else rr[e] = v[e];
          }
          uint32_t lo8 = 0;
          const bool as_lo8 = EPI != EPI_STORE_SPLIT && g.out_lo8;      // mixed pair: the residual as one e5m2 byte (common.h)
          if (as_lo8) lo8 = split_lo8x4<T>(rr, hi);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { T h, l; split16<T>(rr[e], h, l); hi[e] = h; lo[e] = l; }
          }
          if (m < M) {
            const size_t ldo = g.ldo ? (size_t)g.ldo : 2 * (size_t)N;
            T* row = (T*)g.out + (size_t)m * ldo + nbase + c * 4;
            __builtin_nontemporal_store(hi, (v4*)row);
            if (as_lo8) __builtin_nontemporal_store(lo8, (uint32_t*)((char*)g.out + (size_t)m * (2 * ldo) + 2 * N + nbase + c * 4));
            else __builtin_nontemporal_store(lo, (v4*)(row + N));
            if constexpr (EPI == EPI_GELU_SPLIT) { if (g.out2) __builtin_nontemporal_store(u16, (v4*)((T*)g.out2 + o)); }
          }
        } else {  // EPI_STORE32
          if (m < M) __builtin_nontemporal_store(v, (f32x4*)((float*)g.out + o));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

}  // namespace mvlpt
