// Stand-alone micro-reproducer of the round-5 "timing-dependent folded consumer" (VERDICT r5 item 1): a gfx950 hazard hipcc (ROCm 7.2)
// does not know.
//
// The failing epilogue read a row's {rstd, -rstd*mean} pair with ds_read_b64, a row segment of accumulators with ds_read_b128, waited with
// the COUNTED s_waitcnt lgkmcnt(1) hipcc emits (the younger read still in flight) and consumed the pair in the very next instruction,
// v_pk_fma_f32 with an op_sel broadcast.  This program replays that instruction sequence from inline asm (nothing left to the compiler) in
// a list of VARIANTS, millions of times, with the destination registers of the loads pre-filled with a NaN pattern, alone and while a
// register-only MFMA kernel runs on another stream (the one synthetic partner — besides the text tower — that triggered the fault in
// the real kernel; tools/fold_consumer_probe.py).  Every result is compared bit for bit with the recomputation from the loaded values
// read back long after all waits; wrong results are counted per (variant, half of the packed pair, 16-lane group) and classified
// (consistent with the broadcast coefficient read as ZERO / NaN pre-fill / anything else).
//
//   python tools/partners_gen.py      (builds tools/_build/pkfma_hazard too)   or
//   hipcc --offload-arch=gfx950 -O2 -Wno-inline-asm -o tools/_build/pkfma_hazard tools/pkfma_hazard.hip
//   tools/_build/pkfma_hazard [iters=400] [rounds=20]
// Result on MI355X: profiles/r06_pkfma_hazard_micro.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NVAR = 17, NCNT = 6;      // counters per (variant, lane group): wrong LOW, wrong HIGH, wrong loads, LOW ~ coefficient 0, LOW NaN, LOW other
static const char* kVarName[NVAR] = {
  " 0 v_pk_fma_f32 + op_sel straight behind s_waitcnt lgkmcnt(1)            [the round-5 failing form]",
  " 1 scalar v_fma_f32 straight behind the same wait                         [the shipped form]",
  " 2 v_pk_fma_f32 + op_sel behind s_waitcnt lgkmcnt(0) (nothing in flight)",
  " 3 as 0 with s_nop 0 between the wait and the packed op",
  " 4 as 0 with s_nop 1 between the wait and the packed op",
  " 5 as 0 with one unrelated VALU instruction between the wait and the packed op",
  " 6 v_pk_fma_f32 WITHOUT op_sel (plain packed operands) straight behind lgkmcnt(1)",
  " 7 v_pk_fma_f32 + op_sel_hi:[0,1,1] (LOW element of the pair broadcast) straight behind lgkmcnt(1)",
  " 8 v_pk_mul_f32 + op_sel straight behind lgkmcnt(1)",
  " 9 as 0, the younger reads are two ds_read_b64 (wait lgkmcnt(2))",
  "10 as 0 with GLOBAL loads and s_waitcnt vmcnt(1)",
  "11 as 0, the pair arrives as the first half of a ds_read_b128",
  "12 as 0 with 8 wait states behind lgkmcnt(1): only the second packed op (in place on the younger load) sits straight behind its wait",
  "13 no load at all: v_pk_mul_f32 writes a pair, v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (halves swapped) reads it in the NEXT instruction",
  "14 as 13 with one unrelated packed FMA between producer and consumer   [a pair hipcc emitted in assemble_tokens_kernel<3, packed>]",
  "15 as 13 with two unrelated VALU instructions between",
  "16 as 13 with the consumer's halves NOT swapped (plain packed add) in the next instruction",
};

__device__ __forceinline__ float tab_fa(int row, int it) { return 1.0f + 0.001f * (float)((row * 7 + it) % 97); }
__device__ __forceinline__ float tab_fcc(int row, int it) { return 0.5f - 0.002f * (float)((row * 13 + it * 3) % 89); }
__device__ __forceinline__ float scr_val(int wave, int row, int col, int it) { return 0.25f * (float)((wave * 31 + row * 17 + col * 3 + it * 5) % 101) - 12.0f; }

// registers: v[12:15] colsum, v[16:19] bias2, v[20:23] the coefficient pair (+2 spare), v[24:27] the accumulator segment, v[28:31] results
#define PRE \
  "v_mov_b32 v12, %12\n\tv_mov_b32 v13, %13\n\tv_mov_b32 v14, %14\n\tv_mov_b32 v15, %15\n\t" \
  "v_mov_b32 v16, %16\n\tv_mov_b32 v17, %17\n\tv_mov_b32 v18, %18\n\tv_mov_b32 v19, %19\n\t" \
  "v_mov_b32 v20, 0x7fc0beef\n\tv_mov_b32 v21, 0x7fc0beef\n\tv_mov_b32 v22, 0x7fc0beef\n\tv_mov_b32 v23, 0x7fc0beef\n\t" \
  "v_mov_b32 v24, 0x7fc0beef\n\tv_mov_b32 v25, 0x7fc0beef\n\tv_mov_b32 v26, 0x7fc0beef\n\tv_mov_b32 v27, 0x7fc0beef\n\t" \
  "s_nop 4\n\t"
#define LOADS "ds_read_b64 v[20:21], %10\n\tds_read_b128 v[24:27], %11\n\t"
#define T_PKSEL \
  "v_pk_fma_f32 v[28:29], v[12:13], v[20:21], v[16:17] op_sel:[0,1,0]\n\t" \
  "v_pk_fma_f32 v[30:31], v[14:15], v[20:21], v[18:19] op_sel:[0,1,0]\n\t"
#define R_PKSEL \
  "v_pk_fma_f32 v[28:29], v[20:21], v[24:25], v[28:29] op_sel_hi:[0,1,1]\n\t" \
  "v_pk_fma_f32 v[30:31], v[20:21], v[26:27], v[30:31] op_sel_hi:[0,1,1]\n\t"
#define POST \
  "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\t" \
  "v_mov_b32 %0, v20\n\tv_mov_b32 %1, v21\n\tv_mov_b32 %2, v24\n\tv_mov_b32 %3, v25\n\tv_mov_b32 %4, v26\n\tv_mov_b32 %5, v27\n\t" \
  "v_mov_b32 %6, v28\n\tv_mov_b32 %7, v29\n\tv_mov_b32 %8, v30\n\tv_mov_b32 %9, v31\n\t"
#define OPERANDS \
  : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) \
  : "v"(tab_addr), "v"(scr_addr), "v"(cols[0]), "v"(cols[1]), "v"(cols[2]), "v"(cols[3]), "v"(colb[0]), "v"(colb[1]), "v"(colb[2]), "v"(colb[3]), "v"(gtab), "v"(gscr) \
  : "memory", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31"

template <int V>
__global__ __launch_bounds__(256, 2) void probe(unsigned* counters, int iters, const float* gmem) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, rq = lane >> 4;
  char* scr = smem + wave * 4608;          // 16 rows x 272 B per wave (the fp32-staged epilogue's scratch)
  char* tab = smem + 32768;                // 128 rows x {fa, fcc} (variant 11: 16-byte rows {fa, fcc, 0, 0})
  f32x4 cols, colb;
  for (int e = 0; e < 4; ++e) { cols[e] = 0.01f * (float)((c * 4 + e) % 23) - 0.1f; colb[e] = 0.03f * (float)((c * 4 + e) % 11) - 0.15f; }
  unsigned cnt[NCNT] = {};
  for (int it = 0; it < iters; ++it) {
    if (tid < 128) {
      if constexpr (V == 11) *(f32x4*)(tab + tid * 16) = f32x4{tab_fa(tid, it), tab_fcc(tid, it), 0.f, 0.f};
      else *(f32x2*)(tab + tid * 8) = f32x2{tab_fa(tid, it), tab_fcc(tid, it)};
    }
    {
      const int fr = lane & 15, fg = lane >> 4;
      for (int j = 0; j < 4; ++j) {
        f32x4 v;
        for (int e = 0; e < 4; ++e) v[e] = scr_val(wave, fr, j * 16 + fg * 4 + e, it);
        *(f32x4*)(scr + fr * 272 + (j * 16 + fg * 4) * 4) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
      const int r = p * 4 + rq;                           // row of the wave's 16-row pass
      const int trow = (wave * 16 + r + it * 16) & 127;   // table row
      const unsigned tab_addr = (unsigned)(size_t)(tab + trow * (V == 11 ? 16 : 8)), scr_addr = (unsigned)(size_t)(scr + r * 272 + c * 16);
      const float* gtab = gmem + trow * 2;                                  // variant 10: the same two operands from global memory
      const float* gscr = gmem + 256 + ((wave * 16 + r) * 64 + c * 4);
      float a0, a1, b0, b1, b2, b3, r0, r1, r2, r3;
      if constexpr (V == 0) asm volatile(PRE LOADS "s_waitcnt lgkmcnt(1)\n\t" T_PKSEL "s_waitcnt lgkmcnt(0)\n\t" R_PKSEL POST OPERANDS);
      else if constexpr (V == 1)
        asm volatile(PRE LOADS "s_waitcnt lgkmcnt(1)\n\t"
                     "v_fma_f32 v28, v21, v12, v16\n\tv_fma_f32 v29, v21, v13, v17\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_fmac_f32 v28, v20, v24\n\tv_fma_f32 v30, v21, v14, v18\n\tv_fmac_f32 v29, v20, v25\n\t"
                     "v_fma_f32 v31, v21, v15, v19\n\tv_fmac_f32 v30, v20, v26\n\tv_fmac_f32 v31, v20, v27\n\t" POST OPERANDS);
      else if constexpr (V == 2) asm volatile(PRE LOADS "s_waitcnt lgkmcnt(0)\n\t" T_PKSEL R_PKSEL POST OPERANDS);
      else if constexpr (V == 3) asm volatile(PRE LOADS "s_waitcnt lgkmcnt(1)\n\ts_nop 0\n\t" T_PKSEL "s_waitcnt lgkmcnt(0)\n\ts_nop 0\n\t" R_PKSEL POST OPERANDS);
      else if constexpr (V == 4) asm volatile(PRE LOADS "s_waitcnt lgkmcnt(1)\n\ts_nop 1\n\t" T_PKSEL "s_waitcnt lgkmcnt(0)\n\ts_nop 1\n\t" R_PKSEL POST OPERANDS);
      else if constexpr (V == 5) asm volatile(PRE LOADS "s_waitcnt lgkmcnt(1)\n\tv_mov_b32 v22, v12\n\t" T_PKSEL "s_waitcnt lgkmcnt(0)\n\tv_mov_b32 v23, v12\n\t" R_PKSEL POST OPERANDS);
      else if constexpr (V == 6)      // plain packed: p = cols * {fa, fcc} + colb ; r = B * cols + p
        asm volatile(PRE LOADS "s_waitcnt lgkmcnt(1)\n\t"
                     "v_pk_fma_f32 v[28:29], v[12:13], v[20:21], v[16:17]\n\tv_pk_fma_f32 v[30:31], v[14:15], v[20:21], v[18:19]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_pk_fma_f32 v[28:29], v[24:25], v[12:13], v[28:29]\n\tv_pk_fma_f32 v[30:31], v[26:27], v[14:15], v[30:31]\n\t" POST OPERANDS);
      else if constexpr (V == 7)      // q = fa * cols + colb (LOW element broadcast) ; r = B * cols + q
        asm volatile(PRE LOADS "s_waitcnt lgkmcnt(1)\n\t"
                     "v_pk_fma_f32 v[28:29], v[20:21], v[12:13], v[16:17] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[30:31], v[20:21], v[14:15], v[18:19] op_sel_hi:[0,1,1]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_pk_fma_f32 v[28:29], v[24:25], v[12:13], v[28:29]\n\tv_pk_fma_f32 v[30:31], v[26:27], v[14:15], v[30:31]\n\t" POST OPERANDS);
      else if constexpr (V == 8)      // m = cols * fcc (HIGH element broadcast, v_pk_mul) ; r = fa * B + m
        asm volatile(PRE LOADS "s_waitcnt lgkmcnt(1)\n\t"
                     "v_pk_mul_f32 v[28:29], v[12:13], v[20:21] op_sel:[0,1]\n\tv_pk_mul_f32 v[30:31], v[14:15], v[20:21] op_sel:[0,1]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t" R_PKSEL POST OPERANDS);
      else if constexpr (V == 9)
        asm volatile(PRE "ds_read_b64 v[20:21], %10\n\tds_read_b64 v[24:25], %11\n\tds_read_b64 v[26:27], %11 offset:8\n\t"
                     "s_waitcnt lgkmcnt(2)\n\t" T_PKSEL "s_waitcnt lgkmcnt(0)\n\t" R_PKSEL POST OPERANDS);
      else if constexpr (V == 10)
        asm volatile(PRE "global_load_dwordx2 v[20:21], %20, off\n\tglobal_load_dwordx4 v[24:27], %21, off\n\t"
                     "s_waitcnt vmcnt(1)\n\t" T_PKSEL "s_waitcnt vmcnt(0)\n\t" R_PKSEL POST OPERANDS);
      else if constexpr (V == 11)
        asm volatile(PRE "ds_read_b128 v[20:23], %10\n\tds_read_b128 v[24:27], %11\n\t"
                     "s_waitcnt lgkmcnt(1)\n\t" T_PKSEL "s_waitcnt lgkmcnt(0)\n\t" R_PKSEL POST OPERANDS);
      else if constexpr (V >= 13) {
        // p = cols01 * cols23 (pair), then r01 = colb01 + swap(p) [13-15] / colb01 + p [16]; r23 the same once more on v[30:31] (second instance)
        asm volatile(PRE LOADS "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\t"
                     "v_pk_mul_f32 v[22:23], v[12:13], v[14:15]\n\t"
                     ".if %22 == 14\n\tv_pk_fma_f32 v[30:31], v[12:13], v[14:15], v[18:19]\n\t.endif\n\t"
                     ".if %22 == 15\n\tv_mov_b32 v30, v12\n\tv_mov_b32 v31, v13\n\t.endif\n\t"
                     ".if %22 == 16\n\tv_pk_add_f32 v[28:29], v[16:17], v[22:23]\n\t.else\n\tv_pk_add_f32 v[28:29], v[16:17], v[22:23] op_sel:[0,1] op_sel_hi:[1,0]\n\t.endif\n\t"
                     "s_nop 7\n\t"
                     "v_pk_mul_f32 v[22:23], v[14:15], v[12:13]\n\t"
                     ".if %22 == 14\n\tv_pk_fma_f32 v[20:21], v[12:13], v[14:15], v[18:19]\n\t.endif\n\t"
                     ".if %22 == 15\n\tv_mov_b32 v20, v12\n\tv_mov_b32 v21, v13\n\t.endif\n\t"
                     ".if %22 == 16\n\tv_pk_add_f32 v[30:31], v[18:19], v[22:23]\n\t.else\n\tv_pk_add_f32 v[30:31], v[18:19], v[22:23] op_sel:[0,1] op_sel_hi:[1,0]\n\t.endif\n\t"
                     "s_nop 7\n\tds_read_b64 v[20:21], %10\n\t" POST
                     : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                     : "v"(tab_addr), "v"(scr_addr), "v"(cols[0]), "v"(cols[1]), "v"(cols[2]), "v"(cols[3]), "v"(colb[0]), "v"(colb[1]), "v"(colb[2]), "v"(colb[3]), "v"(gtab), "v"(gscr), "n"(V)
                     : "memory", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
      } else      // V == 12
        asm volatile(PRE LOADS "s_waitcnt lgkmcnt(1)\n\ts_nop 7\n\t" T_PKSEL "s_nop 7\n\ts_waitcnt lgkmcnt(0)\n\t" R_PKSEL POST OPERANDS);
      // the loads themselves (read back long after every wait) must equal what was written
      float ea0 = tab_fa(trow, it), ea1 = tab_fcc(trow, it);
      float eb[4];
      for (int e = 0; e < 4; ++e) eb[e] = V == 10 ? gscr[e] : scr_val(wave, r, c * 4 + e, it);
      if (V == 10) { ea0 = gtab[0]; ea1 = gtab[1]; }
      if (!(a0 == ea0 && a1 == ea1 && b0 == eb[0] && b1 == eb[1] && b2 == eb[2] && b3 == eb[3])) ++cnt[2];
      float x[4], z[4];      // expected results, and the results with the broadcast coefficient of the FIRST op read as zero
      const float rr[4] = {r0, r1, r2, r3};
      for (int e = 0; e < 4; ++e) {
        if (V >= 13) {      // r[e]: colb[e] + the product the (possibly swapped) half selects; "coefficient read as 0" = colb[e] + 0
          const float pr[2] = {__fmul_rn(cols[0], cols[2]), __fmul_rn(cols[1], cols[3])};      // (no contraction into an fma)
          const int sel = V == 16 ? (e & 1) : 1 - (e & 1);
          x[e] = __fadd_rn(colb[e], pr[sel]); z[e] = colb[e] + 0.0f;
        } else
        if (V == 6) { x[e] = fmaf(eb[e], cols[e], fmaf(cols[e], (e & 1) ? ea1 : ea0, colb[e])); z[e] = fmaf(eb[e], cols[e], fmaf(cols[e], 0.0f, colb[e])); }
        else if (V == 7) { x[e] = fmaf(eb[e], cols[e], fmaf(ea0, cols[e], colb[e])); z[e] = fmaf(eb[e], cols[e], fmaf(0.0f, cols[e], colb[e])); }
        else if (V == 8) { x[e] = fmaf(ea0, eb[e], cols[e] * ea1); z[e] = fmaf(ea0, eb[e], cols[e] * 0.0f); }
        else { x[e] = fmaf(ea0, eb[e], fmaf(ea1, cols[e], colb[e])); z[e] = fmaf(ea0, eb[e], fmaf(0.0f, cols[e], colb[e])); }
        if (V >= 13 && (e & 1)) continue;      // (forms 13-16 are checked on their LOW halves only)
        if (__float_as_uint(rr[e]) != __float_as_uint(x[e])) {
          ++cnt[e & 1];
          if (!(e & 1)) ++cnt[__float_as_uint(rr[e]) == __float_as_uint(z[e]) ? 3 : (rr[e] != rr[e] ? 4 : 5)];
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  for (int k = 0; k < NCNT; ++k) if (cnt[k]) atomicAdd(&counters[(V * 4 + rq) * NCNT + k], cnt[k]);
}

__global__ __launch_bounds__(64) void partner_mfma(int iters, float* sink) {
  f16x8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 0.01f + e); y[e] = (_Float16)(0.5f - e); }
  f32x4 acc[4] = {};
  for (int i = 0; i < iters; ++i)
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[j], 0, 0, 0);
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) sink[0] = acc[0][0];
}

template <int V>
static void run_variant(unsigned* counters, int iters, const float* gmem, hipStream_t s) {
  constexpr int LDS = 80 * 1024;      // two workgroups per CU, as the failing geometry
  CHECK(hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipLaunchKernelGGL(probe<V>, dim3(480), dim3(256), LDS, s, counters, iters, gmem);
  CHECK(hipGetLastError());
}
template <int... Vs>
static void run_all(unsigned* counters, int iters, const float* gmem, hipStream_t s, std::integer_sequence<int, Vs...>) { (run_variant<Vs>(counters, iters, gmem, s), ...); }

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400, rounds = argc > 2 ? atoi(argv[2]) : 20;
  unsigned* counters; float *sink, *gmem;
  CHECK(hipMalloc(&counters, NVAR * 4 * NCNT * sizeof(unsigned) * 2));
  CHECK(hipMalloc(&sink, 256));
  std::vector<float> hg(256 + 64 * 64);
  for (int r = 0; r < 128; ++r) { hg[2 * r] = 1.0f + 0.001f * (float)((r * 7) % 97); hg[2 * r + 1] = 0.5f - 0.002f * (float)((r * 13) % 89); }
  for (int i = 0; i < 64 * 64; ++i) hg[256 + i] = 0.25f * (float)((i * 7) % 101) - 12.0f;
  CHECK(hipMalloc(&gmem, hg.size() * 4));
  CHECK(hipMemcpy(gmem, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  int exit_code = 0;
  for (int with_partner = 0; with_partner < 2; ++with_partner) {
    unsigned* cnt = counters + with_partner * NVAR * 4 * NCNT;
    CHECK(hipMemset(cnt, 0, NVAR * 4 * NCNT * sizeof(unsigned)));
    for (int rd = 0; rd < rounds; ++rd) {
      if (with_partner) hipLaunchKernelGGL(partner_mfma, dim3(16384 * 8), dim3(64), 0, s1, 300, sink);
      run_all(cnt, iters, gmem, s0, std::make_integer_sequence<int, NVAR>{});
      CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned> h(NVAR * 4 * NCNT);
    CHECK(hipMemcpy(h.data(), cnt, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    const double total = (double)rounds * iters * 4 * 480 * 256 * 2;      // checked elements per half and variant
    printf("== %s: %d rounds x %d iterations x 4 passes, 480 workgroups of 4 waves (%.2e checked elements per half and variant)\n",
           with_partner ? "MFMA partner on another stream" : "alone", rounds, iters, total);
    for (int v = 0; v < NVAR; ++v) {
      auto at = [&](int g, int k) { return h[(v * 4 + g) * NCNT + k]; };
      printf("  variant %s\n     wrong LOW halves by lane group 0-15/16-31/32-47/48-63: %u %u %u %u (coefficient read as 0: %u, NaN: %u, other: %u) | wrong HIGH: %u %u %u %u | wrong loads: %u\n",
             kVarName[v], at(0, 0), at(1, 0), at(2, 0), at(3, 0), at(0, 3) + at(1, 3) + at(2, 3) + at(3, 3), at(0, 4) + at(1, 4) + at(2, 4) + at(3, 4),
             at(0, 5) + at(1, 5) + at(2, 5) + at(3, 5), at(0, 1), at(1, 1), at(2, 1), at(3, 1), at(0, 2) + at(1, 2) + at(2, 2) + at(3, 2));
      if (v == 1 && (at(0, 0) | at(1, 0) | at(2, 0) | at(3, 0) | at(0, 1) | at(1, 1) | at(2, 1) | at(3, 1))) exit_code = 2;      // the shipped form must be clean
    }
  }
  return exit_code;
}
