// MFMA GEMM with two half-tile wave groups per workgroup ("duo"):  C[M,N] = A[M,K] * Bt[N,K]^T (+ fused epilogue).
//
// Same call sites as gemm.hip (clip/model.py:174-176, 183-187: the four linears of a ResidualAttentionBlock in the image tower).
// Why: in the 256x256 kernel of gemm.hip every wave of the workgroup reaches the epilogue at the same time, and for 14-25 % of a
// tile (K = 768: 12 K-stages) the CU's matrix pipes idle while 8 waves convert, transpose through LDS and store.  A second
// accumulator set does not fit in 512 registers, and two independent workgroups per CU load 1.5x the operand bytes per FLOP.
//
// Here the 8 waves are TWO groups of four (one wave of each group per SIMD).  A group owns a 128 x 256 half tile (wave tile
// 128 x 64, as before); both groups work on the SAME 256-column panel of the weight, so the B K-stage in LDS is shared, and the
// groups are shifted in time by about half a tile life: while one group spends its E "epilogue stages", its SIMD partners have
// the matrix pipe to themselves.  The B stream simply cycles k = s mod nk; a group that starts a tile at stage s accumulates its
// K-stages in the rotated order s, s+1, ... (mod nk) — every k exactly once.  One s_barrier per stage for all eight waves, 2-deep
// ring of [A half 0 | A half 1 | B] = 64 KiB stages exactly as in the 256x256 kernel: the same LDS-DMA bytes and fragment reads
// per FLOP.  A workgroup stays on one weight panel for its whole life and walks a contiguous range of 128-row half tiles.
//
//   stage s of a wave:   MFMA role : [request stage s+1] 64 MFMAs on slot s&1 [vmcnt(0)] barrier
//                        EPI  role : [request stage s+1] one quarter / half of the wave's 128x64 block -> scratch -> global
//                                    [vmcnt(stores of this chunk): the request has landed, the stores may stay in flight] barrier
//                        idle      : [request stage s+1] [vmcnt(0)] barrier
// Every wave requests its share of B (4 pieces) whatever its role, and its own group's A half (4 pieces) when the group
// multiplies in stage s+1.  LDS-DMA is issued from inline asm here: hipcc must not see it, or it drains the queue (vmcnt(0)) in
// front of the first use of any ordinary load of the epilogue code.
#include <cstdlib>
#include <hip/hip_ext.h>
#include "kernels.h"
#include "gemm_epi.h"

namespace mvlpt {

// Debug timeline (tools/duo_trace.py): -DMVLPT_GEMM_TRACE builds record (point << 56 | s_memtime) per wave of workgroup 0.
// Points: 1 / 2 / 3 = behind the closing barrier of a stage the wave spent multiplying / in its epilogue / idle; 4 = in front of the
// stage's vmcnt wait.
#ifdef MVLPT_GEMM_TRACE
constexpr int DUO_TR_MAX = 2048;
#define DUO_TR(p) do { if (g.trace && blockIdx.x == 0 && lane == 0 && tr_n < DUO_TR_MAX) \
    g.trace[wave * DUO_TR_MAX + tr_n++] = ((long long)(p) << 56) | ((long long)__builtin_amdgcn_s_memtime() & 0xffffffffffffffLL); } while (0)
#else
#define DUO_TR(p) do { } while (0)
#endif

// async global -> LDS copy, 16 B per lane, from inline asm (M0 = wave-uniform LDS byte address, saved and restored: the register
// belongs to the compiler).  Not counted by hipcc: every wait for it is an explicit s_waitcnt below.
__device__ __forceinline__ void glds16_raw(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

constexpr int DUO_A_HALF = 128 * BK * 2;             // 16 KiB: one group's A rows of a K-stage
constexpr int DUO_B_OFF = 2 * DUO_A_HALF;
constexpr int DUO_STAGE = DUO_B_OFF + 256 * BK * 2;  // 64 KiB
constexpr int DUO_SCR_OFF = 2 * DUO_STAGE;           // epilogue scratch of the group that is in its epilogue stages (4 waves)
constexpr int DUO_BIAS_OFF = DUO_SCR_OFF + 4 * EPI_SCRATCH_PER_WAVE;     // the panel's 256 bias values
constexpr int DUO_LDS = DUO_BIAS_OFF + 1024;

template <typename T, int EPI, int E>
__global__ __launch_bounds__(512, 2) void gemm_duo_kernel(GemmArgs g, int J, int d_req) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using v8 = typename Vec<T>::v8;
  static_assert(E == 2 || E == 4, "epilogue stages per half tile");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  const int M = g.M, N = g.N, K = g.K;
  const int lda = g.lda ? g.lda : K;
  const int ldb = g.ldb ? g.ldb : K;
  const T* __restrict__ A = (const T*)g.A;
  const T* __restrict__ Bt = (const T*)g.Bt;
#ifdef MVLPT_GEMM_TRACE
  int tr_n = 0;
#endif

  // workgroup -> (row chunk j, weight panel n): panels fastest, so the workgroups an XCD runs side by side walk the same rows of A
  const int tilesN = N / 256;
  const int MH = (M + 127) / 128;
  const int G = gridDim.x, b = blockIdx.x;
  const int gq = G >> 3, gr = G & 7, xcd = b & 7;
  const int q = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + (b >> 3);
  const int j = q / tilesN, n = q - j * tilesN;
  const int h0 = (int)((long)j * MH / J), cnt = (int)((long)(j + 1) * MH / J) - h0;
  const int c0 = (cnt + 1) >> 1, c1 = cnt >> 1;        // half tiles of group 0 / group 1 (alternating: h0 + 2t + grp)
  const int c_me = grp ? c1 : c0;
  const int nk = K / BK, P = nk + E;
  int d = d_req > 0 ? d_req : (P >> 1);                // group 1 starts d stages behind: E <= d <= nk keeps the two epilogues apart
  d = d < E ? E : (d > nk ? nk : d);
  const int S = c1 ? (c0 * P > d + c1 * P ? c0 * P : d + c1 * P) : c0 * P;

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // the panel's bias in LDS (the epilogue reads it with ds_read: an ordinary global load there would make hipcc wait for it
  // with vmcnt counts that know nothing of the DMA queue)
  if (tid < 64) {
    f32x4 bvz = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) bvz = *(const f32x4*)(g.bias + n * 256 + tid * 4);
    *(f32x4*)(smem + DUO_BIAS_OFF + tid * 16) = bvz;
  }

  // ---- staging: thread -> (row, 16B chunk) of a 1 KiB LDS slab (8 rows x 128 B), source-side XOR swizzle as in gemm.hip
  const int srow = lane >> 3;
  const int scol = ((lane & 7) ^ srow) * 8;
  const T* ap[4];
  const T* bp[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) bp[it] = Bt + (size_t)(n * 256 + (it * 8 + wave) * 8 + srow) * ldb + scol;
  int a_tile = -1;
  auto set_a_ptrs = [&](int t) {
    const int r0 = (h0 + 2 * t + grp) * 128;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int ar = r0 + (it * 4 + w4) * 8 + srow; ar = ar < M ? ar : M - 1;      // edge rows are re-read, never stored
      ap[it] = A + (size_t)ar * lda + scol;
    }
    a_tile = t;
  };

  // trackers of the stage that is requested next (identical in every wave: wave-uniform scalars)
  int np0 = 0, nt0 = 0, np1 = -d, nt1 = 0, kn = 0, s_req = 0;
  auto request_next = [&]() {
    if (s_req < S) {
      const bool m0 = np0 < nk && nt0 < c0;
      const bool m1 = np1 >= 0 && np1 < nk && nt1 < c1;
      const unsigned base = lds0 + (s_req & 1) * DUO_STAGE;
      if (m0 || m1) {
#pragma unroll
        for (int it = 0; it < 4; ++it) glds16_raw(bp[it] + kn * BK, base + DUO_B_OFF + (it * 8 + wave) * 1024);
      }
      if (grp ? m1 : m0) {
        const int t = grp ? nt1 : nt0;
        if (t != a_tile) set_a_ptrs(t);
#pragma unroll
        for (int it = 0; it < 4; ++it) glds16_raw(ap[it] + kn * BK, base + grp * DUO_A_HALF + (it * 4 + w4) * 1024);
      }
    }
    ++s_req;
    kn = kn + 1 == nk ? 0 : kn + 1;
    if (np0 + 1 == P) { np0 = 0; ++nt0; } else ++np0;
    if (np1 + 1 == P) { np1 = 0; ++nt1; } else ++np1;
  };

  int s = 0;
  auto end_stage = [&](int role) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DUO_TR(role);
    ++s;
  };
  auto idle_stage = [&]() {
    request_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    end_stage(3);
  };

  // ---- fragment addressing (the wave's 128 x 64 block of its group's half tile)
  const int fr = lane & 15, fg = lane >> 4;
  const int a_off = grp * DUO_A_HALF + fr * 128;
  const int b_off = DUO_B_OFF + (w4 * 64 + fr) * 128;
  const int c0k = ((0 + fg) ^ (fr & 7)) * 16;       // k-step 0 chunk
  const int c1k = ((4 + fg) ^ (fr & 7)) * 16;       // k-step 1 chunk

  request_next();                                    // stage 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  if (grp) for (int i = 0; i < d && s < S; ++i) idle_stage();

  char* const scr = smem + DUO_SCR_OFF + w4 * EPI_SCRATCH_PER_WAVE;
  for (int t = 0; t < c_me; ++t) {
    f32x4 acc[2][4][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i >> 2][i & 3][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int p = 0; p < nk; ++p) {
      // the older wave of a SIMD requests before its MFMAs, the younger one behind its third group (gemm.hip, MVLPT_NS2_MODE 1)
      if (grp == 0) request_next();
      const char* base = smem + (s & 1) * DUO_STAGE;
      constexpr int PAIRS = 4, GROUPS = 8;
      v8 bfr[2][4], afr[2][2];
      auto load_b = [&](int ks, v8 (&bf)[4]) {
        const int c = ks ? c1k : c0k;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bf[jj] = *(const v8*)(base + b_off + jj * 2048 + c);
      };
      auto load_a2 = [&](int ks, int pair, v8 (&af)[2]) {
        const int c = ks ? c1k : c0k;
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *(const v8*)(base + a_off + (pair * 2 + i) * 2048 + c);
      };
      load_b(0, bfr[0]);
      load_a2(0, 0, afr[0]);
#pragma unroll
      for (int sg = 0; sg < GROUPS; ++sg) {
        const int ks = sg / PAIRS, pair = sg % PAIRS, cur = sg & 1;
        __builtin_amdgcn_sched_barrier(0);
        {
          const int ai = pair * 2;
          acc[ai >> 2][ai & 3][0] = mfma16<T>(bfr[ks & 1][0], afr[cur][0], acc[ai >> 2][ai & 3][0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (sg + 1 < GROUPS) {
          const int nx = sg + 1, nks = nx / PAIRS, npair = nx % PAIRS;
          if (npair == 0) load_b(nks, bfr[nks & 1]);
          load_a2(nks, npair, afr[nx & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            if (i == 0 && jj == 0) continue;
            const int ai = pair * 2 + i;
            acc[ai >> 2][ai & 3][jj] = mfma16<T>(bfr[ks & 1][jj], afr[cur][i], acc[ai >> 2][ai & 3][jj]);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (sg == 2 && grp) request_next();
      }
      DUO_TR(4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      end_stage(1);
    }

    // ---- epilogue stages: chunk c of E; the SIMD partners (other group) multiply meanwhile
    const int r0 = (h0 + 2 * t + grp) * 128;
    const int nbase = n * 256 + w4 * 64;
    FoldCtx fc{nullptr, nullptr, 0, w4, 4, 0};
    fc.bias_lds = (const __attribute__((address_space(3))) char*)(smem + DUO_BIAS_OFF);
    constexpr int NOUT_MAX = (epi_base(EPI) == EPI_GELU) ? 2 : 1;
#pragma unroll
    for (int c = 0; c < E; ++c) {
      request_next();
      constexpr int HPC = 4 / E;                       // 32-row halves per chunk
      const int hh = (c * HPC) >> 1;
      const int rows_end = r0 + (c + 1) * (128 / E);
      if constexpr (E == 4) {
        if (c & 1) epilogue_store<T, EPI, LinearRows<144>, LinearRows<272>, 1, 2>(g, acc[hh], r0 + hh * 64, nbase, lane, LinearRows<144>{scr}, LinearRows<272>{scr}, fc);
        else epilogue_store<T, EPI, LinearRows<144>, LinearRows<272>, 0, 1>(g, acc[hh], r0 + hh * 64, nbase, lane, LinearRows<144>{scr}, LinearRows<272>{scr}, fc);
      } else {
        epilogue_store<T, EPI, LinearRows<144>, LinearRows<272>, 0, 2>(g, acc[hh], r0 + hh * 64, nbase, lane, LinearRows<144>{scr}, LinearRows<272>{scr}, fc);
      }
      // the request above is older than this chunk's stores: with all of them issued (no row beyond M) it has landed once at
      // most that many operations are outstanding
      const bool two = NOUT_MAX == 2 && g.out2 != nullptr;
      DUO_TR(4);
      if (rows_end <= M) {
        if (two) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 4 * HPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * HPC) : "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      end_stage(2);
    }
  }
  while (s < S) idle_stage();
}

template <typename T, int EPI, int E>
static hipError_t launch_duo_e(const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_duo_kernel<T, EPI, E>, hipFuncAttributeMaxDynamicSharedMemorySize, DUO_LDS);
    attr_set = true;
  }
  const int cus = stream_cus(s);
  const int tilesN = g.N / 256, MH = (g.M + 127) / 128;
  int J = cus / tilesN;
  if (J > MH) J = MH;
  if (J < 1) return hipErrorInvalidValue;
  static const int d_req = getenv("MVLPT_DUO_D") ? atoi(getenv("MVLPT_DUO_D")) : 0;      // experiment: stagger of the groups in stages (0: half a tile life)
  hipExtLaunchKernelGGL((gemm_duo_kernel<T, EPI, E>), dim3(J * tilesN), dim3(512), DUO_LDS, s, ea, eb, 0, g, J, d_req);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_duo_t(int epi, const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  static const int e_store = getenv("MVLPT_DUO_E_STORE") ? atoi(getenv("MVLPT_DUO_E_STORE")) : 2;
  static const int e_gelu = getenv("MVLPT_DUO_E_GELU") ? atoi(getenv("MVLPT_DUO_E_GELU")) : 4;
  switch (epi) {
    case EPI_STORE16: return e_store == 4 ? launch_duo_e<T, EPI_STORE16, 4>(g, s, ea, eb) : launch_duo_e<T, EPI_STORE16, 2>(g, s, ea, eb);
    case EPI_GELU: return e_gelu == 2 ? launch_duo_e<T, EPI_GELU, 2>(g, s, ea, eb) : launch_duo_e<T, EPI_GELU, 4>(g, s, ea, eb);
  }
  return hipErrorInvalidValue;
}

// the problems the duo kernel takes: single operands, whole 256-column panels, enough half tiles per workgroup to amortise the
// stagger of the two groups
bool gemm_duo_takes(int epi, const GemmArgs& g, hipStream_t s) {
  if (!(epi == EPI_STORE16 || epi == EPI_GELU)) return false;
  if (g.a_split || g.fold_part || (g.N % 256) != 0 || g.K < 4 * BK) return false;
  const long cus = stream_cus(s);
  const long tilesN = g.N / 256, MH = (g.M + 127) / 128;
  if (tilesN > cus) return false;
  return MH / (cus / tilesN) >= 4;
}

hipError_t launch_gemm_duo(int dtype, int epi, const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  if (dtype == DT_F16) return launch_duo_t<f16>(epi, g, s, ea, eb);
  if (dtype == DT_BF16) return launch_duo_t<bf16>(epi, g, s, ea, eb);
  return hipErrorInvalidValue;
}

}  // namespace mvlpt
