// MFMA GEMM for the frozen CLIP linears:  C[M,N] = A[M,K] * Bt[N,K]^T  (+ fused epilogue).
//
// Replaces every nn.Linear / conv1-as-GEMM / `x @ proj` call on the hot path (SURVEY.md §2.3 K0,K4,K6,
// K7,K8,K10,K12 forward and their dX twins; reference call sites clip/model.py:174-176,183,207,234 and
// trainers/mvlpt.py:91,128).  Both operands are K-contiguous 16-bit (fp16 or bf16) rows: nn.Linear
// stores W as [N,K] so the forward reads it as-is; the dX GEMM reads the pre-transposed copy packed
// once at load time (weights are frozen, trainers/mvlpt.py:855-858).
//
// gfx950 design: one kernel template, three geometries, BK = 64, MFMA 16x16x32 with fp32 accumulators.
//   huge  : 256x256 tile, 8 waves (2x4) of 128x64 (8x4 accumulators), one workgroup per CU, 2-deep ring (128 KiB):
//           fewest LDS-DMA instructions and LDS reads per FLOP; for the wide GEMMs (QKV, MLP up) with >= 4 rounds.
//   big   : 256x128 tile, 512 threads = 8 waves (4x2), ONE workgroup per CU, 3-deep LDS ring (144 KiB) with
//           waves of 64x64 (4x4 accumulators), COUNTED s_waitcnt vmcnt: the loads of K-stage f+2 stay in flight across the barrier that ends
//           stage f (raw s_barrier + lgkmcnt(0), never __syncthreads, which would drain the DMA queue).
//   small : 128x128 tile, 256 threads = 4 waves (2x2), two workgroups per CU, 2-deep ring, for GEMMs with too
//           few 256x128 tiles to fill 256 CUs (the text tower, CLS-row projections).
// A/B tiles go HBM -> LDS with 16-byte LDS-DMA (global_load_lds).  The LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is applied to
// the per-lane SOURCE address and undone on the ds_read_b128 side: 16-byte chunk c of tile row r lives
// at chunk (c ^ (r & 7)).  MFMA operands are swapped (D = Bfrag x Afrag) so each lane ends up with four
// consecutive output COLUMNS of one row -> 8/16-byte epilogue stores, float4 bias loads.
// Workgroups are remapped so that consecutive tiles (sharing an A panel) run on the same XCD/L2.
#include <cstdlib>
#include <hip/hip_ext.h>
#ifdef MVLPT_GEMM_GLDS_ASM
#define MVLPT_GLDS_ASM 1      // LDS-DMA of this file from inline asm (common.h)
#endif
#include "kernels.h"
#include "gemm_epi.h"

namespace mvlpt {

// Debug timeline (tools/gemm_trace.py): -DMVLPT_GEMM_TRACE builds record s_memtime at fixed points of workgroup 0.
#ifdef MVLPT_GEMM_TRACE
constexpr int TR_MAX = 2048;
#define MVLPT_TR(p) do { if (g.trace && blockIdx.x == 0 && lane == 0 && tr_n < TR_MAX) \
    g.trace[wave * TR_MAX + tr_n++] = ((long long)(p) << 56) | ((long long)__builtin_amdgcn_s_memtime() & 0xffffffffffffffLL); } while (0)
#else
#define MVLPT_TR(p) do { } while (0)
#endif
#ifndef MVLPT_NS2_MODE
#define MVLPT_NS2_MODE 1
#endif
#ifndef MVLPT_FRAG_DEPTH
#define MVLPT_FRAG_DEPTH 2
#endif
#ifndef MVLPT_NS2_POS
#define MVLPT_NS2_POS (GROUPS / 2 - 2)   // after the 3rd of 8 MFMA groups; later positions expose the DMA latency (measured)
#endif

// BM_ x 128 tile, NW waves arranged (NW/2) x 2, NS-deep LDS ring.  Persistent: gridDim.x resident workgroups walk
// the tile list in rounds and keep the LDS-DMA pipeline running ACROSS tile boundaries (the first K-stages of the
// next tile are in flight while the current tile finishes and its epilogue is stored), so the short-K GEMMs of
// this path (K = 768 / 512) do not pay a load bubble per tile.
template <typename T, int EPI, int BM_, int BN_, int NW, int NS, bool MIXED>
__global__ __launch_bounds__(NW * 64, (NW == 2 || BM_ * BN_ / NW > 8192) ? 1 : 2) void gemm_bt_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using v8 = typename Vec<T>::v8;
  constexpr int BN = BN_;
  constexpr int WCN = BN_ / 64, WCM = NW / WCN;       // waves along N / M
  constexpr int WMF = BM_ / WCM / 16;                 // 16-row A fragments per wave (4: 64x64 wave tile, 8: 128x64)
  constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int A_IT = BM_ / 8 / NW, B_IT = BN / 8 / NW, LOADS = A_IT + B_IT;
  constexpr int ST_MIN_ = (WMF / 4) * ((epi_base(EPI) == EPI_STORE16 || epi_base(EPI) == EPI_GELU) ? 8 : 16);
  constexpr bool CAN_FOLD = epi_folds(EPI);
  [[maybe_unused]] char* const xlds = smem + NS * STAGE;        // LayerNorm folding: XLDS_BYTES(_WIDE) behind the ring (when launched with them)
  [[maybe_unused]] char* const xtab = xlds + xlds_tab(g.fold_ntp);
  // Debug builds of the stress tools (tools/fold_consumer_repro.py): every register / every LDS byte of the workgroup holds a
  // signalling pattern before the kernel's first instruction of its own — a read of anything the kernel did not write shows up
  // as a NaN at a position that names it, with or without a partner on the other stream.
#ifdef MVLPT_DBG_POISON_VGPR
  if constexpr (BM_ == 128 && NW == 4 && NS == 2) {      // (v1..v199: the geometry with 256 registers per wave to spare)
#include MVLPT_DBG_POISON_VGPR
  }
#endif
#ifdef MVLPT_DBG_POISON_LDS
  {
    const int bytes = NS * STAGE + (CAN_FOLD ? xlds_bytes(g.fold_ntp) : (epi_ln_producer(EPI) ? XLDS_BYTES : 0));
    for (int i = threadIdx.x; i < bytes / 4; i += NW * 64) ((unsigned*)smem)[i] = 0x7fc0beefu;
    __syncthreads();
  }
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = g.M, N = g.N, K = g.K;
  const int lda = g.lda ? g.lda : (g.a_split ? 2 * K : K);
  const int ldb = g.ldb ? g.ldb : K;
  const T* __restrict__ A = (const T*)g.A;
  const T* __restrict__ Bt = (const T*)g.Bt;
#ifdef MVLPT_GEMM_TRACE
  int tr_n = 0;
#endif

  // XCD-aware order: workgroup b runs on XCD b%8; inside a full round each XCD gets G/8 consecutive tiles
  // (tiles are N-fastest, so neighbours share their A panel in that XCD's L2).
  const int G = gridDim.x, b = blockIdx.x;
  const int tilesN = (N + BN - 1) / BN;
  const int ntiles = ((M + BM_ - 1) / BM_) * tilesN;
  const int gq = G >> 3, gr = G & 7, xcd = b & 7;
  const int b_remap = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + (b >> 3);
  // Tile enumeration: N-fastest (the tiles an XCD runs concurrently share 1-2 A panels).  A grouped, weight-resident
  // enumeration was measured and was not faster (A panels are then re-read from the Infinity Cache once per group).
  auto tile_mn = [&](int t, int& tm, int& tn) { tm = t / tilesN; tn = t - tm * tilesN; };
  auto tile_of = [&](int round) -> int {
    const int base = round * G;
    return base + ((base + G <= ntiles) ? b_remap : b);   // ragged last round: plain order keeps XCDs balanced
  };

  // ---- staging: thread -> (row, 16B chunk) of a 1 KiB LDS slab (8 rows x 128 B); the LDS image is lane-linear,
  //      so the bank swizzle is applied to the SOURCE column: LDS chunk c of row r holds source chunk c ^ (r & 7)
  const int srow = lane >> 3;
  const int scol = ((lane & 7) ^ srow) * 8;
  const T* ap[A_IT];
  const T* bp[B_IT];
  auto set_ptrs = [&](int t) {
    int tm, tn;
    tile_mn(t, tm, tn);
    const int m0 = tm * BM_, n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int ar = m0 + (i * NW + wave) * 8 + srow; ar = ar < M ? ar : M - 1;   // edge rows are re-read, never stored
      ap[i] = A + (size_t)ar * lda + scol;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int br = n0 + (i * NW + wave) * 8 + srow; br = br < N ? br : N - 1;
      bp[i] = Bt + (size_t)br * ldb + scol;
    }
  };
  // load cursor (runs NS-1 K-stages ahead of the compute cursor, across tile boundaries)
  // split-precision A ([M,2K] = [hi | lo], kernels.h): 2K/BK stages, the Bt K-index wraps after K/BK of them.
  // Mixed pair (a_split == 2): K/BK 16-bit stages, then K/128 fp8 stages of the same byte size (128-byte rows = 128
  // k-values); the fp8 planes follow the 16-bit ones in the A and Bt rows, so stage f sits at element offset f*BK of both.
  // (MIXED is a template parameter: with the fp8 loop compiled into every instantiation the single-operand kernels carry
  // ~17 more VGPRs and the 256x256 residual epilogue spills: K = 3072, N = 768 went 267 -> 290 us)
  constexpr bool mixed = MIXED;
  const int nkb = K / BK, nk = mixed ? nkb + nkb / 2 : (g.a_split ? 2 * nkb : nkb);
  int lround = 0, lt = tile_of(0), lkt = 0, lslot = 0;
  if (lt >= ntiles) return;
  set_ptrs(lt);
  auto issue = [&]() -> bool {
    if (lt >= ntiles) return false;
    char* base = smem + lslot * STAGE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) glds16(ap[i] + lkt * BK, base + (i * NW + wave) * 1024);
    const int bk = ((lkt >= nkb && !mixed) ? lkt - nkb : lkt) * BK;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) glds16(bp[i] + bk, base + A_BYTES + (i * NW + wave) * 1024);
    if constexpr (CAN_FOLD) {
      // LayerNorm folding: the {sum, sum of squares} partials of this tile's rows, [row][ntp] float2 = one contiguous block
      // of the global array, copied behind the ring with the tile's LAST K-stage: issued after the barrier that ends the
      // previous tile's epilogue (nk >= NS), landed — like the stage itself — before this tile's epilogue
      if (lkt == nk - 1) {
        const int cpr = g.fold_ntp >> 1;                       // 16-byte chunks per row
        int ltm, ltn;
        tile_mn(lt, ltm, ltn);
        const long first = (long)ltm * BM_ * cpr, last = (long)M * cpr - 1;
        for (int q0 = wave * 64; q0 < BM_ * cpr; q0 += NW * 64) {
          long q = first + q0 + lane; q = q < last ? q : last;
          glds16(g.fold_part + q * 4, xlds + q0 * 16);
        }
        // ... and the tile's slices of W gamma and b + W beta (BN floats each: the epilogue reads them with ds_read)
        const int n0 = ltn * BN + (lane < BN / 4 ? lane : BN / 4 - 1) * 4;
        if (wave == NW - 1) glds16(g.fold_colsum + n0, xtab + XLDS_COLSUM);
        if (wave == NW - 2) glds16(g.bias + n0, xtab + XLDS_BIAS);
      }
    }
    lslot = lslot + 1 == NS ? 0 : lslot + 1;
    if (++lkt == nk) {
      lkt = 0;
      lt = tile_of(++lround);
      if (lt < ntiles) set_ptrs(lt);
    }
    return true;
  };

  // ---- fragment addressing ------------------------------------------------------------------------
  const int wm = wave / WCN, wn = wave % WCN;
  const int fr = lane & 15, fg = lane >> 4;
  const int a_off = (wm * (WMF * 16) + fr) * 128;
  const int b_off = A_BYTES + (wn * 64 + fr) * 128;
  const int c0 = ((0 + fg) ^ (fr & 7)) * 16;        // k-step 0 chunk
  const int c1 = ((4 + fg) ^ (fr & 7)) * 16;        // k-step 1 chunk
  // fp8 stage: the lane's 32 consecutive k-bytes (k = 32 fg .. 32 fg + 31) are source chunks 2 fg and 2 fg + 1
  [[maybe_unused]] const int e0 = ((2 * fg) ^ (fr & 7)) * 16, e1 = ((2 * fg + 1) ^ (fr & 7)) * 16;
  [[maybe_unused]] const int sc_w = 127 - g.w8_exp, sc_a = 127 - Lo8<T>::EXP;      // e8m0 scale bytes: undo the exponents of the two fp8 planes

  // prologue: fill NS-1 ring slots, wait for the first.  `n_issued` K-stages have been requested so far; a wait that
  // must guarantee stage j may leave the n_issued - (j + 1) younger stages in flight (vmcnt retires in order).
  int n_issued = 0, n_done = 0;
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) n_issued += issue() ? 1 : 0;
  auto wait_stage = [&](int younger, bool skip_stores) {
    if (younger >= 2 && NS >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
    else if (younger >= 1 && NS >= 3) {
      if constexpr (LOADS + ST_MIN_ <= 63) {
        if (skip_stores) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS + ST_MIN_) : "memory"); return; }
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  wait_stage(n_issued - 1, false);
  __builtin_amdgcn_s_barrier();

  bool stores_pending = false;
  int slot = 0;
  int t = tile_of(0);
  for (int round = 0; t < ntiles; t = tile_of(++round)) {
    f32x4 acc[WMF / 4][4][4];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i >> 2][i & 3][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One K-stage = [stage_pre: the leading waves' DMA] [MFMA body] [stage_post: trailing DMA, counted wait, barrier].
    // The 16-bit stages and the fp8 stages of a mixed pair run as TWO loops with one body each: with both bodies behind a
    // branch in one loop hipcc stops updating the accumulators in place (the MFMA destinations become fresh registers:
    // +64 VGPRs on a 64x64 wave tile, spills on 128x64).
    const bool dma_first = (NW == 4) || (NS == 2 && MVLPT_NS2_MODE == 0) || (wave < NW / 2);
    constexpr bool DMA_MID = NS == 2 && MVLPT_NS2_MODE == 1 && NW != 4;
    bool issued = false;
    auto stage_pre = [&]() {
      // K-stage f + NS - 1 goes to the slot freed by the last barrier.  An LDS-DMA instruction costs its wave
      // ~100 issue cycles, as much per K-stage as the wave's 32 MFMAs; with two waves per SIMD (8-wave geometry)
      // the older wave issues its DMA BEFORE its MFMAs and the younger one AFTER, so on every SIMD one wave
      // multiplies while its partner is busy with the memory pipe instead of both doing the same thing.
      // With a 2-deep ring the DMA issued in this stage is consumed right after the barrier that ends it, so nobody may
      // issue it at the END of the stage (its whole latency would be exposed): there the younger wave issues in the
      // MIDDLE of its MFMA groups instead (MVLPT_NS2_MODE 1; 0 = every wave first).
      issued = false;
      MVLPT_TR(1);
      if (dma_first) { issued = issue(); MVLPT_TR(2); }
    };
    auto stage_post = [&]() {
      MVLPT_TR(4);
      if (!DMA_MID && !dma_first) { issued = issue(); MVLPT_TR(2); }
      n_issued += issued ? 1 : 0;
      // the NEXT stage (n_done + 1) must have landed (own loads) before the barrier; younger ones may stay in flight.
      // vmcnt retires in order and counts stores: right after an epilogue the youngest operations are its ST_MIN
      // global stores followed by the DMA just issued, so allowing ST_MIN + LOADS operations in flight still
      // guarantees the older DMA has landed without making the wave wait for its own output stores (NS == 3 only).
      wait_stage(n_issued - (n_done + 2), stores_pending && NS == 3 && issued);
      MVLPT_TR(5);
      stores_pending = false;
      ++n_done;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      MVLPT_TR(7);
      slot = slot + 1 == NS ? 0 : slot + 1;
        };
    const int nk16 = mixed ? nkb : nk;
    for (int kt = 0; kt < nk16; ++kt) {
      stage_pre();
      const char* base = smem + slot * STAGE;
      // Register-buffered fragment pipeline: the ds_reads of MFMA group s+1 (two A fragments, plus the four B fragments
      // when the k-step changes) are issued BEFORE the 8 MFMAs of group s, so hipcc's counted lgkmcnt lets the LDS
      // latency run under the matrix pipe instead of in front of every 8-MFMA burst.  (-DMVLPT_FRAG_DEPTH=3, two groups
      // of lookahead, measured 5 % SLOWER on long-K shapes: 8192^3 1.23 vs 1.30 PF in the same run.)
      constexpr int PAIRS = WMF / 2, GROUPS = 2 * PAIRS;
      constexpr int DEPTH = MVLPT_FRAG_DEPTH;          // 2 = double-buffered (one group ahead), 3 = two groups ahead
      v8 bfr[2][4], afr[DEPTH][2];
      auto load_b = [&](int ks, v8 (&bf)[4]) {
        const int c = ks ? c1 : c0;
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *(const v8*)(base + b_off + j * 2048 + c);
      };
      auto load_a2 = [&](int ks, int pair, v8 (&af)[2]) {
        const int c = ks ? c1 : c0;
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *(const v8*)(base + a_off + (pair * 2 + i) * 2048 + c);
      };
      load_b(0, bfr[0]);
#pragma unroll
      for (int p0 = 0; p0 < DEPTH - 1; ++p0) load_a2(p0 / PAIRS, p0 % PAIRS, afr[p0]);
#pragma unroll
      for (int sg = 0; sg < GROUPS; ++sg) {
        const int ks = sg / PAIRS, pair = sg % PAIRS, cur = sg % DEPTH;
        // The prefetch reads are placed AFTER the first MFMA of group sg: hipcc waits in front of the first MFMA that
        // needs LDS data, and with the reads in front of it that wait would also cover the reads just issued.
        __builtin_amdgcn_sched_barrier(0);
        {
          const int ai = pair * 2;
          acc[ai >> 2][ai & 3][0] = mfma16<T>(bfr[ks & 1][0], afr[cur][0], acc[ai >> 2][ai & 3][0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (sg + DEPTH - 1 < GROUPS) {
          const int n = sg + DEPTH - 1, nks = n / PAIRS, npair = n % PAIRS;
          if (npair == 0) load_b(nks, bfr[nks & 1]);
          load_a2(nks, npair, afr[n % DEPTH]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (i == 0 && j == 0) continue;
            const int ai = pair * 2 + i;
            acc[ai >> 2][ai & 3][j] = mfma16<T>(bfr[ks & 1][j], afr[cur][i], acc[ai >> 2][ai & 3][j]);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (DMA_MID && sg == (MVLPT_NS2_POS) && !dma_first) { MVLPT_TR(3); issued = issue(); MVLPT_TR(2); }
      }
      stage_post();
    }
    if constexpr (MIXED)
    for (int kt = nk16; kt < nk; ++kt) {
      stage_pre();
      const char* base = smem + slot * STAGE;
        // fp8 stage of the mixed pair: ONE k-step of 128 on v_mfma_scale_f32_16x16x128_f8f6f4 (32 cycles per instruction:
        // the same matrix time per stage as the 64 16-bit MFMAs below, for twice the K).  B fragments (weights, e4m3) of the
        // whole stage in registers, A fragments (residual bytes, e5m2) in a ring of two: the next one is read behind the
        // first MFMA of the current one.
        i32x8 b8[4], a8[2];
        auto ld8 = [&](int off) -> i32x8 {
          const i32x4 x = *(const i32x4*)(base + off + e0), y = *(const i32x4*)(base + off + e1);
          return __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) b8[j] = ld8(b_off + j * 2048);
        a8[0] = ld8(a_off);
#pragma unroll
        for (int i = 0; i < WMF; ++i) {
          __builtin_amdgcn_sched_barrier(0);
          mfma_lo8(b8[0], a8[i & 1], acc[i >> 2][i & 3][0], sc_w, sc_a);
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 < WMF) a8[(i + 1) & 1] = ld8(a_off + (i + 1) * 2048);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 1; j < 4; ++j)
            mfma_lo8(b8[j], a8[i & 1], acc[i >> 2][i & 3][j], sc_w, sc_a);
          __builtin_amdgcn_sched_barrier(0);
          if (DMA_MID && i == (WMF * 3 / 8 - 1) && !dma_first) { MVLPT_TR(3); issued = issue(); MVLPT_TR(2); }
        }
        mfma_lo8_fence();
      stage_post();
    }
    MVLPT_TR(8);
    // the slot the load cursor will fill next has just been released by the barrier above: use it as scratch,
    // and fence the scratch reads of all waves against that DMA with one more barrier
    int tm, tn;
    tile_mn(t, tm, tn);
    if constexpr (CAN_FOLD) {
      // LayerNorm folding: the rows' {rstd, -rstd * mean} once per tile (the partials landed with the last K-stage), then one
      // barrier; the epilogue reads 8 bytes per row instead of re-deriving them in every lane
      if (tid < BM_) fold_build_coef(g, xlds, xtab, tid);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int hh = 0; hh < WMF / 4; ++hh)
      epilogue_store<T, EPI>(g, acc[hh], tm * BM_ + wm * (WMF * 16) + hh * 64, tn * BN + wn * 64, lane,
                             LinearRows<144>{smem + lslot * STAGE + wave * EPI_SCRATCH_PER_WAVE},
                             LinearRows<272>{smem + lslot * STAGE + wave * EPI_SCRATCH_PER_WAVE},
                             FoldCtx{xlds, xtab, wm * (WMF * 16) + hh * 64, wn, WCN, MIXED ? 2 : (g.ln_split ? 1 : 0)});
    MVLPT_TR(9);
    __builtin_amdgcn_s_barrier();
    if constexpr (epi_ln_producer(EPI)) {
      // the tile's row partials: one 8-byte slot per (row, 128 output columns) = the sum of two wave column blocks, whatever
      // the tile geometry — the statistics a row gets (and the order they are summed in) do not depend on the geometry the
      // launcher picks for the batch size (tests/test_hip_properties.py: a batch in two halves equals the whole bit for bit).
      // The region is rewritten by the next tile's epilogue, nk barriers from here
      if (tid < BM_) {
        const int row = tm * BM_ + tid;
        if (row < M) {
          const float2* p = (const float2*)(xlds + (size_t)tid * WCN * 8);
#pragma unroll
          for (int k = 0; k < WCN / 2; ++k)
            *(float2*)(g.ln_part + ((size_t)row * g.ln_ntp + tn * (WCN / 2) + k) * 2) = float2{p[2 * k].x + p[2 * k + 1].x, p[2 * k].y + p[2 * k + 1].y};
        }
      }
    }
    stores_pending = (tm + 1) * BM_ <= M;
  }
}

#ifdef MVLPT_BREG
// ---------------------------------------------------------------------------------------------- experiment (debug builds only)
// "weight fragments straight from L2 into registers, the whole LDS ring for A".  256x256 tile, 8 waves of 128x64
// as above, but only the A operand travels through LDS (ring of MVLPT_BREG_NS stages of 32 KiB, NS-1 stages ahead); every wave reads
// the B fragments of ITS 64 columns with global_load_dwordx4 into registers, one K-stage ahead (two buffers of 32 VGPRs, the
// K loop unrolled by two so that both are statically indexed).  vmcnt retires in order: the wait for B(f+1) at the end of stage f
// may leave only the A pieces issued behind it in flight.  Dedicated epilogue scratch (a 32-KiB slot does not hold the 36 KiB).
// tools/build_variant.sh breg -DMVLPT_BREG; MVLPT_GEMM_BREG=1 selects it for the single-operand 256x256 launches.
#ifndef MVLPT_BREG_NS
#define MVLPT_BREG_NS 3
#endif
#include <type_traits>
template <typename T, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_breg_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using v8 = typename Vec<T>::v8;
  constexpr int BM_ = 256, BN = 256, NW = 8, NS = MVLPT_BREG_NS, WCN = 4, WMF = 8;
  constexpr int STAGE = BM_ * BK * 2, A_IT = BM_ / 8 / NW;
  constexpr bool CAN_FOLD = epi_folds(EPI);
  char* const scr = smem + NS * STAGE;
  [[maybe_unused]] char* const xlds = scr + NW * EPI_SCRATCH_PER_WAVE;
  [[maybe_unused]] char* const xtab = xlds + xlds_tab(g.fold_ntp);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = g.M, N = g.N, K = g.K;
  const int lda = g.lda ? g.lda : K, ldb = g.ldb ? g.ldb : K;
  const T* __restrict__ A = (const T*)g.A;
  const T* __restrict__ Bt = (const T*)g.Bt;
  const int G = gridDim.x, b = blockIdx.x;
  const int tilesN = N / BN;
  const int ntiles = ((M + BM_ - 1) / BM_) * tilesN;
  const int gq = G >> 3, gr = G & 7, xcd = b & 7;
  const int b_remap = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + (b >> 3);
  auto tile_mn = [&](int t, int& tm, int& tn) { tm = t / tilesN; tn = t - tm * tilesN; };
  auto tile_of = [&](int round) -> int {
    const int base = round * G;
    return base + ((base + G <= ntiles) ? b_remap : b);
  };
  const int srow = lane >> 3, scol = ((lane & 7) ^ srow) * 8;
  // (M % 256 == 0 in this experiment: no edge rows to clamp -> a wave-uniform base per tile + ONE 32-bit lane offset)
  const char* ap_base;
  const unsigned ap_voff = (unsigned)(srow * lda + scol) * 2u;
  auto set_ptrs = [&](int t) {
    int tm, tn;
    tile_mn(t, tm, tn);
    ap_base = (const char*)(A + (size_t)(tm * BM_ + wave * 8) * lda);
  };
  const int nk = K / BK;
  int lround = 0, lt = tile_of(0), lkt = 0, lslot = 0;
  if (lt >= ntiles) return;
  set_ptrs(lt);
  auto issue = [&]() -> bool {
    if (lt >= ntiles) return false;
    char* base = smem + lslot * STAGE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) glds16(ap_base + ((size_t)i * NW * 8 * lda + lkt * BK) * 2 + ap_voff, base + (i * NW + wave) * 1024);
    if constexpr (CAN_FOLD) {
      if (lkt == nk - 1) {
        const int cpr = g.fold_ntp >> 1;
        int ltm, ltn;
        tile_mn(lt, ltm, ltn);
        const long first = (long)ltm * BM_ * cpr, last = (long)M * cpr - 1;
        for (int q0 = wave * 64; q0 < BM_ * cpr; q0 += NW * 64) {
          long q = first + q0 + lane; q = q < last ? q : last;
          glds16(g.fold_part + q * 4, xlds + q0 * 16);
        }
        const int n0 = ltn * BN + (lane < BN / 4 ? lane : BN / 4 - 1) * 4;
        if (wave == NW - 1) glds16(g.fold_colsum + n0, xtab + XLDS_COLSUM);
        if (wave == NW - 2) glds16(g.bias + n0, xtab + XLDS_BIAS);
      }
    }
    lslot = lslot + 1 == NS ? 0 : lslot + 1;
    if (++lkt == nk) {
      lkt = 0;
      lt = tile_of(++lround);
      if (lt < ntiles) set_ptrs(lt);
    }
    return true;
  };
  const int wm = wave / WCN, wn = wave % WCN;
  const int fr = lane & 15, fg = lane >> 4;
  const int a_off = (wm * (WMF * 16) + fr) * 128;
  const int c0 = ((0 + fg) ^ (fr & 7)) * 16, c1 = ((4 + fg) ^ (fr & 7)) * 16;
  // the lane's B rows: fragment j = weight row n0 + wn*64 + j*16 + fr, k-step ks = elements ks*32 + fg*8 .. +7 of the stage
  // (wave-uniform base in SGPRs + ONE 32-bit lane offset: the saddr form of global_load)
  const char* bq_base;
  const unsigned bq_voff = (unsigned)(fr * ldb + fg * 8) * 2u;
  auto set_bptr = [&](int t) {
    int tm, tn;
    tile_mn(t, tm, tn);
    bq_base = (const char*)(Bt + (size_t)(tn * BN + wn * 64) * ldb);
  };
  v8 bq[2][2][4];
  auto load_bq = [&](v8 (&q)[2][4], int kt) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // from inline asm (hipcc's own loads take a 64-bit VGPR address per fragment and spill the fragments): the compiler does not
        // know the result is pending — every use sits behind the explicit vmcnt wait + barrier that ends the stage
        const char* bj = bq_base + ((size_t)j * 16 * ldb + kt * BK) * 2;
        if (ks == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(q[ks][j]) : "v"(bq_voff), "s"(bj) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(q[ks][j]) : "v"(bq_voff), "s"(bj) : "memory");
      }
  };
  int t = tile_of(0);
  set_bptr(t);
  load_bq(bq[0], 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * A_IT) : "memory");
  __builtin_amdgcn_s_barrier();

  int slot = 0;
  for (int round = 0; t < ntiles; t = tile_of(++round)) {
    f32x4 acc[WMF / 4][4][4];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i >> 2][i & 3][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool dma_first = wave < NW / 2;
    const char* bq_next;
    {
      const int nt = tile_of(round + 1);
      int tm, tn;
      tile_mn(nt < ntiles ? nt : t, tm, tn);
      bq_next = (const char*)(Bt + (size_t)(tn * BN + wn * 64) * ldb);
    }
    auto stage = [&](auto curc, int kt) {
      constexpr int CUR = decltype(curc)::value;
      // B of the next stage first (so that the A pieces issued behind it may stay in flight across the wait below)
      // (branch-free: behind the tile's last stage comes stage 0 of the next tile — of this one again when there is none)
      __builtin_amdgcn_sched_barrier(0);
      {
        const bool last = kt + 1 >= nk;
        bq_base = last ? bq_next : bq_base;
        load_bq(bq[CUR ^ 1], last ? 0 : kt + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      bool issued = false;
      if (dma_first) issued = issue();
      const char* base = smem + slot * STAGE;
      constexpr int PAIRS = WMF / 2, GROUPS = 2 * PAIRS;
      v8 afr[2][2];
      auto load_a2 = [&](int ks, int pair, v8 (&af)[2]) {
        const int c = ks ? c1 : c0;
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *(const v8*)(base + a_off + (pair * 2 + i) * 2048 + c);
      };
      load_a2(0, 0, afr[0]);
#pragma unroll
      for (int sg = 0; sg < GROUPS; ++sg) {
        const int ks = sg / PAIRS, pair = sg % PAIRS, cur = sg & 1;
        __builtin_amdgcn_sched_barrier(0);
        {
          const int ai = pair * 2;
          acc[ai >> 2][ai & 3][0] = mfma16<T>(bq[CUR][ks][0], afr[cur][0], acc[ai >> 2][ai & 3][0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (sg + 1 < GROUPS) load_a2((sg + 1) / PAIRS, (sg + 1) % PAIRS, afr[(sg + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (i == 0 && j == 0) continue;
            const int ai = pair * 2 + i;
            acc[ai >> 2][ai & 3][j] = mfma16<T>(bq[CUR][ks][j], afr[cur][i], acc[ai >> 2][ai & 3][j]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!dma_first) issued = issue();
      // A(f+1) (issued NS-2 stages ago) and B(f+1) (this stage) must have landed; only the A pieces issued behind B may stay
      if (issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      slot = slot + 1 == NS ? 0 : slot + 1;
    };
    for (int kt = 0; kt < nk; kt += 2) {
      stage(std::integral_constant<int, 0>{}, kt);
      stage(std::integral_constant<int, 1>{}, kt + 1);
    }
    int tm, tn;
    tile_mn(t, tm, tn);
    if constexpr (CAN_FOLD) {
      if (tid < BM_) fold_build_coef(g, xlds, xtab, tid);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int hh = 0; hh < WMF / 4; ++hh)
      epilogue_store<T, EPI>(g, acc[hh], tm * BM_ + wm * (WMF * 16) + hh * 64, tn * BN + wn * 64, lane,
                             LinearRows<144>{scr + wave * EPI_SCRATCH_PER_WAVE}, LinearRows<272>{scr + wave * EPI_SCRATCH_PER_WAVE},
                             FoldCtx{xlds, xtab, wm * (WMF * 16) + hh * 64, wn, WCN, g.ln_split ? 1 : 0});
    __builtin_amdgcn_s_barrier();
    if constexpr (epi_ln_producer(EPI)) {
      if (tid < BM_) {
        const int row = tm * BM_ + tid;
        if (row < M) {
          const float2* p = (const float2*)(xlds + (size_t)tid * WCN * 8);
#pragma unroll
          for (int k = 0; k < WCN / 2; ++k)
            *(float2*)(g.ln_part + ((size_t)row * g.ln_ntp + tn * (WCN / 2) + k) * 2) = float2{p[2 * k].x + p[2 * k + 1].x, p[2 * k].y + p[2 * k + 1].y};
        }
      }
    }
  }
}
template <typename T, int EPI>
static hipError_t launch_breg(const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  constexpr int LDS = MVLPT_BREG_NS * 256 * BK * 2 + 8 * EPI_SCRATCH_PER_WAVE;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_breg_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const int lds = LDS + (epi_folds(EPI) ? xlds_bytes(g.fold_ntp) : (epi_ln_producer(EPI) ? XLDS_BYTES : 0));
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int cus = stream_cus(s);
  const int tiles = ((g.M + 255) / 256) * (g.N / 256);
  hipExtLaunchKernelGGL((gemm_breg_kernel<T, EPI>), dim3(tiles < cus ? tiles : cus), dim3(512), lds, s, ea, eb, 0, g);
  return hipGetLastError();
}
#endif

// ---------------------------------------------------------------------------------------------- phased variant
// 256x128 tile, 8 waves, 3-deep ring, same data movement as above, but every K-stage is cut into FOUR barrier-
// separated phases  R0 | M0 | R1 | M1  (R = LDS-DMA issue + ds_read of one 32-deep k-step, M = its 16 MFMAs) and waves
// 4-7 run ONE phase behind waves 0-3 (they take one extra barrier up front, waves 0-3 one extra at the end).  Each
// SIMD hosts one wave of either group, so while one wave multiplies, its partner is in its memory phase, instead of
// both contending for the matrix pipe and then both waiting on LDS / DMA.
//   RAW: the DMA of stage f+1 is waited for (own pieces, counted vmcnt) at the END of R1(f); the leading group reads
//        it in R0(f+1), i.e. after the barrier that the trailing group only passes once ITS wait in R1(f) is done.
//   WAR: stage f+2 goes to the slot of stage f-1, last read by the trailing group in its R1(f-1), which is over when
//        the leading group enters R0(f) (one barrier later) and issues the DMA.
//   Epilogue scratch = the wave's OWN DMA slabs of the released slot, so the partner group's DMA of the next stage
//        (issued one phase earlier / later) never touches it; no barrier is needed around the epilogue.
template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bt_phased_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using v8 = typename Vec<T>::v8;
  constexpr int BM_ = 256, BN = 128, NW = 8, NS = 3;
  constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int A_IT = BM_ / 8 / NW, B_IT = BN / 8 / NW, LOADS = A_IT + B_IT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool trail = wave >= NW / 2;
  const int M = g.M, N = g.N, K = g.K;
  const int lda = g.lda ? g.lda : (g.a_split ? 2 * K : K);
  const int ldb = g.ldb ? g.ldb : K;
  const T* __restrict__ A = (const T*)g.A;
  const T* __restrict__ Bt = (const T*)g.Bt;

  const int G = gridDim.x, b = blockIdx.x;
  const int tilesN = N / BN;
  const int ntiles = ((M + BM_ - 1) / BM_) * tilesN;
  const int gq = G >> 3, gr = G & 7, xcd = b & 7;
  const int b_remap = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + (b >> 3);
  auto tile_of = [&](int round) -> int {
    const int base = round * G;
    return base + ((base + G <= ntiles) ? b_remap : b);
  };
  const int srow = lane >> 3;
  const int scol = ((lane & 7) ^ srow) * 8;
  const T* ap[A_IT];
  const T* bp[B_IT];
  auto set_ptrs = [&](int t) {
    const int m0 = (t / tilesN) * BM_, n0 = (t % tilesN) * BN;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int ar = m0 + (i * NW + wave) * 8 + srow; ar = ar < M ? ar : M - 1;
      ap[i] = A + (size_t)ar * lda + scol;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int br = n0 + (i * NW + wave) * 8 + srow;
      bp[i] = Bt + (size_t)br * ldb + scol;
    }
  };
  // split-precision A ([M,2K] = [hi | lo], kernels.h): 2K/BK stages, the Bt K-index wraps after K/BK of them
  const int nkb = K / BK, nk = g.a_split ? 2 * nkb : nkb;
  int lround = 0, lt = tile_of(0), lkt = 0, lslot = 0;
  if (lt >= ntiles) return;
  set_ptrs(lt);
  auto issue = [&]() -> bool {
    if (lt >= ntiles) return false;
    char* base = smem + lslot * STAGE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) glds16(ap[i] + lkt * BK, base + (i * NW + wave) * 1024);
    const int bk = (lkt >= nkb ? lkt - nkb : lkt) * BK;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) glds16(bp[i] + bk, base + A_BYTES + (i * NW + wave) * 1024);
    lslot = lslot + 1 == NS ? 0 : lslot + 1;
    if (++lkt == nk) {
      lkt = 0;
      lt = tile_of(++lround);
      if (lt < ntiles) set_ptrs(lt);
    }
    return true;
  };
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int a_off = (wm * 64 + fr) * 128;
  const int b_off = A_BYTES + (wn * 64 + fr) * 128;
  const int c0 = ((0 + fg) ^ (fr & 7)) * 16;
  const int c1 = ((4 + fg) ^ (fr & 7)) * 16;

  bool more = issue();
  more = issue();
  if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (trail) __builtin_amdgcn_s_barrier();          // the trailing group stays one phase behind from here on

  constexpr int ST_MIN = (EPI == EPI_STORE16 || EPI == EPI_GELU) ? 8 : 16;
  bool stores_pending = false;
  int slot = 0;
  int t = tile_of(0);
  for (int round = 0; t < ntiles; t = tile_of(++round)) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt) {
      const char* base = smem + slot * STAGE;
      v8 af[4], bf[4];
      // ---- R0: DMA of stage f+2 into the slot released two barriers ago; fragments of k-step 0
      const bool issued = issue();
#pragma unroll
      for (int i = 0; i < 4; ++i) { bf[i] = *(const v8*)(base + b_off + i * 2048 + c0); af[i] = *(const v8*)(base + a_off + i * 2048 + c0); }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- M0  (s_setprio(1) around the MFMA clusters was measured: no effect on this structure)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(bf[j], af[i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- R1: fragments of k-step 1; own pieces of stage f+1 must have landed before the barrier
#pragma unroll
      for (int i = 0; i < 4; ++i) { bf[i] = *(const v8*)(base + b_off + i * 2048 + c1); af[i] = *(const v8*)(base + a_off + i * 2048 + c1); }
      if (issued) {
        if (stores_pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS + ST_MIN) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      stores_pending = false;
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- M1
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(bf[j], af[i], acc[i][j]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      slot = slot + 1 == NS ? 0 : slot + 1;
    }
    const int tm = t / tilesN, tn = t - tm * tilesN;
    epilogue_store<T, EPI>(g, acc, tm * BM_ + wm * 64, tn * BN + wn * 64, lane,
                           SlabRows<144>{smem + lslot * STAGE, wave, A_BYTES}, SlabRows<272>{smem + lslot * STAGE, wave, A_BYTES},
                           FoldCtx{nullptr, nullptr, 0, 0, 0, 0});     // (launch_one never sends a folded GEMM here)
    stores_pending = (tm + 1) * BM_ <= M;
  }
  if (!trail) __builtin_amdgcn_s_barrier();         // balance the extra barrier of the trailing group
}

template <typename T, int EPI>
static hipError_t launch_phased(const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  constexpr int LDS = 3 * (256 + 128) * BK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_bt_phased_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int cus = stream_cus(s);
  const int tiles = ((g.M + 255) / 256) * (g.N / 128);
  hipExtLaunchKernelGGL((gemm_bt_phased_kernel<T, EPI>), dim3(tiles < cus ? tiles : cus), dim3(512), LDS, s, ea, eb, 0, g);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- producer / consumer variant
// Problems with no more 128x128 tiles than compute units (the text tower at M = C*L ~ 7.7k rows, N = 512: 244 tiles) leave every
// CU with ONE 4-wave workgroup, i.e. one wave per SIMD, and that wave issues everything itself: per K-stage 8 LDS-DMA instructions
// (~80 issue cycles each), 16 fragment reads and 32 MFMAs (512 cycles) — measured 1 900 cycles per stage (K = 2048, mixed pair).
// Here the tile gets EIGHT waves: waves 0-3 (2x2 blocks of 64x64, one per SIMD) only multiply, waves 4-7 (their SIMD partners)
// only move data: they request K-stage f + 3 into the 4-deep ring (three stages in flight, as before), wait for their own pieces
// of stage f with exact vmcnt counts and publish it with the stage barrier; the multiplying waves never touch vmcnt.
//   RAW: stage f is waited for by its requesters in front of barrier f; the consumers read it behind that barrier.
//   WAR: stage f + 3 goes to the slot of stage f - 1, whose readers arrive at barrier f only when they are done with it; the
//        request is issued behind barrier f.
// One tile per workgroup (grid = tiles <= compute units).  A K-split of the tile over two multiplying wave groups was built first
// and measured 40 % SLOWER (text tower 4.29 -> 6.13 ms): its two 64-KiB double stages leave one stage of look-ahead, and these
// under-filled GEMMs are bound by the latency of their L2 / Infinity-Cache reads (NOTES_experiments.md).
template <typename T, int EPI, bool MIXED>
__global__ __launch_bounds__(512, 1) void gemm_pc_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using v8 = typename Vec<T>::v8;
  constexpr int BM_ = 128, BN = 128, NS = 4, LOADS = 8;
  constexpr int A_BYTES = BM_ * BK * 2, STAGE = A_BYTES + BN * BK * 2;
  [[maybe_unused]] char* const xlds = smem + NS * STAGE;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mover = wave >= 4;
  const int w4 = wave & 3;
  const int M = g.M, N = g.N, K = g.K;
  const int tilesN = N / BN;
  // XCD-aware order (as in the persistent kernels): workgroup b runs on XCD b % 8; each XCD takes a run of consecutive tiles
  // (N-fastest), so the tiles that share an A panel share an L2.  In plain order the 4 column tiles of a panel (N = 512)
  // sit on 4 different XCDs and every A panel crosses the fabric 4 times.
  int t = blockIdx.x;
  if (g.xcd_order) {
    const int G = gridDim.x, gq = G >> 3, gr = G & 7, xcd = t & 7;
    t = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + (t >> 3);
  }
  const int tm = t / tilesN, tn = t - tm * tilesN;
  const int m0 = tm * BM_, n0 = tn * BN;
  const int nkb = K / BK, nk = MIXED ? nkb + nkb / 2 : (g.a_split ? 2 * nkb : nkb);
  const int nk16 = MIXED ? nkb : nk;

  if (mover) {
    const int lda = g.lda ? g.lda : (g.a_split ? 2 * K : K);
    const int ldb = g.ldb ? g.ldb : K;
    const T* __restrict__ A = (const T*)g.A;
    const T* __restrict__ Bt = (const T*)g.Bt;
    const int srow = lane >> 3;
    const int scol = ((lane & 7) ^ srow) * 8;
    const T* ap[4];
    const T* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ar = m0 + (i * 4 + w4) * 8 + srow; ar = ar < M ? ar : M - 1;
      ap[i] = A + (size_t)ar * lda + scol;
      const int br = n0 + (i * 4 + w4) * 8 + srow;
      bp[i] = Bt + (size_t)br * ldb + scol;
    }
    auto issue = [&](int f) {
      char* base = smem + (f & (NS - 1)) * STAGE;
      const int bk = (f >= nkb && !MIXED) ? f - nkb : f;
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16(ap[i] + f * BK, base + (i * 4 + w4) * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16(bp[i] + bk * BK, base + A_BYTES + (i * 4 + w4) * 1024);
    };
#pragma unroll
    for (int f = 0; f < NS - 1; ++f) if (f < nk) issue(f);
    for (int f = 0; f < nk; ++f) {
      // requested so far: stages 0 .. min(f + 2, nk - 1); stage f must have landed, the younger ones may stay in flight
      const int younger = (nk - 1 - f) < 2 ? (nk - 1 - f) : 2;
      if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (f + NS - 1 < nk) issue(f + NS - 1);
    }
    if constexpr (epi_ln_producer(EPI)) __builtin_amdgcn_s_barrier();
    return;
  }

  const int wm = w4 >> 1, wn = w4 & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int a_off = (wm * 64 + fr) * 128;
  const int b_off = A_BYTES + (wn * 64 + fr) * 128;
  const int c0 = ((0 + fg) ^ (fr & 7)) * 16;
  const int c1 = ((4 + fg) ^ (fr & 7)) * 16;
  [[maybe_unused]] const int e0 = ((2 * fg) ^ (fr & 7)) * 16, e1 = ((2 * fg + 1) ^ (fr & 7)) * 16;
  [[maybe_unused]] const int sc_w = 127 - g.w8_exp, sc_a = 127 - Lo8<T>::EXP;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int f = 0; f < nk16; ++f) {
    __builtin_amdgcn_s_barrier();
    const char* base = smem + (f & (NS - 1)) * STAGE;
    v8 bf[2][4], af[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[0][j] = *(const v8*)(base + b_off + j * 2048 + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = *(const v8*)(base + a_off + i * 2048 + c0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = mfma16<T>(bf[ks][0], af[ks][0], acc[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 0) {      // the second k-step's fragments travel under the first one's MFMAs
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[1][j] = *(const v8*)(base + b_off + j * 2048 + c1);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[1][i] = *(const v8*)(base + a_off + i * 2048 + c1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (i == 0 && j == 0) continue;
          acc[i][j] = mfma16<T>(bf[ks][j], af[ks][i], acc[i][j]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if constexpr (MIXED)
  for (int f = nk16; f < nk; ++f) {
    __builtin_amdgcn_s_barrier();
    const char* base = smem + (f & (NS - 1)) * STAGE;
    i32x8 b8[4], a8[2];
    auto ld8 = [&](int off) -> i32x8 {
      const i32x4 x = *(const i32x4*)(base + off + e0), y = *(const i32x4*)(base + off + e1);
      return __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) b8[j] = ld8(b_off + j * 2048);
    a8[0] = ld8(a_off);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      mfma_lo8(b8[0], a8[i & 1], acc[i][0], sc_w, sc_a);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < 4) a8[(i + 1) & 1] = ld8(a_off + (i + 1) * 2048);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 1; j < 4; ++j) mfma_lo8(b8[j], a8[i & 1], acc[i][j], sc_w, sc_a);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_lo8_fence();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // epilogue scratch: the slot of the last stage but one (every request has landed; the four multiplying waves only sync with
  // each other through their own reads, so each waits for its partners with the producer barrier below when there is one)
  char* const scr = smem + ((nk + 1) & (NS - 1)) * STAGE + w4 * EPI_SCRATCH_PER_WAVE;
  const FoldCtx fc{xlds, nullptr, wm * 64, wn, 2, MIXED ? 2 : (g.ln_split ? 1 : 0)};
  epilogue_store<T, EPI>(g, acc, m0 + wm * 64, n0 + wn * 64, lane, LinearRows<144>{scr}, LinearRows<272>{scr}, fc);
  if constexpr (epi_ln_producer(EPI)) {
    __builtin_amdgcn_s_barrier();
    if (tid < BM_) {
      const int row = m0 + tid;
      if (row < M) {
        const float2* pp = (const float2*)(xlds + (size_t)tid * 2 * 8);
        *(float2*)(g.ln_part + ((size_t)row * g.ln_ntp + tn) * 2) = float2{pp[0].x + pp[1].x, pp[0].y + pp[1].y};
      }
    }
  }
}

template <typename T, int EPI, bool MIXED>
static hipError_t launch_pc_m(const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  constexpr int LDS = 4 * (128 + 128) * BK * 2 + XLDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_pc_kernel<T, EPI, MIXED>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int tiles = ((g.M + 127) / 128) * (g.N / 128);
  static const int xcd_order = getenv("MVLPT_PC_XCD_ORDER") ? atoi(getenv("MVLPT_PC_XCD_ORDER")) : 1;
  GemmArgs q = g;
  q.xcd_order = xcd_order;
  hipExtLaunchKernelGGL((gemm_pc_kernel<T, EPI, MIXED>), dim3(tiles), dim3(512), LDS, s, ea, eb, 0, q);
  return hipGetLastError();
}
template <typename T, int EPI>
static hipError_t launch_pc(const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  if constexpr (epi_folds(EPI)) return hipErrorInvalidValue;
  else {
    constexpr int BE = epi_base(EPI);
    if constexpr (BE == EPI_RESID32 || BE == EPI_RESID32_LN || BE == EPI_STORE32 || BE == EPI_GELU_SPLIT || BE == EPI_GELUBWD_SPLIT || BE == EPI_STORE_SPLIT) {
      if (g.a_split == 2) return launch_pc_m<T, EPI, true>(g, s, ea, eb);
    } else if (g.a_split == 2) return hipErrorInvalidValue;
    return launch_pc_m<T, EPI, false>(g, s, ea, eb);
  }
}
// the producer / consumer kernel takes the problem: one tile per workgroup, no more tiles than compute units, no consumer-side
// LayerNorm folding
template <int EPI>
static bool pc_takes(const GemmArgs& g, long cus) {
  if (epi_folds(EPI)) return false;
  return (long)((g.M + 127) / 128) * (g.N / 128) <= cus;
}

// ---------------------------------------------------------------------------------------------- persistent, with movers
// The 256x128 geometry (GEMMs with 1.5 .. 4 rounds of tiles: out-projection of the image tower, the text tower's N = 2048 GEMMs)
// spends as many issue cycles on its LDS-DMA as on its MFMAs: 48 KiB per K-stage are 6 DMA instructions per wave (~450 cycles)
// against 512 cycles of MFMA per wave, and a stage takes 2 800 cycles for 1 024 of matrix work per SIMD (tile timelines,
// profiles/r04_gemm_tile_timelines.txt).  Same split of labour as gemm_pc_kernel, persistent: TWELVE waves — 0-7 (4x2 blocks of
// 64x64, two per SIMD) only read fragments and multiply, 8-11 (one per SIMD) walk the workgroup's tile list two K-stages ahead of
// them (3-deep ring, across tile boundaries: the first two stages of the next tile land under the epilogue) and publish every stage
// with exact vmcnt waits + the stage barrier.  64x64 accumulator blocks keep the kernel under the 168 VGPRs three waves per SIMD
// leave.  Tile end: barrier (the last stage's slot is free: epilogue scratch), epilogue, barrier (the movers may overwrite it).
// LayerNorm folding: the movers bring a consumer tile's row partials and column vectors with its last K-stage (the same number
// of instructions each, so the counted waits stay compile-time constants); rows of up to 6 slots (d <= 768).
template <typename T, int EPI, bool MIXED>
__global__ __launch_bounds__(768, 1) void gemm_pcp_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using v8 = typename Vec<T>::v8;
  constexpr int BM_ = 256, BN = 128, NS = 3, NCW = 8, LOADS = 12;
  constexpr int A_BYTES = BM_ * BK * 2, STAGE = A_BYTES + BN * BK * 2;
  constexpr bool CAN_FOLD = epi_folds(EPI);
  [[maybe_unused]] char* const xlds = smem + NS * STAGE;
  [[maybe_unused]] char* const xtab = xlds + xlds_tab(g.fold_ntp);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = g.M, N = g.N, K = g.K;
  const int G = gridDim.x, b = blockIdx.x;
  const int tilesN = N / BN;
  const int ntiles = ((M + BM_ - 1) / BM_) * tilesN;
  const int gq = G >> 3, gr = G & 7, xcd = b & 7;
  const int b_remap = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + (b >> 3);
  auto tile_of = [&](int round) -> int {
    const int base = round * G;
    return base + ((base + G <= ntiles) ? b_remap : b);
  };
  const int nkb = K / BK, nk = MIXED ? nkb + nkb / 2 : (g.a_split ? 2 * nkb : nkb);
  const int nk16 = MIXED ? nkb : nk;
  if (tile_of(0) >= ntiles) return;

  if (wave >= NCW) {
    // ------------------------------------------------------------------------------------------ data movement
    const int w4 = wave - NCW;
    const int lda = g.lda ? g.lda : (g.a_split ? 2 * K : K);
    const int ldb = g.ldb ? g.ldb : K;
    const T* __restrict__ A = (const T*)g.A;
    const T* __restrict__ Bt = (const T*)g.Bt;
    const int srow = lane >> 3;
    const int scol = ((lane & 7) ^ srow) * 8;
    const T* ap[8];
    const T* bp[4];
    int lround = 0, lt = tile_of(0), lkt = 0, lslot = 0;
    auto set_ptrs = [&](int t) {
      const int m0 = (t / tilesN) * BM_, n0 = (t % tilesN) * BN;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int ar = m0 + (i * 4 + w4) * 8 + srow; ar = ar < M ? ar : M - 1;
        ap[i] = A + (size_t)ar * lda + scol;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int br = n0 + (i * 4 + w4) * 8 + srow;
        bp[i] = Bt + (size_t)br * ldb + scol;
      }
    };
    set_ptrs(lt);
    const int cpr = g.fold_ntp >> 1;                 // CAN_FOLD: 16-byte chunks of partials per row = extra DMA instructions per mover
    // -> 0 nothing left to request, 1 a plain stage (LOADS instructions), 2 a tile's last stage with the folding extras
    auto issue = [&]() -> int {
      if (lt >= ntiles) return 0;
      char* base = smem + lslot * STAGE;
      const int bk = ((lkt >= nkb && !MIXED) ? lkt - nkb : lkt) * BK;
#pragma unroll
      for (int i = 0; i < 8; ++i) glds16(ap[i] + lkt * BK, base + (i * 4 + w4) * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16(bp[i] + bk, base + A_BYTES + (i * 4 + w4) * 1024);
      int kind = 1;
      if constexpr (CAN_FOLD) {
        if (lkt == nk - 1) {
          kind = 2;
          const long first = (long)(lt / tilesN) * BM_ * cpr, last = (long)M * cpr - 1;
          for (int j = 0; j < cpr; ++j) {
            const int q0 = (w4 * cpr + j) * 64;
            long q = first + q0 + lane; q = q < last ? q : last;
            glds16(g.fold_part + q * 4, xlds + q0 * 16);
          }
          const int n0 = (lt % tilesN) * BN + (lane < BN / 4 ? lane : BN / 4 - 1) * 4;
          if (w4 & 1) glds16(g.bias + n0, xtab + XLDS_BIAS);          // movers 1, 3 (the same bytes twice: harmless)
          else glds16(g.fold_colsum + n0, xtab + XLDS_COLSUM);        // movers 0, 2
        }
      }
      lslot = lslot + 1 == NS ? 0 : lslot + 1;
      if (++lkt == nk) {
        lkt = 0;
        lt = tile_of(++lround);
        if (lt < ntiles) set_ptrs(lt);
      }
      return kind;
    };
    (void)issue();
    int ahead = issue();                               // kind of the one stage requested beyond the stage waited for next
    for (int round = 0; tile_of(round) < ntiles; ++round) {
      for (int f = 0; f < nk; ++f) {
        // the stage the multiplying waves take next has landed; the one behind it may stay in flight (vmcnt retires in order)
        if (ahead == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        else if (cpr == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS + 4) : "memory");
        else if (cpr == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS + 3) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");          // (more extras than counted: waits for some of them too)
        __builtin_amdgcn_s_barrier();
        ahead = issue();                               // into the slot of the stage consumed before this barrier
      }
      __builtin_amdgcn_s_barrier();                    // tile end: the multiplying waves take the last slot as scratch ...
      if constexpr (CAN_FOLD) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();                    // ... and give it back
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------- multiplication
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int a_off = (wm * 64 + fr) * 128;
  const int b_off = A_BYTES + (wn * 64 + fr) * 128;
  const int c0 = ((0 + fg) ^ (fr & 7)) * 16;
  const int c1 = ((4 + fg) ^ (fr & 7)) * 16;
  [[maybe_unused]] const int e0 = ((2 * fg) ^ (fr & 7)) * 16, e1 = ((2 * fg + 1) ^ (fr & 7)) * 16;
  [[maybe_unused]] const int sc_w = 127 - g.w8_exp, sc_a = 127 - Lo8<T>::EXP;
  int slot = 0;
  for (int round = 0;; ++round) {
    const int t = tile_of(round);
    if (t >= ntiles) break;
    const int tm = t / tilesN, tn = t - tm * tilesN;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int last_slot = 0;
    for (int f = 0; f < nk16; ++f) {
      __builtin_amdgcn_s_barrier();
      const char* base = smem + slot * STAGE;
      last_slot = slot;
      slot = slot + 1 == NS ? 0 : slot + 1;
      v8 bf[2][4], af[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[0][j] = *(const v8*)(base + b_off + j * 2048 + c0);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[0][i] = *(const v8*)(base + a_off + i * 2048 + c0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = mfma16<T>(bf[ks][0], af[ks][0], acc[0][0]);
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[1][j] = *(const v8*)(base + b_off + j * 2048 + c1);
#pragma unroll
          for (int i = 0; i < 4; ++i) af[1][i] = *(const v8*)(base + a_off + i * 2048 + c1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (i == 0 && j == 0) continue;
            acc[i][j] = mfma16<T>(bf[ks][j], af[ks][i], acc[i][j]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (MIXED)
    for (int f = nk16; f < nk; ++f) {
      __builtin_amdgcn_s_barrier();
      const char* base = smem + slot * STAGE;
      last_slot = slot;
      slot = slot + 1 == NS ? 0 : slot + 1;
      i32x8 b8[4], a8[2];
      auto ld8 = [&](int off) -> i32x8 {
        const i32x4 x = *(const i32x4*)(base + off + e0), y = *(const i32x4*)(base + off + e1);
        return __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
      };
#pragma unroll
      for (int j = 0; j < 4; ++j) b8[j] = ld8(b_off + j * 2048);
      a8[0] = ld8(a_off);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        mfma_lo8(b8[0], a8[i & 1], acc[i][0], sc_w, sc_a);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < 4) a8[(i + 1) & 1] = ld8(a_off + (i + 1) * 2048);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 1; j < 4; ++j) mfma_lo8(b8[j], a8[i & 1], acc[i][j], sc_w, sc_a);
        __builtin_amdgcn_sched_barrier(0);
      }
      mfma_lo8_fence();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                      // every multiplying wave is done with the last stage: its slot is scratch
    if constexpr (CAN_FOLD) {
      if (tid < BM_) fold_build_coef(g, xlds, xtab, tid);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    epilogue_store<T, EPI>(g, acc, tm * BM_ + wm * 64, tn * BN + wn * 64, lane,
                           LinearRows<144>{smem + last_slot * STAGE + wave * EPI_SCRATCH_PER_WAVE},
                           LinearRows<272>{smem + last_slot * STAGE + wave * EPI_SCRATCH_PER_WAVE},
                           FoldCtx{xlds, xtab, wm * 64, wn, 2, MIXED ? 2 : (g.ln_split ? 1 : 0)});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (epi_ln_producer(EPI)) {
      if (tid < BM_) {
        const int row = tm * BM_ + tid;
        if (row < M) {
          const float2* pp = (const float2*)(xlds + (size_t)tid * 2 * 8);
          *(float2*)(g.ln_part + ((size_t)row * g.ln_ntp + tn) * 2) = float2{pp[0].x + pp[1].x, pp[0].y + pp[1].y};
        }
      }
    }
  }
}

template <typename T, int EPI, bool MIXED>
static hipError_t launch_pcp_m(const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  constexpr int LDS = 3 * (256 + 128) * BK * 2 + XLDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_pcp_kernel<T, EPI, MIXED>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int cus = stream_cus(s);
  const int tiles = ((g.M + 255) / 256) * (g.N / 128);
  hipExtLaunchKernelGGL((gemm_pcp_kernel<T, EPI, MIXED>), dim3(tiles < cus ? tiles : cus), dim3(768), LDS, s, ea, eb, 0, g);
  return hipGetLastError();
}
template <typename T, int EPI>
static hipError_t launch_pcp(const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  constexpr int BE = epi_base(EPI);
  if constexpr (BE == EPI_RESID32 || BE == EPI_RESID32_LN || BE == EPI_STORE32 || BE == EPI_GELU_SPLIT || BE == EPI_GELUBWD_SPLIT || BE == EPI_STORE_SPLIT) {
    if (g.a_split == 2) return launch_pcp_m<T, EPI, true>(g, s, ea, eb);
  } else if (g.a_split == 2) return hipErrorInvalidValue;
  return launch_pcp_m<T, EPI, false>(g, s, ea, eb);
}

template <typename T, int EPI, int BM_, int BN_, int NW, int NS, bool MIXED>
static hipError_t launch_geo_m(const GemmArgs& g, int wg_per_cu, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  constexpr int LDS = NS * (BM_ + BN_) * BK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_bt_kernel<T, EPI, BM_, BN_, NW, NS, MIXED>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS + (LDS + XLDS_BYTES_WIDE <= 160 * 1024 ? XLDS_BYTES_WIDE : XLDS_BYTES));
    attr_set = true;
  }
  // LayerNorm folding: 16 KiB behind the ring (the consumer's row partials / the producer's per-tile column-block sums)
  const int lds = LDS + (epi_folds(EPI) ? xlds_bytes(g.fold_ntp) : (epi_ln_producer(EPI) ? XLDS_BYTES : 0));
  if (lds > 160 * 1024) return hipErrorInvalidValue;      // (launch_one keeps 8-slot consumers off the 3-deep 256x128 ring)

  int cus = stream_cus(s);
#ifdef MVLPT_DEBUG_CUS
  if (getenv("MVLPT_DBG_CUS")) cus = atoi(getenv("MVLPT_DBG_CUS"));      // CU-scaling measurement, debug builds only
#endif
  const int tiles = ((g.M + BM_ - 1) / BM_) * ((g.N + BN_ - 1) / BN_);
  int resident = cus * wg_per_cu;
  // experiment: at most MVLPT_GEMM_MAXTILES tiles per workgroup (a CU is handed back to the dispatcher that often); 0 = persistent
  static const int maxtiles = getenv("MVLPT_GEMM_MAXTILES") ? atoi(getenv("MVLPT_GEMM_MAXTILES")) : 0;
  if (maxtiles > 0 && (tiles + maxtiles - 1) / maxtiles > resident) resident = (tiles + maxtiles - 1) / maxtiles;
  hipExtLaunchKernelGGL((gemm_bt_kernel<T, EPI, BM_, BN_, NW, NS, MIXED>), dim3(tiles < resident ? tiles : resident), dim3(NW * 64), lds, s,
                        ea, eb, 0, g);
  return hipGetLastError();
}
template <typename T, int EPI, int BM_, int BN_, int NW, int NS>
static hipError_t launch_geo(const GemmArgs& g, int wg_per_cu, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  // the mixed pair only exists with the epilogues a split tower uses (fp32 outputs and the pair-producing ones)
  constexpr int BE = epi_base(EPI);
  if constexpr (BE == EPI_RESID32 || BE == EPI_RESID32_LN || BE == EPI_STORE32 || BE == EPI_GELU_SPLIT || BE == EPI_GELUBWD_SPLIT || BE == EPI_STORE_SPLIT) {
    if (g.a_split == 2) return launch_geo_m<T, EPI, BM_, BN_, NW, NS, true>(g, wg_per_cu, s, ea, eb);
  } else if (g.a_split == 2) return hipErrorInvalidValue;
  return launch_geo_m<T, EPI, BM_, BN_, NW, NS, false>(g, wg_per_cu, s, ea, eb);
}

// rows [m_lo, m_lo + rows) of the problem as a GEMM of its own (all operands are row-major with leading dimension
// K or N, so a row range is a contiguous sub-problem)
template <int EPI>
static GemmArgs row_slice(const GemmArgs& g, int m_lo, int rows) {
  GemmArgs r = g;
  const size_t ok = (size_t)m_lo * (g.lda ? (size_t)g.lda : (size_t)g.K * (g.a_split ? 2 : 1)), on = (size_t)m_lo * g.N;
  constexpr int BE = epi_base(EPI);
  constexpr size_t OB = (BE == EPI_RESID32 || BE == EPI_RESID32_LN || BE == EPI_STORE32 || BE == EPI_GELU_SPLIT || BE == EPI_GELUBWD_SPLIT || BE == EPI_STORE_SPLIT) ? 4 : 2;
  r.A = (const char*)g.A + ok * 2;
  r.M = rows;
  r.out = (char*)g.out + on * OB;
  if (g.out2) r.out2 = (char*)g.out2 + on * 2;
  if (g.aux) r.aux = (const char*)g.aux + on * 2;
  if (g.resid) r.resid = g.resid + on;
  if (g.rp_hi_in) { r.rp_hi_in = (const char*)g.rp_hi_in + on * 2; r.rp_lo_in = g.rp_lo_in + on; r.rp_lo_out = g.rp_lo_out + on; }
  return r;
}

template <typename T, int EPI>
static hipError_t launch_one(const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb, int* tile_m, int* tile_n) {
  // (256x256 with FOUR waves of 256x64 — 17 % less LDS-read traffic per FLOP, one wave per SIMD — was measured 17-36 % slower
  // than the eight-wave kernel: 8192^3 1.09 vs 1.31 PF; nothing hides a wave's own LDS-DMA issue and ds_read latency.)
  // geometry by tile count: 256x256 (128x64 wave tiles, least LDS-DMA / LDS-read traffic per FLOP) needs >= 4 full
  // rounds of 256 resident workgroups to amortise its tail; 256x128 needs >= 1.5 rounds; otherwise 128x128.
  const long t128 = (long)((g.M + 255) / 256) * ((g.N + 127) / 128);
  const long t256 = (long)((g.M + 255) / 256) * (g.N / 256);
  static const int geo = getenv("MVLPT_GEMM_GEO") ? atoi(getenv("MVLPT_GEMM_GEO")) : 2;   // experiment switch (0,1,2)
  // phased 256x128 variant: measured +3..5 % on long-K GEMMs (MLP down-projection, K = 4d), -4..6 % on K = d
  static const int phased = getenv("MVLPT_GEMM_PHASED") ? atoi(getenv("MVLPT_GEMM_PHASED")) : 2;   // 0 off, 1 all, 2 long K
  // 256x256 needs >= 4 rounds of tiles, or >= 2 rounds when K is long (a ragged last round then costs less than the
  // smaller geometry's extra LDS traffic: N = 768, K = 3072: 315 -> 297 us with 2.3 rounds)
  const int Keff = g.a_split == 2 ? g.K + g.K / 2 : (g.a_split ? 2 * g.K : g.K);
  // the round counts below are per compute unit the stream can use (a CU-partitioned stream: mvlpt_stream_create_cus)
  const long cus = stream_cus(s);
  // ... or when the 256x256 tiles fill exactly one round (0.75 .. 1 tile per CU: the text tower's N = 2048 GEMMs at M = 7 700 are
  // 248 tiles): one tile time instead of two rounds of the 256x128 geometry
  static const int one_round_on = getenv("MVLPT_GEMM_ONE_ROUND") ? atoi(getenv("MVLPT_GEMM_ONE_ROUND")) : 1;
  static const int one_round_pct = getenv("MVLPT_GEMM_ONE_ROUND_PCT") ? atoi(getenv("MVLPT_GEMM_ONE_ROUND_PCT")) : 75;   // minimum fill of the round
  const bool one_round = one_round_on && g.N % 256 == 0 && t256 <= cus && 100 * t256 >= one_round_pct * cus;
  const bool big = g.N % 256 == 0 && (t256 >= 4 * cus || (t256 >= 2 * cus && Keff >= 2048) || one_round);
  const bool r15 = 2 * t128 >= 3 * cus;      // >= 1.5 rounds of 256x128 tiles
  // (the phased kernel has no fp8 stages: mixed pairs take the plain 256x128 geometry)
  // (nor the LayerNorm-folding fields: folded GEMMs take the plain geometries)
  constexpr bool folded = epi_folds(EPI) || epi_ln_producer(EPI);
  if (g.a_split != 2 && !folded && r15 && (phased == 1 || (phased == 2 && Keff >= 2048 && !big))) {
    *tile_m = 256; *tile_n = 128;
    if constexpr (!folded) return ea == (hipEvent_t)-1 ? hipSuccess : launch_phased<T, EPI>(g, s, ea, eb);
  }
  if (geo >= 2 && big) {
    *tile_m = 256; *tile_n = 256;
#ifdef MVLPT_BREG
    static const int breg = getenv("MVLPT_GEMM_BREG") ? atoi(getenv("MVLPT_GEMM_BREG")) : 0;
    if (breg && g.a_split == 0 && g.M % 256 == 0 && g.K % 128 == 0 && g.K / BK >= MVLPT_BREG_NS) return ea == (hipEvent_t)-1 ? hipSuccess : launch_breg<T, EPI>(g, s, ea, eb);
#endif
    return ea == (hipEvent_t)-1 ? hipSuccess : launch_geo<T, EPI, 256, 256, 8, 2>(g, 1, s, ea, eb);
  }
  // (a folded consumer with 8-slot rows needs 20 KiB behind its ring: the 3-deep 256x128 ring has 16 left -> 256x256 or 128x128)
  const bool wide_fold = epi_folds(EPI) && g.fold_ntp > 6;
  if (geo >= 2 && wide_fold && r15 && g.N % 256 == 0) {
    *tile_m = 256; *tile_n = 256;
    return ea == (hipEvent_t)-1 ? hipSuccess : launch_geo<T, EPI, 256, 256, 8, 2>(g, 1, s, ea, eb);
  }
  // 256x128 with data-movement waves (gemm_pcp_kernel): 1 on (default), 0 the self-serving 8-wave kernel
  static const int pcp = getenv("MVLPT_GEMM_PCP") ? atoi(getenv("MVLPT_GEMM_PCP")) : 1;
  if (geo >= 1 && r15 && !wide_fold && pcp && (!epi_folds(EPI) || (g.fold_ntp == 4 || g.fold_ntp == 6))) {
    *tile_m = 256; *tile_n = 128;
    return ea == (hipEvent_t)-1 ? hipSuccess : launch_pcp<T, EPI>(g, s, ea, eb);
  }
  if (geo >= 1 && r15 && !wide_fold) {
    *tile_m = 256; *tile_n = 128;
    return ea == (hipEvent_t)-1 ? hipSuccess : launch_geo<T, EPI, 256, 128, 8, 3>(g, 1, s, ea, eb);
  }
  // small problems (text tower: M = C*L ~ 7.7k rows) put at most one workgroup on a CU, so nothing hides the
  // LDS-DMA latency of a 2-deep ring: use a 4-deep ring (128 KiB, three K-stages in flight) instead
  static const int deep = getenv("MVLPT_GEMM_DEEP") ? atoi(getenv("MVLPT_GEMM_DEEP")) : 1;   // split operands only.  round 1 (single operands): text tower alone -7 %, overlapped step +1.4 %
  // (its 128 KiB workgroups cannot share a CU with the image-tower kernels they overlap with) -> off.  Round 2 (split operands
  // double every K of the text tower): tower alone 5.57 -> 5.10 ms, overlapped step 15.22 -> 15.00 ms -> on.
  const long t_small = (long)((g.M + 127) / 128) * (g.N / 128);
  // dedicated data-movement waves (gemm_pc_kernel above): split operands 1 (default), every operand kind 2, off 0
  static const int pcw = getenv("MVLPT_GEMM_PC") ? atoi(getenv("MVLPT_GEMM_PC")) : 1;
  if (pcw && (g.a_split || pcw >= 2) && pc_takes<EPI>(g, cus)) {
    *tile_m = 128; *tile_n = 128;
    return ea == (hipEvent_t)-1 ? hipSuccess : launch_pc<T, EPI>(g, s, ea, eb);
  }
  if (deep && g.a_split && t_small <= cus) {
    *tile_m = 128; *tile_n = 128;
    return ea == (hipEvent_t)-1 ? hipSuccess : launch_geo<T, EPI, 128, 128, 4, 4>(g, 1, s, ea, eb);
  }
  *tile_m = 128; *tile_n = 128;
  return ea == (hipEvent_t)-1 ? hipSuccess : launch_geo<T, EPI, 128, 128, 4, 2>(g, 2, s, ea, eb);
}

template <typename T, int EPI>
static hipError_t launch_t(const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  // Tail splitting (experiment, off by default).  The persistent grid runs rounds of `cus` big tiles; a ragged last round (e.g. 9.23 rounds for the
  // MLP up-projection) leaves most CUs idle for a full tile time.  When the last round is less than ~70 % full, the
  // rows of whole rounds go to the big geometry and the remaining rows are a second, small-tile launch (128x128,
  // two workgroups per CU), which finishes in about half a big-tile time.
  int bm = 0, bn = 0;
  (void)launch_one<T, EPI>(g, s, (hipEvent_t)-1, nullptr, &bm, &bn);      // query the geometry only
  static const int split = getenv("MVLPT_GEMM_TAILSPLIT") ? atoi(getenv("MVLPT_GEMM_TAILSPLIT")) : 0;   // measured: +2 % / -8 % by shape -> off
  if (split && bm == 256) {
    const int cus = stream_cus(s);
    const long tn = g.N / bn, tm = (g.M + bm - 1) / bm, tiles = tm * tn;
    const long full = tiles / cus;
    const double frac = (double)(tiles - full * cus) / cus;
    if (full >= 2 && frac > 0.02 && frac < 0.7) {
      const long tm_main = full * cus / tn;                     // M tiles that fit the whole rounds
      const int m_main = (int)(tm_main * bm);
      if (m_main > 0 && m_main < g.M) {
        const GemmArgs a = row_slice<EPI>(g, 0, m_main), b = row_slice<EPI>(g, m_main, g.M - m_main);
        int x, y;
        hipError_t e = launch_one<T, EPI>(a, s, ea, nullptr, &x, &y);
        if (e != hipSuccess) return e;
        return launch_geo<T, EPI, 128, 128, 4, 2>(b, 2, s, nullptr, eb);
      }
    }
  }
  int x, y;
  return launch_one<T, EPI>(g, s, ea, eb, &x, &y);
}

template <typename T>
static hipError_t launch_epi(const GemmArgs& g, int epi, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  if (g.fold_part) {
    switch (epi) {
      case EPI_STORE16: return launch_t<T, EPI_STORE16_FOLD>(g, s, ea, eb);
      case EPI_GELU: return launch_t<T, EPI_GELU_FOLD>(g, s, ea, eb);
      case EPI_STORE_SPLIT: return launch_t<T, EPI_STORE_SPLIT_FOLD>(g, s, ea, eb);
      case EPI_GELU_SPLIT: return launch_t<T, EPI_GELU_SPLIT_FOLD>(g, s, ea, eb);
    }
    return hipErrorInvalidValue;
  }
  switch (epi) {
    case EPI_STORE16: return launch_t<T, EPI_STORE16>(g, s, ea, eb);
    case EPI_GELU: return launch_t<T, EPI_GELU>(g, s, ea, eb);
    case EPI_RESID32: return launch_t<T, EPI_RESID32>(g, s, ea, eb);
    case EPI_GELUBWD: return launch_t<T, EPI_GELUBWD>(g, s, ea, eb);
    case EPI_STORE32: return launch_t<T, EPI_STORE32>(g, s, ea, eb);
    case EPI_GELU_SPLIT: return launch_t<T, EPI_GELU_SPLIT>(g, s, ea, eb);
    case EPI_GELUBWD_SPLIT: return launch_t<T, EPI_GELUBWD_SPLIT>(g, s, ea, eb);
    case EPI_STORE_SPLIT: return launch_t<T, EPI_STORE_SPLIT>(g, s, ea, eb);
    case EPI_RESID32_LN: return launch_t<T, EPI_RESID32_LN>(g, s, ea, eb);
    case EPI_RESIDP_LN:      // the packed stream is an fp16 format
      if constexpr (__is_same(T, f16)) return launch_t<T, EPI_RESIDP_LN>(g, s, ea, eb);
      else return hipErrorInvalidValue;
  }
  return hipErrorInvalidValue;
}

template <typename T>
static int tile_n_epi(const GemmArgs& g, int epi, hipStream_t s) {
  int bm = 0, bn = 0;
  switch (epi) {      // (the geometry does not depend on the epilogue except for the phased routing, which folded GEMMs skip)
    case EPI_RESID32_LN: case EPI_RESIDP_LN: (void)launch_one<T, EPI_RESID32_LN>(g, s, (hipEvent_t)-1, nullptr, &bm, &bn); break;
    default: (void)launch_one<T, EPI_RESID32>(g, s, (hipEvent_t)-1, nullptr, &bm, &bn); break;
  }
  return bn;
}
int gemm_tile_n(int dtype, int epi, const GemmArgs& g, hipStream_t s) {
  return dtype == DT_BF16 ? tile_n_epi<bf16>(g, epi, s) : tile_n_epi<f16>(g, epi, s);
}

// K must be a multiple of 64 and N of 128 (every CLIP width is; conv K is zero-padded); M is arbitrary.
hipError_t launch_gemm(int dtype, int epi, const GemmArgs& g, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K % BK) != 0 || (g.N % 128) != 0) return hipErrorInvalidValue;
  if (g.a_split == 2 && ((g.K % 128) != 0 || g.ldb < g.K + g.K / 2)) return hipErrorInvalidValue;
  if (g.ldb && (g.ldb < g.K || (g.ldb % 8) != 0)) return hipErrorInvalidValue;
  if (g.lda && (g.lda < (g.a_split ? 2 : 1) * g.K || (g.lda % 8) != 0)) return hipErrorInvalidValue;
  if (g.ldo && (g.ldo < 2 * g.N || (g.ldo % 8) != 0 || !(epi == EPI_GELU_SPLIT || epi == EPI_GELUBWD_SPLIT || epi == EPI_STORE_SPLIT))) return hipErrorInvalidValue;
  if ((epi == EPI_RESID32 || epi == EPI_RESID32_LN) && !g.resid) return hipErrorInvalidValue;
  if (epi == EPI_RESID32_LN && (!g.ln_gamma || !g.ln_x16 || !g.ln_part || g.ln_ntp <= 0 || (g.ln_ntp & 1))) return hipErrorInvalidValue;
  if (epi == EPI_RESIDP_LN && (dtype != DT_F16 || g.a_split || !g.rp_hi_in || !g.rp_lo_in || !g.rp_lo_out || !g.ln_part || g.ln_ntp <= 0 || (g.ln_ntp & 1)))
    return hipErrorInvalidValue;
  if (g.fold_part) {
    const bool can = epi == EPI_STORE16 || epi == EPI_GELU || epi == EPI_STORE_SPLIT || epi == EPI_GELU_SPLIT;
    const int nk = g.a_split == 2 ? g.K / BK + g.K / 128 : (g.a_split ? 2 : 1) * (g.K / BK);
    // the row partials ride with the LAST K-stage of a tile: it must be issued inside the tile's own K loop (ring depth <= 4)
    if (!can || !g.fold_colsum || !g.bias || g.fold_nt <= 0 || g.fold_nt > g.fold_ntp || g.fold_ntp > FOLD_MAX_NTP || (g.fold_ntp & 1) || nk < 4)
      return hipErrorInvalidValue;
  }
  if ((epi == EPI_GELUBWD || epi == EPI_GELUBWD_SPLIT) && !g.aux) return hipErrorInvalidValue;
  static const int duo = getenv("MVLPT_GEMM_DUO") ? atoi(getenv("MVLPT_GEMM_DUO")) : 0;
  if (duo && gemm_duo_takes(epi, g, s)) return launch_gemm_duo(dtype, epi, g, s, ea, eb);
  if (dtype == DT_F16) return launch_epi<f16>(g, epi, s, ea, eb);
  if (dtype == DT_BF16) return launch_epi<bf16>(g, epi, s, ea, eb);
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------- fp32 GEMM
// One wave per 16x16 output tile, straight from global memory (the operands are a few hundred KB and
// L2-resident): each lane loads 4 consecutive k of its row (16 B), 4 x v_mfma_f32_16x16x4_f32 per step.
// A/B k-slot of lane l in MFMA j is  k0 + 4*(l>>4) + j  on both operands (any consistent map is exact).
__global__ __launch_bounds__(256) void sgemm_bt_kernel(const float* __restrict__ A, const float* __restrict__ Bt,
                                                       float* __restrict__ Cm, int M, int N, int K, const float* alpha_dev) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tilesN = (N + 15) / 16;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= ((M + 15) / 16) * tilesN) return;
  const int m0 = (tile / tilesN) * 16, n0 = (tile % tilesN) * 16;
  const int fr = lane & 15, fg = lane >> 4;
  int ar = m0 + fr; ar = ar < M ? ar : M - 1;
  int br = n0 + fr; br = br < N ? br : N - 1;
  const float* ap = A + (size_t)ar * K + fg * 4;
  const float* bp = Bt + (size_t)br * K + fg * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 16) {
    const f32x4 a4 = *(const f32x4*)(ap + k0);
    const f32x4 b4 = *(const f32x4*)(bp + k0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b4[j], a4[j], acc, 0, 0, 0);
  }
  // swapped operands: lane holds C[m0 + fr][n0 + 4*fg + 0..3]
  const int m = m0 + fr, n = n0 + fg * 4;
  if (m < M && n < N) {
    const float alpha = alpha_dev ? alpha_dev[0] : 1.0f;
    *(f32x4*)(Cm + (size_t)m * N + n) = acc * alpha;
  }
}
hipError_t launch_sgemm_bt(const float* A, const float* Bt, float* C, int M, int N, int K, const float* alpha_dev, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % 16) || (N % 4)) return hipErrorInvalidValue;
  const int tiles = ((M + 15) / 16) * ((N + 15) / 16);
  hipLaunchKernelGGL(sgemm_bt_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, A, Bt, C, M, N, K, alpha_dev);
  return hipGetLastError();
}

}  // namespace mvlpt
