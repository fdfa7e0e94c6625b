// Multi-head attention core, head_dim 64, forward and backward (dQ,dK,dV), for the two CLIP towers:
//   vision  : non-causal, L = 1 + n_vpt + grid^2 (50 / 197..205), nn.MultiheadAttention in
//             clip/model.py:181-183 called from trainers/mvlpt.py:72,83,85
//   text    : additive causal mask (clip/model.py:324-330), L <= 77 (CUT_CONTEXTLEN shortens it)
// scale = 1/sqrt(64); softmax statistics and all accumulation in fp32; P and dS feed the MFMA as 16-bit.
//
// gfx950 design: sequences are short, so a whole head's K and V fit in LDS (<= 64 KiB): one workgroup
// (4 waves) per (sequence, head), each wave owns 16-query tiles.  Scores are computed TRANSPOSED,
// S^T = K Q^T, so every lane owns ONE query column (q = lane & 15) and four keys per 16-key tile: the
// softmax row-reduction is in-register plus two wavefront shuffles (xor 16, 32), and the P^T accumulator
// registers ARE the B operand of the next MFMA (O^T = V^T P^T) with no cross-lane movement: the k-slot
// (lane>>4)*8 + j of a 32-key block is defined as key 4*(lane>>4) + j of its first 16-key tile for j<4
// and of its second tile for j>=4.  K and V rows are stored 128 B wide with the 16-B chunk index XOR (row & 7)
// (conflict-free ds_read_b128); the forward reads V^T fragments straight out of the row-major V image with the
// hardware transpose read ds_read_b64_tr_b16 (and so do the backward kernels for K^T, Q^T, dO^T): no transposed
// copies are ever staged.  All LDS images are filled by LDS-DMA; per-wave Q/K/V/dO fragments that are used as B
// operands are fetched from global memory once, before the DMA wait.
#include <hip/hip_ext.h>
#include "attn_common.h"

namespace mvlpt {

constexpr int ATT_MAX_NKT = 38;  // 38 tiles * 16 keys = 608: ViT-L/14@336 (577 + prompts); K+V images fill 152 of 160 KiB LDS
int attn_max_len() { return ATT_MAX_NKT * 16; }

// ======================================================================================= forward
template <typename T, int NKT, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using v8 = typename Vec<T>::v8;
  using v4 = typename Vec<T>::v4;
  constexpr int LP = NKT * 16;
  constexpr int BLK = NKT < 16 ? NKT : 16;      // key tiles per softmax block (online softmax across blocks, L > 256)
  constexpr bool PRELOAD = NKT <= 16;
  char* sK = smem;                            // [LP][64] row-major, 16-B chunks swizzled by (row & 7)
  char* sV = smem + LP * 128;                 // same image for V; transposed on the fly by ds_read_b64_tr_b16
  const int L = a.L, H = a.H, d = H * 64;
  const int n = blockIdx.x / H, h = blockIdx.x % H;
  const size_t ld = (size_t)3 * d;
  const T* base = (const T*)a.qkv + (size_t)n * L * ld + h * 64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int nqt = a.q_rows > 0 ? ((a.q_rows < L ? a.q_rows : L) + 15) >> 4 : (L + 15) >> 4;
  // short sequences: every Q fragment this wave will need is requested before the K/V DMA is waited for
  constexpr int MAXQ = PRELOAD ? (NKT + 3) / 4 : 1;
  v8 qf[MAXQ][2];
  if constexpr (PRELOAD) {
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
      int qr = (wave + 4 * i) * 16 + fr;
      qr = qr < L ? qr : L - 1;
      const T* qp = base + (size_t)qr * ld + fg * 8;
      qf[i][0] = *(const v8*)qp;
      qf[i][1] = *(const v8*)(qp + 32);
    }
  }
  stage_rows_dma<T>(sK, base + d, ld, L, LP, wave, lane, a.flags & 1);
  stage_rows_dma<T>(sV, base + 2 * d, ld, L, LP, wave, lane, a.flags & 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  constexpr float SC = 0.125f * 1.4426950408889634f;   // 1/sqrt(64) * log2(e): softmax in base 2
  auto process = [&](const int qt, const v8 q0, const v8 q1) {
    const int qrow = qt * 16 + fr;
    const int nkt = CAUSAL ? (qt + 1 < NKT ? qt + 1 : NKT) : NKT;   // key tiles that can be unmasked
    float mrun = -INFINITY, sum = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < NKT; k0 += BLK) {
      if (k0 >= nkt) break;
      f32x4 s[BLK];
      float mx = -INFINITY;
#pragma unroll
      for (int u = 0; u < BLK; ++u) {
        const int kt = k0 + u;
        s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kt < NKT && kt < nkt) {
          s[u] = mfma16<T>(frag_rows<T>(sK, kt, 0, fr, fg), q0, s[u]);
          s[u] = mfma16<T>(frag_rows<T>(sK, kt, 1, fr, fg), q1, s[u]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + fg * 4 + r;
            const bool ok = key < L && (!CAUSAL || key <= qrow);
            s[u][r] = ok ? s[u][r] * SC : -INFINITY;
            mx = fmaxf(mx, s[u][r]);
          }
        }
      }
      mx = quad_max(mx);
      const float mnew = fmaxf(mrun, mx);           // finite from the first block on: key 0 is never masked
      const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);   // first block: exp2(-inf) = 0 on zero accumulators
      mrun = mnew;
      sum *= alpha;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
      for (int u = 0; u < BLK; ++u) {
        if (k0 + u < NKT && k0 + u < nkt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s[u][r] = __builtin_amdgcn_exp2f(s[u][r] - mnew); sum += s[u][r]; }
        }
      }
#pragma unroll
      for (int u = 0; u < BLK; u += 2) {
        if (k0 + u < NKT && k0 + u < nkt) {
          if (u + 1 < BLK) {
            const v8 pf = pack8<T>(s[u], s[u + 1 < BLK ? u + 1 : u]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16<T>(frag_vt<T>(sV, (k0 + u) >> 1, dt, fr, fg), pf, o[dt]);
          } else {                                     // odd tile count: the last 16 keys form half a block
            const v8 pf = pack8<T>(s[u], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16<T>(frag_vt_half<T>(sV, (k0 + u) >> 1, dt, fr, fg), pf, o[dt]);
          }
        }
      }
    }
    sum = quad_sum(sum);
    const float inv = 1.0f / sum;
    if (qrow < L) {
      T* op = (T*)a.out + ((size_t)n * L + qrow) * d + h * 64 + fg * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        v4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(o[dt][e] * inv);
        if (a.flags & 2) __builtin_nontemporal_store(w, (v4*)(op + dt * 16));
        else *(v4*)(op + dt * 16) = w;
      }
      // natural-log LSE of the scaled scores (the backward recomputes P = exp(s/8 - lse))
      if (a.lse && fg == 0) a.lse[((size_t)n * H + h) * L + qrow] = (mrun + log2f(sum)) * 0.6931471805599453f;
    }
  };
  if constexpr (PRELOAD) {
#pragma unroll
    for (int qi = 0; qi < MAXQ; ++qi) {
      if (wave + 4 * qi >= nqt) break;
      process(wave + 4 * qi, qf[qi][0], qf[qi][1]);
    }
  } else {
    for (int qt = wave; qt < nqt; qt += 4) {
      const int qrow = qt * 16 + fr;
      const T* qp = base + (size_t)(qrow < L ? qrow : L - 1) * ld + fg * 8;
      process(qt, *(const v8*)qp, *(const v8*)(qp + 32));
    }
  }
}

// ======================================================================================= forward, persistent over heads
// The headline's shape (ViT-B/16: L = 197..208, 13 key tiles, not causal, full sequences, N * H >> CUs).  attn_fwd_kernel above runs
// one 4-wave workgroup per (image, head): staging (52 KiB), arithmetic and stores of a head are serial inside the workgroup and only
// overlap across the three workgroups of a CU, and 13 query tiles on 4 waves leave a quarter of three waves' time unused: 81.5 us for
// 310 MB = 3.8 TB/s.  Here ONE workgroup per CU walks the (image, head) list — the scheme that bought +14 % for the pair kernel
// (attention32.hip attn32p_fwd_kernel): seven waves with two query tiles each (14 slots for 13 tiles), three 26-KiB buffers in
// rotation so that K of head i+1 lands during all of head i and V of head i+1 behind its S phase (the buffer K_i just left), Q of head
// i+1 fetched into the registers Q_i leaves.  Both tiles of a wave share every K / V^T fragment (half the LDS reads per score).
// Waits are counted by hand (LDS-DMA from inline asm: dma_raw): at the top of a head everything but the previous head's output stores.
constexpr int PNT = 13, PNW = 7, PROWS = PNT * 16, PIMG = PROWS * 128, PLDS = 3 * PIMG;
template <typename T>
__device__ __forceinline__ void stage_p(char* buf, const T* src, size_t ld, int L, int wave, int lane) {
  const int srow = lane >> 3, chunk = (lane & 7) ^ srow;
#pragma unroll
  for (int i = 0; i < 4; ++i) {      // 26 slabs of 8 rows on 7 waves: four requests each (the last two waves repeat slab 25: same bytes)
    int sl = wave + PNW * i;
    sl = sl < PROWS / 8 ? sl : PROWS / 8 - 1;
    int row = sl * 8 + srow;
    row = row < L ? row : L - 1;
    dma_raw<16>(src + (size_t)row * ld + chunk * 8, buf + sl * 1024);
  }
}
template <typename T>
__global__ __launch_bounds__(PNW * 64) void attn_fwdp_kernel(AttnArgs a, int total) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  using v8 = typename Vec<T>::v8;
  using v4 = typename Vec<T>::v4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
  const int L = a.L, H = a.H, d = H * 64, G = gridDim.x;
  const size_t ld = 3 * (size_t)d;
  const bool two = wave + PNW < PNT;         // the wave's second tile exists (waves 0-5)
  auto head_base = [&](int item) -> const T* {
    const int n = item / H, h = item - n * H;
    return (const T*)a.qkv + (size_t)n * L * ld + h * 64;
  };
  auto load_q = [&](const T* base, v8 (&Q)[2][2]) {
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int q = (wave + o * PNW) * 16 + fr;
      const T* qp = base + (size_t)(q < L ? q : L - 1) * ld + fg * 8;
      Q[o][0] = *(const v8*)qp;
      Q[o][1] = *(const v8*)(qp + 32);
    }
  };
  int item = blockIdx.x;
  if (item >= total) return;
  int kbuf = 0;
  v8 Q[2][2];
  {
    const T* base = head_base(item);
    stage_p<T>(sm, base + d, ld, L, wave, lane);               // K_0
    stage_p<T>(sm + PIMG, base + 2 * d, ld, L, wave, lane);    // V_0
    load_q(base, Q);
  }
  constexpr float SC2 = 0.125f * 1.4426950408889634f;           // 1/sqrt(64) * log2(e): softmax in base 2
  const bool want_lse = a.lse != nullptr;
  for (bool first = true; item < total; item += G, first = false) {
    const int n = item / H, h = item - n * H;
    char* const Kb = sm + kbuf * PIMG;
    char* const Vb = sm + (kbuf + 1 >= 3 ? kbuf - 2 : kbuf + 1) * PIMG;
    char* const Nb = sm + (kbuf + 2 >= 3 ? kbuf - 1 : kbuf + 2) * PIMG;
    const int next = item + G;
    // everything this wave requested has landed, except (vmcnt retires in order) its output stores of the previous head
    if (first || want_lse) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (two) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // everybody's pieces of K_i, V_i are in; everybody is done with V_(i-1)
    // (the compiler's own vmcnt waits for the Q fragments must sit HERE, in front of the DMA it does not see: attention32.hip)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) asm volatile("" ::"v"(Q[o][ks]));
    if (next < total) stage_p<T>(Nb, head_base(next) + d, ld, L, wave, lane);          // K_(i+1)
    // ---- S^T = K Q^T of both own tiles against every key tile; the K fragments of tile kt + 1 travel under tile kt's MFMAs
    f32x4 S[2][PNT];
    v8 kf[2], kn[2];
    auto load_k = [&](int kt, v8 (&f)[2]) { f[0] = frag_rows<T>(Kb, kt, 0, fr, fg); f[1] = frag_rows<T>(Kb, kt, 1, fr, fg); };
    load_k(0, kf);
#pragma unroll
    for (int kt = 0; kt < PNT; ++kt) {
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < PNT) load_k(kt + 1, kn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        S[o][kt] = mfma16<T>(kf[0], Q[o][0], f32x4{0.f, 0.f, 0.f, 0.f});
        S[o][kt] = mfma16<T>(kf[1], Q[o][1], S[o][kt]);
      }
      __builtin_amdgcn_sched_barrier(0);
      kf[0] = kn[0]; kf[1] = kn[1];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // K_i is dead
    if (next < total) {
      const T* nb = head_base(next);
      stage_p<T>(Kb, nb + 2 * d, ld, L, wave, lane);                                    // V_(i+1)
      load_q(nb, Q);                                                                    // Q_(i+1): Q_i is dead too
    }
    // ---- softmax of both tiles (only the last key tile can hold keys past the sequence)
    float mx[2], sum[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
#pragma unroll
      for (int r = 0; r < 4; ++r) S[o][PNT - 1][r] = (PNT - 1) * 16 + 4 * fg + r < L ? S[o][PNT - 1][r] : -INFINITY;
      mx[o] = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < PNT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[o] = fmaxf(mx[o], S[o][kt][r]);
      mx[o] = quad_max(mx[o]);
      const float msc = mx[o] * SC2;
      sum[o] = 0.f;
#pragma unroll
      for (int kt = 0; kt < PNT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { S[o][kt][r] = __builtin_amdgcn_exp2f(fmaf(S[o][kt][r], SC2, -msc)); sum[o] += S[o][kt][r]; }
      sum[o] = quad_sum(sum[o]);
    }
    // ---- O^T = V^T P^T: every V^T fragment feeds both tiles; the fragment of the next (block, column tile) travels under the MFMAs
    f32x4 O[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int i = 0; i < 4; ++i) O[o][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NKB = (PNT + 1) / 2;
    v8 vf, vn;
    auto load_v = [&](int kb, int dt) -> v8 { return 2 * kb + 1 >= PNT ? frag_vt_half<T>(Vb, kb, dt, fr, fg) : frag_vt<T>(Vb, kb, dt, fr, fg); };
    vf = load_v(0, 0);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      v8 pf[2];
#pragma unroll
      for (int o = 0; o < 2; ++o) pf[o] = 2 * kb + 1 < PNT ? pack8<T>(S[o][2 * kb], S[o][2 * kb + 1 < PNT ? 2 * kb + 1 : 0]) : pack8<T>(S[o][2 * kb], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        __builtin_amdgcn_sched_barrier(0);
        if (dt + 1 < 4) vn = load_v(kb, dt + 1);
        else if (kb + 1 < NKB) vn = load_v(kb + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int o = 0; o < 2; ++o) O[o][dt] = mfma16<T>(vf, pf[o], O[o][dt]);
        __builtin_amdgcn_sched_barrier(0);
        vf = vn;
      }
    }
    // ---- outputs: four 8-byte stores per existing tile (+ lse)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      if (o == 1 && !two) continue;
      const int q = (wave + o * PNW) * 16 + fr;
      if (q < L) {
        const float inv = 1.0f / sum[o];
        T* op = (T*)a.out + ((size_t)n * L + q) * d + h * 64 + fg * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          v4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(O[o][dt][e] * inv);
          if (a.flags & 2) __builtin_nontemporal_store(w, (v4*)(op + dt * 16));
          else *(v4*)(op + dt * 16) = w;
        }
        if (want_lse && fg == 0) a.lse[((size_t)n * H + h) * L + q] = (mx[o] * SC2 + log2f(sum[o])) * 0.6931471805599453f;
      }
    }
    kbuf = kbuf + 2 >= 3 ? kbuf - 1 : kbuf + 2;
  }
}

// ======================================================================================= backward A: dQ (+ delta)
// per query tile:  S^T, P^T = exp(S^T*scale - lse), dP^T = V dO^T, dS^T = P^T (dP^T - delta) * scale,
//                  dQ^T = K^T dS^T.   LDS: K rows, V rows (DMA-staged); K^T fragments by transpose-read.
template <typename T, int NKT, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using v8 = typename Vec<T>::v8;
  using v4 = typename Vec<T>::v4;
  constexpr int LP = NKT * 16;
  constexpr int MAXQ = (NKT + 3) / 4;
  constexpr int DSB = NKT > 16 ? 4 : (NKT / 2 < 8 ? NKT / 2 : 8);   // 32-key blocks of dS^T held at a time (register budget)
  char* sK = smem;
  char* sV = smem + LP * 128;
  const int L = a.L, H = a.H, d = H * 64;
  const int n = blockIdx.x / H, h = blockIdx.x % H;
  const size_t ld = (size_t)3 * d;
  const T* base = (const T*)a.qkv + (size_t)n * L * ld + h * 64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int nqt = (L + 15) >> 4;

  // per-wave operands straight from global; short sequences request all of them before the K/V DMA is waited for
  struct QOps { v8 q0, q1, do0, do1; float dl, lse; };
  auto load_ops = [&](int qt) -> QOps {
    QOps r;
    int qr = qt * 16 + fr;
    qr = qr < L ? qr : L - 1;
    const size_t tok = (size_t)n * L + qr;
    const T* qp = base + (size_t)qr * ld + fg * 8;
    const T* dop = (const T*)a.dout + tok * d + h * 64 + fg * 8;
    const T* op = (const T*)a.out + tok * d + h * 64 + fg * 8;
    r.q0 = *(const v8*)qp; r.q1 = *(const v8*)(qp + 32);
    r.do0 = *(const v8*)dop; r.do1 = *(const v8*)(dop + 32);
    const v8 o0 = *(const v8*)op, o1 = *(const v8*)(op + 32);
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) t += to_f32<T>(r.do0[e]) * to_f32<T>(o0[e]) + to_f32<T>(r.do1[e]) * to_f32<T>(o1[e]);
    r.dl = quad_sum(t);                                        // delta = rowsum(dO * O)
    r.lse = a.lse[((size_t)n * H + h) * L + qr];
    return r;
  };
  constexpr bool PRELOAD = NKT <= 16;
  QOps pre[PRELOAD ? MAXQ : 1];
  if constexpr (PRELOAD) {
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) pre[i] = load_ops(wave + 4 * i);
  }
  stage_rows_dma<T>(sK, base + d, ld, L, LP, wave, lane);
  stage_rows_dma<T>(sV, base + 2 * d, ld, L, LP, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto process = [&](const int qt, const QOps& O_) {
    const int qrow = qt * 16 + fr;
    if (qrow < L && fg == 0) a.delta[((size_t)n * H + h) * L + qrow] = O_.dl;
    const int nkt = CAUSAL ? (qt + 1 < NKT ? qt + 1 : NKT) : NKT;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb0 = 0; kb0 < NKT / 2; kb0 += DSB) {
      if (2 * kb0 >= nkt) break;
      v8 dsf[DSB];     // dS^T packed to 16-bit as soon as a 32-key block is done
#pragma unroll
      for (int w = 0; w < DSB; ++w) {
        const int kb = kb0 + w;
        f32x4 dsv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int kt = 2 * kb + u;
          dsv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (kt < NKT && kt < nkt) {
            f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
            sv = mfma16<T>(frag_rows<T>(sK, kt, 0, fr, fg), O_.q0, sv);
            sv = mfma16<T>(frag_rows<T>(sK, kt, 1, fr, fg), O_.q1, sv);
            dp = mfma16<T>(frag_rows<T>(sV, kt, 0, fr, fg), O_.do0, dp);
            dp = mfma16<T>(frag_rows<T>(sV, kt, 1, fr, fg), O_.do1, dp);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kt * 16 + fg * 4 + r;
              const bool ok = key < L && (!CAUSAL || key <= qrow);
              const float p = ok ? __expf(sv[r] * 0.125f - O_.lse) : 0.f;
              dsv[u][r] = p * (dp[r] - O_.dl) * 0.125f;
            }
          }
        }
        dsf[w] = pack8<T>(dsv[0], dsv[1]);
      }
#pragma unroll
      for (int w = 0; w < DSB; ++w) {
        const int kb = kb0 + w;
        if (2 * kb < NKT && 2 * kb < nkt) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16<T>(frag_vt<T>(sK, kb, dt, fr, fg), dsf[w], dq[dt]);
        }
      }
    }
    if (qrow < L) {
      T* gp = (T*)a.dqkv + ((size_t)n * L + qrow) * ld + h * 64 + fg * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        v4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(dq[dt][e]);
        *(v4*)(gp + dt * 16) = w;
      }
    }
  };
  if constexpr (PRELOAD) {
#pragma unroll
    for (int qi = 0; qi < MAXQ; ++qi) {
      if (wave + 4 * qi >= nqt) break;
      process(wave + 4 * qi, pre[qi]);
    }
  } else {
    for (int qt = wave; qt < nqt; qt += 4) process(qt, load_ops(qt));
  }
}

// ======================================================================================= backward B: dK, dV
// per key tile (col = key = lane&15), looping over 32-query blocks:
//   S = Q K^T (rows = queries), P, dP = dO V^T, dS ;  dV^T += dO^T P ;  dK^T += Q^T dS.
// LDS: Q rows, dO rows (DMA-staged; Q^T / dO^T fragments by transpose-read), lse[LP], delta[LP].
template <typename T, int NKT, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using v8 = typename Vec<T>::v8;
  using v4 = typename Vec<T>::v4;
  constexpr int LP = NKT * 16;
  constexpr int MAXK = (NKT + 3) / 4;
  char* sQ = smem;
  char* sdO = smem + LP * 128;
  float* sLse = (float*)(smem + 2 * LP * 128);
  float* sDel = sLse + LP;
  const int L = a.L, H = a.H, d = H * 64;
  const int n = blockIdx.x / H, h = blockIdx.x % H;
  const size_t ld = (size_t)3 * d;
  const T* base = (const T*)a.qkv + (size_t)n * L * ld + h * 64;
  const T* dob = (const T*)a.dout + (size_t)n * L * d + h * 64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int nkt_all = (L + 15) >> 4;

  struct KOps { v8 k0, k1, v0, v1; };
  auto load_kv = [&](int kt) -> KOps {
    KOps r;
    int kr = kt * 16 + fr;
    kr = kr < L ? kr : L - 1;
    const T* kp = base + (size_t)kr * ld + d + fg * 8;
    r.k0 = *(const v8*)kp; r.k1 = *(const v8*)(kp + 32);
    r.v0 = *(const v8*)(kp + d); r.v1 = *(const v8*)(kp + d + 32);
    return r;
  };
  constexpr bool PRELOAD = NKT <= 16;
  KOps pre[PRELOAD ? MAXK : 1];
  if constexpr (PRELOAD) {
#pragma unroll
    for (int i = 0; i < MAXK; ++i) pre[i] = load_kv(wave + 4 * i);
  }
  for (int i = threadIdx.x; i < LP; i += 256) {
    sLse[i] = i < L ? a.lse[((size_t)n * H + h) * L + i] : 0.f;
    sDel[i] = i < L ? a.delta[((size_t)n * H + h) * L + i] : 0.f;
  }
  stage_rows_dma<T>(sQ, base, ld, L, LP, wave, lane);
  stage_rows_dma<T>(sdO, dob, (size_t)d, L, LP, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto process = [&](const int kt, const KOps& K_) {
    const int key = kt * 16 + fr;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int qb0 = CAUSAL ? (kt >> 1) : 0;     // query tiles below the key tile are fully masked
#pragma unroll
    for (int qb = 0; qb < NKT / 2; ++qb) {
      if (qb >= qb0 && qb * 32 < L) {
        f32x4 p[2], ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int qt = 2 * qb + u;
          f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
          sv = mfma16<T>(frag_rows<T>(sQ, qt, 0, fr, fg), K_.k0, sv);
          sv = mfma16<T>(frag_rows<T>(sQ, qt, 1, fr, fg), K_.k1, sv);
          dp = mfma16<T>(frag_rows<T>(sdO, qt, 0, fr, fg), K_.v0, dp);
          dp = mfma16<T>(frag_rows<T>(sdO, qt, 1, fr, fg), K_.v1, dp);
          const f32x4 l4 = *(const f32x4*)(sLse + qt * 16 + fg * 4);
          const f32x4 d4 = *(const f32x4*)(sDel + qt * 16 + fg * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + fg * 4 + r;
            const bool ok = q < L && key < L && (!CAUSAL || key <= q);
            const float pv = ok ? __expf(sv[r] * 0.125f - l4[r]) : 0.f;
            p[u][r] = pv;
            ds[u][r] = pv * (dp[r] - d4[r]) * 0.125f;
          }
        }
        const v8 pf = pack8<T>(p[0], p[1]);
        const v8 dsf = pack8<T>(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = mfma16<T>(frag_vt<T>(sdO, qb, dt, fr, fg), pf, dv[dt]);
          dk[dt] = mfma16<T>(frag_vt<T>(sQ, qb, dt, fr, fg), dsf, dk[dt]);
        }
      }
    }
    if (key < L) {
      T* gp = (T*)a.dqkv + ((size_t)n * L + key) * ld + d + h * 64 + fg * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        v4 wk, wv;
#pragma unroll
        for (int e = 0; e < 4; ++e) { wk[e] = from_f32<T>(dk[dt][e]); wv[e] = from_f32<T>(dv[dt][e]); }
        *(v4*)(gp + dt * 16) = wk;
        *(v4*)(gp + d + dt * 16) = wv;
      }
    }
  };
  if constexpr (PRELOAD) {
#pragma unroll
    for (int ki = 0; ki < MAXK; ++ki) {
      if (wave + 4 * ki >= nkt_all) break;
      process(wave + 4 * ki, pre[ki]);
    }
  } else {
    for (int kt = wave; kt < nkt_all; kt += 4) process(kt, load_kv(kt));
  }
}

// ======================================================================================= backward, CLS query only
// Last block of the image tower when it needs a backward (VPT / UPT): only x[:, 0, :] of that block is consumed
// (trainers/mvlpt.py:88), so only the CLS query carries a gradient.  Per (sequence, head), in fp32, one pass over K and V:
//   delta = dO.O,  p_j = exp(q.k_j / 8 - lse),  dp_j = dO.v_j,  ds_j = p_j (dp_j - delta) / 8,
//   dV_j = p_j dO,  dK_j = ds_j q,  dQ_0 = sum_j ds_j k_j,  dQ_{i>0} = 0.
// Four lanes share a key (16 of the 64 head dimensions each), a wave covers 16 keys, a block 64 keys per sweep.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_cls_kernel(const T* __restrict__ qkv, const T* __restrict__ o_cls,
                                                           const T* __restrict__ do_cls, const float* __restrict__ lse,
                                                           T* __restrict__ dqkv, int N, int L, int H) {
  using v8 = typename Vec<T>::v8;
  __shared__ float sq[64], sdo[64], red[4][64], sdelta;
  const int n = blockIdx.x / H, h = blockIdx.x % H, d = H * 64, tid = threadIdx.x;
  const size_t ld = (size_t)3 * d;
  const T* base = qkv + (size_t)n * L * ld + h * 64;
  T* gbase = dqkv + (size_t)n * L * ld + h * 64;
  if (tid < 64) {
    sq[tid] = to_f32<T>(base[tid]);
    const float g = to_f32<T>(do_cls[(size_t)n * d + h * 64 + tid]);
    sdo[tid] = g;
    const float v = wave_sum(g * to_f32<T>(o_cls[(size_t)n * d + h * 64 + tid]));
    if (tid == 0) sdelta = v;
  }
  __syncthreads();
  const float delta = sdelta, l0 = lse[((size_t)n * H + h) * L];
  const int lane = tid & 63, wave = tid >> 6, kg = lane >> 2, ch = lane & 3;
  float q16[16], g16[16], dqa[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { q16[e] = sq[ch * 16 + e]; g16[e] = sdo[ch * 16 + e]; dqa[e] = 0.f; }
  for (int j0 = wave * 16; j0 < L; j0 += 64) {
    const int j = j0 + kg;
    const int jc = j < L ? j : L - 1;
    const T* kp = base + (size_t)jc * ld + d + ch * 16;
    const v8 k0 = *(const v8*)kp, k1 = *(const v8*)(kp + 8);
    const v8 v0 = *(const v8*)(kp + d), v1 = *(const v8*)(kp + d + 8);
    float k16[16], sdot = 0.f, dp = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      k16[e] = to_f32<T>(k0[e]); k16[e + 8] = to_f32<T>(k1[e]);
      dp += g16[e] * to_f32<T>(v0[e]) + g16[e + 8] * to_f32<T>(v1[e]);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) sdot += q16[e] * k16[e];
    sdot += __shfl_xor(sdot, 1, 64); sdot += __shfl_xor(sdot, 2, 64);
    dp += __shfl_xor(dp, 1, 64); dp += __shfl_xor(dp, 2, 64);
    const float pj = j < L ? __expf(sdot * 0.125f - l0) : 0.f;
    const float ds = pj * (dp - delta) * 0.125f;
    v8 wk[2], wv[2], wz[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      dqa[e] += ds * k16[e];
      wk[e >> 3][e & 7] = from_f32<T>(ds * q16[e]);
      wv[e >> 3][e & 7] = from_f32<T>(pj * g16[e]);
      wz[e >> 3][e & 7] = from_f32<T>(0.f);
    }
    if (j < L) {
      T* gp = gbase + (size_t)j * ld + ch * 16;
      if (j > 0) { *(v8*)gp = wz[0]; *(v8*)(gp + 8) = wz[1]; }            // dQ of the other queries
      *(v8*)(gp + d) = wk[0]; *(v8*)(gp + d + 8) = wk[1];
      *(v8*)(gp + 2 * d) = wv[0]; *(v8*)(gp + 2 * d + 8) = wv[1];
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float v = dqa[e];
    v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (kg == 0) red[wave][ch * 16 + e] = v;
  }
  __syncthreads();
  if (tid < 64) gbase[tid] = from_f32<T>(red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

hipError_t launch_attn_bwd_cls(int dtype, const void* qkv, const void* o_cls, const void* do_cls, const float* lse, void* dqkv,
                               int N, int L, int H, hipStream_t s) {
  if (N <= 0 || L <= 0 || H <= 0) return hipErrorInvalidValue;
  if (dtype == DT_F16)
    hipLaunchKernelGGL(attn_bwd_cls_kernel<f16>, dim3(N * H), dim3(256), 0, s, (const f16*)qkv, (const f16*)o_cls, (const f16*)do_cls, lse,
                       (f16*)dqkv, N, L, H);
  else if (dtype == DT_BF16)
    hipLaunchKernelGGL(attn_bwd_cls_kernel<bf16>, dim3(N * H), dim3(256), 0, s, (const bf16*)qkv, (const bf16*)o_cls, (const bf16*)do_cls,
                       lse, (bf16*)dqkv, N, L, H);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ======================================================================================= launchers
static thread_local hipEvent_t t_ev_a = nullptr, t_ev_b = nullptr;      // launch_attn_fwd's optional dispatch events
template <typename T, int NKT, bool CAUSAL>
static hipError_t fwd_t(const AttnArgs& a, hipStream_t s) {
  constexpr int LP = NKT * 16;
  constexpr int lds = 2 * LP * 128;
  static bool set = false;
  if (!set) { hipFuncSetAttribute((const void*)attn_fwd_kernel<T, NKT, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); set = true; }
  hipExtLaunchKernelGGL((attn_fwd_kernel<T, NKT, CAUSAL>), dim3(a.N * a.H), dim3(256), lds, s, t_ev_a, t_ev_b, 0, a);
  return hipGetLastError();
}
template <typename T, int NKT, bool CAUSAL>
static hipError_t bwd_t(const AttnBwdArgs& a, hipStream_t s) {
  constexpr int LP = NKT * 16;
  constexpr int lds_a = 2 * LP * 128;
  constexpr int lds_b = 2 * LP * 128 + 2 * LP * 4;
  static bool set = false;
  if (!set) {
    hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<T, NKT, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a);
    hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<T, NKT, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
    set = true;
  }
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, NKT, CAUSAL>), dim3(a.N * a.H), dim3(256), lds_a, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, NKT, CAUSAL>), dim3(a.N * a.H), dim3(256), lds_b, s, a);
  return hipGetLastError();
}

#define MVLPT_NKT_SWITCH(FN, ARGS)                                  \
  switch (nkt) {                                                    \
    case 2: return FN<T, 2, CAUSAL> ARGS;                           \
    case 4: return FN<T, 4, CAUSAL> ARGS;                           \
    case 6: return FN<T, 6, CAUSAL> ARGS;                           \
    case 8: return FN<T, 8, CAUSAL> ARGS;                           \
    case 10: return FN<T, 10, CAUSAL> ARGS;                         \
    case 12: return FN<T, 12, CAUSAL> ARGS;                         \
    case 14: return FN<T, 14, CAUSAL> ARGS;                         \
    case 16: return FN<T, 16, CAUSAL> ARGS;                         \
  }                                                                 \
  if constexpr (!CAUSAL) {   /* long sequences exist only in the vision tower (ViT-L/14: 257 / 577 tokens) */ \
    if (nkt == 18) return FN<T, 18, CAUSAL> ARGS;                   \
    if (nkt == 38) return FN<T, 38, CAUSAL> ARGS;                   \
  }                                                                 \
  return hipErrorInvalidValue;

template <typename T, bool CAUSAL> static hipError_t fwd_n(int nkt, const AttnArgs& a, hipStream_t s) {
  // 13 tiles (L = 193..208, ViT-B/16 with up to 11 prompts): 2 x 13 x 2 KiB = 52 KiB of LDS, so THREE workgroups fit a CU
  if constexpr (!CAUSAL) { if (nkt == 13) return fwd_t<T, 13, CAUSAL>(a, s); }
  MVLPT_NKT_SWITCH(fwd_t, (a, s))
}
template <typename T, bool CAUSAL> static hipError_t bwd_n(int nkt, const AttnBwdArgs& a, hipStream_t s) { MVLPT_NKT_SWITCH(bwd_t, (a, s)) }

static int nkt_for(int L) {
  int t = (L + 15) / 16; t += t & 1;
  if (t > 18) return 38;
  return t < 2 ? 2 : t;
}

// MVLPT_ATTN_STREAM: 0 = resident kernels only, 1 = streaming kernels always, unset = by sequence length
static int stream_mode() {
  static const int m = [] { const char* e = getenv("MVLPT_ATTN_STREAM"); return e ? atoi(e) : 2; }();
  return m;
}
static bool use_stream(int L, bool bwd) {
  const int m = stream_mode();
  if (m == 0) return false;
  if (m == 1) return true;
  (void)bwd;
  return L > 256;
}

hipError_t launch_attn_fwd(int dtype, const AttnArgs& a_, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
  t_ev_a = ea; t_ev_b = eb;
  // bit 0: non-temporal K/V staging (one workgroup reads them once), bit 1: non-temporal output stores  (+0.5 % on the step)
  static const int flags = [] { const char* e = getenv("MVLPT_ATTN_FLAGS"); return e ? atoi(e) : 3; }();
  AttnArgs a = a_;
  a.flags = flags;
  if (a.L <= 0 || a.L > attn_max_len() || a.N <= 0) return hipErrorInvalidValue;
  if (use_stream(a.L, false)) return launch_attn_fwd_stream(dtype, a, s);
  static const int odd_ok = [] { const char* e = getenv("MVLPT_ATTN_ODD"); return e ? atoi(e) : 1; }();
  // persistent variant for the headline's shape: 13 key tiles, full sequences, at least two heads per compute unit of the stream
  static const int persist = [] { const char* e = getenv("MVLPT_ATTN_PERSIST"); return e ? atoi(e) : 1; }();
  if (persist && !a.causal && a.q_rows <= 0 && (a.L + 15) / 16 == PNT && (dtype == DT_F16 || dtype == DT_BF16)) {
    const int total = a.N * a.H, cus = stream_cus(s);
    if (total >= 2 * cus) {
      static bool set = false;
      if (!set) {
        hipFuncSetAttribute((const void*)attn_fwdp_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
        hipFuncSetAttribute((const void*)attn_fwdp_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
        set = true;
      }
      if (dtype == DT_F16) hipExtLaunchKernelGGL((attn_fwdp_kernel<f16>), dim3(cus), dim3(PNW * 64), PLDS, s, ea, eb, 0, a, total);
      else hipExtLaunchKernelGGL((attn_fwdp_kernel<bf16>), dim3(cus), dim3(PNW * 64), PLDS, s, ea, eb, 0, a, total);
      return hipGetLastError();
    }
  }
  int nkt = nkt_for(a.L);
  if (odd_ok && !a.causal && (a.L + 15) / 16 == 13) nkt = 13;
  if (dtype == DT_F16) return a.causal ? fwd_n<f16, true>(nkt, a, s) : fwd_n<f16, false>(nkt, a, s);
  if (dtype == DT_BF16) return a.causal ? fwd_n<bf16, true>(nkt, a, s) : fwd_n<bf16, false>(nkt, a, s);
  return hipErrorInvalidValue;
}
hipError_t launch_attn_bwd(int dtype, const AttnBwdArgs& a, hipStream_t s) {
  if (a.L <= 0 || a.L > attn_max_len() || a.N <= 0) return hipErrorInvalidValue;
  if (use_stream(a.L, true)) return launch_attn_bwd_stream(dtype, a, s);
  const int nkt = nkt_for(a.L);
  if (dtype == DT_F16) return a.causal ? bwd_n<f16, true>(nkt, a, s) : bwd_n<f16, false>(nkt, a, s);
  if (dtype == DT_BF16) return a.causal ? bwd_n<bf16, true>(nkt, a, s) : bwd_n<bf16, false>(nkt, a, s);
  return hipErrorInvalidValue;
}

}  // namespace mvlpt
