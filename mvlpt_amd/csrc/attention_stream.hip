// Streaming attention kernels (any sequence length): instead of holding a whole head's K and V in LDS
// (attention.hip: 57-152 KiB per workgroup, 1-2 workgroups per CU, load -> compute -> store with nothing to overlap),
// a workgroup owns 64 queries (or 64 keys in the dK/dV kernel) and streams the other operand through a
// DOUBLE-BUFFERED 32 KiB LDS ring in 64-row chunks filled by LDS-DMA: chunk c+1 is in flight while chunk c is
// multiplied, five workgroups fit a CU, and the softmax runs online across chunks (lane-local thanks to the
// transposed S^T = K Q^T formulation, see attention.hip).  Workgroups of one (sequence, head) are placed on the same
// XCD (block b runs on XCD b % 8) so the chunks they all stream are served by that XCD's L2, not re-fetched from HBM.
//   forward : attn_fwd_stream_kernel   — grid (heads x 64-query chunks), streams K, V
//   backward: attn_dq_stream_kernel    — same ownership, streams K, V, writes dQ and delta = rowsum(dO * O)
//             attn_dkv_stream_kernel   — owns 64 keys, streams Q, dO (+ lse, delta), writes dK, dV
#include "attn_common.h"

namespace mvlpt {

constexpr int CK = 64;                 // rows per streamed chunk (4 MFMA tiles)
constexpr int CHUNK_BYTES = CK * 128;  // one 64 x 64 16-bit image

// block -> (sequence*head index, chunk) with all chunks of a head on one XCD
__device__ __forceinline__ bool map_block(int nheads, int nchunks, int& nh, int& ch) {
  const int b = blockIdx.x, x = b & 7, r = b >> 3;
  nh = (r / nchunks) * 8 + x;
  ch = r % nchunks;
  return nh < nheads;
}
static int stream_grid(int nheads, int nchunks) { return ((nheads + 7) / 8) * 8 * nchunks; }

// DMA one 64-row chunk (rows row0 .. row0+63 of `src`, row stride ld) into `dst`; 2 slabs per wave
template <typename T>
__device__ __forceinline__ void stage_chunk(char* dst, const T* src, size_t ld, int row0, int L, int wave, int lane) {
  const int srow = lane >> 3, chunk = (lane & 7) ^ srow;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int sl = wave + 4 * i;
    int row = row0 + sl * 8 + srow;
    row = row < L ? row : L - 1;
    dma_raw<16>(src + (size_t)row * ld + chunk * 8, dst + sl * 1024);
  }
}

// ======================================================================================= forward
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_stream_kernel(AttnArgs a, int nqc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K chunk | V chunk]
  using v8 = typename Vec<T>::v8;
  using v4 = typename Vec<T>::v4;
  const int L = a.L, H = a.H, d = H * 64;
  int nh, qc;
  if (!map_block(a.N * H, nqc, nh, qc)) return;
  const int n = nh / H, h = nh % H;
  const size_t ld = (size_t)3 * d;
  const T* base = (const T*)a.qkv + (size_t)n * L * ld + h * 64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int qt = qc * 4 + wave;
  const int qrow = qt * 16 + fr;
  const bool active = qt * 16 < L;
  int nch = (L + CK - 1) / CK;
  if (CAUSAL) nch = nch < qc + 1 ? nch : qc + 1;               // later chunks are fully masked for these queries

  auto issue = [&](int c) {
    char* buf = smem + (c & 1) * 2 * CHUNK_BYTES;
    stage_chunk<T>(buf, base + d, ld, c * CK, L, wave, lane);
    stage_chunk<T>(buf + CHUNK_BYTES, base + 2 * d, ld, c * CK, L, wave, lane);
  };
  issue(0);
  const T* qp = base + (size_t)(qrow < L ? qrow : L - 1) * ld + fg * 8;
  const v8 q0 = *(const v8*)qp, q1 = *(const v8*)(qp + 32);
  // consume the operands here so that the compiler's own vmcnt wait for them sits BEFORE the loop; inside the loop
  // only the counted waits below may appear (the compiler does not see the asm-issued DMA, its waits would drain it)
  asm volatile("" ::"v"(q0), "v"(q1));

  constexpr float SC = 0.125f * 1.4426950408889634f;
  float mrun = -INFINITY, sum = 0.f;
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c = 0; c < nch; ++c) {
    if (c > 0) __builtin_amdgcn_s_barrier();                    // every wave is done with the buffer chunk c+1 overwrites
    if (c + 1 < nch) { issue(c + 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // all pieces of chunk c have landed
    if (!active) continue;
    const char* sK = smem + (c & 1) * 2 * CHUNK_BYTES;
    const char* sV = sK + CHUNK_BYTES;
    f32x4 s[4];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      s[u] = mfma16<T>(frag_rows<T>(sK, u, 0, fr, fg), q0, s[u]);
      s[u] = mfma16<T>(frag_rows<T>(sK, u, 1, fr, fg), q1, s[u]);
      // raw scores (the scale sits in the exponent's FMA); the mask only where a key can be >= L or on the causal diagonal
      if (CAUSAL || c * CK + (u + 1) * 16 > L) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = c * CK + u * 16 + fg * 4 + r;
          const bool ok = key < L && (!CAUSAL || key <= qrow);
          s[u][r] = ok ? s[u][r] : -INFINITY;
        }
      }
      mx = fmaxf(mx, fmaxf(fmaxf(s[u][0], s[u][1]), fmaxf(s[u][2], s[u][3])));
    }
    mx = quad_max(mx) * SC;
    const float mnew = fmaxf(mrun, mx);                         // finite from chunk 0 on (key 0 is never masked)
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
    mrun = mnew;
    sum *= alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[u][r] = __builtin_amdgcn_exp2f(fmaf(s[u][r], SC, -mnew)); sum += s[u][r]; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const v8 pf = pack8<T>(s[2 * kb], s[2 * kb + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16<T>(frag_vt<T>(sV, kb, dt, fr, fg), pf, o[dt]);
    }
  }
  if (!active) return;
  sum = quad_sum(sum);
  const float inv = 1.0f / sum;
  if (qrow < L) {
    T* op = (T*)a.out + ((size_t)n * L + qrow) * d + h * 64 + fg * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      v4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(o[dt][e] * inv);
      *(v4*)(op + dt * 16) = w;
    }
    if (a.lse && fg == 0) a.lse[((size_t)n * H + h) * L + qrow] = (mrun + log2f(sum)) * 0.6931471805599453f;
  }
}

// ======================================================================================= backward: dQ (+ delta)
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_dq_stream_kernel(AttnBwdArgs a, int nqc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using v8 = typename Vec<T>::v8;
  using v4 = typename Vec<T>::v4;
  const int L = a.L, H = a.H, d = H * 64;
  int nh, qc;
  if (!map_block(a.N * H, nqc, nh, qc)) return;
  const int n = nh / H, h = nh % H;
  const size_t ld = (size_t)3 * d;
  const T* base = (const T*)a.qkv + (size_t)n * L * ld + h * 64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int qt = qc * 4 + wave;
  const int qrow = qt * 16 + fr;
  const bool active = qt * 16 < L;
  int nch = (L + CK - 1) / CK;
  if (CAUSAL) nch = nch < qc + 1 ? nch : qc + 1;

  auto issue = [&](int c) {
    char* buf = smem + (c & 1) * 2 * CHUNK_BYTES;
    stage_chunk<T>(buf, base + d, ld, c * CK, L, wave, lane);
    stage_chunk<T>(buf + CHUNK_BYTES, base + 2 * d, ld, c * CK, L, wave, lane);
  };
  issue(0);
  const int qr = qrow < L ? qrow : L - 1;
  const size_t tok = (size_t)n * L + qr;
  const T* qp = base + (size_t)qr * ld + fg * 8;
  const T* dop = (const T*)a.dout + tok * d + h * 64 + fg * 8;
  const T* op = (const T*)a.out + tok * d + h * 64 + fg * 8;
  const v8 q0 = *(const v8*)qp, q1 = *(const v8*)(qp + 32);
  const v8 do0 = *(const v8*)dop, do1 = *(const v8*)(dop + 32);
  const v8 o0 = *(const v8*)op, o1 = *(const v8*)(op + 32);
  float dl = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) dl += to_f32<T>(do0[e]) * to_f32<T>(o0[e]) + to_f32<T>(do1[e]) * to_f32<T>(o1[e]);
  dl = quad_sum(dl);
  const float lse = a.lse[((size_t)n * H + h) * L + qr];
  asm volatile("" ::"v"(q0), "v"(q1), "v"(do0), "v"(do1), "v"(dl), "v"(lse));   // see the forward kernel
  if (active && qrow < L && fg == 0) a.delta[((size_t)n * H + h) * L + qrow] = dl;

  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nch; ++c) {
    if (c > 0) __builtin_amdgcn_s_barrier();
    if (c + 1 < nch) { issue(c + 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!active) continue;
    const char* sK = smem + (c & 1) * 2 * CHUNK_BYTES;
    const char* sV = sK + CHUNK_BYTES;
    f32x4 ds[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
      sv = mfma16<T>(frag_rows<T>(sK, u, 0, fr, fg), q0, sv);
      sv = mfma16<T>(frag_rows<T>(sK, u, 1, fr, fg), q1, sv);
      dp = mfma16<T>(frag_rows<T>(sV, u, 0, fr, fg), do0, dp);
      dp = mfma16<T>(frag_rows<T>(sV, u, 1, fr, fg), do1, dp);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = c * CK + u * 16 + fg * 4 + r;
        const bool ok = key < L && (!CAUSAL || key <= qrow);
        const float p = ok ? __expf(sv[r] * 0.125f - lse) : 0.f;
        ds[u][r] = p * (dp[r] - dl) * 0.125f;
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const v8 dsf = pack8<T>(ds[2 * kb], ds[2 * kb + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16<T>(frag_vt<T>(sK, kb, dt, fr, fg), dsf, dq[dt]);
    }
  }
  if (active && qrow < L) {
    T* gp = (T*)a.dqkv + ((size_t)n * L + qrow) * ld + h * 64 + fg * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      v4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(dq[dt][e]);
      *(v4*)(gp + dt * 16) = w;
    }
  }
}

// ======================================================================================= backward: dK, dV
// owns 64 keys (one 16-key tile per wave, column = key = lane & 15); streams 64-query chunks of Q and dO plus their
// lse / delta (256 B each, one 4-byte LDS-DMA per wave: every wave requests the same words so that all waves have
// the same number of DMA operations in flight).
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_dkv_stream_kernel(AttnBwdArgs a, int nkc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][Q chunk | dO chunk | lse(64 f32) | delta(64 f32)]
  using v8 = typename Vec<T>::v8;
  using v4 = typename Vec<T>::v4;
  constexpr int BUF = 2 * CHUNK_BYTES + 512;
  const int L = a.L, H = a.H, d = H * 64;
  int nh, kc;
  if (!map_block(a.N * H, nkc, nh, kc)) return;
  const int n = nh / H, h = nh % H;
  const size_t ld = (size_t)3 * d;
  const T* base = (const T*)a.qkv + (size_t)n * L * ld + h * 64;
  const T* dob = (const T*)a.dout + (size_t)n * L * d + h * 64;
  const float* lse_g = a.lse + ((size_t)n * H + h) * L;
  const float* del_g = a.delta + ((size_t)n * H + h) * L;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int kt = kc * 4 + wave;
  const int key = kt * 16 + fr;
  const bool active = kt * 16 < L;
  const int nch = (L + CK - 1) / CK;
  const int c_first = CAUSAL ? kc : 0;                         // query chunks before the key chunk are fully masked

  auto issue = [&](int c) {
    char* buf = smem + (c & 1) * BUF;
    stage_chunk<T>(buf, base, ld, c * CK, L, wave, lane);
    stage_chunk<T>(buf + CHUNK_BYTES, dob, (size_t)d, c * CK, L, wave, lane);
    int q = c * CK + lane;
    q = q < L ? q : L - 1;
    dma_raw<4>(lse_g + q, buf + 2 * CHUNK_BYTES);
    dma_raw<4>(del_g + q, buf + 2 * CHUNK_BYTES + 256);
  };
  issue(c_first);
  const int kr = key < L ? key : L - 1;
  const T* kp = base + (size_t)kr * ld + d + fg * 8;
  const v8 k0 = *(const v8*)kp, k1 = *(const v8*)(kp + 32);
  const v8 v0 = *(const v8*)(kp + d), v1 = *(const v8*)(kp + d + 32);
  asm volatile("" ::"v"(k0), "v"(k1), "v"(v0), "v"(v1));                        // see the forward kernel

  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int c = c_first; c < nch; ++c) {
    if (c > c_first) __builtin_amdgcn_s_barrier();
    if (c + 1 < nch) { issue(c + 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!active) continue;
    const char* sQ = smem + (c & 1) * BUF;
    const char* sdO = sQ + CHUNK_BYTES;
    const float* sLse = (const float*)(sQ + 2 * CHUNK_BYTES);
    const float* sDel = sLse + 64;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x4 p[2], ds[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * qb + u;
        f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
        sv = mfma16<T>(frag_rows<T>(sQ, t, 0, fr, fg), k0, sv);
        sv = mfma16<T>(frag_rows<T>(sQ, t, 1, fr, fg), k1, sv);
        dp = mfma16<T>(frag_rows<T>(sdO, t, 0, fr, fg), v0, dp);
        dp = mfma16<T>(frag_rows<T>(sdO, t, 1, fr, fg), v1, dp);
        const f32x4 l4 = *(const f32x4*)(sLse + t * 16 + fg * 4);
        const f32x4 d4 = *(const f32x4*)(sDel + t * 16 + fg * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = c * CK + t * 16 + fg * 4 + r;
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
  if (active && key < L) {
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
}

// ======================================================================================= launchers
template <typename T, bool CAUSAL>
static hipError_t fwd_stream_t(const AttnArgs& a, hipStream_t s) {
  const int rows = a.q_rows > 0 ? (a.q_rows < a.L ? a.q_rows : a.L) : a.L;
  const int nqc = (rows + CK - 1) / CK;
  hipLaunchKernelGGL((attn_fwd_stream_kernel<T, CAUSAL>), dim3(stream_grid(a.N * a.H, nqc)), dim3(256), 4 * CHUNK_BYTES, s, a, nqc);
  return hipGetLastError();
}
template <typename T, bool CAUSAL>
static hipError_t bwd_stream_t(const AttnBwdArgs& a, hipStream_t s) {
  const int nc = (a.L + CK - 1) / CK;
  hipLaunchKernelGGL((attn_dq_stream_kernel<T, CAUSAL>), dim3(stream_grid(a.N * a.H, nc)), dim3(256), 4 * CHUNK_BYTES, s, a, nc);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((attn_dkv_stream_kernel<T, CAUSAL>), dim3(stream_grid(a.N * a.H, nc)), dim3(256),
                     2 * (2 * CHUNK_BYTES + 512), s, a, nc);
  return hipGetLastError();
}

hipError_t launch_attn_fwd_stream(int dtype, const AttnArgs& a, hipStream_t s) {
  if (a.L <= 0 || a.N <= 0) return hipErrorInvalidValue;
  if (dtype == DT_F16) return a.causal ? fwd_stream_t<f16, true>(a, s) : fwd_stream_t<f16, false>(a, s);
  if (dtype == DT_BF16) return a.causal ? fwd_stream_t<bf16, true>(a, s) : fwd_stream_t<bf16, false>(a, s);
  return hipErrorInvalidValue;
}
hipError_t launch_attn_bwd_stream(int dtype, const AttnBwdArgs& a, hipStream_t s) {
  if (a.L <= 0 || a.N <= 0) return hipErrorInvalidValue;
  if (dtype == DT_F16) return a.causal ? bwd_stream_t<f16, true>(a, s) : bwd_stream_t<f16, false>(a, s);
  if (dtype == DT_BF16) return a.causal ? bwd_stream_t<bf16, true>(a, s) : bwd_stream_t<bf16, false>(a, s);
  return hipErrorInvalidValue;
}

}  // namespace mvlpt
