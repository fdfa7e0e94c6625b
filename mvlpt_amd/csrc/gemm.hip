// MFMA GEMM for the frozen CLIP linears:  C[M,N] = A[M,K] * Bt[N,K]^T  (+ fused epilogue).
//
// Replaces every nn.Linear / conv1-as-GEMM / `x @ proj` call on the hot path (SURVEY.md §2.3 K0,K4,K6,
// K7,K8,K10,K12 forward and their dX twins; reference call sites clip/model.py:174-176,183,207,234 and
// trainers/mvlpt.py:91,128).  Both operands are K-contiguous 16-bit (fp16 or bf16) rows: nn.Linear
// stores W as [N,K] so the forward reads it as-is; the dX GEMM reads the pre-transposed copy packed
// once at load time (weights are frozen, trainers/mvlpt.py:855-858).
//
// gfx950 design: 128x128x64 tile, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32
// accumulators (fp32).  A/B tiles go HBM -> LDS with 16-byte LDS-DMA (global_load_lds), double
// buffered.  The LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is applied to
// the per-lane SOURCE address and undone on the ds_read_b128 side: 16-byte chunk c of tile row r lives
// at chunk (c ^ (r & 7)).  MFMA operands are swapped (D = Bfrag x Afrag) so each lane ends up with four
// consecutive output COLUMNS of one row -> 8/16-byte epilogue stores, float4 bias loads.
// Workgroups are remapped so that consecutive tiles (sharing an A panel) run on the same XCD/L2.
#include "kernels.h"

namespace mvlpt {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + B
constexpr int GEMM_LDS = 2 * STAGE_BYTES;      // double buffered: 64 KiB -> 2 workgroups / CU

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bt_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using v8 = typename Vec<T>::v8;
  using v4 = typename Vec<T>::v4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = g.M, N = g.N, K = g.K;

  // XCD-aware bijective remap: workgroup b runs on XCD b%8; give each XCD a contiguous run of tiles.
  const int nwg = gridDim.x, b = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const int tilesN = (N + BN - 1) / BN;
  const int m0 = (t / tilesN) * BM, n0 = (t % tilesN) * BN;

  const T* __restrict__ A = (const T*)g.A;
  const T* __restrict__ Bt = (const T*)g.Bt;

  // ---- staging: thread -> (row, 16B chunk) of a 1 KiB LDS slab (8 rows x 128 B) ------------------
  const int srow = lane >> 3;                       // row inside the slab == (tile row & 7)
  const int scol = ((lane & 7) ^ srow) * 8;         // SOURCE chunk (elements) for LDS chunk lane&7
  const T* ap[4];
  const T* bp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 8 + srow;
    int ar = m0 + row; ar = ar < M ? ar : M - 1;    // edge tiles: re-read the last row (never stored)
    int br = n0 + row; br = br < N ? br : N - 1;
    ap[i] = A + (size_t)ar * K + scol;
    bp[i] = Bt + (size_t)br * K + scol;
  }
  auto stage = [&](int s, int kt) {
    char* base = smem + s * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(ap[i] + kt * BK, base + (i * 4 + wave) * 1024);
      glds16(bp[i] + kt * BK, base + TILE_BYTES + (i * 4 + wave) * 1024);
    }
  };

  // ---- fragment addressing ------------------------------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int a_off = (wm * 64 + fr) * 128;           // + i*16*128
  const int b_off = TILE_BYTES + (wn * 64 + fr) * 128;
  const int c0 = ((0 + fg) ^ (fr & 7)) * 16;        // k-step 0 chunk
  const int c1 = ((4 + fg) ^ (fr & 7)) * 16;        // k-step 1 chunk

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* base = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks ? c1 : c0;
      v8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *(const v8*)(base + a_off + i * 2048 + c);
        bf[i] = *(const v8*)(base + b_off + i * 2048 + c);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(bf[j], af[i], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = .. + fr][n = .. + 4*fg + 0..3] -------------------------------------
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fg * 4;
      if (n >= N) continue;
      f32x4 v = acc[i][j];
      if (g.bias) {
        const f32x4 bv = *(const f32x4*)(g.bias + n);
        v += bv;
      }
      const size_t o = (size_t)m * N + n;
      if constexpr (EPI == EPI_STORE16) {
        v4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(v[e]);
        *(v4*)((T*)g.out + o) = w;
      } else if constexpr (EPI == EPI_GELU) {
        v4 w;
        if (g.out2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(v[e]);
          *(v4*)((T*)g.out2 + o) = w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(quick_gelu(v[e]));
        *(v4*)((T*)g.out + o) = w;
      } else if constexpr (EPI == EPI_RESID32) {
        const f32x4 rv = *(const f32x4*)(g.resid + o);
        *(f32x4*)((float*)g.out + o) = v + rv;
      } else if constexpr (EPI == EPI_GELUBWD) {
        const v4 u = *(const v4*)((const T*)g.aux + o);
        v4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(v[e] * quick_gelu_grad(to_f32<T>(u[e])));
        *(v4*)((T*)g.out + o) = w;
      } else {  // EPI_STORE32
        *(f32x4*)((float*)g.out + o) = v;
      }
    }
  }
}

template <typename T, int EPI>
static hipError_t launch_t(const GemmArgs& g, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_bt_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    attr_set = true;
  }
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm_bt_kernel<T, EPI>), dim3(tiles), dim3(256), GEMM_LDS, s, g);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_epi(const GemmArgs& g, int epi, hipStream_t s) {
  switch (epi) {
    case EPI_STORE16: return launch_t<T, EPI_STORE16>(g, s);
    case EPI_GELU: return launch_t<T, EPI_GELU>(g, s);
    case EPI_RESID32: return launch_t<T, EPI_RESID32>(g, s);
    case EPI_GELUBWD: return launch_t<T, EPI_GELUBWD>(g, s);
    case EPI_STORE32: return launch_t<T, EPI_STORE32>(g, s);
  }
  return hipErrorInvalidValue;
}

// K must be a multiple of 64 and N of 4 (callers pad); M is arbitrary.
hipError_t launch_gemm(int dtype, int epi, const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K % BK) != 0 || (g.N % 4) != 0) return hipErrorInvalidValue;
  if (epi == EPI_RESID32 && !g.resid) return hipErrorInvalidValue;
  if (epi == EPI_GELUBWD && !g.aux) return hipErrorInvalidValue;
  if (dtype == DT_F16) return launch_epi<f16>(g, epi, s);
  if (dtype == DT_BF16) return launch_epi<bf16>(g, epi, s);
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
