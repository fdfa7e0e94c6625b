// Input pipeline on the device: decoded uint8 HWC images -> the [B,3,R,R] normalised tensor the image tower reads.
// Replaces, bit for bit, what the reference does per image on CPU worker processes with PIL + torchvision
// (Dassl build_transform, configs/trainers/MVLPT/vit_b16.yaml:8-13: random_resized_crop (bicubic) / random_flip /
// normalize; ELEVATER eval: trainers/vision_benchmark/evaluation/feature.py:538-553 Resize(BICUBIC) [+ CenterCrop]):
//   crop box -> Pillow's two-pass antialiased bicubic resample for 8-bit images (22-bit fixed-point coefficients,
//   8-bit intermediate image, horizontal pass first, a pass whose size does not change is skipped) -> optional
//   window of the resized image (CenterCrop) -> optional horizontal flip -> u8/255 -> (x - mean)/std.
// Three kernels per batch, all HBM-bound byte work (no MFMA):
//   pp_coeffs_kernel      per image and pass: tap range + fixed-point weights of every output index, in DOUBLE with
//                         FMA contraction off so that the integers equal the ones Pillow's C code computes
//   pp_horizontal_kernel  one block per (image, source row): u8 x int32 taps -> 8-bit intermediate rows
//   pp_vertical_kernel    one block per (image, output row): taps down the columns (coalesced across x), then
//                         ToTensor / Normalize / flip, written as three coalesced CHW planes (fp32 / f16 / bf16)
#include "kernels.h"

namespace mvlpt {

constexpr int PP_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ double pp_bicubic(double x) {
#pragma clang fp contract(off)
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// table layout per (image, pass): int32 bounds[2 * n_out] then int32 kk[ks_max][n_out] (tap-major: coalesced reads)
__global__ __launch_bounds__(256) void pp_coeffs_kernel(const PpDesc* __restrict__ descs, int32_t* __restrict__ tables,
                                                        size_t table_stride, int ks_max, int out_h, int out_w) {
#pragma clang fp contract(off)
  const PpDesc d = descs[blockIdx.x];
  const int pass = blockIdx.y;                                   // 0 horizontal, 1 vertical
  const int in_size = pass == 0 ? d.crop_w : d.crop_h;
  const int rs = pass == 0 ? d.resize_w : d.resize_h;
  const int off = pass == 0 ? d.out_left : d.out_top;
  const int n_out = pass == 0 ? out_w : out_h;
  int32_t* bounds = tables + ((size_t)blockIdx.x * 2 + pass) * table_stride;
  int32_t* kk = bounds + 2 * n_out;
  double scale = (double)in_size / rs, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const double ss = 1.0 / filterscale;
  for (int i = threadIdx.x; i < n_out; i += blockDim.x) {
    if (in_size == rs) {                                         // pass skipped by Pillow: identity tap
      bounds[2 * i] = off + i;
      bounds[2 * i + 1] = -1;
      continue;
    }
    const int xx = off + i;
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += pp_bicubic((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < xmax; ++x) {
      double v = pp_bicubic((x + xmin - center + 0.5) * ss);
      if (ww != 0.0) v = v / ww;
      kk[(size_t)x * n_out + i] = v < 0 ? (int32_t)(-0.5 + v * (double)(1 << PP_PRECISION_BITS))
                                        : (int32_t)(0.5 + v * (double)(1 << PP_PRECISION_BITS));
    }
    bounds[2 * i] = xmin;
    bounds[2 * i + 1] = xmax;
  }
}

__device__ __forceinline__ uint8_t pp_clip8(int32_t v) {
  v >>= PP_PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[b][y][i][c] (u8), y over the crop rows the vertical pass reads, i over the output window columns
__global__ __launch_bounds__(256) void pp_horizontal_kernel(const uint8_t* __restrict__ src, const PpDesc* __restrict__ descs,
                                                            const int32_t* __restrict__ tables, size_t table_stride,
                                                            uint8_t* __restrict__ tmp, size_t tmp_stride, int out_h, int out_w) {
  const int b = blockIdx.y, y = blockIdx.x;
  const PpDesc d = descs[b];
  if (y >= d.crop_h) return;
  const int32_t* hb = tables + ((size_t)b * 2 + 0) * table_stride;
  const int32_t* hk = hb + 2 * out_w;
  const int32_t* vb = tables + ((size_t)b * 2 + 1) * table_stride;
  // rows the vertical pass touches: [first tap of the first window row, last tap of the last window row]
  const int v0 = vb[0], vl = vb[2 * (out_h - 1)], vn = vb[2 * (out_h - 1) + 1];
  const int vend = vn < 0 ? vl + 1 : vl + vn;
  if (y < v0 || y >= vend) return;
  const uint8_t* row = src + d.offset + ((size_t)(d.crop_top + y) * d.width + d.crop_left) * 3;
  uint8_t* out = tmp + (size_t)b * tmp_stride + (size_t)y * out_w * 3;
  for (int i = threadIdx.x; i < out_w; i += blockDim.x) {
    const int xmin = hb[2 * i], n = hb[2 * i + 1];
    uint8_t r0, r1, r2;
    if (n < 0) {
      r0 = row[xmin * 3]; r1 = row[xmin * 3 + 1]; r2 = row[xmin * 3 + 2];
    } else {
      int32_t s0 = 1 << (PP_PRECISION_BITS - 1), s1 = s0, s2 = s0;
      const uint8_t* p = row + (size_t)xmin * 3;
      for (int x = 0; x < n; ++x) {
        const int32_t k = hk[(size_t)x * out_w + i];
        s0 += (int32_t)p[3 * x] * k; s1 += (int32_t)p[3 * x + 1] * k; s2 += (int32_t)p[3 * x + 2] * k;
      }
      r0 = pp_clip8(s0); r1 = pp_clip8(s1); r2 = pp_clip8(s2);
    }
    out[3 * i] = r0; out[3 * i + 1] = r1; out[3 * i + 2] = r2;
  }
}

template <typename TO> __device__ __forceinline__ void pp_store(TO* p, float v) { *p = (TO)v; }

template <typename TO>
__global__ __launch_bounds__(256) void pp_vertical_kernel(const PpDesc* __restrict__ descs, const int32_t* __restrict__ tables,
                                                          size_t table_stride, const uint8_t* __restrict__ tmp,
                                                          size_t tmp_stride, TO* __restrict__ out, uint8_t* __restrict__ out_u8,
                                                          int out_h, int out_w, float m0, float m1, float m2, float d0,
                                                          float d1, float d2) {
  const int b = blockIdx.y, j = blockIdx.x;                       // output row j of the window
  const PpDesc d = descs[b];
  const int32_t* vb = tables + ((size_t)b * 2 + 1) * table_stride;
  const int32_t* vk = vb + 2 * out_h;
  const int ymin = vb[2 * j], n = vb[2 * j + 1];
  const uint8_t* t = tmp + (size_t)b * tmp_stride;
  const size_t plane = (size_t)out_h * out_w;
  TO* o = out ? out + (size_t)b * 3 * plane + (size_t)j * out_w : nullptr;
  for (int i = threadIdx.x; i < out_w; i += blockDim.x) {
    uint8_t r0, r1, r2;
    if (n < 0) {
      const uint8_t* p = t + ((size_t)ymin * out_w + i) * 3;
      r0 = p[0]; r1 = p[1]; r2 = p[2];
    } else {
      int32_t s0 = 1 << (PP_PRECISION_BITS - 1), s1 = s0, s2 = s0;
      for (int y = 0; y < n; ++y) {
        const int32_t k = vk[(size_t)y * out_h + j];
        const uint8_t* p = t + ((size_t)(ymin + y) * out_w + i) * 3;
        s0 += (int32_t)p[0] * k; s1 += (int32_t)p[1] * k; s2 += (int32_t)p[2] * k;
      }
      r0 = pp_clip8(s0); r1 = pp_clip8(s1); r2 = pp_clip8(s2);
    }
    const int x = d.flip ? out_w - 1 - i : i;
    if (out_u8) {                                                 // resized 8-bit image (parity tests), HWC
      uint8_t* q = out_u8 + (((size_t)b * out_h + j) * out_w + x) * 3;
      q[0] = r0; q[1] = r1; q[2] = r2;
    }
    if (o) {
      // ToTensor: u8 -> fp32, / 255;  Normalize: (x - mean) / std  — two correctly rounded fp32 divisions, no FMA
      pp_store<TO>(o + x, ((float)r0 / 255.0f - m0) / d0);
      pp_store<TO>(o + plane + x, ((float)r1 / 255.0f - m1) / d1);
      pp_store<TO>(o + 2 * plane + x, ((float)r2 / 255.0f - m2) / d2);
    }
  }
}

hipError_t launch_preprocess(const uint8_t* src, const PpDesc* descs_dev, int B, int max_crop_h, int ks_max, int out_h, int out_w,
                             int32_t* tables, size_t table_stride, uint8_t* tmp, size_t tmp_stride, const float* mean,
                             const float* stdv, void* out, int out_dtype, uint8_t* out_u8, hipStream_t s) {
  hipLaunchKernelGGL(pp_coeffs_kernel, dim3(B, 2), dim3(256), 0, s, descs_dev, tables, table_stride, ks_max, out_h, out_w);
  hipLaunchKernelGGL(pp_horizontal_kernel, dim3(max_crop_h, B), dim3(256), 0, s, src, descs_dev, tables, table_stride, tmp,
                     tmp_stride, out_h, out_w);
#define MVLPT_PP_V(TO) hipLaunchKernelGGL(pp_vertical_kernel<TO>, dim3(out_h, B), dim3(256), 0, s, descs_dev, tables, table_stride, tmp, \
                                          tmp_stride, (TO*)out, out_u8, out_h, out_w, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2])
  if (out_dtype == DT_F32) MVLPT_PP_V(float);
  else if (out_dtype == DT_F16) MVLPT_PP_V(f16);
  else if (out_dtype == DT_BF16) MVLPT_PP_V(bf16);
  else return hipErrorInvalidValue;
#undef MVLPT_PP_V
  return hipGetLastError();
}

}  // namespace mvlpt
