/* TEST INFRASTRUCTURE — CPU restatement of the image preprocessing on the input side of the hot path.
 *
 * Reference call sites: Dassl `build_transform` driven by configs/trainers/MVLPT/vit_b16.yaml:8-13
 * (INPUT.TRANSFORMS random_resized_crop / random_flip / normalize, INTERPOLATION bicubic, CLIP PIXEL_MEAN/STD) and
 * trainers/vision_benchmark/evaluation/feature.py:538-553 (Resize(BICUBIC) [+ CenterCrop] + ToTensor + Normalize).
 * Both run torchvision transforms on PIL images, i.e. the arithmetic is Pillow's (third-party, not under
 * /root/reference; this container has Pillow 12.2.0): `Image.crop` + `Image.resize(size, BICUBIC)` =
 * libImaging/Resample.c ImagingResample for 8-bit images, followed by ToTensor (u8 / 255 in fp32) and Normalize
 * ((x - mean) / std in fp32).  The published algorithm restated here:
 *   - per output index xx: scale = in/out, filterscale = max(scale, 1), support = 2 * filterscale (bicubic),
 *     center = (xx + 0.5) * scale, taps xmin = (int)(center - support + 0.5) clamped to 0,
 *     xmax = (int)(center + support + 0.5) clamped to the input size, weights bicubic((x + xmin - center + 0.5) /
 *     filterscale) with a = -0.5, normalised by their sum (double precision), converted to 22-bit fixed point
 *     with round-half-away-from-zero;
 *   - horizontal pass first, then vertical, each accumulating u8 * int32 coefficients from 1 << 21 and clipping
 *     (acc >> 22) to 0..255, with an 8-bit intermediate image; a pass whose input and output sizes are equal is skipped.
 * Pinned by tests/golden/preprocess.npz (outputs of Pillow itself, oracle/make_golden.py `preprocess`).
 * Only tests/, __graft_entry__.smoke() and bench/tools CPU-baseline legs may load this library. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRECISION_BITS (32 - 8 - 2)

static double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

/* coefficients for resampling `in_size` samples to `out_size`; returns ksize; bounds[2*xx] = xmin, [2*xx+1] = count */
static int precompute(int in_size, int out_size, int** bounds_out, int32_t** kk_out) {
  double scale = (double)in_size / out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  int* bounds = (int*)malloc(sizeof(int) * 2 * out_size);
  int32_t* kk = (int32_t*)malloc(sizeof(int32_t) * (size_t)out_size * ksize);
  double* k = (double*)malloc(sizeof(double) * ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < ksize; ++x) {
      double v = 0.0;
      if (x < xmax) v = (ww != 0.0) ? k[x] / ww : k[x];
      kk[(size_t)xx * ksize + x] = v < 0 ? (int32_t)(-0.5 + v * (1 << PRECISION_BITS)) : (int32_t)(0.5 + v * (1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  free(k);
  *bounds_out = bounds;
  *kk_out = kk;
  return ksize;
}

static uint8_t clip8(int32_t v) {
  v >>= PRECISION_BITS;                      /* arithmetic shift, as in Pillow's lookup table */
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* src: uint8 HWC (3 channels, row stride 3*width).  The crop box is resized to (rh, rw) exactly as
 * `img.crop(box).resize((rw, rh), BICUBIC)` does; `out_u8` receives the window [ot, ot+oh) x [ol, ol+ow) of that
 * resized image (HWC).  Returns 0, or -1 on bad arguments. */
int resample_crop_u8(const uint8_t* src, int height, int width, int crop_top, int crop_left, int crop_h, int crop_w,
                     int rh, int rw, int ot, int ol, int oh, int ow, uint8_t* out_u8) {
  if (crop_top < 0 || crop_left < 0 || crop_h <= 0 || crop_w <= 0 || crop_top + crop_h > height || crop_left + crop_w > width ||
      rh <= 0 || rw <= 0 || ot < 0 || ol < 0 || ot + oh > rh || ol + ow > rw)
    return -1;
  const uint8_t* base = src + ((size_t)crop_top * width + crop_left) * 3;
  const size_t stride = (size_t)width * 3;
  /* horizontal pass (all crop rows; Pillow restricts it to the rows the vertical pass reads — same values) */
  uint8_t* tmp = NULL;
  const uint8_t* hsrc = base;
  size_t hstride = stride;
  if (rw != crop_w) {
    int* b; int32_t* kk;
    const int ksize = precompute(crop_w, rw, &b, &kk);
    tmp = (uint8_t*)malloc((size_t)crop_h * rw * 3);
    for (int y = 0; y < crop_h; ++y)
      for (int xx = 0; xx < rw; ++xx) {
        const int xmin = b[2 * xx], n = b[2 * xx + 1];
        const int32_t* k = kk + (size_t)xx * ksize;
        for (int c = 0; c < 3; ++c) {
          int32_t ss = 1 << (PRECISION_BITS - 1);
          for (int x = 0; x < n; ++x) ss += (int32_t)base[y * stride + (size_t)(x + xmin) * 3 + c] * k[x];
          tmp[((size_t)y * rw + xx) * 3 + c] = clip8(ss);
        }
      }
    free(b); free(kk);
    hsrc = tmp;
    hstride = (size_t)rw * 3;
  }
  /* vertical pass */
  if (rh != crop_h) {
    int* b; int32_t* kk;
    const int ksize = precompute(crop_h, rh, &b, &kk);
    for (int yy = ot; yy < ot + oh; ++yy) {
      const int ymin = b[2 * yy], n = b[2 * yy + 1];
      const int32_t* k = kk + (size_t)yy * ksize;
      for (int xx = ol; xx < ol + ow; ++xx)
        for (int c = 0; c < 3; ++c) {
          int32_t ss = 1 << (PRECISION_BITS - 1);
          for (int y = 0; y < n; ++y) ss += (int32_t)hsrc[(size_t)(y + ymin) * hstride + (size_t)xx * 3 + c] * k[y];
          out_u8[((size_t)(yy - ot) * ow + (xx - ol)) * 3 + c] = clip8(ss);
        }
    }
    free(b); free(kk);
  } else {
    for (int yy = ot; yy < ot + oh; ++yy)
      memcpy(out_u8 + (size_t)(yy - ot) * ow * 3, hsrc + (size_t)yy * hstride + (size_t)ol * 3, (size_t)ow * 3);
  }
  free(tmp);
  return 0;
}

/* ToTensor + (optional horizontal flip) + Normalize: out[c][y][x] = (u8 / 255 - mean[c]) / std[c], fp32 CHW */
void to_tensor_normalize(const uint8_t* img_hwc, int h, int w, int flip, const float* mean, const float* std, float* out_chw) {
  for (int c = 0; c < 3; ++c)
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const int sx = flip ? w - 1 - x : x;
        const float v = (float)img_hwc[((size_t)y * w + sx) * 3 + c] / 255.0f;
        out_chw[((size_t)c * h + y) * w + x] = (v - mean[c]) / std[c];
      }
}
