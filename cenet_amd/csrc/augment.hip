// augment.hip — the ACDC training-time augmentation on the device (src/datasets/dataset_acdc.py:15-48).
//
// The reference resamples every training slice on the host with numpy / scipy, one sample at a time, in the DataLoader's process
// (30 samples/s measured, against ~1 700 images/s for the step).  Here the whole training set lives in HBM (1 312 slices of
// ~256 x 216 fp32 + uint8 labels: 0.35 GB of 288) and a batch is gathered AND augmented by three launches; the host only draws
// the random numbers, in the reference's order (cenet_amd/data.py DeviceAugmenter.draw).
//
//   stage 1  aug_geom        quarter turns + flip (np.rot90 / np.flip, :15-22: pure index maps) or the whole-degree rotation
//                            (ndimage.rotate(order=0, reshape=False), :25-29: nearest gather, zero outside) -> a staged image
//                            (fp64) and label (uint8) of the sample's own size
//   stage 2  aug_prefilter   scipy's cubic B-spline prefilter (spline_filter, mirror boundary), axis 0 then axis 1, in place
//   stage 3  aug_zoom        zoom(order=3) for the image, zoom(order=0) for the label (:43-44), or a copy when the staged size
//                            already is the output size
//
// Everything is computed in fp64 with scipy's own operation order and WITHOUT fused multiply-adds (the pragma below), from
// matrices / offsets / zoom factors / pole powers the host computed with numpy — so labels are bit-identical to the reference's
// and images agree to the last float32 bit in practice (tests/test_augment.py holds them to 1e-6 and counts exact matches).
#include "common.h"
#include "../../include/cenet_hip.h"

#pragma clang fp contract(off)

#define AUG_TAB 8   // per sample: offset (elements into the pools), H, W, mode, k, axis, zoom flag, unused
#define AUG_DP 10   // per sample: m00 m01 m10 m11 off0 off1 | pole^(Ha-1) pole^(Wa-1) | zoom_y zoom_x
#define AUG_POLE (-0.26794919243112270647)  // sqrt(3) - 2

__device__ __forceinline__ void aug_staged_size(const long* t, int& Ha, int& Wa) {
  const int H = (int)t[1], W = (int)t[2];
  const bool swap = t[3] == 1 && (t[4] & 1);
  Ha = swap ? W : H;
  Wa = swap ? H : W;
}

// grid (chunks, B)
__global__ __launch_bounds__(256) void aug_geom_kernel(const float* __restrict__ pool_img, const unsigned char* __restrict__ pool_lab,
                                                      const long* __restrict__ tab, const double* __restrict__ dp,
                                                      double* __restrict__ stage_img, unsigned char* __restrict__ stage_lab,
                                                      long stride) {
  const int b = blockIdx.y;
  const long* t = tab + (long)b * AUG_TAB;
  const int H = (int)t[1], W = (int)t[2], mode = (int)t[3], k = (int)t[4], axis = (int)t[5];
  int Ha, Wa;
  aug_staged_size(t, Ha, Wa);
  const float* src = pool_img + t[0];
  const unsigned char* srl = pool_lab + t[0];
  double* di = stage_img + (long)b * stride;
  unsigned char* dl = stage_lab + (long)b * stride;
  const double* m = dp + (long)b * AUG_DP;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < Ha * Wa; p += gridDim.x * 256) {
    int i = p / Wa, j = p - i * Wa;
    int si = i, sj = j;
    bool inside = true;
    if (mode == 1) {
      // out = flip(rot90(a, k), axis): undo the flip, then the quarter turns
      if (axis == 0) i = Ha - 1 - i;
      else j = Wa - 1 - j;
      switch (k & 3) {
        case 0: si = i, sj = j; break;
        case 1: si = j, sj = W - 1 - i; break;
        case 2: si = H - 1 - i, sj = W - 1 - j; break;
        default: si = H - 1 - j, sj = i; break;
      }
    } else if (mode == 2) {
      // scipy NI_GeometricTransform: coordinate = sum over output axes of o * matrix, then + shift; outside [0, len - 1] -> cval;
      // order 0 samples floor(c + 0.5)
      double cy = 0.0, cx = 0.0;
      cy += (double)i * m[0];
      cy += (double)j * m[1];
      cy += m[4];
      cx += (double)i * m[2];
      cx += (double)j * m[3];
      cx += m[5];
      inside = !(cy < 0.0 || cy > (double)(H - 1) || cx < 0.0 || cx > (double)(W - 1));
      si = (int)floor(cy + 0.5);
      sj = (int)floor(cx + 0.5);
    }
    di[p] = inside ? (double)src[(long)si * W + sj] : 0.0;
    dl[p] = inside ? srl[(long)si * W + sj] : (unsigned char)0;
  }
}

// one line of scipy's spline_filter1d(order=3), mirror boundary: c[0 .. n) at stride s, zn = pole^(n-1)
__device__ __forceinline__ void aug_prefilter_line(double* c, int n, long s, double zn) {
  if (n < 2) return;
  const double z = AUG_POLE;
  const double gain = (1.0 - z) * (1.0 - 1.0 / z);
  for (int i = 0; i < n; ++i) c[i * s] *= gain;
  double zi = z;
  double c0 = c[0] + zn * c[(n - 1) * s];
  for (int i = 1; i < n - 1; ++i) {
    c0 += (zi + zn * zn / zi) * c[i * s];
    zi *= z;
  }
  c[0] = c0 / (1.0 - zn * zn);
  for (int i = 1; i < n; ++i) c[i * s] += z * c[(i - 1) * s];
  c[(n - 1) * s] = (z * c[(n - 2) * s] + c[(n - 1) * s]) * z / (z * z - 1.0);
  for (int i = n - 2; i >= 0; --i) c[i * s] = z * (c[(i + 1) * s] - c[i * s]);
}

// AXIS 0: thread = one column (lines run down the rows, neighbouring threads read neighbouring addresses); AXIS 1: thread = one
// row.  grid (chunks of lines, B).  Samples that are not resized skip the filter.
template <int AXIS>
__global__ __launch_bounds__(64) void aug_prefilter_kernel(const long* __restrict__ tab, const double* __restrict__ dp,
                                                          double* __restrict__ stage_img, long stride) {
  const int b = blockIdx.y;
  const long* t = tab + (long)b * AUG_TAB;
  if (!t[6]) return;
  int Ha, Wa;
  aug_staged_size(t, Ha, Wa);
  const int line = blockIdx.x * 64 + threadIdx.x;
  double* c = stage_img + (long)b * stride;
  if (AXIS == 0) {
    if (line < Wa) aug_prefilter_line(c + line, Ha, Wa, dp[(long)b * AUG_DP + 6]);
  } else {
    if (line < Ha) aug_prefilter_line(c + (long)line * Wa, Wa, 1, dp[(long)b * AUG_DP + 7]);
  }
}

__device__ __forceinline__ void aug_w3(double t, double* w) {  // cubic B-spline weights of the taps floor - 1 .. floor + 2
  w[1] = (t * t * (t - 2.0) * 3.0 + 4.0) / 6.0;
  const double z = 1.0 - t;
  w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
  w[0] = z * z * z / 6.0;
  w[3] = 1.0 - w[0] - w[1] - w[2];
}
__device__ __forceinline__ int aug_mirror(int idx, int n) {  // scipy NI_ZoomShift tap mapping (all non-grid modes)
  if (n <= 1) return 0;
  const int s2 = 2 * n - 2;
  if (idx < 0) {
    idx = s2 * (-idx / s2) + idx;
    idx = idx <= 1 - n ? idx + s2 : -idx;
  } else if (idx >= n) {
    idx -= s2 * (idx / s2);
    if (idx >= n) idx = s2 - idx;
  }
  return idx;
}

// grid (chunks, B)
__global__ __launch_bounds__(256) void aug_zoom_kernel(const long* __restrict__ tab, const double* __restrict__ dp,
                                                      const double* __restrict__ stage_img,
                                                      const unsigned char* __restrict__ stage_lab, long stride,
                                                      float* __restrict__ out_img, float* __restrict__ out_lab, int OH, int OW) {
  const int b = blockIdx.y;
  const long* t = tab + (long)b * AUG_TAB;
  int Ha, Wa;
  aug_staged_size(t, Ha, Wa);
  const double* c = stage_img + (long)b * stride;
  const unsigned char* l = stage_lab + (long)b * stride;
  const double zy = dp[(long)b * AUG_DP + 8], zx = dp[(long)b * AUG_DP + 9];
  float* oi = out_img + (long)b * OH * OW;
  float* ol = out_lab + (long)b * OH * OW;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < OH * OW; p += gridDim.x * 256) {
    const int oy = p / OW, ox = p - oy * OW;
    if (!t[6]) {  // (Ha, Wa) == (OH, OW)
      oi[p] = (float)c[p];
      ol[p] = (float)l[p];
      continue;
    }
    const double cy = (double)oy * zy, cx = (double)ox * zx;
    // scipy map_coordinate(mode='constant'): a coordinate outside [0, len - 1] — by one ulp is enough, and (OH - 1)·((Ha - 1)/(OH - 1))
    // does round above Ha - 1 for some sizes (Ha = 58, 63, 232, 248 ... at OH = 224) — gives cval = 0, image and label alike
    if (cy > (double)(Ha - 1) || cx > (double)(Wa - 1)) {
      oi[p] = 0.f;
      ol[p] = 0.f;
      continue;
    }
    const int fy = (int)floor(cy), fx = (int)floor(cx);
    double wy[4], wx[4];
    aug_w3(cy - (double)fy, wy);
    aug_w3(cx - (double)fx, wx);
    int yy[4], xx[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      yy[a] = aug_mirror(fy - 1 + a, Ha);
      xx[a] = aug_mirror(fx - 1 + a, Wa);
    }
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += c[(long)yy[a] * Wa + xx[e]] * wy[a] * wx[e];
    oi[p] = (float)acc;
    ol[p] = (float)l[(long)(int)floor(cy + 0.5) * Wa + (int)floor(cx + 0.5)];
  }
}

extern "C" int cenet_augment_acdc(const float* pool_img, const unsigned char* pool_lab, const long* tab, const double* dp,
                                  double* stage_img, unsigned char* stage_lab, long stage_stride, int max_h, int max_w,
                                  float* out_img, float* out_lab, int B, int OH, int OW, hipStream_t stream) {
  if (!pool_img || !pool_lab || !tab || !dp || !stage_img || !stage_lab || !out_img || !out_lab) return CENET_EINVAL;
  if (B <= 0 || OH <= 1 || OW <= 1 || max_h <= 0 || max_w <= 0 || stage_stride < (long)max_h * max_w) return CENET_EINVAL;
  const int big = max_h > max_w ? max_h : max_w;
  CENET_LAUNCH(aug_geom_kernel, dim3(cdiv((long)max_h * max_w, 256 * 4), B), dim3(256), stream, pool_img, pool_lab, tab, dp, stage_img,
               stage_lab, stage_stride);
  CENET_LAUNCH(aug_prefilter_kernel<0>, dim3(cdiv(big, 64), B), dim3(64), stream, tab, dp, stage_img, stage_stride);
  CENET_LAUNCH(aug_prefilter_kernel<1>, dim3(cdiv(big, 64), B), dim3(64), stream, tab, dp, stage_img, stage_stride);
  CENET_LAUNCH(aug_zoom_kernel, dim3(cdiv((long)OH * OW, 256 * 4), B), dim3(256), stream, tab, dp, (const double*)stage_img,
               (const unsigned char*)stage_lab, stage_stride, out_img, out_lab, OH, OW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
