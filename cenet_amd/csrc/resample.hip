// resample.hip — spatial resampling on NCHW planes (HBM-bound; one thread per output element, threads along x).
//   bilinear  (align_corners True/False, explicit coordinate scale)  dseb.py:67-68, cfam.py:217,232, blocks.py:210, out.py:74
//   nearest x2                                                       blocks.py:304
//   adaptive average pool                                            cfam.py:213
//   max pool 2x2 stride 2                                            out.py:43
// Planes are addressed ptr[b*sb + c*HW + p] (channel-slice views).  PyTorch coordinate conventions are restated
// exactly (area_pixel_compute_source_index): align: src = scale*dst ; else src = max(scale*(dst+0.5)-0.5, 0).
#include "common.h"
#include "../../include/cenet_hip.h"

__device__ __forceinline__ void bil_coord(int dst, float scale, int align, int in, int& i0, int& i1, float& l1) {
  float src = align ? scale * dst : fmaxf(scale * (dst + 0.5f) - 0.5f, 0.f);
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - i0;
  if (l1 > 1.f) l1 = 1.f;
}

// grid (B*C, chunks)
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, long sxb, float* __restrict__ y, long syb,
                                                          int C, int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                          int align) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const float* xp = x + (long)b * sxb + (long)c * Hi * Wi;
  float* yp = y + (long)b * syb + (long)c * Ho * Wo;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Ho * Wo; p += gridDim.y * 256) {
    const int oy = p / Wo, ox = p - oy * Wo;
    int y0, y1, x0, x1;
    float ly, lx;
    bil_coord(oy, sh, align, Hi, y0, y1, ly);
    bil_coord(ox, sw, align, Wi, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    yp[p] = hy * (hx * xp[y0 * Wi + x0] + lx * xp[y0 * Wi + x1]) + ly * (hx * xp[y1 * Wi + x0] + lx * xp[y1 * Wi + x1]);
  }
}

// candidate output range [lo, hi) whose taps can touch input index i (conservative; exact weights are recomputed)
__device__ __forceinline__ void bil_range(int i, float scale, int align, int out, int& lo, int& hi) {
  if (scale < 1e-6f) {
    lo = 0;
    hi = out;
    return;
  }
  const float off = align ? 0.f : 0.5f;
  float a = (i - 1 + off) / scale - off, b = (i + 1 + off) / scale - off;
  lo = (int)floorf(a);      // exact bounds are floor(a)+1 .. ceil(b)-1; one index of slack each side for rounding
  hi = (int)ceilf(b) + 1;
  if (lo < 0) lo = 0;
  if (hi > out) hi = out;
}

// gather form of the backward: one thread per INPUT pixel sums the output gradients whose taps touch it
// MAXR = most output columns whose taps can touch one input column (range from bil_range): 6 covers scale >= 0.5 (x2
// up-sampling and every down-sampling), 10 covers x4; anything wider takes the unbounded loop
template <int BIL_MAXR>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, long sgb, float* __restrict__ dx,
                                                          long sdb, int C, int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                          int align) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const float* gp = dy + (long)b * sgb + (long)c * Ho * Wo;
  float* dp = dx + (long)b * sdb + (long)c * Hi * Wi;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Hi * Wi; p += gridDim.y * 256) {
    const int iy = p / Wi, ix = p - iy * Wi;
    int ylo, yhi, xlo, xhi;
    bil_range(iy, sh, align, Ho, ylo, yhi);
    bil_range(ix, sw, align, Wo, xlo, xhi);
    float acc = 0.f;
    if (xhi - xlo <= BIL_MAXR) {
      // separable weights: the x taps of this input column are computed once, not once per output row
      float wxs[BIL_MAXR];
#pragma unroll
      for (int k = 0; k < BIL_MAXR; ++k) {
        wxs[k] = 0.f;
        if (xlo + k < xhi) {
          int x0, x1;
          float lx;
          bil_coord(xlo + k, sw, align, Wi, x0, x1, lx);
          wxs[k] = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        }
      }
      for (int oy = ylo; oy < yhi; ++oy) {
        int y0, y1;
        float ly;
        bil_coord(oy, sh, align, Hi, y0, y1, ly);
        const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
        if (wy == 0.f) continue;
        float row = 0.f;
#pragma unroll
        for (int k = 0; k < BIL_MAXR; ++k)
          if (xlo + k < xhi) row += wxs[k] * gp[oy * Wo + xlo + k];
        acc += wy * row;
      }
    } else {
      for (int oy = ylo; oy < yhi; ++oy) {
        int y0, y1;
        float ly;
        bil_coord(oy, sh, align, Hi, y0, y1, ly);
        const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int ox = xlo; ox < xhi; ++ox) {
          int x0, x1;
          float lx;
          bil_coord(ox, sw, align, Wi, x0, x1, lx);
          const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
          if (wx != 0.f) row += wx * gp[oy * Wo + ox];
        }
        acc += wy * row;
      }
    }
    dp[p] = acc;
  }
}

__global__ __launch_bounds__(256) void nearest2x_fwd_kernel(const float* __restrict__ x, long sxb, float* __restrict__ y, long syb,
                                                           int C, int Hi, int Wi) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const float* xp = x + (long)b * sxb + (long)c * Hi * Wi;
  float* yp = y + (long)b * syb + (long)c * Ho * Wo;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Ho * Wo; p += gridDim.y * 256) {
    const int oy = p / Wo, ox = p - oy * Wo;
    yp[p] = xp[(oy >> 1) * Wi + (ox >> 1)];
  }
}
__global__ __launch_bounds__(256) void nearest2x_bwd_kernel(const float* __restrict__ dy, long sgb, float* __restrict__ dx,
                                                           long sdb, int C, int Hi, int Wi) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const int Wo = 2 * Wi;
  const float* gp = dy + (long)b * sgb + (long)c * 4 * Hi * Wi;
  float* dp = dx + (long)b * sdb + (long)c * Hi * Wi;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Hi * Wi; p += gridDim.y * 256) {
    const int iy = p / Wi, ix = p - iy * Wi;
    const float* g = gp + (2 * iy) * Wo + 2 * ix;
    dp[p] = g[0] + g[1] + g[Wo] + g[Wo + 1];
  }
}

__device__ __forceinline__ int ap_start(int o, int in, int out) { return (o * in) / out; }
__device__ __forceinline__ int ap_end(int o, int in, int out) { return ((o + 1) * in + out - 1) / out; }

__global__ __launch_bounds__(64) void adaptive_avgpool_fwd_kernel(const float* __restrict__ x, long sxb, float* __restrict__ y,
                                                                 long syb, int C, int Hi, int Wi, int Ho, int Wo) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const float* xp = x + (long)b * sxb + (long)c * Hi * Wi;
  float* yp = y + (long)b * syb + (long)c * Ho * Wo;
  for (int p = threadIdx.x; p < Ho * Wo; p += 64) {
    const int oy = p / Wo, ox = p - oy * Wo;
    const int ys = ap_start(oy, Hi, Ho), ye = ap_end(oy, Hi, Ho), xs = ap_start(ox, Wi, Wo), xe = ap_end(ox, Wi, Wo);
    float s = 0.f;
    for (int iy = ys; iy < ye; ++iy)
      for (int ix = xs; ix < xe; ++ix) s += xp[iy * Wi + ix];
    yp[p] = s / (float)((ye - ys) * (xe - xs));
  }
}
__global__ __launch_bounds__(256) void adaptive_avgpool_bwd_kernel(const float* __restrict__ dy, long sgb, float* __restrict__ dx,
                                                                  long sdb, int C, int Hi, int Wi, int Ho, int Wo) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const float* gp = dy + (long)b * sgb + (long)c * Ho * Wo;
  float* dp = dx + (long)b * sdb + (long)c * Hi * Wi;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Hi * Wi; p += gridDim.y * 256) {
    const int iy = p / Wi, ix = p - iy * Wi;
    float s = 0.f;
    for (int oy = 0; oy < Ho; ++oy) {
      const int ys = ap_start(oy, Hi, Ho), ye = ap_end(oy, Hi, Ho);
      if (iy < ys || iy >= ye) continue;
      for (int ox = 0; ox < Wo; ++ox) {
        const int xs = ap_start(ox, Wi, Wo), xe = ap_end(ox, Wi, Wo);
        if (ix < xs || ix >= xe) continue;
        s += gp[oy * Wo + ox] / (float)((ye - ys) * (xe - xs));
      }
    }
    dp[p] = s;
  }
}

// y = scale[c] * maxpool2x2(x)   (out.py:70: self.w * self.rb(x)); scale may be null
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long syb,
                                                          const float* __restrict__ scale, int C, int Hi, int Wi) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const int Ho = Hi / 2, Wo = Wi / 2;
  const float* xp = x + (long)bc * Hi * Wi;
  float* yp = y + (long)b * syb + (long)c * Ho * Wo;
  const float sc = scale ? scale[c] : 1.f;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Ho * Wo; p += gridDim.y * 256) {
    const int oy = p / Wo, ox = p - oy * Wo;
    const float* q = xp + (2 * oy) * Wi + 2 * ox;
    yp[p] = sc * fmaxf(fmaxf(q[0], q[1]), fmaxf(q[Wi], q[Wi + 1]));
  }
}
// dx: gradient to the first maximal element of each window (PyTorch tie rule); dscale[c] += sum dy*maxpool(x)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, long sgb,
                                                          float* __restrict__ dx, const float* __restrict__ scale,
                                                          float* __restrict__ dscale, int C, int Hi, int Wi) {
  __shared__ float red[16];
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const int Ho = Hi / 2, Wo = Wi / 2;
  const float* xp = x + (long)bc * Hi * Wi;
  const float* gp = dy + (long)b * sgb + (long)c * Ho * Wo;
  float* dp = dx + (long)bc * Hi * Wi;
  const float sc = scale ? scale[c] : 1.f;
  float ds = 0.f;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Ho * Wo; p += gridDim.y * 256) {
    const int oy = p / Wo, ox = p - oy * Wo;
    const int base = (2 * oy) * Wi + 2 * ox;
    const int off[4] = {0, 1, Wi, Wi + 1};
    int am = 0;
    float mv = xp[base];
#pragma unroll
    for (int t = 1; t < 4; ++t) {
      float v = xp[base + off[t]];
      if (v > mv) {
        mv = v;
        am = t;
      }
    }
    const float g = gp[p];
    ds += g * mv;
#pragma unroll
    for (int t = 0; t < 4; ++t) dp[base + off[t]] = (t == am) ? g * sc : 0.f;
  }
  if (dscale) {
    ds = block_sum(ds, red);
    if (threadIdx.x == 0) atomicAdd(&dscale[c], ds);
  }
}

static inline int chunks_for(int n) {
  int ch = cdiv(n, 1024);
  return ch > 64 ? 64 : (ch < 1 ? 1 : ch);
}

extern "C" int cenet_bilinear_fwd_f32(const float* x, long sxb, float* y, long syb, int B, int C, int Hi, int Wi, int Ho,
                                      int Wo, float scale_h, float scale_w, int align_corners, hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CENET_EINVAL;
  CENET_LAUNCH(bilinear_fwd_kernel, dim3(B * C, chunks_for(Ho * Wo)), dim3(256), stream, x, sxb, y, syb, C, Hi, Wi, Ho, Wo,
               scale_h, scale_w, align_corners);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_bilinear_bwd_f32(const float* dy, long sgb, float* dx, long sdb, int B, int C, int Hi, int Wi, int Ho,
                                      int Wo, float scale_h, float scale_w, int align_corners, hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CENET_EINVAL;
  if (scale_w >= 0.5f)
    CENET_LAUNCH((bilinear_bwd_kernel<6>), dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi, Ho,
                 Wo, scale_h, scale_w, align_corners);
  else
    CENET_LAUNCH((bilinear_bwd_kernel<10>), dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi, Ho,
                 Wo, scale_h, scale_w, align_corners);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_nearest2x_fwd_f32(const float* x, long sxb, float* y, long syb, int B, int C, int Hi, int Wi,
                                       hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0) return CENET_EINVAL;
  CENET_LAUNCH(nearest2x_fwd_kernel, dim3(B * C, chunks_for(4 * Hi * Wi)), dim3(256), stream, x, sxb, y, syb, C, Hi, Wi);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_nearest2x_bwd_f32(const float* dy, long sgb, float* dx, long sdb, int B, int C, int Hi, int Wi,
                                       hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0) return CENET_EINVAL;
  CENET_LAUNCH(nearest2x_bwd_kernel, dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_adaptive_avgpool_fwd_f32(const float* x, long sxb, float* y, long syb, int B, int C, int Hi, int Wi,
                                              int Ho, int Wo, hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CENET_EINVAL;
  CENET_LAUNCH(adaptive_avgpool_fwd_kernel, dim3(B * C), dim3(64), stream, x, sxb, y, syb, C, Hi, Wi, Ho, Wo);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_adaptive_avgpool_bwd_f32(const float* dy, long sgb, float* dx, long sdb, int B, int C, int Hi, int Wi,
                                              int Ho, int Wo, hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CENET_EINVAL;
  CENET_LAUNCH(adaptive_avgpool_bwd_kernel, dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi,
               Ho, Wo);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_maxpool2_fwd_f32(const float* x, float* y, long syb, const float* scale, int B, int C, int Hi, int Wi,
                                      hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 1 || Wi <= 1 || (Hi & 1) || (Wi & 1)) return CENET_EINVAL;
  CENET_LAUNCH(maxpool2_fwd_kernel, dim3(B * C, chunks_for(Hi * Wi / 4)), dim3(256), stream, x, y, syb, scale, C, Hi, Wi);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_maxpool2_bwd_acc_f32(const float* x, const float* dy, long sgb, float* dx, const float* scale,
                                          float* dscale_acc, int B, int C, int Hi, int Wi, hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 1 || Wi <= 1 || (Hi & 1) || (Wi & 1)) return CENET_EINVAL;
  CENET_LAUNCH(maxpool2_bwd_kernel, dim3(B * C, chunks_for(Hi * Wi / 4)), dim3(256), stream, x, dy, sgb, dx, scale, dscale_acc,
               C, Hi, Wi);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
