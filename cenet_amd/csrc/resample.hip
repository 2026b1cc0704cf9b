// resample.hip — spatial resampling on NCHW planes (HBM-bound; one thread per output element, threads along x).
//   bilinear  (align_corners True/False, explicit coordinate scale)  dseb.py:67-68, cfam.py:217,232, blocks.py:210, out.py:74
//   nearest x2                                                       blocks.py:304
//   adaptive average pool                                            cfam.py:213
//   max pool 2x2 stride 2                                            out.py:43
// Planes are addressed ptr[b*sb + c*HW + p] (channel-slice views).  PyTorch coordinate conventions are restated
// exactly (area_pixel_compute_source_index): align: src = scale*dst ; else src = max(scale*(dst+0.5)-0.5, 0).
// Templates over the activation storage type T (float / bf16_t) and, for the large maps, over OPT = outputs per thread
// (2 adjacent x positions for bf16 when the output width is even, so that a lane stores 4 bytes).
#include "common.h"
#include <cstdlib>
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
template <typename T, int OPT>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, long sxb, T* __restrict__ y, long syb,
                                                          int C, int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                          int align) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const T* xp = x + (long)b * sxb + (long)c * Hi * Wi;
  T* yp = y + (long)b * syb + (long)c * Ho * Wo;
  for (int pp = blockIdx.y * 256 + threadIdx.x; pp < Ho * Wo / OPT; pp += gridDim.y * 256) {
    const int p = pp * OPT;
    const int oy = p / Wo, ox0 = p - oy * Wo;
    int y0, y1;
    float ly;
    bil_coord(oy, sh, align, Hi, y0, y1, ly);
    const float hy = 1.f - ly;
    float o[OPT];
#pragma unroll
    for (int e = 0; e < OPT; ++e) {
      int x0, x1;
      float lx;
      bil_coord(ox0 + e, sw, align, Wi, x0, x1, lx);
      const float hx = 1.f - lx;
      o[e] = hy * (hx * ldf(xp + y0 * Wi + x0) + lx * ldf(xp + y0 * Wi + x1)) +
             ly * (hx * ldf(xp + y1 * Wi + x0) + lx * ldf(xp + y1 * Wi + x1));
    }
    stv<OPT>(yp + p, o);
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
template <typename T, int BIL_MAXR>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ dy, long sgb, T* __restrict__ dx,
                                                          long sdb, int C, int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                          int align, const T* __restrict__ dx_add) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const T* gp = dy + (long)b * sgb + (long)c * Ho * Wo;
  T* dp = dx + (long)b * sdb + (long)c * Hi * Wi;
  const T* ap = dx_add ? dx_add + (long)b * sdb + (long)c * Hi * Wi : nullptr;  // (laid out like dx)
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
          if (xlo + k < xhi) row += wxs[k] * ldf(gp + oy * Wo + xlo + k);
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
          if (wx != 0.f) row += wx * ldf(gp + oy * Wo + ox);
        }
        acc += wy * row;
      }
    }
    stf(dp + p, ap ? acc + ldf(ap + p) : acc);
  }
}

// one thread per INPUT pixel: writes its 2x2 output block (two 2-element row stores)
template <typename T>
__global__ __launch_bounds__(256) void nearest2x_fwd_kernel(const T* __restrict__ x, long sxb, T* __restrict__ y, long syb,
                                                           int C, int Hi, int Wi) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const int Wo = 2 * Wi;
  const T* xp = x + (long)b * sxb + (long)c * Hi * Wi;
  T* yp = y + (long)b * syb + (long)c * 4 * Hi * Wi;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Hi * Wi; p += gridDim.y * 256) {
    const int iy = p / Wi, ix = p - iy * Wi;
    const T v = xp[p];
    const T vv[2] = {v, v};
    T* o = yp + (2 * iy) * Wo + 2 * ix;  // even element offset inside an even-sized plane: 2-element aligned when the base is
    memcpy(o, vv, 2 * sizeof(T));
    memcpy(o + Wo, vv, 2 * sizeof(T));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void nearest2x_bwd_kernel(const T* __restrict__ dy, long sgb, T* __restrict__ dx,
                                                           long sdb, int C, int Hi, int Wi) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const int Wo = 2 * Wi;
  const T* gp = dy + (long)b * sgb + (long)c * 4 * Hi * Wi;
  T* dp = dx + (long)b * sdb + (long)c * Hi * Wi;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Hi * Wi; p += gridDim.y * 256) {
    const int iy = p / Wi, ix = p - iy * Wi;
    const T* g = gp + (2 * iy) * Wo + 2 * ix;
    float r0[2], r1[2];
    ldv<2>(r0, g);
    ldv<2>(r1, g + Wo);
    stf(dp + p, r0[0] + r0[1] + r1[0] + r1[1]);
  }
}

__device__ __forceinline__ int ap_start(int o, int in, int out) { return (o * in) / out; }
__device__ __forceinline__ int ap_end(int o, int in, int out) { return ((o + 1) * in + out - 1) / out; }

template <typename T>
__global__ __launch_bounds__(64) void adaptive_avgpool_fwd_kernel(const T* __restrict__ x, long sxb, T* __restrict__ y,
                                                                 long syb, int C, int Hi, int Wi, int Ho, int Wo) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const T* xp = x + (long)b * sxb + (long)c * Hi * Wi;
  T* yp = y + (long)b * syb + (long)c * Ho * Wo;
  for (int p = threadIdx.x; p < Ho * Wo; p += 64) {
    const int oy = p / Wo, ox = p - oy * Wo;
    const int ys = ap_start(oy, Hi, Ho), ye = ap_end(oy, Hi, Ho), xs = ap_start(ox, Wi, Wo), xe = ap_end(ox, Wi, Wo);
    float s = 0.f;
    for (int iy = ys; iy < ye; ++iy)
      for (int ix = xs; ix < xe; ++ix) s += ldf(xp + iy * Wi + ix);
    stf(yp + p, s / (float)((ye - ys) * (xe - xs)));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void adaptive_avgpool_bwd_kernel(const T* __restrict__ dy, long sgb, T* __restrict__ dx,
                                                                  long sdb, int C, int Hi, int Wi, int Ho, int Wo) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const T* gp = dy + (long)b * sgb + (long)c * Ho * Wo;
  T* dp = dx + (long)b * sdb + (long)c * Hi * Wi;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Hi * Wi; p += gridDim.y * 256) {
    const int iy = p / Wi, ix = p - iy * Wi;
    float s = 0.f;
    // pooling DOWN (Hi >= Ho): the windows that hold row iy are among the bin floor(iy Ho / Hi) and its two neighbours (windows
    // overlap by at most one row) — 3 x 3 candidates instead of all Ho x Wo bins with four integer divisions each
    int oy_lo = 0, oy_hi = Ho, ox_lo = 0, ox_hi = Wo;
    if (Hi >= Ho) {
      const int o = (int)(((long)iy * Ho) / Hi);
      oy_lo = o > 0 ? o - 1 : 0;
      oy_hi = o + 2 < Ho ? o + 2 : Ho;
    }
    if (Wi >= Wo) {
      const int o = (int)(((long)ix * Wo) / Wi);
      ox_lo = o > 0 ? o - 1 : 0;
      ox_hi = o + 2 < Wo ? o + 2 : Wo;
    }
    for (int oy = oy_lo; oy < oy_hi; ++oy) {
      const int ys = ap_start(oy, Hi, Ho), ye = ap_end(oy, Hi, Ho);
      if (iy < ys || iy >= ye) continue;
      for (int ox = ox_lo; ox < ox_hi; ++ox) {
        const int xs = ap_start(ox, Wi, Wo), xe = ap_end(ox, Wi, Wo);
        if (ix < xs || ix >= xe) continue;
        s += ldf(gp + oy * Wo + ox) / (float)((ye - ys) * (xe - xs));
      }
    }
    stf(dp + p, s);
  }
}

// y = scale[c] * maxpool2x2(x)   (out.py:70: self.w * self.rb(x)); scale may be null
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long syb,
                                                          const float* __restrict__ scale, int C, int Hi, int Wi) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const int Ho = Hi / 2, Wo = Wi / 2;
  const T* xp = x + (long)bc * Hi * Wi;
  T* yp = y + (long)b * syb + (long)c * Ho * Wo;
  const float sc = scale ? scale[c] : 1.f;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Ho * Wo; p += gridDim.y * 256) {
    const int oy = p / Wo, ox = p - oy * Wo;
    const T* q = xp + (2 * oy) * Wi + 2 * ox;
    float r0[2], r1[2];
    ldv<2>(r0, q);
    ldv<2>(r1, q + Wi);
    stf(yp + p, sc * fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r1[0], r1[1])));
  }
}
// dx: gradient to the first maximal element of each window (PyTorch tie rule); dscale[c] += sum dy*maxpool(x)
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, long sgb,
                                                          T* __restrict__ dx, const float* __restrict__ scale,
                                                          float* __restrict__ dscale, int C, int Hi, int Wi) {
  __shared__ float red[16];
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const int Ho = Hi / 2, Wo = Wi / 2;
  const T* xp = x + (long)bc * Hi * Wi;
  const T* gp = dy + (long)b * sgb + (long)c * Ho * Wo;
  T* dp = dx + (long)bc * Hi * Wi;
  const float sc = scale ? scale[c] : 1.f;
  float ds = 0.f;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Ho * Wo; p += gridDim.y * 256) {
    const int oy = p / Wo, ox = p - oy * Wo;
    const int base = (2 * oy) * Wi + 2 * ox;
    float w4[4];
    ldv<2>(w4, xp + base);
    ldv<2>(w4 + 2, xp + base + Wi);
    int am = 0;
    float mv = w4[0];
#pragma unroll
    for (int t = 1; t < 4; ++t) {
      if (w4[t] > mv) {
        mv = w4[t];
        am = t;
      }
    }
    const float g = ldf(gp + p);
    ds += g * mv;
#pragma unroll
    for (int t = 0; t < 4; ++t) w4[t] = (t == am) ? g * sc : 0.f;
    stv<2>(dp + base, w4);
    stv<2>(dp + base + Wi, w4 + 2);
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
// one element per thread — measured per kernel: it helps the x2 align-corners backward (50 -> 41 us: four 8-byte row reads per
// iteration were serialised by the run-time trip count) and hurts the forward / x0.5 kernels (37 -> 45 us: four times the workgroups)
static inline int chunks_one(int n) {
  int ch = cdiv(n, 256);
  return ch > 1024 ? 1024 : (ch < 1 ? 1 : ch);
}

template <typename T>
static int bilinear_fwd_impl(const T* x, long sxb, T* y, long syb, int B, int C, int Hi, int Wi, int Ho, int Wo, float scale_h,
                             float scale_w, int align_corners, hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CENET_EINVAL;
  if (sizeof(T) == 2 && (Wo & 1) == 0 && (syb & 1) == 0 && ((uintptr_t)y & 3) == 0)
    CENET_LAUNCH((bilinear_fwd_kernel<T, 2>), dim3(B * C, chunks_for(Ho * Wo / 2)), dim3(256), stream, x, sxb, y, syb, C, Hi, Wi, Ho,
                 Wo, scale_h, scale_w, align_corners);
  else
    CENET_LAUNCH((bilinear_fwd_kernel<T, 1>), dim3(B * C, chunks_for(Ho * Wo)), dim3(256), stream, x, sxb, y, syb, C, Hi, Wi, Ho, Wo,
                 scale_h, scale_w, align_corners);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(bilinear_fwd, (const T* x, long sxb, T* y, long syb, int B, int C, int Hi, int Wi, int Ho, int Wo, float scale_h,
                          float scale_w, int align_corners, hipStream_t stream),
           (x, sxb, y, syb, B, C, Hi, Wi, Ho, Wo, scale_h, scale_w, align_corners, stream))

// ---- exact x2 / x0.5 resizes with align_corners = False (the FEA scale 0.5 of the ACDC preset, dseb.py:27-30, and the final
// x2 of the output head): fixed tap patterns instead of the general gather with its per-candidate coordinate arithmetic ----
// x2 up-sampling: output 2i reads (i-1: .25, i: .75), output 2i+1 reads (i: .75, i+1: .25), indices clamped at the borders, so
//   dx[i] = .25 g[2i-1] + .75 g[2i] + .75 g[2i+1] + .25 g[2i+2]   with g[0] (g[2n-1]) counted in full at i = 0 (i = n-1)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_up2_bwd_kernel(const T* __restrict__ dy, long sgb, T* __restrict__ dx, long sdb,
                                                              int C, int Hi, int Wi) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const int Wo = 2 * Wi;
  const T* gp = dy + (long)b * sgb + (long)c * 4 * Hi * Wi;
  T* dp = dx + (long)b * sdb + (long)c * Hi * Wi;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Hi * Wi; p += gridDim.y * 256) {
    const int iy = p / Wi, ix = p - iy * Wi;
    float wy[4] = {0.25f, 0.75f, 0.75f, 0.25f}, wx[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    if (iy == 0) wy[0] = 0.f, wy[1] = 1.f;
    if (iy == Hi - 1) wy[3] = 0.f, wy[2] = 1.f;
    if (ix == 0) wx[0] = 0.f, wx[1] = 1.f;
    if (ix == Wi - 1) wx[3] = 0.f, wx[2] = 1.f;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int oy = 2 * iy - 1 + r;
      if (wy[r] == 0.f) continue;
      const T* row = gp + (long)oy * Wo + 2 * ix - 1;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (wx[k] != 0.f) s += wx[k] * ldf(row + k);
      acc += wy[r] * s;
    }
    stf(dp + p, acc);
  }
}
// x0.5 down-sampling: output o reads (2o: .5, 2o+1: .5), so dx[i] = .25 g[i/2][j/2]
template <typename T>
__global__ __launch_bounds__(256) void bilinear_down2_bwd_kernel(const T* __restrict__ dy, long sgb, T* __restrict__ dx, long sdb,
                                                                int C, int Ho, int Wo, const T* __restrict__ dx_add) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const T* gp = dy + (long)b * sgb + (long)c * Ho * Wo;
  T* dp = dx + (long)b * sdb + (long)c * 4 * Ho * Wo;
  const T* ap = dx_add ? dx_add + (long)b * sdb + (long)c * 4 * Ho * Wo : nullptr;  // (laid out like dx)
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Ho * Wo; p += gridDim.y * 256) {
    const int oy = p / Wo, ox = p - oy * Wo;
    const float v = 0.25f * ldf(gp + p);
    const long off = (long)(2 * oy) * (2 * Wo) + 2 * ox;
    T* o = dp + off;
    float q[4] = {v, v, v, v};
    if (ap) {
      float t[4];
      ldv<2>(t, ap + off);
      ldv<2>(t + 2, ap + off + 2 * Wo);
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] += t[e];
    }
    stv<2>(o, q);
    stv<2>(o + 2 * Wo, q + 2);
  }
}

// Strong up-sampling backward (bf16; the 7x7 -> 49x49 pooled branch of cfam.py:231-236): the gather form above gives one thread
// per input pixel — 49 busy threads per plane, each walking ~250 candidate outputs (70-80 us for 0.3 M elements).  Here a
// workgroup owns a plane and its threads take (input pixel, candidate output ROW) pairs: the row's taps are summed with the
// separable x weights, weighted by wy and added to the pixel's fp32 bin in LDS (~16 adds per bin).
#define BIL_LDS_MAX 1024
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_rows_kernel(const T* __restrict__ dy, long sgb, T* __restrict__ dx, long sdb,
                                                               int C, int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                               int align, int maxrows) {
  __shared__ float bins[BIL_LDS_MAX];
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const T* gp = dy + (long)b * sgb + (long)c * Ho * Wo;
  T* dp = dx + (long)b * sdb + (long)c * Hi * Wi;
  for (int i = threadIdx.x; i < Hi * Wi; i += 256) bins[i] = 0.f;
  __syncthreads();
  for (int it = threadIdx.x; it < Hi * Wi * maxrows; it += 256) {
    const int i = it / maxrows, rr = it - i * maxrows;
    const int iy = i / Wi, ix = i - iy * Wi;
    int ylo, yhi, xlo, xhi;
    bil_range(iy, sh, align, Ho, ylo, yhi);
    const int oy = ylo + rr;
    if (oy >= yhi) continue;
    int y0, y1;
    float ly;
    bil_coord(oy, sh, align, Hi, y0, y1, ly);
    const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
    if (wy == 0.f) continue;
    bil_range(ix, sw, align, Wo, xlo, xhi);
    float row = 0.f;
    for (int ox = xlo; ox < xhi; ++ox) {
      int x0, x1;
      float lx;
      bil_coord(ox, sw, align, Wi, x0, x1, lx);
      const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
      if (wx != 0.f) row += wx * ldf(gp + oy * Wo + ox);
    }
    atomicAdd(&bins[i], wy * row);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Hi * Wi; i += 256) stf(dp + i, bins[i]);
}

// The same problem class when the whole plane pair is small (Hi, Wi <= 8, Ho, Wo <= 64: the 7 x 7 -> 49 x 49 pooled branch): bilinear
// resampling is separable, dx = Wy^T dy Wx with Wy [Ho][Hi], Wx [Wo][Wi] (two non-zeros per row), so the backward of a plane is two
// small dense products through LDS — 49 x 49 x 7 + 49 x 7 x 7 multiply-adds and 2 x 49 coordinate evaluations per PLANE instead of
// ~16 coordinate evaluations per (pixel, candidate row) item: 19 -> ~5 us per call.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_sep_kernel(const T* __restrict__ dy, long sgb, T* __restrict__ dx, long sdb,
                                                              int C, int Hi, int Wi, int Ho, int Wo, float sh, float sw, int align) {
  __shared__ float G[64 * 64], Tm[64 * 8], wxs[64 * 8], wys[64 * 8];
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const T* gp = dy + (long)b * sgb + (long)c * Ho * Wo;
  T* dp = dx + (long)b * sdb + (long)c * Hi * Wi;
  for (int i = threadIdx.x; i < Ho * Wo; i += 256) G[i] = ldf(gp + i);
  for (int i = threadIdx.x; i < 64 * 8; i += 256) wxs[i] = wys[i] = 0.f;
  __syncthreads();
  if (threadIdx.x < Wo) {
    int x0, x1;
    float lx;
    bil_coord(threadIdx.x, sw, align, Wi, x0, x1, lx);
    wxs[threadIdx.x * 8 + x0] += 1.f - lx;  // (x0 == x1 at the border: both weights land on the same input)
    wxs[threadIdx.x * 8 + x1] += lx;
  } else if (threadIdx.x >= 64 && threadIdx.x - 64 < Ho) {
    const int oy = threadIdx.x - 64;
    int y0, y1;
    float ly;
    bil_coord(oy, sh, align, Hi, y0, y1, ly);
    wys[oy * 8 + y0] += 1.f - ly;
    wys[oy * 8 + y1] += ly;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Ho * Wi; i += 256) {  // Tm[oy][ix] = sum_ox dy[oy][ox] Wx[ox][ix]
    const int oy = i / Wi, ix = i - oy * Wi;
    float t = 0.f;
    for (int ox = 0; ox < Wo; ++ox) t += G[oy * Wo + ox] * wxs[ox * 8 + ix];
    Tm[oy * 8 + ix] = t;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Hi * Wi; i += 256) {  // dx[iy][ix] = sum_oy Wy[oy][iy] Tm[oy][ix]
    const int iy = i / Wi, ix = i - iy * Wi;
    float t = 0.f;
    for (int oy = 0; oy < Ho; ++oy) t += wys[oy * 8 + iy] * Tm[oy * 8 + ix];
    stf(dp + i, t);
  }
}

// Up-sampling by two with align_corners (nn.UpsamplingBilinear2d(scale_factor=2): the head's UpConv, blocks.py:210, 56 -> 112),
// backward, bf16: the gather form walks ~5 x 7 candidate outputs per input pixel through predicated 2-byte loads (113 us for a 51 MB
// gradient).  Here a thread owns FOUR consecutive input pixels of a row: input i is touched by outputs 2i - 2 .. 2i + 3 only (source
// coordinate o (Hi - 1) / (Ho - 1), slightly below o / 2), so the thread needs output rows 2 iy - 2 .. 2 iy + 3 and the 16 columns
// 2 ix - 4 .. 2 ix + 11, four 8-byte loads per row; the weights are PyTorch's (bil_coord), zero where an output does not touch.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_up2ac_bwd_kernel(const T* __restrict__ dy, long sgb, T* __restrict__ dx, long sdb,
                                                                int C, int Hi, int Wi, float sh, float sw) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const int Ho = 2 * Hi, Wo = 2 * Wi, Wq = Wi / 4;
  const T* gp = dy + (long)b * sgb + (long)c * Ho * Wo;
  T* dp = dx + (long)b * sdb + (long)c * Hi * Wi;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < Hi * Wq; p += gridDim.y * 256) {
    const int iy = p / Wq, ix = 4 * (p - iy * Wq);
    const int c0 = 2 * ix - 4;  // first column of the 16-column window (a multiple of 4: 8-byte aligned rows)
    // column weights: wx[e][k] = weight of output column c0 + k for input ix + e
    float wx[4][16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ox = c0 + k;
      int x0 = -9, x1 = -9;
      float lx = 0.f;
      if (ox >= 0 && ox < Wo) bil_coord(ox, sw, 1, Wi, x0, x1, lx);
#pragma unroll
      for (int e = 0; e < 4; ++e) wx[e][k] = (x0 == ix + e ? 1.f - lx : 0.f) + (x1 == ix + e ? lx : 0.f);
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int oy = 2 * iy - 2 + r;
      if (oy < 0 || oy >= Ho) continue;
      int y0, y1;
      float ly;
      bil_coord(oy, sh, 1, Hi, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      float g[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ox = c0 + 4 * q;
        if (ox >= 0 && ox + 3 < Wo) {
          ldv<4>(g + 4 * q, gp + (long)oy * Wo + ox);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) g[4 * q + t] = 0.f;  // (Wo % 4 == 0 and ox % 4 == 0: a quad is inside or outside as a whole)
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float row = 0.f;
#pragma unroll
        for (int k = 2 * e + 2; k < 2 * e + 8; ++k) row += wx[e][k] * g[k];  // outputs 2 (ix + e) - 2 .. + 3
        acc[e] += wy * row;
      }
    }
    stv<4>(dp + (long)iy * Wi + ix, acc);
  }
}

template <typename T>
static int bilinear_bwd_impl(const T* dy, long sgb, T* dx, long sdb, int B, int C, int Hi, int Wi, int Ho, int Wo, float scale_h,
                             float scale_w, int align_corners, hipStream_t stream, const T* dx_add = nullptr) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CENET_EINVAL;
  if (dx_add && sizeof(T) == 2 && !align_corners && Hi == 2 * Ho && Wi == 2 * Wo && scale_h == 2.0f && scale_w == 2.0f &&
      ((((uintptr_t)dx | (uintptr_t)dx_add) & 3) == 0) && (sdb & 1) == 0) {
    CENET_LAUNCH((bilinear_down2_bwd_kernel<T>), dim3(B * C, chunks_for(Ho * Wo)), dim3(256), stream, dy, sgb, dx, sdb, C, Ho, Wo,
                 dx_add);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (dx_add) {  // with an addend: the general kernel (the other specialised forms have no such operand)
    if (scale_w >= 0.5f)
      CENET_LAUNCH((bilinear_bwd_kernel<T, 6>), dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi, Ho,
                   Wo, scale_h, scale_w, align_corners, dx_add);
    else
      CENET_LAUNCH((bilinear_bwd_kernel<T, 10>), dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi, Ho,
                   Wo, scale_h, scale_w, align_corners, dx_add);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  // (bf16 tensors only: the fp32 parity mode keeps the one summation order of the general kernel)
  if (sizeof(T) == 2 && !align_corners && Ho == 2 * Hi && Wo == 2 * Wi && scale_h == 0.5f && scale_w == 0.5f && Hi > 1 && Wi > 1) {
    CENET_LAUNCH((bilinear_up2_bwd_kernel<T>), dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (sizeof(T) == 2 && Hi <= 8 && Wi <= 8 && Ho <= 64 && Wo <= 64 && scale_h > 1e-6f && scale_w > 1e-6f && !getenv("CENET_BIL_NO_SEP")) {
    CENET_LAUNCH((bilinear_bwd_sep_kernel<T>), dim3(B * C), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi, Ho, Wo, scale_h, scale_w,
                 align_corners);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (sizeof(T) == 2 && Hi * Wi <= BIL_LDS_MAX && scale_h > 1e-6f && scale_h <= 0.3f && scale_w > 1e-6f && scale_w <= 0.3f) {
    const int maxrows = (int)(2.f / scale_h) + 4;  // bil_range spans (i-1 .. i+1) / scale plus a slack index each side
    CENET_LAUNCH((bilinear_bwd_rows_kernel<T>), dim3(B * C), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi, Ho, Wo, scale_h,
                 scale_w, align_corners, maxrows);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (sizeof(T) == 2 && align_corners && Ho == 2 * Hi && Wo == 2 * Wi && (Wi & 3) == 0 && Hi >= 4 && Wi >= 4 &&
      ((((uintptr_t)dy | (uintptr_t)dx) & 7) == 0) && ((sgb | sdb) & 3) == 0 && !getenv("CENET_BIL_NO_UP2AC")) {
    CENET_LAUNCH((bilinear_up2ac_bwd_kernel<T>), dim3(B * C, chunks_one(Hi * Wi / 4)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi,
                 scale_h, scale_w);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (sizeof(T) == 2 && !align_corners && Hi == 2 * Ho && Wi == 2 * Wo && scale_h == 2.0f && scale_w == 2.0f) {
    CENET_LAUNCH((bilinear_down2_bwd_kernel<T>), dim3(B * C, chunks_for(Ho * Wo)), dim3(256), stream, dy, sgb, dx, sdb, C, Ho, Wo,
                 (const T*)nullptr);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (scale_w >= 0.5f)
    CENET_LAUNCH((bilinear_bwd_kernel<T, 6>), dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi, Ho,
                 Wo, scale_h, scale_w, align_corners, (const T*)nullptr);
  else
    CENET_LAUNCH((bilinear_bwd_kernel<T, 10>), dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi, Ho,
                 Wo, scale_h, scale_w, align_corners, (const T*)nullptr);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(bilinear_bwd, (const T* dy, long sgb, T* dx, long sdb, int B, int C, int Hi, int Wi, int Ho, int Wo, float scale_h,
                          float scale_w, int align_corners, hipStream_t stream),
           (dy, sgb, dx, sdb, B, C, Hi, Wi, Ho, Wo, scale_h, scale_w, align_corners, stream))
// dx = backward(dy) + dx_add (dx_add laid out like dx): the input of the resampling has further consumers (dseb.py:63-76: the FEA
// down-samplings read the tensor the attention and the combine also read) whose gradients arrive here as one addend
template <typename T>
static int bilinear_bwd_add_impl(const T* dy, long sgb, T* dx, long sdb, const T* dx_add, int B, int C, int Hi, int Wi, int Ho,
                                 int Wo, float scale_h, float scale_w, int align_corners, hipStream_t stream) {
  return bilinear_bwd_impl<T>(dy, sgb, dx, sdb, B, C, Hi, Wi, Ho, Wo, scale_h, scale_w, align_corners, stream, dx_add);
}
CENET_TWIN(bilinear_bwd_add, (const T* dy, long sgb, T* dx, long sdb, const T* dx_add, int B, int C, int Hi, int Wi, int Ho, int Wo,
                              float scale_h, float scale_w, int align_corners, hipStream_t stream),
           (dy, sgb, dx, sdb, dx_add, B, C, Hi, Wi, Ho, Wo, scale_h, scale_w, align_corners, stream))

template <typename T>
static int nearest2x_fwd_impl(const T* x, long sxb, T* y, long syb, int B, int C, int Hi, int Wi, hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0) return CENET_EINVAL;
  if ((syb & 1) || ((uintptr_t)y & (2 * sizeof(T) - 1))) return CENET_EINVAL;  // output rows are written as element pairs
  CENET_LAUNCH((nearest2x_fwd_kernel<T>), dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, x, sxb, y, syb, C, Hi, Wi);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(nearest2x_fwd, (const T* x, long sxb, T* y, long syb, int B, int C, int Hi, int Wi, hipStream_t stream),
           (x, sxb, y, syb, B, C, Hi, Wi, stream))

template <typename T>
static int nearest2x_bwd_impl(const T* dy, long sgb, T* dx, long sdb, int B, int C, int Hi, int Wi, hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0) return CENET_EINVAL;
  if ((sgb & 1) || ((uintptr_t)dy & (2 * sizeof(T) - 1))) return CENET_EINVAL;  // gradient rows are read as element pairs
  CENET_LAUNCH((nearest2x_bwd_kernel<T>), dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(nearest2x_bwd, (const T* dy, long sgb, T* dx, long sdb, int B, int C, int Hi, int Wi, hipStream_t stream),
           (dy, sgb, dx, sdb, B, C, Hi, Wi, stream))

template <typename T>
static int adaptive_avgpool_fwd_impl(const T* x, long sxb, T* y, long syb, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                     hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CENET_EINVAL;
  CENET_LAUNCH((adaptive_avgpool_fwd_kernel<T>), dim3(B * C), dim3(64), stream, x, sxb, y, syb, C, Hi, Wi, Ho, Wo);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(adaptive_avgpool_fwd, (const T* x, long sxb, T* y, long syb, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                  hipStream_t stream), (x, sxb, y, syb, B, C, Hi, Wi, Ho, Wo, stream))

template <typename T>
static int adaptive_avgpool_bwd_impl(const T* dy, long sgb, T* dx, long sdb, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                     hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CENET_EINVAL;
  CENET_LAUNCH((adaptive_avgpool_bwd_kernel<T>), dim3(B * C, chunks_for(Hi * Wi)), dim3(256), stream, dy, sgb, dx, sdb, C, Hi, Wi,
               Ho, Wo);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(adaptive_avgpool_bwd, (const T* dy, long sgb, T* dx, long sdb, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                  hipStream_t stream), (dy, sgb, dx, sdb, B, C, Hi, Wi, Ho, Wo, stream))

template <typename T>
static int maxpool2_fwd_impl(const T* x, T* y, long syb, const float* scale, int B, int C, int Hi, int Wi, hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 1 || Wi <= 1 || (Hi & 1) || (Wi & 1)) return CENET_EINVAL;
  if ((uintptr_t)x & (2 * sizeof(T) - 1)) return CENET_EINVAL;  // windows are read as element pairs
  CENET_LAUNCH((maxpool2_fwd_kernel<T>), dim3(B * C, chunks_for(Hi * Wi / 4)), dim3(256), stream, x, y, syb, scale, C, Hi, Wi);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(maxpool2_fwd, (const T* x, T* y, long syb, const float* scale, int B, int C, int Hi, int Wi, hipStream_t stream),
           (x, y, syb, scale, B, C, Hi, Wi, stream))

template <typename T>
static int maxpool2_bwd_acc_impl(const T* x, const T* dy, long sgb, T* dx, const float* scale, float* dscale_acc, int B, int C,
                                 int Hi, int Wi, hipStream_t stream) {
  if (B <= 0 || C <= 0 || Hi <= 1 || Wi <= 1 || (Hi & 1) || (Wi & 1)) return CENET_EINVAL;
  if (((uintptr_t)x | (uintptr_t)dx) & (2 * sizeof(T) - 1)) return CENET_EINVAL;
  CENET_LAUNCH((maxpool2_bwd_kernel<T>), dim3(B * C, chunks_for(Hi * Wi / 4)), dim3(256), stream, x, dy, sgb, dx, scale,
               dscale_acc, C, Hi, Wi);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(maxpool2_bwd_acc, (const T* x, const T* dy, long sgb, T* dx, const float* scale, float* dscale_acc, int B, int C,
                              int Hi, int Wi, hipStream_t stream), (x, dy, sgb, dx, scale, dscale_acc, B, C, Hi, Wi, stream))
