// dwconv.hip — depthwise 3x3 convolution (stride 1, padding = dilation), forward / data-grad / weight-grad.
//   token layout  [B, N=H*W, C] : PVTv2 Mlp DWConv (+bias, +GELU)            pvtv2.py:42-43,359-370
//   NCHW planes               : CFAM Mlp dwconv (+bias,+GELU) cfam.py:150-151; dilated SepConvBN depthwise
//                               blocks.py:142-150,173; EUCB depthwise blocks.py:305
// HBM-bound: one thread per output element; NCHW threads run along x (coalesced rows, neighbours from L1/L2),
// token-layout threads run along C.  The data-gradient is the same kernel with the 3x3 taps flipped.
#include "common.h"
#include "../../include/cenet_hip.h"

// grid (B*C, chunks). y_pre = conv(x)+bias ; if a != nullptr: a = act(y_pre)
__global__ __launch_bounds__(256) void dw3x3_nchw_kernel(const float* __restrict__ x, long sxb, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, long syb,
                                                        float* __restrict__ a, long sab, int C, int H, int W, int dil,
                                                        int flip, int act, float slope) {
  const int bc = blockIdx.x;
  const int b = bc / C, c = bc - b * C;
  const int HW = H * W;
  float wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = w[c * 9 + (flip ? 8 - t : t)];
  const float bv = bias ? bias[c] : 0.f;
  const float* xp = x + (long)b * sxb + (long)c * HW;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < HW; p += gridDim.y * 256) {
    const int py = p / W, px = p - py * W;
    float acc = bv;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = py + (ky - 1) * dil;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = px + (kx - 1) * dil;
        if (ix < 0 || ix >= W) continue;
        acc += wt[ky * 3 + kx] * xp[iy * W + ix];
      }
    }
    y[(long)b * syb + (long)c * HW + p] = acc;
    if (a) a[(long)b * sab + (long)c * HW + p] = act_fwd(act, acc, slope);
  }
}

// token layout [B, H*W, C]: thread = one channel (lanes along C: every load/store is a contiguous 1 KB row),
// workgroup = (256-channel slab, strip of DW_SW pixels along x, image).  A 3x3 register window slides along the
// strip, so each output costs 3 new loads instead of 9 and there is no per-element div/mod.
#define DW_SW 8
__global__ __launch_bounds__(256) void dw3x3_tok_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ a, int C, int H, int W, int flip, int act,
                                                       float slope) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int strips_per_row = (W + DW_SW - 1) / DW_SW;
  const int py = blockIdx.y / strips_per_row;
  const int px0 = (blockIdx.y - py * strips_per_row) * DW_SW;
  const long img = (long)blockIdx.z * H * W * C;
  const float* xb = x + img + c;
  float wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = w[c * 9 + (flip ? 8 - t : t)];
  const float bv = bias ? bias[c] : 0.f;
  float win[3][3];  // win[ky][slot]: columns px-1, px, px+1
  auto load_col = [&](int ix, float col[3]) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = py + ky - 1;
      col[ky] = (ix >= 0 && ix < W && iy >= 0 && iy < H) ? xb[((long)iy * W + ix) * C] : 0.f;
    }
  };
  float c0[3], c1[3], c2[3];
  load_col(px0 - 1, c0);
  load_col(px0, c1);
#pragma unroll
  for (int i = 0; i < DW_SW; ++i) {
    const int px = px0 + i;
    load_col(px + 1, c2);
    if (px < W) {
      float acc = bv;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) acc += wt[ky * 3] * c0[ky] + wt[ky * 3 + 1] * c1[ky] + wt[ky * 3 + 2] * c2[ky];
      const long e = img + ((long)py * W + px) * C + c;
      y[e] = acc;
      if (a) a[e] = act_fwd(act, acc, slope);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      c0[ky] = c1[ky];
      c1[ky] = c2[ky];
    }
  }
  (void)win;
}

// weight/bias gradient, NCHW: grid (C, splits); dw[c,t] += sum_{b,p} dy[b,c,p] * x[b,c,p+off_t]; db[c] += sum dy
__global__ __launch_bounds__(256) void dw3x3_wgrad_nchw_kernel(const float* __restrict__ x, long sxb,
                                                              const float* __restrict__ dy, long sgb,
                                                              float* __restrict__ dw, float* __restrict__ db, int B, int C,
                                                              int H, int W, int dil) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const int HW = H * W;
  const long total = (long)B * HW;
  float acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = 0.f;
  for (int w__ = blockIdx.y; w__ < B * ((HW + 1023) / 1024); w__ += gridDim.y)  // (image, 1024-pixel chunk) items: no per-element division
  for (int b = w__ / ((HW + 1023) / 1024), p = (w__ - b * ((HW + 1023) / 1024)) * 1024 + threadIdx.x,
           pend__ = ((w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 < HW) ? (w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 : HW;
       p < pend__; p += 256) {
    const int py = p / W, px = p - py * W;
    const float g = dy[(long)b * sgb + (long)c * HW + p];
    const float* xp = x + (long)b * sxb + (long)c * HW;
    acc[9] += g;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = py + (ky - 1) * dil;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = px + (kx - 1) * dil;
        if (ix < 0 || ix >= W) continue;
        acc[ky * 3 + kx] += g * xp[iy * W + ix];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    float s = block_sum(acc[t], red);
    if (threadIdx.x == 0) {
      if (t < 9) atomicAdd(&dw[c * 9 + t], s);
      else if (db) atomicAdd(&db[c], s);
    }
  }
}

// weight/bias gradient, token layout: thread = channel, workgroup = (256-channel slab, DW_WROWS image rows, image).
// The 3x3 input window slides along each row (3 new loads + 1 gradient load per pixel); ten float atomics per thread
// at the end.
#define DW_WROWS 4
__global__ __launch_bounds__(256) void dw3x3_wgrad_tok_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ dw, float* __restrict__ db, int C, int H,
                                                             int W) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const long img = (long)blockIdx.z * H * W * C;
  const float* xb = x + img + c;
  const float* gb = dy + img + c;
  float acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = 0.f;
  const int y0 = blockIdx.y * DW_WROWS;
  for (int py = y0; py < y0 + DW_WROWS && py < H; ++py) {
    float c0[3], c1[3], c2[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = py + ky - 1;
      c0[ky] = 0.f;
      c1[ky] = (iy >= 0 && iy < H) ? xb[((long)iy * W) * C] : 0.f;
    }
    for (int px = 0; px < W; ++px) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = py + ky - 1;
        c2[ky] = (px + 1 < W && iy >= 0 && iy < H) ? xb[((long)iy * W + px + 1) * C] : 0.f;
      }
      const float g = gb[((long)py * W + px) * C];
      acc[9] += g;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        acc[ky * 3] += g * c0[ky];
        acc[ky * 3 + 1] += g * c1[ky];
        acc[ky * 3 + 2] += g * c2[ky];
        c0[ky] = c1[ky];
        c1[ky] = c2[ky];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) atomicAdd(&dw[c * 9 + t], acc[t]);
  if (db) atomicAdd(&db[c], acc[9]);
}

static inline int plane_chunks(int HW) {
  int ch = cdiv(HW, 1024);
  return ch > 64 ? 64 : ch;
}

extern "C" int cenet_dwconv3x3_nchw_f32(const float* x, long sxb, const float* w, const float* bias, float* y, long syb,
                                        float* a, long sab, int B, int C, int H, int W, int dil, int flip, int act,
                                        float slope, hipStream_t stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || dil <= 0) return CENET_EINVAL;
  CENET_LAUNCH(dw3x3_nchw_kernel, dim3(B * C, plane_chunks(H * W)), dim3(256), stream, x, sxb, w, bias, y, syb, a, sab, C, H,
               W, dil, flip, act, slope);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_dwconv3x3_tok_f32(const float* x, const float* w, const float* bias, float* y, float* a, int B, int C,
                                       int H, int W, int flip, int act, float slope, hipStream_t stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  const int strips = H * ((W + DW_SW - 1) / DW_SW);
  if (strips > 65535 || B > 65535) return CENET_EUNSUPPORTED;
  CENET_LAUNCH(dw3x3_tok_kernel, dim3(cdiv(C, 256), strips, B), dim3(256), stream, x, w, bias, y, a, C, H, W, flip, act, slope);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_dwconv3x3_wgrad_nchw_acc_f32(const float* x, long sxb, const float* dy, long sgb, float* dw_acc,
                                                  float* dbias_acc, int B, int C, int H, int W, int dil,
                                                  hipStream_t stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  long total = (long)B * H * W;
  long want = 1024 / C;
  long maxs = (total + 2047) / 2048;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  CENET_LAUNCH(dw3x3_wgrad_nchw_kernel, dim3(C, (unsigned)want), dim3(256), stream, x, sxb, dy, sgb, dw_acc, dbias_acc, B, C,
               H, W, dil);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_dwconv3x3_wgrad_tok_acc_f32(const float* x, const float* dy, float* dw_acc, float* dbias_acc, int B,
                                                 int C, int H, int W, hipStream_t stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  CENET_LAUNCH(dw3x3_wgrad_tok_kernel, dim3(cdiv(C, 256), cdiv(H, DW_WROWS), B), dim3(256), stream, x, dy, dw_acc,
               dbias_acc, C, H, W);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
