// dwconv.hip — depthwise 3x3 convolution (stride 1, padding = dilation), forward / data-grad / weight-grad.
//   token layout  [B, N=H*W, C] : PVTv2 Mlp DWConv (+bias, +GELU)            pvtv2.py:42-43,359-370
//   NCHW planes               : CFAM Mlp dwconv (+bias,+GELU) cfam.py:150-151; dilated SepConvBN depthwise
//                               blocks.py:142-150,173; EUCB depthwise blocks.py:305
// HBM-bound: one thread per output element; NCHW threads run along x (coalesced rows, neighbours from L1/L2),
// token-layout threads run along C.  The data-gradient is the same kernel with the 3x3 taps flipped.
// Templates over the activation storage type T (float / bf16_t, common.h); weights, bias and their gradients are fp32.
#include "common.h"
#include <cstdlib>
#include "../../include/cenet_hip.h"

// grid (B*C, chunks). y_pre = conv(x)+bias ; if a != nullptr: a = act(y_pre)
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_nchw_kernel(const T* __restrict__ x, long sxb, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ y, long syb,
                                                        T* __restrict__ a, long sab, int C, int H, int W, int dil,
                                                        int flip, int act, float slope) {
  const int bc = blockIdx.x;
  const int b = bc / C, c = bc - b * C;
  const int HW = H * W;
  float wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = w[c * 9 + (flip ? 8 - t : t)];
  const float bv = bias ? bias[c] : 0.f;
  const T* xp = x + (long)b * sxb + (long)c * HW;
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
        acc += wt[ky * 3 + kx] * ldf(xp + iy * W + ix);
      }
    }
    if (y) stf(y + (long)b * syb + (long)c * HW + p, acc);
    if (a) stf(a + (long)b * sab + (long)c * HW + p, act_fwd(act, acc, slope));
  }
}

// token layout [B, H*W, C]: thread = one channel (lanes along C: every load/store is a contiguous 1 KB row),
// workgroup = (256-channel slab, strip of DW_SW pixels along x, image).  A 3x3 register window slides along the
// strip, so each output costs 3 new loads instead of 9 and there is no per-element div/mod.
#define DW_SW 8
// DW_SR output rows per thread: (DW_SR + 2) input rows x (DW_SW + 2) columns feed DW_SR x DW_SW outputs — 1.9 loads per output
// at DW_SR = 4 (4.1 TB/s at 56x56x512), 2.5 at DW_SR = 2 (more workgroups for the small maps), 3.75 for single rows
// CPT channels per thread (1 for fp32; 2 for bf16, so that a lane still moves 4 bytes per access)
template <typename T, int DW_SR, int CPT>
__global__ __launch_bounds__(256) void dw3x3_tok_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ y,
                                                       T* __restrict__ a, int C, int H, int W, int flip, int act,
                                                       float slope) {
  const cenet_bid bid = cenet_xcd_block();  // row-neighbour strips share their halo rows through one L2
  const int c = (bid.x * 256 + threadIdx.x) * CPT;
  if (c >= C) return;
  const int strips_per_row = (W + DW_SW - 1) / DW_SW;
  const int py = (bid.y / strips_per_row) * DW_SR;
  const int px0 = (bid.y % strips_per_row) * DW_SW;
  const long img = (long)bid.z * H * W * C;
  const T* xb = x + img + c;
  float wt[CPT][9], bv[CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) {
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[e][t] = w[(c + e) * 9 + (flip ? 8 - t : t)];
    bv[e] = bias ? bias[c + e] : 0.f;
  }
  // col[slot][r]: input rows py-1 .. py+DW_SR of one column; three columns slide along the strip
  float col[3][DW_SR + 2][CPT];
  auto load_col = [&](int ix, float (*dst)[CPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < DW_SR + 2; ++r) {
      const int iy = py + r - 1;
      if (ix >= 0 && ix < W && iy >= 0 && iy < H) {
        ldv<CPT>(dst[r], xb + ((long)iy * W + ix) * C);
      } else {
#pragma unroll
        for (int e = 0; e < CPT; ++e) dst[r][e] = 0.f;
      }
    }
  };
  load_col(px0 - 1, col[0]);
  load_col(px0, col[1]);
#pragma unroll
  for (int i = 0; i < DW_SW; ++i) {
    const int px = px0 + i;
    load_col(px + 1, col[(i + 2) % 3]);
    if (px < W) {
      const float(*c0)[CPT] = col[i % 3];
      const float(*c1)[CPT] = col[(i + 1) % 3];
      const float(*c2)[CPT] = col[(i + 2) % 3];
#pragma unroll
      for (int r = 0; r < DW_SR; ++r)
        if (py + r < H) {
          float acc[CPT], av[CPT];
#pragma unroll
          for (int e = 0; e < CPT; ++e) {
            acc[e] = bv[e];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
              acc[e] += wt[e][ky * 3] * c0[r + ky][e] + wt[e][ky * 3 + 1] * c1[r + ky][e] + wt[e][ky * 3 + 2] * c2[r + ky][e];
            av[e] = act_fwd(act, acc[e], slope);
          }
          const long o = img + ((long)(py + r) * W + px) * C + c;
          if (y) stv<CPT>(y + o, acc);
          if (a) stv<CPT>(a + o, av);
        }
    }
  }
}

// weight/bias gradient, NCHW: grid (C, splits); dw[c,t] += sum_{b,p} dy[b,c,p] * x[b,c,p+off_t]; db[c] += sum dy
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_wgrad_nchw_kernel(const T* __restrict__ x, long sxb,
                                                              const T* __restrict__ dy, long sgb,
                                                              float* __restrict__ dw, float* __restrict__ db, int B, int C,
                                                              int H, int W, int dil) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const int HW = H * W;
  const long total = (long)B * HW;
  float acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = 0.f;
  auto pixel = [&](int b, int p) __attribute__((always_inline)) {
    const int py = p / W, px = p - py * W;
    const float g = ldf(dy + (long)b * sgb + (long)c * HW + p);
    const T* xp = x + (long)b * sxb + (long)c * HW;
    acc[9] += g;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = py + (ky - 1) * dil;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = px + (kx - 1) * dil;
        if (ix < 0 || ix >= W) continue;
        acc[ky * 3 + kx] += g * ldf(xp + iy * W + ix);
      }
    }
  };
  if (HW < 256) {
    // small planes (7x7: 49 pixels): the (image, pixel) pairs of the channel flat over the threads — a pass per image left 49
    // of 256 lanes busy and B dependent load rounds per workgroup (94 us for 2048 channels x 7 x 7 x 32 images)
    for (long e = (long)blockIdx.y * 256 + threadIdx.x; e < total; e += (long)gridDim.y * 256) {
      const int b = (int)(e / HW);
      pixel(b, (int)(e - (long)b * HW));
    }
  } else {
    for (int w__ = blockIdx.y; w__ < B * ((HW + 1023) / 1024); w__ += gridDim.y)  // (image, 1024-pixel chunk) items: no per-element division
      for (int b = w__ / ((HW + 1023) / 1024), p = (w__ - b * ((HW + 1023) / 1024)) * 1024 + threadIdx.x,
               pend__ = ((w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 < HW) ? (w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 : HW;
           p < pend__; p += 256)
        pixel(b, p);
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
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_wgrad_tok_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                             float* __restrict__ dw, float* __restrict__ db, int C, int H,
                                                             int W) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const long img = (long)blockIdx.z * H * W * C;
  const T* xb = x + img + c;
  const T* gb = dy + img + c;
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
      c1[ky] = (iy >= 0 && iy < H) ? ldf(xb + ((long)iy * W) * C) : 0.f;
    }
    for (int px = 0; px < W; ++px) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = py + ky - 1;
        c2[ky] = (px + 1 < W && iy >= 0 && iy < H) ? ldf(xb + ((long)iy * W + px + 1) * C) : 0.f;
      }
      const float g = ldf(gb + ((long)py * W + px) * C);
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

// ---- 16-byte token-layout kernels (C % 4 == 0): a thread owns 4 consecutive channels ------------------------------------
struct q4 {
  float v[4];
};
template <typename T>
__device__ __forceinline__ q4 ldq(const T* p) {
  q4 r;
  ld4v(r.v, p);
  return r;
}
__device__ __forceinline__ q4 zq() { return q4{{0.f, 0.f, 0.f, 0.f}}; }

// weight / bias gradient: workgroup = 4 waves = 4 image rows x (64 channel quads); each wave slides the 3x3 window along
// its row (many rows in flight per CU), the four waves meet in LDS and one set of atomics leaves the workgroup
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_wgrad_tok_v4_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                float* __restrict__ dw, float* __restrict__ db, int C, int H,
                                                                int W) {
  __shared__ float red[4][10][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const cenet_bid bid = cenet_xcd_block();
  const int c = (bid.x * 64 + lane) * 4;
  const int py = bid.y * 4 + wave;
  const long img = (long)bid.z * H * W * C;
  float acc[4][10];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[e][t] = 0.f;
  if (c < C && py < H) {
    const T* xb = x + img + c;
    const T* gb = dy + img + c;
    q4 c0[3], c1[3], c2[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = py + ky - 1;
      c0[ky] = zq();
      c1[ky] = (iy >= 0 && iy < H) ? ldq(xb + ((long)iy * W) * C) : zq();
    }
#pragma unroll 2
    for (int px = 0; px < W; ++px) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = py + ky - 1;
        c2[ky] = (px + 1 < W && iy >= 0 && iy < H) ? ldq(xb + ((long)iy * W + px + 1) * C) : zq();
      }
      const q4 g = ldq(gb + ((long)py * W + px) * C);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[e][9] += g.v[e];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          acc[e][ky * 3] += g.v[e] * c0[ky].v[e];
          acc[e][ky * 3 + 1] += g.v[e] * c1[ky].v[e];
          acc[e][ky * 3 + 2] += g.v[e] * c2[ky].v[e];
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        c0[ky] = c1[ky];
        c1[ky] = c2[ky];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int t = 0; t < 10; ++t) red[wave][t][lane * 4 + e] = acc[e][t];
  __syncthreads();
  const int cc = bid.x * 256 + threadIdx.x;  // one channel per thread now
  if (cc < C) {
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      const float sm = red[0][t][threadIdx.x] + red[1][t][threadIdx.x] + red[2][t][threadIdx.x] + red[3][t][threadIdx.x];
      if (t < 9) atomicAdd(&dw[cc * 9 + t], sm);
      else if (db) atomicAdd(&db[cc], sm);
    }
  }
}

// GELU / GELU' of the bf16 kernels: common.h's gelu_as / gelu_as_grad (Abramowitz & Stegun 7.1.26 erf, |error| <= 1.5e-7, far below a
// bf16 ulp, hardware reciprocal and exponential): the tile / plane kernels are VALU-bound, and libm's erff + expf cost more than the
// nine taps (round 6: the previous form here divided in IEEE arithmetic — ten instructions — and carried the 0.5 u (1 + erf) chain).
// Only the bf16 kernels use it; the fp32 (parity) kernels keep erff.
template <int ACT>
__device__ __forceinline__ float dw_act(float u, int act, float slope) {
  if (ACT == ACT_NONE) return u;
  if (ACT == ACT_GELU) return gelu_as(u);
  return act_fwd(act, u, slope);
}
template <int ACT>
__device__ __forceinline__ float dw_act_grad(float u, int act, float slope) {
  if (ACT == ACT_NONE) return 1.f;
  if (ACT == ACT_GELU) return gelu_as_grad(u);
  return act_bwd(act, u, slope);
}

// ---- 16-byte NCHW kernels (W % 4 == 0): a thread owns 4 consecutive pixels of one row --------------------------------
// x window of tap column kx for the quad at (iy, px..px+3): unaligned 16-byte load inside the row, masked scalars at its ends
template <typename T>
__device__ __forceinline__ q4 ld_row4(const T* row, int ix0, int W) {
  if (ix0 >= 0 && ix0 + 3 < W) return ldq(row + ix0);
  q4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r.v[e] = (ix0 + e >= 0 && ix0 + e < W) ? ldf(row + ix0 + e) : 0.f;
  return r;
}

// grid (B*C, chunks), any block size; y_pre = conv(x)+bias ; if a != nullptr: a = act(y_pre)
// ACT: ACT_NONE / ACT_GELU compiled in (GELU with the 1.5e-7 erf above, bf16 tensors only), -1 = run-time switch
template <typename T, int ACT>
__global__ __launch_bounds__(256) void dw3x3_nchw_v4_kernel(const T* __restrict__ x, long sxb, const float* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ y, long syb,
                                                           T* __restrict__ a, long sab, int C, int H, int W, int dil,
                                                           int flip, int act, float slope) {
  const int bc = blockIdx.x;
  const int b = bc / C, c = bc - b * C;
  const int HW = H * W, nq = HW >> 2, wq = W >> 2;
  float wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = w[c * 9 + (flip ? 8 - t : t)];
  const float bv = bias ? bias[c] : 0.f;
  const T* xp = x + (long)b * sxb + (long)c * HW;
  for (int q = blockIdx.y * blockDim.x + threadIdx.x; q < nq; q += gridDim.y * blockDim.x) {
    const int py = q / wq, px = (q - py * wq) * 4;
    q4 acc = q4{{bv, bv, bv, bv}};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = py + (ky - 1) * dil;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const q4 v = ld_row4(xp + iy * W, px + (kx - 1) * dil, W);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc.v[e] += wt[ky * 3 + kx] * v.v[e];
      }
    }
    if (y) st4v(y + (long)b * syb + (long)c * HW + 4 * q, acc.v);
    if (a) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc.v[e] = dw_act<ACT>(acc.v[e], act, slope);
      st4v(a + (long)b * sab + (long)c * HW + 4 * q, acc.v);
    }
  }
}

// grid (C, splits): flat walk over the channel's B*HW/4 quads
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_wgrad_nchw_v4_kernel(const T* __restrict__ x, long sxb,
                                                                 const T* __restrict__ dy, long sgb,
                                                                 float* __restrict__ dw, float* __restrict__ db, int B, int C,
                                                                 int H, int W, int dil) {
  __shared__ float red[4][10];
  const int c = blockIdx.x;
  const int HW = H * W, nq = HW >> 2, wq = W >> 2, total = B * nq;
  float acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = 0.f;
  for (int qg = blockIdx.y * 256 + threadIdx.x; qg < total; qg += gridDim.y * 256) {
    const int b = qg / nq, q = qg - b * nq;
    const int py = q / wq, px = (q - py * wq) * 4;
    const q4 g = ldq(dy + (long)b * sgb + (long)c * HW + 4 * q);
    const T* xp = x + (long)b * sxb + (long)c * HW;
    acc[9] += (g.v[0] + g.v[1]) + (g.v[2] + g.v[3]);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = py + (ky - 1) * dil;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const q4 v = ld_row4(xp + iy * W, px + (kx - 1) * dil, W);
        acc[ky * 3 + kx] += (g.v[0] * v.v[0] + g.v[1] * v.v[1]) + (g.v[2] * v.v[2] + g.v[3] * v.v[3]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    const float sm = wave_sum(acc[t]);
    if (lane == 0) red[wave][t] = sm;
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    const float sm = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (threadIdx.x < 9) atomicAdd(&dw[c * 9 + threadIdx.x], sm);
    else if (db) atomicAdd(&db[c], sm);
  }
}

template <typename T>
static inline bool al16p(const void* a, const void* b, const void* c) {
  return ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & (4 * sizeof(T) - 1)) == 0);
}

static inline int plane_chunks(int HW) {
  int ch = cdiv(HW, 1024);
  return ch > 64 ? 64 : ch;
}

// y (pre-activation) may be NULL when only the activated output a is wanted
// ---- bf16 NCHW, planes resident in LDS (H * W % 4 == 0, (H + 2 dil) * (W + 2 dil) <= DWP_MAX) ------------------------------
// The quad kernels above read every tap window from L1 with unaligned 8-byte loads and ran the decoder's big depthwise convs
// (CFAM Mlp: 256 ch @56x56 ... 2048 ch @7x7; EUCB) at ~1.1-1.3 TB/s.  Here a workgroup stages whole (batch, channel) planes —
// as many as fit — into a zero-bordered fp32 LDS tile with aligned 8-byte loads (everything in flight at once), then every
// thread computes output quads from LDS; the nine weights of a plane are workgroup data.  Forward (+bias, +activation,
// optional pre-activation store), data gradient (mirrored taps) and weight gradient (10 partial sums per thread -> LDS atomics
// per plane -> one global atomic per (channel, tap) and workgroup).
#define DWP_MAX 4624  // floats of LDS per workgroup: one 64x64 plane with dilation 2, or 57 padded 7x7 planes
struct DwPlaneArgs {
  const bf16_t* x;
  long sxb;
  const float* w;
  const float* bias;
  bf16_t* y;
  long syb;
  bf16_t* a;
  long sab;
  const bf16_t* dy;  // weight gradient
  long sgb;
  float* dw;
  float* db;
  int BC, C, H, W, dil, flip, act, ppw;
  float slope;
};

template <int MODE>  // 0: forward / data gradient, 2: weight gradient
__device__ __forceinline__ void dwp_stage(const DwPlaneArgs& a, float* tile, int p0, int np, int PS, int PWd) {
  const int HW = a.H * a.W, nq = HW >> 2;
  for (int i = threadIdx.x; i < np * PS; i += 256) tile[i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < np * nq; i += 256) {
    const int pl = i / nq, q = i - pl * nq;
    const int bc = p0 + pl, b = bc / a.C, c = bc - b * a.C;
    float v[4];
    ld4v(v, a.x + (long)b * a.sxb + (long)c * HW + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // (a quad may straddle two rows when W % 4 != 0: 14x14 maps)
      const int p = 4 * q + e, py = p / a.W, px = p - py * a.W;
      tile[pl * PS + (py + a.dil) * PWd + px + a.dil] = v[e];
    }
  }
  __syncthreads();
}

// Fast form of the two bodies below for W % 4 == 0 and at most 1024 quads per workgroup (28x28: five planes, 56x56: one): a
// thread OWNS its <= 4 quads — their loads are issued together on clamped addresses, every index is computed once with float
// reciprocals (the general bodies spend ~6 integer divisions per quad and iteration: ~240 instructions around 36 multiply-adds,
// behind loops whose trip count is a run-time value), a quad never straddles a row, and with dilation 1 the 18 distinct tap values
// of a quad are read once (the address arithmetic is compile-time, so the 36 reads of the general form fold).
struct DwpOwn {
  int toff[4], pl[4];   // tile offset of the quad's first element (tap (1, 1)); plane
  long go[4];           // element offset of the quad inside its image batch: c * HW + 4 q
  int b[4];
  bool ok[4];
};
__device__ __forceinline__ bool dwp_fast_ok(const DwPlaneArgs& a, int np) {
  return (a.W & 3) == 0 && np * ((a.H * a.W) >> 2) <= 1024;
}
template <bool D1>
__device__ __forceinline__ void dwp_own(const DwPlaneArgs& a, int p0, int np, int PS, int PWd, DwpOwn& o) {
  const int HW = a.H * a.W, nq = HW >> 2, d = D1 ? 1 : a.dil, Q = np * nq;
  const float invW = 1.f / (float)a.W, invnq = 1.f / (float)nq, invC = 1.f / (float)a.C;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = threadIdx.x + 256 * k;
    o.ok[k] = i < Q;
    const int ii = o.ok[k] ? i : 0;
    const int pl = np == 1 ? 0 : (int)(((float)ii + 0.5f) * invnq), q = ii - pl * nq;
    const int bc = p0 + pl, b = (int)(((float)bc + 0.5f) * invC), c = bc - b * a.C;
    const int py = (int)(((float)(4 * q) + 0.5f) * invW), px = 4 * q - py * a.W;
    o.pl[k] = pl;
    o.b[k] = b;
    o.go[k] = (long)c * HW + 4 * q;
    o.toff[k] = pl * PS + (py + d) * PWd + px + d;
  }
}
// zero-fills the tile, then stages src planes (batch stride sb) into it
__device__ __forceinline__ void dwp_stage_own(const DwpOwn& o, const bf16_t* src, long sb, float* tile, int nfl) {
  for (int i = threadIdx.x; i < nfl; i += 256) tile[i] = 0.f;
  float v[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ld4v(v[k], src + (long)o.b[k] * sb + o.go[k]);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (o.ok[k]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[o.toff[k] + e] = v[k][e];
    }
  __syncthreads();
}

template <int ACT, bool D1>
__device__ __forceinline__ void dwp_fwd_fast(const DwPlaneArgs& a, int p0, int np, float* tile, float (*wts)[10]) {
  const int d = D1 ? 1 : a.dil;
  const int PWd = a.W + 2 * d, PS = (a.H + 2 * d) * PWd;
  DwpOwn o;
  dwp_own<D1>(a, p0, np, PS, PWd, o);
  dwp_stage_own(o, a.x, a.sxb, tile, np * PS);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!o.ok[k]) continue;
    const float* w = wts[o.pl[k]];
    const float* t0 = tile + o.toff[k] - d * PWd - d;
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = w[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float wv = w[ky * 3 + kx];
        const float* tp = t0 + ky * d * PWd + kx * d;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += wv * tp[e];
      }
    if (a.y) st4v(a.y + (long)o.b[k] * a.syb + o.go[k], acc);
    if (a.a) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = dw_act<ACT>(acc[e], a.act, a.slope);
      st4v(a.a + (long)o.b[k] * a.sab + o.go[k], acc);
    }
  }
}

template <int ACT>
__device__ __forceinline__ void dwp_fwd_body(const DwPlaneArgs& a, int bx) {
  __shared__ float tile[DWP_MAX];
  __shared__ float wts[64][10];
  const int HW = a.H * a.W, nq = HW >> 2, d = a.dil;
  const int PWd = a.W + 2 * d, PS = (a.H + 2 * d) * PWd;
  const int p0 = bx * a.ppw;
  const int np = p0 + a.ppw <= a.BC ? a.ppw : a.BC - p0;
  for (int i = threadIdx.x; i < np * 10; i += 256) {
    const int pl = i / 10, t = i - pl * 10, c = (p0 + pl) % a.C;
    wts[pl][t] = t < 9 ? a.w[c * 9 + (a.flip ? 8 - t : t)] : (a.bias ? a.bias[c] : 0.f);
  }
  if (dwp_fast_ok(a, np)) {  // (workgroup-uniform)
    if (d == 1) dwp_fwd_fast<ACT, true>(a, p0, np, tile, wts);
    else dwp_fwd_fast<ACT, false>(a, p0, np, tile, wts);
    return;
  }
  dwp_stage<0>(a, tile, p0, np, PS, PWd);
  for (int i = threadIdx.x; i < np * nq; i += 256) {
    const int pl = i / nq, q = i - pl * nq;
    const int bc = p0 + pl, b = bc / a.C, c = bc - b * a.C;
    int off[4];  // tap (ky, kx) of output element e sits at tile[off[e] + ky*d*PWd + kx*d]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = 4 * q + e, py = p / a.W, px = p - py * a.W;
      off[e] = pl * PS + py * PWd + px;
    }
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = wts[pl][9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float wv = wts[pl][ky * 3 + kx];
        const int ro = ky * d * PWd + kx * d;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += wv * tile[off[e] + ro];
      }
    const long o = (long)c * HW + 4 * q;
    if (a.y) st4v(a.y + (long)b * a.syb + o, acc);
    if (a.a) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = dw_act<ACT>(acc[e], a.act, a.slope);
      st4v(a.a + (long)b * a.sab + o, acc);
    }
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void dw3x3_nchw_plane_kernel(DwPlaneArgs a) {
  dwp_fwd_body<ACT>(a, blockIdx.x);
}

__device__ __forceinline__ void dwp_wgrad_body(const DwPlaneArgs& a, int bx) {
  __shared__ float tile[DWP_MAX];
  __shared__ float sums[64][10];
  const int HW = a.H * a.W, nq = HW >> 2, d = a.dil;
  const int PWd = a.W + 2 * d, PS = (a.H + 2 * d) * PWd;
  const int p0 = bx * a.ppw;
  const int np = p0 + a.ppw <= a.BC ? a.ppw : a.BC - p0;
  if (np == 1 && dwp_fast_ok(a, 1)) {  // (workgroup-uniform) one plane per workgroup: the ten sums are workgroup reductions
    DwpOwn o;
    if (d == 1) dwp_own<true>(a, p0, 1, PS, PWd, o);
    else dwp_own<false>(a, p0, 1, PS, PWd, o);
    float g[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ld4v(g[k], a.dy + (long)o.b[k] * a.sgb + o.go[k]);
    dwp_stage_own(o, a.x, a.sxb, tile, PS);
    float acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!o.ok[k]) continue;
      const float* t0 = tile + o.toff[k] - d * PWd - d;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[9] += g[k][e];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* tp = t0 + ky * d * PWd + kx * d;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[ky * 3 + kx] += g[k][e] * tp[e];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      const float sm = wave_sum(acc[t]);
      if (lane == 0) sums[wave][t] = sm;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
      const float sm = sums[0][threadIdx.x] + sums[1][threadIdx.x] + sums[2][threadIdx.x] + sums[3][threadIdx.x];
      const int c = p0 % a.C;
      if (threadIdx.x < 9) atomicAdd(&a.dw[c * 9 + threadIdx.x], sm);
      else if (a.db) atomicAdd(&a.db[c], sm);
    }
    return;
  }
  for (int i = threadIdx.x; i < np * 10; i += 256) sums[i / 10][i % 10] = 0.f;
  dwp_stage<2>(a, tile, p0, np, PS, PWd);
  // work items = (plane, chunk of 64 quads), dealt to the four waves in turn: the plane is wave-uniform, so the ten partial sums
  // are folded across the wave with shuffles and ONE lane adds them to the plane's LDS slots when the plane changes
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cpp = (nq + 63) >> 6, csz = (nq + cpp - 1) / cpp, items = np * cpp;  // equal chunks (196 quads: 4 x 49, not 3 x 64 + 4)
  float acc[10];
  int cur = -1;
  auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      float v = acc[t];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) atomicAdd(&sums[cur][t], v);
    }
  };
  for (int j = wave; j < items; j += 4) {
    const int pl = j / cpp, q = (j - pl * cpp) * csz + lane;
    if (pl != cur) {  // (wave-uniform)
      if (cur >= 0) flush();
      cur = pl;
#pragma unroll
      for (int t = 0; t < 10; ++t) acc[t] = 0.f;
    }
    if (lane >= csz || q >= nq) continue;
    const int bc = p0 + pl, b = bc / a.C, c = bc - b * a.C;
    float g[4];
    ld4v(g, a.dy + (long)b * a.sgb + (long)c * HW + 4 * q);
    int off[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = 4 * q + e, py = p / a.W, px = p - py * a.W;
      off[e] = pl * PS + py * PWd + px;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[9] += g[e];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ro = ky * d * PWd + kx * d;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[ky * 3 + kx] += g[e] * tile[off[e] + ro];
      }
  }
  if (cur >= 0) flush();
  __syncthreads();
  for (int i = threadIdx.x; i < np * 10; i += 256) {
    const int pl = i / 10, t = i - pl * 10, c = (p0 + pl) % a.C;
    if (t < 9) atomicAdd(&a.dw[c * 9 + t], sums[pl][t]);
    else if (a.db) atomicAdd(&a.db[c], sums[pl][9]);
  }
}

__global__ __launch_bounds__(256) void dw3x3_wgrad_nchw_plane_kernel(DwPlaneArgs a) { dwp_wgrad_body(a, blockIdx.x); }

// ---- several depthwise convs in ONE launch (round 4): the three dilated SepConvBN branches of a CFAM block (cfam.py:208-212) read
// channel slices of one tensor with different dilations; as three launches of 2 - 25 MB each they sat at their fill / drain
// latency (15 - 17 us apiece, forward, data gradient and weight gradient alike).  blockIdx.y selects the branch's argument set;
// workgroups beyond a branch's own plane count leave at once.
#define DWP_MULTI_MAX 4
struct DwPlaneMulti {
  DwPlaneArgs a[DWP_MULTI_MAX];
};
__global__ __launch_bounds__(256) void dw3x3_nchw_plane_multi_kernel(DwPlaneMulti m) {
  const DwPlaneArgs& a = m.a[blockIdx.y];
  if ((int)blockIdx.x * a.ppw >= a.BC) return;
  dwp_fwd_body<ACT_NONE>(a, blockIdx.x);
}
__global__ __launch_bounds__(256) void dw3x3_wgrad_nchw_plane_multi_kernel(DwPlaneMulti m) {
  const DwPlaneArgs& a = m.a[blockIdx.y];
  if ((int)blockIdx.x * a.ppw >= a.BC) return;
  dwp_wgrad_body(a, blockIdx.x);
}

// planes per workgroup: as many padded planes as the tile holds, at most 64, and at least ~1024 workgroups where possible
static inline int dwp_ppw(int BC, int H, int W, int dil) {
  const int ps = (H + 2 * dil) * (W + 2 * dil);
  if (ps > DWP_MAX || ((H * W) & 3)) return 0;
  int ppw = DWP_MAX / ps;
  if (ppw > 64) ppw = 64;
  while (ppw > 1 && cdiv(BC, ppw) < 1024) --ppw;
  return ppw;
}

template <typename T>
static int dwconv3x3_nchw_impl(const T* x, long sxb, const float* w, const float* bias, T* y, long syb, T* a, long sab, int B,
                               int C, int H, int W, int dil, int flip, int act, float slope, hipStream_t stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || dil <= 0 || (!y && !a)) return CENET_EINVAL;
  static const bool plane_off = getenv("CENET_DW_NO_PLANE") != nullptr;
  if (sizeof(T) == 2 && !plane_off && ((sxb | syb | sab) & 3) == 0 && quad_aligned<T>(x) && (!y || quad_aligned<T>(y)) &&
      (!a || quad_aligned<T>(a)) && ((long)H * W) % 4 == 0) {
    const int ppw = dwp_ppw(B * C, H, W, dil);
    if (ppw > 0) {
      DwPlaneArgs p;
      p.x = (const bf16_t*)x; p.sxb = sxb; p.w = w; p.bias = bias; p.y = (bf16_t*)y; p.syb = syb; p.a = (bf16_t*)a; p.sab = sab;
      p.dy = nullptr; p.sgb = 0; p.dw = p.db = nullptr;
      p.BC = B * C; p.C = C; p.H = H; p.W = W; p.dil = dil; p.flip = flip; p.act = act; p.ppw = ppw; p.slope = slope;
      const dim3 grid(cdiv(B * C, ppw));
      if (!a || act == ACT_NONE) CENET_LAUNCH((dw3x3_nchw_plane_kernel<ACT_NONE>), grid, dim3(256), stream, p);
      else if (act == ACT_GELU) CENET_LAUNCH((dw3x3_nchw_plane_kernel<ACT_GELU>), grid, dim3(256), stream, p);
      else CENET_LAUNCH((dw3x3_nchw_plane_kernel<-1>), grid, dim3(256), stream, p);
      CENET_CHECK_LAUNCH();
      return CENET_OK;
    }
  }
  if ((W & 3) == 0 && ((sxb | syb | sab) & 3) == 0 && al16p<T>(x, y, a)) {
    const int nq = H * W / 4;
    const int th = nq <= 64 ? 64 : (nq <= 128 ? 128 : 256);
    int ch = cdiv(nq, th * 2);
    if (ch > 16) ch = 16;
    if (sizeof(T) == 2 && (!a || act == ACT_NONE))
      CENET_LAUNCH((dw3x3_nchw_v4_kernel<T, ACT_NONE>), dim3(B * C, ch), dim3(th), stream, x, sxb, w, bias, y, syb, a, sab, C, H, W,
                   dil, flip, act, slope);
    else if (sizeof(T) == 2 && act == ACT_GELU)
      CENET_LAUNCH((dw3x3_nchw_v4_kernel<T, ACT_GELU>), dim3(B * C, ch), dim3(th), stream, x, sxb, w, bias, y, syb, a, sab, C, H, W,
                   dil, flip, act, slope);
    else
      CENET_LAUNCH((dw3x3_nchw_v4_kernel<T, -1>), dim3(B * C, ch), dim3(th), stream, x, sxb, w, bias, y, syb, a, sab, C, H, W, dil,
                   flip, act, slope);
  } else {
    CENET_LAUNCH((dw3x3_nchw_kernel<T>), dim3(B * C, plane_chunks(H * W)), dim3(256), stream, x, sxb, w, bias, y, syb, a, sab, C,
                 H, W, dil, flip, act, slope);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(dwconv3x3_nchw, (const T* x, long sxb, const float* w, const float* bias, T* y, long syb, T* a, long sab, int B, int C,
                            int H, int W, int dil, int flip, int act, float slope, hipStream_t stream),
           (x, sxb, w, bias, y, syb, a, sab, B, C, H, W, dil, flip, act, slope, stream))

// ---- bf16 token layout, LDS-tiled (C % 8 == 0) ----------------------------------------------------------------------------
// The sliding-window kernels above keep one or two 4-byte loads per lane in flight and reach 1 - 2 TB/s on the 100 MB
// tensors of stage 1 (the tensor is only ~6x the bandwidth-delay product of the chip, so bytes in flight decide).  Here a
// workgroup owns a (TH x TW)-pixel tile of a 128-channel slab: the (TH+2) x (TW+2) halo tile goes HBM -> LDS with
// global_load_lds_dwordx4 (40 KB in flight per workgroup, no staging registers, out-of-image pixels read a zero block),
// then thread = (8 channels, one of 16 pixel slots) reads its nine taps with ds_read_b128.
//   MODE 0: out2 = act(conv(x) + bias), out = conv(x) + bias (either may be NULL); FLIP = data gradient (taps mirrored)
//   MODE 2: backward of act(conv(x) + bias) given g = dL/d(act): recomputes u = conv(x) + bias from the x tile (u is not
//           stored by the forward pass), writes gu = g * act'(u), and accumulates dw[c][t] += gu * x[tap t], db[c] += gu
//           in registers over the workgroup's tiles; one LDS reduction and one set of float atomics per workgroup.
#ifdef CENET_HOSTSIM_BUILD
typedef unsigned dw_u4 __attribute__((vector_size(16)));
typedef unsigned dw_u2 __attribute__((vector_size(8)));
#else
typedef unsigned dw_u4 __attribute__((ext_vector_type(4)));
typedef unsigned dw_u2 __attribute__((ext_vector_type(2)));
#endif
struct DwTileArgs {
  const bf16_t* x;
  const bf16_t* g;
  const float* w;
  const float* bias;
  bf16_t* out;
  bf16_t* out2;
  float* dw;
  float* db;
  int C, H, W, flip, act, tiles_x, ntiles, tpw;
  float slope;
};

// 16 (CPT = 8) or 8 (CPT = 4) bytes of one pixel's channels from LDS / HBM, as packed bf16 pairs
template <int CPT>
struct DwPk {
  unsigned u[CPT / 2];
};
template <int CPT>
__device__ __forceinline__ DwPk<CPT> dw_ld(const void* p) {
  DwPk<CPT> r;
  if (CPT == 8) {
    const dw_u4 v = *(const dw_u4*)p;
    r.u[0] = v[0], r.u[1] = v[1], r.u[CPT / 2 - 2] = v[2], r.u[CPT / 2 - 1] = v[3];
  } else {
    const dw_u2 v = *(const dw_u2*)p;
    r.u[0] = v[0], r.u[1] = v[1];
  }
  return r;
}
template <int CPT>
__device__ __forceinline__ void dw_st(void* p, const unsigned* q) {
  if (CPT == 8) {
    dw_u4 v;
    v[0] = q[0], v[1] = q[1], v[2] = q[CPT / 2 - 2], v[3] = q[CPT / 2 - 1];
    *(dw_u4*)p = v;
  } else {
    dw_u2 v;
    v[0] = q[0], v[1] = q[1];
    *(dw_u2*)p = v;
  }
}

// ACT: ACT_NONE / ACT_GELU compiled in, -1 = the runtime switch of common.h.  CPT channels per thread: 8 (16-byte accesses)
// for the forward / data-gradient modes; 4 for MODE 2, whose 10 accumulators per channel would not fit beside the weights
template <int TH, int TW, int MODE, int ACT>
__global__ __launch_bounds__(256, 3) void dw3x3_tok_tile_kernel(DwTileArgs a) {
  constexpr int CPT = MODE == 2 ? 4 : 8, NG = 128 / CPT, SLOTS = 256 / NG, GPW = 64 / NG < 1 ? 1 : 64 / NG;
  constexpr int PW = TW + 2, PT = (TH + 2) * PW, NI = (PT + 3) / 4;  // halo tile: pixels, LDS-DMA instructions (4 pixels each)
  constexpr int NOUT = TH * TW, NPASS = (NOUT + SLOTS - 1) / SLOTS;
  constexpr int NACC = CPT * 10;
  constexpr int RED = MODE == 2 ? 4 * NG * NACC * 4 : 0;
  constexpr int LDSB = NI * 1024 > RED ? NI * 1024 : RED;
  constexpr bool FLIP = MODE == 1;  // data gradient: tap (ky, kx) reads the mirrored neighbour
  static_assert(GPW == 4 || GPW == 2, "pixel slots per wave");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDSB];
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef CENET_HOSTSIM_BUILD
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const cenet_bid bid = cenet_xcd_block();
  const int grp = tid % NG, slot = tid / NG;
  const int c0 = bid.x * 128, c = c0 + grp * CPT;
  const bool cok = c < a.C;
  const long img = (long)bid.z * a.H * a.W * a.C;
  float wt[CPT * 9], bv[CPT];  // wt[e * 9 + t]: consecutive floats of the weight tensor, 16-byte aligned (c % 4 == 0)
#pragma unroll
  for (int i = 0; i < CPT * 9 / 4; ++i) {
    const f4 v = cok ? ld4(a.w + (long)c * 9 + 4 * i) : f4{{0.f, 0.f, 0.f, 0.f}};
    wt[4 * i] = v.v[0], wt[4 * i + 1] = v.v[1], wt[4 * i + 2] = v.v[2], wt[4 * i + 3] = v.v[3];
  }
#pragma unroll
  for (int i = 0; i < CPT / 4; ++i) {
    const f4 v = (cok && MODE != 1 && a.bias) ? ld4(a.bias + c + 4 * i) : f4{{0.f, 0.f, 0.f, 0.f}};
    bv[4 * i] = v.v[0], bv[4 * i + 1] = v.v[1], bv[4 * i + 2] = v.v[2], bv[4 * i + 3] = v.v[3];
  }
  float acc[MODE == 2 ? CPT : 1][10];
  if (MODE == 2) {
#pragma unroll
    for (int e = 0; e < CPT; ++e)
#pragma unroll
      for (int t = 0; t < 10; ++t) acc[e][t] = 0.f;
  }
  const int t_begin = bid.y * a.tpw;
  const int t_end = t_begin + a.tpw < a.ntiles ? t_begin + a.tpw : a.ntiles;
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int y0 = (tile / a.tiles_x) * TH, x0 = (tile % a.tiles_x) * TW;
    if (tile != t_begin) __syncthreads();  // every thread is done reading the previous tile
    for (int i = wave; i < NI; i += 4) {
      const int pix = 4 * i + (lane >> 4);
      const int ty = pix / PW, tx = pix - ty * PW;
      const int iy = y0 - 1 + ty, ix = x0 - 1 + tx;
      const int cc = c0 + (lane & 15) * 8;
      const bool ok = pix < PT && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && cc < a.C;
      const void* src = ok ? (const void*)(a.x + img + ((long)iy * a.W + ix) * a.C + cc) : (const void*)ring_zero16;
      ring_glds16(src, lds + i * 1024, lane);
    }
    ring_wait_vm<0>();
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < NPASS; ++p) {
      const int o = p * SLOTS + slot;
      const int oy = o / TW, ox = o - oy * TW;
      const bool ok = cok && o < NOUT && y0 + oy < a.H && x0 + ox < a.W;
      if (!ok) continue;
      const unsigned char* tp = lds + (oy * PW + ox) * 256 + grp * (2 * CPT);
      float u[CPT];
#pragma unroll
      for (int e = 0; e < CPT; ++e) u[e] = bv[e];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const DwPk<CPT> r = dw_ld<CPT>(tp + (FLIP ? ((2 - ky) * PW + 2 - kx) : (ky * PW + kx)) * 256);
#pragma unroll
          for (int h = 0; h < CPT / 2; ++h) {
            u[2 * h] += wt[(2 * h) * 9 + ky * 3 + kx] * cenet_bf2f(r.u[h] & 0xFFFFu);
            u[2 * h + 1] += wt[(2 * h + 1) * 9 + ky * 3 + kx] * cenet_bf2f(r.u[h] >> 16);
          }
        }
      const long oo = img + ((long)(y0 + oy) * a.W + x0 + ox) * a.C + c;
      if (MODE != 2) {
        unsigned q[CPT / 2];
        if (a.out) {
#pragma unroll
          for (int h = 0; h < CPT / 2; ++h) q[h] = cenet_pack_bf2(u[2 * h], u[2 * h + 1]);
          dw_st<CPT>(a.out + oo, q);
        }
        if (MODE == 0 && a.out2) {
#pragma unroll
          for (int h = 0; h < CPT / 2; ++h)
            q[h] = cenet_pack_bf2(dw_act<ACT>(u[2 * h], a.act, a.slope), dw_act<ACT>(u[2 * h + 1], a.act, a.slope));
          dw_st<CPT>(a.out2 + oo, q);
        }
      } else {
        const DwPk<CPT> gq = dw_ld<CPT>(a.g + oo);
        unsigned q[CPT / 2];
        float gu[CPT];
#pragma unroll
        for (int h = 0; h < CPT / 2; ++h) {
          gu[2 * h] = cenet_bf2f(gq.u[h] & 0xFFFFu) * dw_act_grad<ACT>(u[2 * h], a.act, a.slope);
          gu[2 * h + 1] = cenet_bf2f(gq.u[h] >> 16) * dw_act_grad<ACT>(u[2 * h + 1], a.act, a.slope);
          q[h] = cenet_pack_bf2(gu[2 * h], gu[2 * h + 1]);
        }
        dw_st<CPT>(a.out + oo, q);
#pragma unroll
        for (int e = 0; e < CPT; ++e) acc[e][9] += gu[e];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const DwPk<CPT> r = dw_ld<CPT>(tp + (ky * PW + kx) * 256);
#pragma unroll
            for (int h = 0; h < CPT / 2; ++h) {
              acc[2 * h][ky * 3 + kx] += gu[2 * h] * cenet_bf2f(r.u[h] & 0xFFFFu);
              acc[2 * h + 1][ky * 3 + kx] += gu[2 * h + 1] * cenet_bf2f(r.u[h] >> 16);
            }
          }
      }
    }
  }
  if (MODE == 2) {
    // lanes that differ by a multiple of NG hold the same channels: fold them, then the four waves meet in LDS
    float* red = (float*)lds;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < CPT; ++e)
#pragma unroll
      for (int t = 0; t < 10; ++t) {
        float v = acc[e][t];
        if (GPW == 4) v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        acc[e][t] = v;
      }
    if (lane < NG) {
#pragma unroll
      for (int e = 0; e < CPT; ++e)
#pragma unroll
        for (int t = 0; t < 10; ++t) red[(wave * NG + lane) * NACC + e * 10 + t] = acc[e][t];
    }
    __syncthreads();
    for (int i = tid; i < NG * NACC; i += 256) {
      const int gi = i / NACC, r = i - gi * NACC, e = r / 10, t = r - e * 10;
      const int ch = c0 + gi * CPT + e;
      if (ch < a.C) {
        const float sm = red[i] + red[NG * NACC + i] + red[2 * NG * NACC + i] + red[3 * NG * NACC + i];
        if (t < 9) atomicAdd(&a.dw[ch * 9 + t], sm);
        else if (a.db) atomicAdd(&a.db[ch], sm);
      }
    }
  }
}

// ---- bf16 token layout, WHOLE small planes (14 x 14 / 7 x 7: PVTv2 stages 3 / 4, pvtv2.py:364-370 at C = 1280 / 2048) -------------
// The tile kernel above spends ~60 vector instructions per output element (nine 16-byte tap reads per pixel, every tap converted
// nine times, a one-pixel-per-pass loop) and at these sizes runs ONE round of workgroups, so nothing hides its set-up: 22 / 31 /
// 14 us for 8 M elements (forward / activation backward / data gradient) where 56 x 56 x 512 costs 1.6 ps per element.  Here a
// workgroup owns the plane of one image for a slab of 64 channels — (HW + 2)^2 zero-bordered pixels x 128 B in LDS by LDS-DMA,
// no halo exchange, no ragged tiles — and a thread owns a channel PAIR of one image row: it slides a three-row window along the
// row (3 ds_read_b32 + 6 conversions per column for 2 x 9 FMAs per output pixel), keeps the row's outputs in registers and
// stores 4 bytes per pixel (a half wave writes the 128 contiguous bytes of a pixel's slab).  MODE as above; MODE 2 makes a second
// sweep over the window for the weight gradient (acc[ky][kx] += gu[x - kx] * h[ky][x]) and folds rows in LDS.
template <int HW, int MODE, int ACT>
__global__ __launch_bounds__(HW == 14 ? 448 : 256, HW == 14 ? 4 : 3) void dw3x3_tok_plane_kernel(DwTileArgs a) {
  constexpr int PW = HW + 2, NPIX = PW * PW, NI = (NPIX + 7) / 8, NT = HW == 14 ? 448 : 256, NW = NT / 64;
  constexpr bool FLIP = MODE == 1;
  constexpr int RED = MODE == 2 ? NW * 32 * 20 * 4 : 0;
  constexpr int LDSB = NI * 1024 > RED ? NI * 1024 : RED;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDSB];
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef CENET_HOSTSIM_BUILD
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int c0 = blockIdx.x * 64;
  const long img = (long)blockIdx.y * HW * HW * a.C;
  const int cp = tid & 31, row = tid >> 5;
  const bool rok = row < HW;
  const int ch = c0 + 2 * cp;
  // plane -> LDS: one DMA instruction = 8 pixels x 128 B; border pixels (and the tail past the plane) read the zero block
  for (int i = wave; i < NI; i += NW) {
    const int pix = 8 * i + (lane >> 3);
    const int py = pix / PW, px = pix - py * PW;
    const bool ok = pix < NPIX && py >= 1 && py <= HW && px >= 1 && px <= HW;
    const void* src = ok ? (const void*)(a.x + img + ((long)(py - 1) * HW + px - 1) * a.C + c0 + (lane & 7) * 8) : (const void*)ring_zero16;
    ring_glds16(src, lds + i * 1024, lane);
  }
  float wt[2][9], bv[2] = {0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[e][t] = a.w[(long)(ch + e) * 9 + t];
  if (MODE != 1 && a.bias) bv[0] = a.bias[ch], bv[1] = a.bias[ch + 1];
  unsigned gq[HW];
  if (MODE == 2 && rok) {
#pragma unroll
    for (int i = 0; i < HW; ++i) gq[i] = *(const unsigned*)(a.g + img + ((long)row * HW + i) * a.C + ch);
  }
  ring_wait_vm<0>();
  __syncthreads();
  float u[HW][2];
  float acc[MODE == 2 ? 2 : 1][10];
  if (MODE == 2) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int t = 0; t < 10; ++t) acc[e][t] = 0.f;
  }
  if (rok) {
    const unsigned char* rp = lds + (row * PW) * 128 + cp * 4;  // window rows row .. row + 2 of the bordered plane
#pragma unroll
    for (int i = 0; i < HW; ++i) u[i][0] = bv[0], u[i][1] = bv[1];
#pragma unroll
    for (int kxx = 0; kxx < PW; ++kxx) {
      float h0[3], h1[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const unsigned v = *(const unsigned*)(rp + (ky * PW + kxx) * 128);
        h0[ky] = __uint_as_float(v << 16);
        h1[ky] = __uint_as_float(v & 0xFFFF0000u);
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int i = kxx - kx;
        if (i >= 0 && i < HW) {
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int t = FLIP ? (2 - ky) * 3 + 2 - kx : ky * 3 + kx;
            u[i][0] += wt[0][t] * h0[ky];
            u[i][1] += wt[1][t] * h1[ky];
          }
        }
      }
    }
    const long oo = img + (long)row * HW * a.C + ch;
    if (MODE != 2) {
      if (a.out) {
#pragma unroll
        for (int i = 0; i < HW; ++i) *(unsigned*)(a.out + oo + (long)i * a.C) = cenet_pack_bf2(u[i][0], u[i][1]);
      }
      if (MODE == 0 && a.out2) {
#pragma unroll
        for (int i = 0; i < HW; ++i)
          *(unsigned*)(a.out2 + oo + (long)i * a.C) =
              cenet_pack_bf2(dw_act<ACT>(u[i][0], a.act, a.slope), dw_act<ACT>(u[i][1], a.act, a.slope));
      }
    } else {
#pragma unroll
      for (int i = 0; i < HW; ++i) {
        u[i][0] = __uint_as_float(gq[i] << 16) * dw_act_grad<ACT>(u[i][0], a.act, a.slope);  // u becomes gu
        u[i][1] = __uint_as_float(gq[i] & 0xFFFF0000u) * dw_act_grad<ACT>(u[i][1], a.act, a.slope);
        *(unsigned*)(a.out + oo + (long)i * a.C) = cenet_pack_bf2(u[i][0], u[i][1]);
        acc[0][9] += u[i][0];
        acc[1][9] += u[i][1];
      }
      // (the window is read from LDS AGAIN: without this fence the compiler keeps all 3 x PW x 2 values of the first sweep in
      // registers — 218 of them at HW = 14, one workgroup per CU)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int kxx = 0; kxx < PW; ++kxx) {
        float h0[3], h1[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const unsigned v = *(const unsigned*)(rp + (ky * PW + kxx) * 128);
          h0[ky] = __uint_as_float(v << 16);
          h1[ky] = __uint_as_float(v & 0xFFFF0000u);
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int i = kxx - kx;
          if (i >= 0 && i < HW) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              acc[0][ky * 3 + kx] += u[i][0] * h0[ky];
              acc[1][ky * 3 + kx] += u[i][1] * h1[ky];
            }
          }
        }
      }
    }
  }
  if (MODE == 2) {
    // the two rows of a wave meet by one exchange with lane ^ 32, the waves in LDS (the plane is dead), then one atomic per
    // (channel, tap) and workgroup — 32 images add to an address
    float* red = (float*)lds;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int t = 0; t < 10; ++t) {
        const float v = acc[e][t] + __shfl_xor(acc[e][t], 32);
        if (lane < 32) red[(wave * 32 + lane) * 20 + e * 10 + t] = v;
      }
    __syncthreads();
    for (int i = tid; i < 32 * 20; i += NT) {
      float sm = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < NW; ++w_) sm += red[w_ * 640 + i];
      const int c_ = c0 + 2 * (i / 20) + (i % 20) / 10, t = i % 10;
      if (t < 9) atomicAdd(&a.dw[c_ * 9 + t], sm);
      else if (a.db) atomicAdd(&a.db[c_], sm);
    }
  }
}

template <int HW, int MODE>
static void dw_plane_go(const DwTileArgs& a, int B, hipStream_t stream) {
  const dim3 grid(a.C / 64, B), block(HW == 14 ? 448 : 256);
  if (MODE == 1 || a.act == ACT_NONE) CENET_LAUNCH((dw3x3_tok_plane_kernel<HW, MODE, ACT_NONE>), grid, block, stream, a);
  else if (a.act == ACT_GELU) CENET_LAUNCH((dw3x3_tok_plane_kernel<HW, MODE, ACT_GELU>), grid, block, stream, a);
  else CENET_LAUNCH((dw3x3_tok_plane_kernel<HW, MODE, -1>), grid, block, stream, a);
}

template <int MODE>
static int dw_tile_launch(DwTileArgs a, int B, hipStream_t stream) {
  const int slabs = cdiv(a.C, 128);
  // tile shape by map size; tiles per workgroup: 1 (forward / data gradient), 2 on the largest maps for the weight gradient
#define DW_TILE_GO(TH_, TW_)                                                                                     \
  {                                                                                                              \
    a.tiles_x = cdiv(a.W, TW_);                                                                                  \
    a.ntiles = a.tiles_x * cdiv(a.H, TH_);                                                                       \
    a.tpw = 1;                                                                                                   \
    if (MODE == 2) {                                                                                             \
      static const char* e = getenv("CENET_DW_TPW");                                                              \
      a.tpw = e ? atoi(e) : (a.ntiles >= 16 ? 2 : 1); /* measured: more tiles per workgroup only cost parallelism */                                      \
      if (a.tpw > a.ntiles) a.tpw = a.ntiles;                                                                    \
    }                                                                                                            \
    const dim3 grid(slabs, cdiv(a.ntiles, a.tpw), B);                                                            \
    if (grid.y > 65535 || grid.z > 65535) return CENET_EUNSUPPORTED;                                             \
    if (MODE == 1 || a.act == ACT_NONE)                                                                          \
      CENET_LAUNCH((dw3x3_tok_tile_kernel<TH_, TW_, MODE, ACT_NONE>), grid, dim3(256), stream, a);               \
    else if (a.act == ACT_GELU)                                                                                  \
      CENET_LAUNCH((dw3x3_tok_tile_kernel<TH_, TW_, MODE, ACT_GELU>), grid, dim3(256), stream, a);               \
    else                                                                                                         \
      CENET_LAUNCH((dw3x3_tok_tile_kernel<TH_, TW_, MODE, -1>), grid, dim3(256), stream, a);                     \
  }
  static const bool plane_off = getenv("CENET_DW_NO_PLANE") != nullptr;  // measurement aid
  if (!plane_off && a.H == a.W && (a.H == 14 || a.H == 7) && (a.C & 63) == 0 && B <= 65535) {
    if (a.H == 14) dw_plane_go<14, MODE>(a, B, stream);
    else dw_plane_go<7, MODE>(a, B, stream);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (a.W % 14 == 0) DW_TILE_GO(8, 14)
  else if (a.W <= 8 && a.H <= 8) DW_TILE_GO(8, 8)
  else DW_TILE_GO(8, 16)
#undef DW_TILE_GO
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

// backward of act(DW3x3(x) + bias) from the saved INPUT: gu = g * act'(conv(x) + bias), dw += gu (*) x, db += sum gu
extern "C" int cenet_dwconv3x3_tok_bwd_pre_bf16(const bf16_t* x, const bf16_t* g, const float* w, const float* bias, bf16_t* gu,
                                                float* dw_acc, float* dbias_acc, int B, int C, int H, int W, int act, float slope,
                                                hipStream_t stream) {
  if (!x || !g || !w || !gu || !dw_acc || B <= 0 || C <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  if ((C & 7) != 0 || ((((uintptr_t)x | (uintptr_t)g | (uintptr_t)gu) & 15) != 0)) return CENET_EUNSUPPORTED;
  DwTileArgs a;
  a.x = x; a.g = g; a.w = w; a.bias = bias; a.out = gu; a.out2 = nullptr; a.dw = dw_acc; a.db = dbias_acc;
  a.C = C; a.H = H; a.W = W; a.flip = 0; a.act = act; a.slope = slope;
  return dw_tile_launch<2>(a, B, stream);
}

template <typename T>
static int dwconv3x3_tok_impl(const T* x, const float* w, const float* bias, T* y, T* a, int B, int C, int H, int W, int flip,
                              int act, float slope, hipStream_t stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (!y && !a)) return CENET_EINVAL;
  if (sizeof(T) == 2 && (C & 7) == 0 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)a) & 15) == 0)) {
    DwTileArgs t;
    t.x = (const bf16_t*)x; t.g = nullptr; t.w = w; t.bias = bias; t.out = (bf16_t*)y; t.out2 = (bf16_t*)a; t.dw = t.db = nullptr;
    t.C = C; t.H = H; t.W = W; t.flip = flip; t.act = act; t.slope = slope;
    if (flip) {
      if (bias || a) return CENET_EUNSUPPORTED;  // the mirrored form is the data gradient: no bias, no activation
      return dw_tile_launch<1>(t, B, stream);
    }
    return dw_tile_launch<0>(t, B, stream);
  }
  const int sr = H >= 28 ? 4 : 2;
  const int strips = ((H + sr - 1) / sr) * ((W + DW_SW - 1) / DW_SW);
  if (strips > 65535 || B > 65535) return CENET_EUNSUPPORTED;
  // (a 16-byte-per-thread fp32 variant of this kernel measured slower: 2.6 vs 3.0 TB/s at 56x56x512)
  // bf16: two channels per thread (4-byte accesses) when C is even and the tensors are 4-byte aligned
  constexpr int CP2 = sizeof(T) == 2 ? 2 : 1;
  const bool two = CP2 == 2 && (C & 1) == 0 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)a) & 3) == 0);
  if (two) {
    const dim3 grid(cdiv(C, 512), strips, B);
    if (sr == 4) CENET_LAUNCH((dw3x3_tok_kernel<T, 4, CP2>), grid, dim3(256), stream, x, w, bias, y, a, C, H, W, flip, act, slope);
    else CENET_LAUNCH((dw3x3_tok_kernel<T, 2, CP2>), grid, dim3(256), stream, x, w, bias, y, a, C, H, W, flip, act, slope);
  } else {
    const dim3 grid(cdiv(C, 256), strips, B);
    if (sr == 4) CENET_LAUNCH((dw3x3_tok_kernel<T, 4, 1>), grid, dim3(256), stream, x, w, bias, y, a, C, H, W, flip, act, slope);
    else CENET_LAUNCH((dw3x3_tok_kernel<T, 2, 1>), grid, dim3(256), stream, x, w, bias, y, a, C, H, W, flip, act, slope);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(dwconv3x3_tok, (const T* x, const float* w, const float* bias, T* y, T* a, int B, int C, int H, int W, int flip,
                           int act, float slope, hipStream_t stream), (x, w, bias, y, a, B, C, H, W, flip, act, slope, stream))

// ---- bf16 NCHW weight gradient, ONE CHANNEL ACROSS IMAGES per workgroup (round 6) ---------------------------------------------------
// dw3x3_wgrad_nchw_plane_kernel gives every (image, channel) plane its own workgroup: at 28x28x512 / 56x56x256 (the CFAM Mlp of
// dec2 / dec1) that is 16 384 / 8 192 workgroups whose ~10 us of fixed work — zero-fill + stage + two barriers + ten wave
// reductions + ten global atomics for 784 x 9 multiply-adds — is the kernel (83 / 51 us for 25 / 51 MB), and 32 images add to every
// (channel, tap) address.  Here a workgroup owns channel c for a GROUP of images: it walks them in passes of `np` planes (as many
// as give each thread <= 4 quads), the ten sums stay in registers across the passes, and one reduction + ten atomics end the
// workgroup: 8x fewer workgroups, 8x fewer atomics, the border of the LDS tile zeroed once.
template <bool D1>
__global__ __launch_bounds__(256) void dw3x3_wgrad_nchw_chan_kernel(DwPlaneArgs a, int B, int ipg, int np) {
  __shared__ float tile[DWP_MAX];
  __shared__ float red[4][10];
  const int c = blockIdx.x, b_begin = blockIdx.y * ipg;
  const int b_end = b_begin + ipg < B ? b_begin + ipg : B;
  const int HW = a.H * a.W, nq = HW >> 2, d = D1 ? 1 : a.dil;
  const int PWd = a.W + 2 * d, PS = (a.H + 2 * d) * PWd;
  const float invW = 1.f / (float)a.W, invnq = 1.f / (float)nq;
  int toff[4], pl[4];
  long go[4];
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = threadIdx.x + 256 * k;
    ok[k] = i < np * nq;
    const int ii = ok[k] ? i : 0;
    pl[k] = np == 1 ? 0 : (int)(((float)ii + 0.5f) * invnq);
    const int q = ii - pl[k] * nq;
    const int py = (int)(((float)(4 * q) + 0.5f) * invW), px = 4 * q - py * a.W;  // (W % 4 == 0: a quad never straddles a row)
    go[k] = (long)c * HW + 4 * q;
    toff[k] = pl[k] * PS + (py + d) * PWd + px + d;
  }
  for (int i = threadIdx.x; i < np * PS; i += 256) tile[i] = 0.f;  // (the borders stay zero: every pass rewrites the interiors)
  float acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = 0.f;
  // the NEXT pass's loads are issued before the current pass is computed (bf16 quads kept packed: 16 registers per pass)
  uint2 gq[4], xq[4], gn[4], xn[4];
  auto fetch = [&](int b0, uint2* gd, uint2* xd) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // every load of the pass in flight together, on clamped addresses
      const int bi = b0 + pl[k];
      const int bc = bi < b_end ? bi : b_end - 1;
      gd[k] = *(const uint2*)(a.dy + (long)bc * a.sgb + go[k]);
      xd[k] = *(const uint2*)(a.x + (long)bc * a.sxb + go[k]);
    }
  };
  fetch(b_begin, gn, xn);
  for (int b0 = b_begin; b0 < b_end; b0 += np) {
    float g[4][4];
    bool val[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gq[k] = gn[k];
      xq[k] = xn[k];
      val[k] = ok[k] && b0 + pl[k] < b_end;
    }
    __syncthreads();  // the zero fill (first pass) / every thread is done reading the previous pass's planes
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (ok[k]) {
        tile[toff[k] + 0] = val[k] ? __uint_as_float(xq[k].x << 16) : 0.f;
        tile[toff[k] + 1] = val[k] ? __uint_as_float(xq[k].x & 0xFFFF0000u) : 0.f;
        tile[toff[k] + 2] = val[k] ? __uint_as_float(xq[k].y << 16) : 0.f;
        tile[toff[k] + 3] = val[k] ? __uint_as_float(xq[k].y & 0xFFFF0000u) : 0.f;
      }
    if (b0 + np < b_end) fetch(b0 + np, gn, xn);  // (workgroup-uniform)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g[k][0] = __uint_as_float(gq[k].x << 16), g[k][1] = __uint_as_float(gq[k].x & 0xFFFF0000u);
      g[k][2] = __uint_as_float(gq[k].y << 16), g[k][3] = __uint_as_float(gq[k].y & 0xFFFF0000u);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!val[k]) continue;
      const float* t0 = tile + toff[k] - d * PWd - d;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[9] += g[k][e];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* tp = t0 + ky * d * PWd + kx * d;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[ky * 3 + kx] += g[k][e] * tp[e];
        }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    const float sm = wave_sum(acc[t]);
    if (lane == 0) red[wave][t] = sm;
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    const float sm = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (threadIdx.x < 9) atomicAdd(&a.dw[c * 9 + threadIdx.x], sm);
    else if (a.db) atomicAdd(&a.db[c], sm);
  }
}

template <typename T>
static int dwconv3x3_wgrad_nchw_acc_impl(const T* x, long sxb, const T* dy, long sgb, float* dw_acc, float* dbias_acc, int B,
                                         int C, int H, int W, int dil, hipStream_t stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  static const bool plane_off = getenv("CENET_DW_NO_PLANE") != nullptr;
  if (sizeof(T) == 2 && !plane_off && ((sxb | sgb) & 3) == 0 && quad_aligned<T>(x) && quad_aligned<T>(dy)) {
    const int ppw = dwp_ppw(B * C, H, W, dil);
    if (ppw > 0) {
      DwPlaneArgs p;
      p.x = (const bf16_t*)x; p.sxb = sxb; p.w = nullptr; p.bias = nullptr; p.y = p.a = nullptr; p.syb = p.sab = 0;
      p.dy = (const bf16_t*)dy; p.sgb = sgb; p.dw = dw_acc; p.db = dbias_acc;
      p.BC = B * C; p.C = C; p.H = H; p.W = W; p.dil = dil; p.flip = 0; p.act = 0; p.ppw = ppw; p.slope = 0.f;
      // one channel across a group of images per workgroup where a plane is at most 1024 quads of whole rows and the batch is deep
      // enough to share a workgroup: ~1 000 - 2 000 workgroups
      static const bool chan_off = getenv("CENET_DW_WGRAD_NO_CHAN") != nullptr;  // measurement aid
      const int nq = (H * W) >> 2, ps = (H + 2 * dil) * (W + 2 * dil);
      if (!chan_off && (W & 3) == 0 && nq <= 1024 && B >= 4 && C <= 65535) {
        int np = 1024 / nq;
        if (np > DWP_MAX / ps) np = DWP_MAX / ps;
        if (np > B) np = B;
        int ng = 2048 / C;  // image groups
        if (ng < 1) ng = 1;
        if (ng > cdiv(B, np)) ng = cdiv(B, np);
        const int ipg = cdiv(cdiv(B, ng), np) * np;  // whole passes per group
        if (dil == 1) CENET_LAUNCH(dw3x3_wgrad_nchw_chan_kernel<true>, dim3(C, cdiv(B, ipg)), dim3(256), stream, p, B, ipg, np);
        else CENET_LAUNCH(dw3x3_wgrad_nchw_chan_kernel<false>, dim3(C, cdiv(B, ipg)), dim3(256), stream, p, B, ipg, np);
        CENET_CHECK_LAUNCH();
        return CENET_OK;
      }
      CENET_LAUNCH(dw3x3_wgrad_nchw_plane_kernel, dim3(cdiv(B * C, ppw)), dim3(256), stream, p);
      CENET_CHECK_LAUNCH();
      return CENET_OK;
    }
  }
  long total = (long)B * H * W;
  long want = 1024 / C;
  long maxs = (total + 2047) / 2048;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  if ((W & 3) == 0 && ((sxb | sgb) & 3) == 0 && al16p<T>(x, dy, nullptr)) {
    long w4 = 2048 / C, m4 = (total / 4 + 1023) / 1024;  // ~2048 workgroups, each >= 1024 quads deep
    if (w4 > m4) w4 = m4;
    if (w4 < 1) w4 = 1;
    if (w4 > 256) w4 = 256;
    CENET_LAUNCH((dw3x3_wgrad_nchw_v4_kernel<T>), dim3(C, (unsigned)w4), dim3(256), stream, x, sxb, dy, sgb, dw_acc, dbias_acc,
                 B, C, H, W, dil);
  } else {
    CENET_LAUNCH((dw3x3_wgrad_nchw_kernel<T>), dim3(C, (unsigned)want), dim3(256), stream, x, sxb, dy, sgb, dw_acc, dbias_acc, B,
                 C, H, W, dil);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(dwconv3x3_wgrad_nchw_acc, (const T* x, long sxb, const T* dy, long sgb, float* dw_acc, float* dbias_acc, int B, int C,
                                      int H, int W, int dil, hipStream_t stream),
           (x, sxb, dy, sgb, dw_acc, dbias_acc, B, C, H, W, dil, stream))

/* n <= 4 bias-free, activation-free depthwise 3x3 convs (forward, or data gradient with flip = 1) of bf16 NCHW channel slices in ONE
 * launch: branch i reads x[i] (batch stride sxb[i]), writes y[i] (syb[i]), C[i] channels, dilation dil[i]; all on H x W maps of B
 * images.  CENET_EUNSUPPORTED unless every branch takes the plane-in-LDS path (H W % 4 == 0, the padded plane fits the tile). */
extern "C" int cenet_dwconv3x3_nchw_multi_bf16(const bf16_t* const* x, const long* sxb, const float* const* w, bf16_t* const* y,
                                               const long* syb, const int* C, const int* dil, int n, int B, int H, int W, int flip,
                                               hipStream_t stream) {
  if (n < 1 || n > DWP_MULTI_MAX || !x || !sxb || !w || !y || !syb || !C || !dil || B <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  if (((long)H * W) % 4 != 0) return CENET_EUNSUPPORTED;
  DwPlaneMulti m;
  unsigned gx = 0;
  for (int i = 0; i < n; ++i) {
    if (!x[i] || !w[i] || !y[i] || C[i] <= 0 || dil[i] <= 0) return CENET_EINVAL;
    if (((sxb[i] | syb[i]) & 3) != 0 || !quad_aligned<bf16_t>(x[i]) || !quad_aligned<bf16_t>(y[i])) return CENET_EUNSUPPORTED;
    const int ppw = dwp_ppw(B * C[i], H, W, dil[i]);
    if (ppw <= 0) return CENET_EUNSUPPORTED;
    DwPlaneArgs& p = m.a[i];
    p.x = x[i]; p.sxb = sxb[i]; p.w = w[i]; p.bias = nullptr; p.y = y[i]; p.syb = syb[i]; p.a = nullptr; p.sab = 0;
    p.dy = nullptr; p.sgb = 0; p.dw = p.db = nullptr;
    p.BC = B * C[i]; p.C = C[i]; p.H = H; p.W = W; p.dil = dil[i]; p.flip = flip; p.act = ACT_NONE; p.ppw = ppw; p.slope = 0.f;
    const unsigned g = (unsigned)cdiv(B * C[i], ppw);
    if (g > gx) gx = g;
  }
  for (int i = n; i < DWP_MULTI_MAX; ++i) m.a[i] = m.a[0];
  CENET_LAUNCH(dw3x3_nchw_plane_multi_kernel, dim3(gx, n), dim3(256), stream, m);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

/* the weight gradients of the same n branches in ONE launch: dw[i][C[i], 9] += dy[i] (*) x[i] */
extern "C" int cenet_dwconv3x3_wgrad_nchw_multi_bf16(const bf16_t* const* x, const long* sxb, const bf16_t* const* dy, const long* sgb,
                                                     float* const* dw_acc, const int* C, const int* dil, int n, int B, int H, int W,
                                                     hipStream_t stream) {
  if (n < 1 || n > DWP_MULTI_MAX || !x || !sxb || !dy || !sgb || !dw_acc || !C || !dil || B <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  if (((long)H * W) % 4 != 0) return CENET_EUNSUPPORTED;
  DwPlaneMulti m;
  unsigned gx = 0;
  for (int i = 0; i < n; ++i) {
    if (!x[i] || !dy[i] || !dw_acc[i] || C[i] <= 0 || dil[i] <= 0) return CENET_EINVAL;
    if (((sxb[i] | sgb[i]) & 3) != 0 || !quad_aligned<bf16_t>(x[i]) || !quad_aligned<bf16_t>(dy[i])) return CENET_EUNSUPPORTED;
    const int ppw = dwp_ppw(B * C[i], H, W, dil[i]);
    if (ppw <= 0) return CENET_EUNSUPPORTED;
    DwPlaneArgs& p = m.a[i];
    p.x = x[i]; p.sxb = sxb[i]; p.w = nullptr; p.bias = nullptr; p.y = p.a = nullptr; p.syb = p.sab = 0;
    p.dy = dy[i]; p.sgb = sgb[i]; p.dw = dw_acc[i]; p.db = nullptr;
    p.BC = B * C[i]; p.C = C[i]; p.H = H; p.W = W; p.dil = dil[i]; p.flip = 0; p.act = 0; p.ppw = ppw; p.slope = 0.f;
    const unsigned g = (unsigned)cdiv(B * C[i], ppw);
    if (g > gx) gx = g;
  }
  for (int i = n; i < DWP_MULTI_MAX; ++i) m.a[i] = m.a[0];
  CENET_LAUNCH(dw3x3_wgrad_nchw_plane_multi_kernel, dim3(gx, n), dim3(256), stream, m);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

template <typename T>
static int dwconv3x3_wgrad_tok_acc_impl(const T* x, const T* dy, float* dw_acc, float* dbias_acc, int B, int C, int H, int W,
                                        hipStream_t stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  if ((C & 3) == 0 && al16p<T>(x, dy, nullptr)) {
    CENET_LAUNCH((dw3x3_wgrad_tok_v4_kernel<T>), dim3(cdiv(C, 256), cdiv(H, 4), B), dim3(256), stream, x, dy, dw_acc, dbias_acc,
                 C, H, W);
  } else {
    CENET_LAUNCH((dw3x3_wgrad_tok_kernel<T>), dim3(cdiv(C, 256), cdiv(H, DW_WROWS), B), dim3(256), stream, x, dy, dw_acc,
                 dbias_acc, C, H, W);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(dwconv3x3_wgrad_tok_acc, (const T* x, const T* dy, float* dw_acc, float* dbias_acc, int B, int C, int H, int W,
                                     hipStream_t stream), (x, dy, dw_acc, dbias_acc, B, C, H, W, stream))
