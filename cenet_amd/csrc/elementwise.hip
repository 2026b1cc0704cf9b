// elementwise.hip — HBM-bound glue kernels of the CENet path (layout changes, gates, residual mixes, reducers).
//   token <-> NCHW transposes                 pvtv2.py:320-321, 93, 366-368
//   SiLU(g)*SiLU(v)                           cfam.py:302
//   (1-w)x + w p                              nlb.py:147
//   x + layer_scale * y                       cfam.py:368-373
//   add (+LeakyReLU)                          unet.py:212-213, decoders.py:96,100,104
//   DSEB combine: (FEA(y)+y) + diff*y         dseb.py:40-50,63-76,156-163
//   differential-attention combine + RMSNorm  multihead_diffattn.py:112-123, rms_norm.py:15-22
//   per-channel / per-column gradient reducers (bias, layer-scale, FEA weight gradients)
// Every kernel is a template over the activation storage type T (float: parity mode, bf16_t: throughput mode) — arithmetic is
// fp32 either way, parameters / statistics / parameter gradients are always fp32 — and the pointwise ones over a vector
// width V (8 / 4 / 1 elements per thread and access, picked on the host from the length and the pointer alignment).
#include "common.h"
#include <cstdlib>
#include "../../include/cenet_hip.h"

#define EW_GRID(total) dim3((unsigned)((((total) + 255) / 256) > 8192 ? 8192 : (((total) + 255) / 256)))
// launch a pointwise kernel template KERNEL<T, V> over n elements with the widest legal vector width
#define EW_LAUNCH_V(KERNEL, n, vw, ...)                                                                          \
  do {                                                                                                           \
    if ((vw) == 8) CENET_LAUNCH((KERNEL<T, 8>), EW_GRID((n) / 8), dim3(256), stream, __VA_ARGS__, (long)((n) / 8)); \
    else if ((vw) == 4) CENET_LAUNCH((KERNEL<T, 4>), EW_GRID((n) / 4), dim3(256), stream, __VA_ARGS__, (long)((n) / 4)); \
    else CENET_LAUNCH((KERNEL<T, 1>), EW_GRID(n), dim3(256), stream, __VA_ARGS__, (long)(n));                    \
  } while (0)

// ---- batched 2-D transpose: y[b][j][i] = x[b][i][j], x: [R x Cc] per batch -------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ x, long sxb, T* __restrict__ y, long syb, int R,
                                                       int Cc, const T* __restrict__ add) {
  __shared__ T tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const T* xb = x + (long)b * sxb;
  T* yb = y + (long)b * syb;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < Cc) ? xb[(long)r * Cc + c] : (T)0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < Cc) {
      // add (laid out like y): y = x^T + add — a gradient that reached the transposed tensor's source along another path
      if (add) stf(&yb[(long)c * R + r], ldf(&tile[tx][ty + 8 * i]) + ldf(add + (long)b * syb + (long)c * R + r));
      else yb[(long)c * R + r] = tile[tx][ty + 8 * i];
    }
  }
}
// bf16, both extents multiples of 2: a thread moves 2x2 blocks, so both sides are 4-byte accesses (64 x 64 tiles)
__global__ __launch_bounds__(256) void transpose_bf16x2_kernel(const bf16_t* __restrict__ x, long sxb, bf16_t* __restrict__ y,
                                                              long syb, int R, int Cc, const bf16_t* __restrict__ add) {
  __shared__ unsigned tile[64][33];  // [row][col pair]
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const bf16_t* xb = x + (long)b * sxb;
  bf16_t* yb = y + (long)b * syb;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + 2 * tx;
    unsigned v = 0;
    if (r < R && c < Cc) memcpy(&v, xb + (long)r * Cc + c, 4);
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // output row c = c0 + (ty + 8 i) * ... : thread writes the pair (r, r+1) of output row c
    const int cl = ty + 8 * i;            // local column 0..63
    const int c = c0 + cl, r = r0 + 2 * tx;
    if (c < Cc && r < R) {
      const unsigned a = tile[2 * tx][cl >> 1], bq = tile[2 * tx + 1][cl >> 1];
      const unsigned lo = (cl & 1) ? (a >> 16) : (a & 0xFFFFu), hi = (cl & 1) ? (bq >> 16) : (bq & 0xFFFFu);
      unsigned v = lo | (hi << 16);
      if (add) {
        unsigned w;
        memcpy(&w, add + (long)b * syb + (long)c * R + r, 4);
        v = cenet_pack_bf2(cenet_bf2f(lo) + cenet_bf2f(w & 0xFFFFu), cenet_bf2f(hi) + cenet_bf2f(w >> 16));
      }
      memcpy(yb + (long)c * R + r, &v, 4);
    }
  }
}

// ---- strided batch copy: y[b*syb + i] = x[b*sxb + i], i < n (channel-slice concat / split) ----------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void copy_batched_kernel(const T* __restrict__ x, long sxb, T* __restrict__ y, long syb,
                                                          int accumulate, long n) {
  const int b = blockIdx.y;
  const T* xb = x + (long)b * sxb;
  T* yb = y + (long)b * syb;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v[V];
    ldv<V>(v, xb + i * V);
    if (accumulate) {
      float o[V];
      ldv<V>(o, yb + i * V);
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] = o[e] + v[e];
    }
    stv<V>(yb + i * V, v);
  }
}

// ---- channel concat / split of up to four NCHW tensors in ONE launch (torch.cat(xs, dim=1) and its backward): part j holds
// n[j] = C_j * HW contiguous elements per image and sits at element offset off[j] inside an image of the joined tensor.
// grid (blocks, B, parts).  One launch per part (copy_batched) cost 6-7 us each for a few MB: 57 launches per step.
template <typename T>
struct CatArgs {
  T* part[4];
  const T* add[4];  // SPLIT only: part[j] = slice + add[j] (nullptr: plain copy) — the gradient that reached the part by another path
  long n[4], off[4];
  T* joined;
  long sjb;  // elements per image of the joined tensor
};
template <typename T, int V, bool SPLIT>
__global__ __launch_bounds__(256) void cat_channels_kernel(CatArgs<T> a) {
  const int j = blockIdx.z, b = blockIdx.y;
  const long n = a.n[j] / V;
  T* pb = a.part[j] + (long)b * a.n[j];
  T* jb = a.joined + (long)b * a.sjb + a.off[j];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v[V];
    ldv<V>(v, (SPLIT ? jb : pb) + i * V);
    if (SPLIT && a.add[j]) {
      float w[V];
      ldv<V>(w, a.add[j] + (long)b * a.n[j] + i * V);
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] += w[e];
    }
    stv<V>((SPLIT ? pb : jb) + i * V, v);
  }
}

// ---- space-to-depth of a token map: the kernel == stride spatial-reduction conv of pvtv2.py:93-95 reads each input
// pixel exactly once, so gathering the s x s patches into rows turns it into a dense GEMM with both operands k-contiguous.
// tok [B, Ho*S, Wo*S, C] <-> patch [B*Ho*Wo, C*S*S], k = (c, ky, kx) as in the conv weight [Cout, C, S, S].
// One thread moves the S*S values of one (patch, channel): token side coalesced over c, patch side S contiguous elements.
template <typename T, int S, bool INV>
__global__ __launch_bounds__(256) void patch_tok_kernel(const T* __restrict__ src, T* __restrict__ dst, int C, int Ho, int Wo,
                                                       long total) {
  const long Wd = (long)Wo * S;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long r = idx / C;
    const int ox = (int)(r % Wo);
    const long t = r / Wo;
    const int oy = (int)(t % Ho);
    const long b = t / Ho;
    const long tbase = (((b * Ho + oy) * S) * Wd + (long)ox * S) * C + c;
    const long pbase = (r * C + c) * (S * S);
#pragma unroll
    for (int ky = 0; ky < S; ++ky) {
      T v[S];
      if (INV) {
        memcpy(v, src + pbase + ky * S, S * sizeof(T));  // S * sizeof(T)-byte aligned: pbase is a multiple of S*S
#pragma unroll
        for (int kx = 0; kx < S; ++kx) dst[tbase + (ky * Wd + kx) * C] = v[kx];
      } else {
#pragma unroll
        for (int kx = 0; kx < S; ++kx) v[kx] = src[tbase + (ky * Wd + kx) * C];
        memcpy(dst + pbase + ky * S, v, S * sizeof(T));
      }
    }
  }
}

// Overlapping patches (the 3x3 stride-2 pad-1 patch-embedding convs of stages 2-4, pvtv2.py:164): tok [B, H*W, C] ->
// patch rows [B*Ho*Wo, C*K*K], k = (c, ky, kx) as in the conv weight, zeros outside the map; and the transpose (every input
// pixel GATHERS the patch entries that cover it: no atomics).  With the rows materialised (2.25x the map, 29 MB at stage 2)
// forward, weight gradient and data gradient are plain k-contiguous GEMMs for the LDS-DMA ring kernel instead of implicit
// GEMMs with per-element gathers (64 / 77 / 161 us -> ~12 us each at stage 2) plus these two ~10 us passes.
template <typename T, int K>
__global__ __launch_bounds__(256) void im2col_tok_kernel(const T* __restrict__ x, T* __restrict__ xp, int C, int H, int W, int Ho,
                                                        int Wo, int stride, int pad, long total) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long r = idx / C;
    const int ox = (int)(r % Wo);
    const long t = r / Wo;
    const int oy = (int)(t % Ho);
    const long b = t / Ho;
    T v[K * K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        T e;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) e = x[((b * H + iy) * W + ix) * C + c];
        else stf(&e, 0.f);
        v[ky * K + kx] = e;
      }
    T* o = xp + (r * C + c) * (K * K);
#pragma unroll
    for (int i = 0; i < K * K; ++i) o[i] = v[i];
  }
}
template <typename T, int K>
__global__ __launch_bounds__(256) void col2im_tok_kernel(const T* __restrict__ gp, T* __restrict__ dx, int C, int H, int W, int Ho,
                                                        int Wo, int stride, int pad, long total) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long r = idx / C;
    const int ix = (int)(r % W);
    const long t = r / W;
    const int iy = (int)(t % H);
    const long b = t / H;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int ny = iy + pad - ky;
      if (ny < 0 || ny % stride != 0) continue;
      const int oy = ny / stride;
      if (oy >= Ho) continue;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int nx = ix + pad - kx;
        if (nx < 0 || nx % stride != 0) continue;
        const int ox = nx / stride;
        if (ox >= Wo) continue;
        acc += ldf(gp + (((b * Ho + oy) * Wo + ox) * C + c) * (K * K) + ky * K + kx);
      }
    }
    stf(dx + idx, acc);
  }
}

// ---- the three kernels above with a row per blockIdx.y (round 5): the flat forms decompose a 64-bit element index with six 64-bit
// divisions per thread (~600 instructions in front of 4 .. 64 two-byte moves); here (image, output row) come from blockIdx.y
// (workgroup-uniform arithmetic) and (column, channel) from ONE float-reciprocal division of a small index.  Rows <= 65535.
template <typename T, int S, bool INV>
__global__ __launch_bounds__(256) void patch_tok_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int C, int Ho, int Wo) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Wo * C) return;
  const int ox = (int)(((float)i + 0.5f) * (1.f / (float)C)), c = i - ox * C;
  const int by = blockIdx.y, b = by / Ho, oy = by - b * Ho;
  const long Wd = (long)Wo * S;
  const long r = (long)by * Wo + ox;
  const long tbase = ((((long)b * Ho + oy) * S) * Wd + (long)ox * S) * C + c;
  const long pbase = (r * C + c) * (S * S);
#pragma unroll
  for (int ky = 0; ky < S; ++ky) {
    T v[S];
    if (INV) {
      memcpy(v, src + pbase + ky * S, S * sizeof(T));
#pragma unroll
      for (int kx = 0; kx < S; ++kx) dst[tbase + (ky * Wd + kx) * C] = v[kx];
    } else {
#pragma unroll
      for (int kx = 0; kx < S; ++kx) v[kx] = src[tbase + (ky * Wd + kx) * C];
      memcpy(dst + pbase + ky * S, v, S * sizeof(T));
    }
  }
}
template <typename T, int K>
__global__ __launch_bounds__(256) void im2col_tok_rows_kernel(const T* __restrict__ x, T* __restrict__ xp, int C, int H, int W, int Ho,
                                                             int Wo, int stride, int pad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Wo * C) return;
  const int ox = (int)(((float)i + 0.5f) * (1.f / (float)C)), c = i - ox * C;
  const int by = blockIdx.y, b = by / Ho, oy = by - b * Ho;
  T v[K * K];
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
      T e = x[in ? (((long)b * H + iy) * W + ix) * C + c : 0];  // (unconditional load on a clamped address)
      if (!in) stf(&e, 0.f);
      v[ky * K + kx] = e;
    }
  T* o = xp + (((long)by * Wo + ox) * C + c) * (K * K);
#pragma unroll
  for (int t = 0; t < K * K; ++t) o[t] = v[t];
}
template <typename T, int K, bool S2>
__global__ __launch_bounds__(256) void col2im_tok_rows_kernel(const T* __restrict__ gp, T* __restrict__ dx, int C, int H, int W, int Ho,
                                                             int Wo, int stride, int pad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * C) return;
  const int ix = (int)(((float)i + 0.5f) * (1.f / (float)C)), c = i - ix * C;
  const int by = blockIdx.y, b = by / H, iy = by - b * H;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int ny = iy + pad - ky;
    const int oy = S2 ? ny >> 1 : ny / stride;
    const bool yok = ny >= 0 && (S2 ? (ny & 1) == 0 : ny % stride == 0) && oy < Ho;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int nx = ix + pad - kx;
      const int ox = S2 ? nx >> 1 : nx / stride;
      const bool ok = yok && nx >= 0 && (S2 ? (nx & 1) == 0 : nx % stride == 0) && ox < Wo;
      const float g = ldf(gp + (ok ? ((((long)b * Ho + oy) * Wo + ox) * C + c) * (K * K) + ky * K + kx : 0));
      acc += ok ? g : 0.f;
    }
  }
  stf(dx + ((long)by * W + ix) * C + c, acc);
}

// ---- y[b, i] = s[b] * x[b, i] ------------------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void scale_batch_kernel(const T* __restrict__ x, const float* __restrict__ s,
                                                         T* __restrict__ y, long n) {
  const int b = blockIdx.y;
  const float sv = s[b];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v[V];
    ldv<V>(v, x + (b * n + i) * V);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = sv * v[e];
    stv<V>(y + (b * n + i) * V, v);
  }
}

// ---- dx = dy * act'(pre) -----------------------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ pre, const T* __restrict__ dy, T* __restrict__ dx,
                                                     int act, float slope, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float p[V], g[V];
    ldv<V>(p, pre + i * V);
    ldv<V>(g, dy + i * V);
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = g[e] * act_bwd(act, p[e], slope);
    stv<V>(dx + i * V, g);
  }
}
template <typename T, int V>
__global__ __launch_bounds__(256) void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int act, float slope, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v[V];
    ldv<V>(v, x + i * V);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = act_fwd(act, v[e], slope);
    stv<V>(y + i * V, v);
  }
}

// ---- y = silu(a)*silu(b) -----------------------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void silu_mul_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                                                          long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float av[V], bv[V];
    ldv<V>(av, a + i * V);
    ldv<V>(bv, b + i * V);
#pragma unroll
    for (int e = 0; e < V; ++e) av[e] = act_fwd(ACT_SILU, av[e], 0.f) * act_fwd(ACT_SILU, bv[e], 0.f);
    stv<V>(y + i * V, av);
  }
}
template <typename T, int V>
__global__ __launch_bounds__(256) void silu_mul_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                          const T* __restrict__ dy, T* __restrict__ da, T* __restrict__ db,
                                                          long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float g[V], av[V], bv[V], oa[V], ob[V];
    ldv<V>(g, dy + i * V);
    ldv<V>(av, a + i * V);
    ldv<V>(bv, b + i * V);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      oa[e] = g[e] * act_fwd(ACT_SILU, bv[e], 0.f) * act_bwd(ACT_SILU, av[e], 0.f);
      ob[e] = g[e] * act_fwd(ACT_SILU, av[e], 0.f) * act_bwd(ACT_SILU, bv[e], 0.f);
    }
    stv<V>(da + i * V, oa);
    stv<V>(db + i * V, ob);
  }
}

// ---- z = (1-w)x + w p ; w is a device scalar ---------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void mix_fwd_kernel(const T* __restrict__ x, const T* __restrict__ p,
                                                     const float* __restrict__ w, T* __restrict__ z, long n) {
  const float wv = w[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float xv[V], pv[V];
    ldv<V>(xv, x + i * V);
    ldv<V>(pv, p + i * V);
#pragma unroll
    for (int e = 0; e < V; ++e) xv[e] = (1.f - wv) * xv[e] + wv * pv[e];
    stv<V>(z + i * V, xv);
  }
}
template <typename T, int V>
__global__ __launch_bounds__(256) void mix_bwd_kernel(const T* __restrict__ x, const T* __restrict__ p,
                                                     const float* __restrict__ w, const T* __restrict__ dz, T* __restrict__ dx,
                                                     T* __restrict__ dp, float* __restrict__ dw, long n) {
  __shared__ float red[16];
  const float wv = w[0];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float g[V], xv[V], pv[V], ox[V], op[V];
    ldv<V>(g, dz + i * V);
    ldv<V>(xv, x + i * V);
    ldv<V>(pv, p + i * V);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      ox[e] = (1.f - wv) * g[e];
      op[e] = wv * g[e];
      s += g[e] * (pv[e] - xv[e]);
    }
    stv<V>(dx + i * V, ox);
    stv<V>(dp + i * V, op);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(dw, s);
}

// ---- out = x + ls[c]*y (NCHW) ; dy = ls[c]*g ; HWv = plane length in units of V -----------------------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void scale_residual_fwd_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                                const float* __restrict__ ls, T* __restrict__ out, int C,
                                                                int HWv) {
  const int bc = blockIdx.x, c = bc % C;
  const float s = ls[c];
  const long base = (long)bc * HWv * V;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < HWv; p += gridDim.y * 256) {
    float xv[V], yv[V];
    ldv<V>(xv, x + base + (long)p * V);
    ldv<V>(yv, y + base + (long)p * V);
#pragma unroll
    for (int e = 0; e < V; ++e) xv[e] = xv[e] + s * yv[e];
    stv<V>(out + base + (long)p * V, xv);
  }
}
template <typename T, int V>
__global__ __launch_bounds__(256) void scale_chan_kernel(const T* __restrict__ g, const float* __restrict__ ls,
                                                        T* __restrict__ out, int C, int HWv) {
  const int bc = blockIdx.x, c = bc % C;
  const float s = ls[c];
  const long base = (long)bc * HWv * V;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < HWv; p += gridDim.y * 256) {
    float v[V];
    ldv<V>(v, g + base + (long)p * V);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = s * v[e];
    stv<V>(out + base + (long)p * V, v);
  }
}

// ---- out[c] += sum_{b,p} a[b,c,p] * (bb ? bb[b,c,p] : 1)  (grid C x splits) -------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void chan_dot_kernel(const T* __restrict__ a, long sab, const T* __restrict__ bb, long sbb,
                                                      float* __restrict__ out, int B, int HW) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  float s = 0.f;
  if (HW < 256) {
    // small planes (7x7): the channel's (image, pixel) pairs flat over the threads (a pass per image left 49 of 256 lanes busy
    // and B dependent load rounds per workgroup)
    const long total = (long)B * HW;
    for (long e = (long)blockIdx.y * 256 + threadIdx.x; e < total; e += (long)gridDim.y * 256) {
      const int b = (int)(e / HW), p = (int)(e - (long)b * HW);
      const float av = ldf(a + (long)b * sab + (long)c * HW + p);
      s += bb ? av * ldf(bb + (long)b * sbb + (long)c * HW + p) : av;
    }
  } else {
    for (int w__ = blockIdx.y; w__ < B * ((HW + 1023) / 1024); w__ += gridDim.y)  // (image, 1024-pixel chunk) items: no per-element division
      for (int b = w__ / ((HW + 1023) / 1024), p = (w__ - b * ((HW + 1023) / 1024)) * 1024 + threadIdx.x,
               pend__ = ((w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 < HW) ? (w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 : HW;
           p < pend__; p += 256) {
        const float av = ldf(a + (long)b * sab + (long)c * HW + p);
        s += bb ? av * ldf(bb + (long)b * sbb + (long)c * HW + p) : av;
      }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(&out[c], s);
}

// quad form (HW % 4 == 0): flat grid-stride walk over the channel's B*HW/4 quads, two independent loads in flight
template <typename T>
__global__ __launch_bounds__(256) void chan_dot_v4_kernel(const T* __restrict__ a, long sab, const T* __restrict__ bb, long sbb,
                                                         float* __restrict__ out, int B, int HW) {
  __shared__ float red[16];
  const int c = blockIdx.x, nq = HW >> 2, total = B * nq, stride = gridDim.y * 256;
  const float inv_nq = cenet_inv_small(nq, total);
  float s = 0.f;
  for (int q0 = blockIdx.y * 256 + threadIdx.x; q0 < total; q0 += 2 * stride) {
    float av[2][4], bv[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = q0 + u * stride;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        av[u][e] = 0.f;
        bv[u][e] = 1.f;
      }
      if (q < total) {
        const int b = cenet_div_small(q, nq, inv_nq), qi = q - b * nq;
        ld4v(av[u], a + (long)b * sab + (long)c * HW + 4 * qi);
        if (bb) ld4v(bv[u], bb + (long)b * sbb + (long)c * HW + 4 * qi);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += av[u][e] * bv[u][e];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(&out[c], s);
}

// ---- out[c] += sum_r a[r, c]  (row-major [R, C]) ---------------------------------------------------------------------
// scalar form: block = 64 columns x 4 row-lanes, grid (col tiles, row chunks)
#define CS_ROWS 256
template <typename T>
__global__ __launch_bounds__(256) void col_sum_kernel(const T* __restrict__ a, float* __restrict__ out, long R, int C) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long r0 = (long)blockIdx.y * CS_ROWS;
  float s = 0.f;
  if (c < C)
    for (long r = r0 + rl; r < r0 + CS_ROWS && r < R; r += 4) s += ldf(a + r * C + c);
  part[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) atomicAdd(&out[c], part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl]);
}
// quad form (C % 4 == 0): a thread owns one column quad, TPR threads cover a row slab of 4*TPR columns and the other
// 256/TPR thread groups take interleaved rows, 4 independent loads in flight each
template <typename T, int TPR>
__global__ __launch_bounds__(256) void col_sum_v4_kernel(const T* __restrict__ a, float* __restrict__ out, long R, int C,
                                                        int rows_per_block) {
  constexpr int RL = 256 / TPR;
  __shared__ float part[RL][TPR * 4];
  const int q = threadIdx.x % TPR, rl = threadIdx.x / TPR;
  const int c = (blockIdx.x * TPR + q) * 4;
  const long r0 = (long)blockIdx.y * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    long r = r0 + rl;
    for (; r + 3 * RL < r1; r += 4 * RL) {
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ld4v(v[u], a + (r + u * RL) * C + c);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += v[u][e];
    }
    for (; r < r1; r += RL) {
      float v[4];
      ld4v(v, a + r * C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[rl][q * 4 + e] = s[e];
  __syncthreads();
  for (int i = threadIdx.x; i < TPR * 4; i += 256) {
    const int cc = blockIdx.x * TPR * 4 + i;
    if (cc < C) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < RL; ++j) t += part[j][i];
      atomicAdd(&out[cc], t);
    }
  }
}

// ---- out = act(a + b) ; backward from the output sign (LeakyReLU/ReLU preserve sign) -------------------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void add_act_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                         int act, float slope, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float av[V], bv[V];
    ldv<V>(av, a + i * V);
    ldv<V>(bv, b + i * V);
#pragma unroll
    for (int e = 0; e < V; ++e) av[e] = act_fwd(act, av[e] + bv[e], slope);
    stv<V>(out + i * V, av);
  }
}
template <typename T, int V>
__global__ __launch_bounds__(256) void lrelu_bwd_from_out_kernel(const T* __restrict__ out, const T* __restrict__ dy,
                                                                T* __restrict__ dx, float slope, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float o[V], g[V];
    ldv<V>(o, out + i * V);
    ldv<V>(g, dy + i * V);
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = o[e] > 0.f ? g[e] : g[e] * slope;
    stv<V>(dx + i * V, g);
  }
}

// ---- DSEB combine ---------------------------------------------------------------------------------------------------
// z = 2y + w[c]*edge + diff*y,  edge = (1/m) sum_{i<j} | e_i - e_j |,  e_s = | y - r_s |  (r_s == nullptr: scale 1.0 -> e_s = 0)
template <typename T>
struct DsebArgs {
  const T* y;
  const T* r[3];
  const float* w;
  const T* diff;
  T* z;
  // backward
  const T* dz;
  T* dy;
  T* dr[3];
  T* ddiff;
  float* dw;
  int n, C, HW;  // HW in units of V
  float ycoef;
};
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

template <typename T, int V>
__global__ __launch_bounds__(256) void dseb_combine_fwd_kernel(DsebArgs<T> a) {
  const int bc = blockIdx.x, c = bc % a.C;
  const float wc = a.w[c];
  const long base = (long)bc * a.HW * V;
  const float inv_m = 1.f / (float)(a.n * (a.n - 1) / 2);
  for (int p = blockIdx.y * 256 + threadIdx.x; p < a.HW; p += gridDim.y * 256) {
    const long o = base + (long)p * V;
    float yv[V], rv[3][V], dv[V], zv[V];
    ldv<V>(yv, a.y + o);
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (s < a.n && a.r[s]) ldv<V>(rv[s], a.r[s] + o);
    if (a.diff) ldv<V>(dv, a.diff + o);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float e[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) e[s] = (s < a.n && a.r[s]) ? fabsf(yv[k] - rv[s][k]) : 0.f;
      float edge = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i + 1; j < 3; ++j)
          if (j < a.n) edge += fabsf(e[i] - e[j]);
      edge *= inv_m;
      zv[k] = a.ycoef * yv[k] + wc * edge + (a.diff ? dv[k] * yv[k] : 0.f);
    }
    stv<V>(a.z + o, zv);
  }
}
template <typename T, int V>
__global__ __launch_bounds__(256) void dseb_combine_bwd_kernel(DsebArgs<T> a) {
  __shared__ float red[16];
  const int bc = blockIdx.x, c = bc % a.C;
  const float wc = a.w[c];
  const long base = (long)bc * a.HW * V;
  const float inv_m = 1.f / (float)(a.n * (a.n - 1) / 2);
  float dws = 0.f;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < a.HW; p += gridDim.y * 256) {
    const long o = base + (long)p * V;
    float yv[V], gv[V], dfv[V], rv[3][V], dyo[V], dro[3][V], ddo[V];
    ldv<V>(yv, a.y + o);
    ldv<V>(gv, a.dz + o);
    if (a.diff) ldv<V>(dfv, a.diff + o);
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (s < a.n && a.r[s]) ldv<V>(rv[s], a.r[s] + o);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float g = gv[k], df = a.diff ? dfv[k] : 0.f;
      float e[3], sy[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (s < a.n && a.r[s]) {
          const float d = yv[k] - rv[s][k];
          e[s] = fabsf(d);
          sy[s] = sgn(d);
        } else {
          e[s] = 0.f;
          sy[s] = 0.f;
        }
      }
      float edge = 0.f, de[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i + 1; j < 3; ++j)
          if (j < a.n) {
            const float d = e[i] - e[j];
            edge += fabsf(d);
            const float sd = sgn(d);
            de[i] += sd;
            de[j] -= sd;
          }
      edge *= inv_m;
      dws += g * edge;
      float dyv = a.ycoef * g + g * df;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        dro[s][k] = 0.f;
        if (s < a.n && a.r[s]) {
          const float t = wc * g * inv_m * de[s] * sy[s];
          dyv += t;
          dro[s][k] = -t;
        }
      }
      dyo[k] = dyv;
      ddo[k] = g * yv[k];
    }
    stv<V>(a.dy + o, dyo);
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (s < a.n && a.r[s]) stv<V>(a.dr[s] + o, dro[s]);
    if (a.ddiff) stv<V>(a.ddiff + o, ddo);
  }
  dws = block_sum(dws, red);
  if (threadIdx.x == 0) atomicAdd(&a.dw[c], dws);
}

// ---- differential attention: lambda, combine + RMSNorm ------------------------------------------------------------
// lam[0] = exp(<q1,k1>) - exp(<q2,k2>) + lambda_init ; lam[1] = exp(<q1,k1>) ; lam[2] = exp(<q2,k2>)
__global__ __launch_bounds__(64) void diffattn_lambda_fwd_kernel(const float* q1, const float* k1, const float* q2,
                                                                const float* k2, float lambda_init, float* lam, int hd) {
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < hd; i += 64) {
    s1 += q1[i] * k1[i];
    s2 += q2[i] * k2[i];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (threadIdx.x == 0) {
    const float l1 = expf(s1), l2 = expf(s2);
    lam[0] = l1 - l2 + lambda_init;
    lam[1] = l1;
    lam[2] = l2;
  }
}
__global__ __launch_bounds__(64) void diffattn_lambda_bwd_kernel(const float* q1, const float* k1, const float* q2,
                                                                const float* k2, const float* lam, const float* dlam,
                                                                float* dq1, float* dk1, float* dq2, float* dk2, int hd) {
  const float g1 = dlam[0] * lam[1], g2 = -dlam[0] * lam[2];
  for (int i = threadIdx.x; i < hd; i += 64) {
    atomicAdd(&dq1[i], g1 * k1[i]);
    atomicAdd(&dk1[i], g1 * q1[i]);
    atomicAdd(&dq2[i], g2 * k2[i]);
    atomicAdd(&dk2[i], g2 * q2[i]);
  }
}
// U [B, 2H, N, dv] -> out [B, N, H*dv]: a = U[2h] - lam*U[2h+1]; out = a * rsqrt(mean(a^2)+eps) * post.
// A (b,h,n) vector is handled by a SUB-wave of `sub` lanes (16/32/64 >= min(dv,64)), so short vectors fill the wave.
__device__ __forceinline__ float subwave_sum(float v, int sub) {
  for (int o = sub >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <typename T>
__global__ __launch_bounds__(256) void diffattn_combine_fwd_kernel(const T* __restrict__ U, const float* __restrict__ lam,
                                                                  T* __restrict__ out, int H, int N, int dv, float eps,
                                                                  float post, long nvec, int sub) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int vpw = 64 / sub, sl = lane & (sub - 1);
  const long vec = ((long)blockIdx.x * 4 + wave) * vpw + lane / sub;
  const bool ok = vec < nvec;
  const int n = ok ? (int)(vec % N) : 0;
  const long bh = ok ? vec / N : 0;
  const int h = (int)(bh % H);
  const long b = bh / H;
  const float lm = lam[0];
  const T* u0 = U + (((b * 2 * H) + 2 * h) * N + n) * (long)dv;
  const T* u1 = u0 + (long)N * dv;
  float ss = 0.f;
  if (ok)
    for (int d = sl; d < dv; d += sub) {
      const float av = ldf(u0 + d) - lm * ldf(u1 + d);
      ss += av * av;
    }
  const float r = rsqrtf(subwave_sum(ss, sub) / dv + eps) * post;
  if (ok) {
    T* o = out + (b * N + n) * (long)(H * dv) + (long)h * dv;
    for (int d = sl; d < dv; d += sub) stf(o + d, (ldf(u0 + d) - lm * ldf(u1 + d)) * r);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void diffattn_combine_bwd_kernel(const T* __restrict__ U, const float* __restrict__ lam,
                                                                  const T* __restrict__ dout, T* __restrict__ dU,
                                                                  float* __restrict__ dlam, int H, int N, int dv, float eps,
                                                                  float post, long nvec, int sub) {
  __shared__ float part[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int vpw = 64 / sub, sl = lane & (sub - 1);
  const float lm = lam[0];
  float dl = 0.f;
  // grid-stride over the (b, h, n) vectors: one dlam atomic per workgroup, and few workgroups (the atomics all hit one
  // address and serialise at ~12 ns each)
  for (long v0 = (long)blockIdx.x * 4 * vpw; v0 < nvec; v0 += (long)gridDim.x * 4 * vpw) {
    const long vec = v0 + wave * vpw + lane / sub;
    const bool ok = vec < nvec;
    const int n = ok ? (int)(vec % N) : 0;
    const long bh = ok ? vec / N : 0;
    const int h = (int)(bh % H);
    const long b = bh / H;
    const long off0 = (((b * 2 * H) + 2 * h) * N + n) * (long)dv;
    const T* u0 = U + off0;
    const T* u1 = u0 + (long)N * dv;
    const T* g = dout + (b * N + n) * (long)(H * dv) + (long)h * dv;
    float ss = 0.f, ga = 0.f;
    if (ok)
      for (int d = sl; d < dv; d += sub) {
        const float av = ldf(u0 + d) - lm * ldf(u1 + d);
        ss += av * av;
        ga += ldf(g + d) * av;
      }
    ss = subwave_sum(ss, sub);
    ga = subwave_sum(ga, sub);
    const float r = rsqrtf(ss / dv + eps);
    const float k = r * r * ga / dv;
    if (ok) {
      T* d0 = dU + off0;
      T* d1 = d0 + (long)N * dv;
      for (int d = sl; d < dv; d += sub) {
        const float u1v = ldf(u1 + d);
        const float av = ldf(u0 + d) - lm * u1v;
        const float da = post * r * (ldf(g + d) - av * k);
        stf(d0 + d, da);
        stf(d1 + d, -lm * da);
        dl -= da * u1v;
      }
    }
  }
  dl = wave_sum(dl);
  if (lane == 0) part[wave] = dl;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dlam, part[0] + part[1] + part[2] + part[3]);
}

// ---- 16-byte forms of the two combine kernels (bf16, dv a multiple of 8): a lane owns 8 consecutive features of a (b, h, n)
// vector, LPV = dv / 8 lanes share one, the values stay in registers between the statistics and the write (the scalar forms
// above read every element twice through 2-byte loads: 62 us for 154 MB at 56x56).
template <int LPV>
__device__ __forceinline__ float lpv_sum(float v) {  // sum over aligned groups of LPV lanes, every lane of the group gets it
#ifdef CENET_HOSTSIM_BUILD
#pragma unroll
  for (int o = LPV >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
#else
  if (LPV >= 2) v += dpp_f<0xb1, 0xf>(0.f, v);  // quad_perm [1, 0, 3, 2]
  if (LPV >= 4) v += dpp_f<0x4e, 0xf>(0.f, v);  // quad_perm [2, 3, 0, 1]
#pragma unroll
  for (int o = 4; o < LPV; o <<= 1) v += __shfl_xor(v, o);
#endif
  return v;
}
// grid (ceil(N * LPV / 256), B * H): (batch, pair) from blockIdx.y, the token from a shift — the flat forms spent four 64-bit
// divisions per thread (~400 instructions) on eight elements
template <int LPV>
__global__ __launch_bounds__(256) void diffattn_combine_fwd_v8_kernel(const bf16_t* __restrict__ U, const float* __restrict__ lam,
                                                                     bf16_t* __restrict__ out, int H, int N, float eps, float post,
                                                                     int dv) {
  // (dv == 8 LPV, or LPV == 32 with dv / 8 < 32 active lanes per vector: head dimension 80 -> dv = 160 = 20 lanes)
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int nn = t / LPV, sl = t - nn * LPV;
  const bool act = 8 * sl < dv;
  const bool ok = nn < N && act;  // (LPV divides 64: the lanes of a vector leave together)
  const int n = nn < N ? nn : 0;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const float lm = lam[0];
  const bf16_t* u0 = U + ((((long)b * 2 * H) + 2 * h) * N + n) * (long)dv + (act ? 8 * sl : 0);
  float a0[8], a1[8];
  ldv<8>(a0, u0);
  ldv<8>(a1, u0 + (long)N * dv);
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a0[e] -= lm * a1[e];
    ss += act ? a0[e] * a0[e] : 0.f;
  }
  const float r = rsqrtf(lpv_sum<LPV>(ss) / dv + eps) * post;
#pragma unroll
  for (int e = 0; e < 8; ++e) a0[e] *= r;
  if (ok) stv<8>(out + ((long)b * N + n) * (long)(H * dv) + (long)h * dv + 8 * sl, a0);
}
// grid (ceil(N * LPV / 2048), B * H): eight vectors per thread, two at a time with their loads issued together; ONE dlam atomic per
// workgroup (they all hit one address and retire at ~12 ns each: 3 200 workgroups were 38 us of atomics behind 16 us of traffic)
template <int LPV>
__global__ __launch_bounds__(256) void diffattn_combine_bwd_v8_kernel(const bf16_t* __restrict__ U, const float* __restrict__ lam,
                                                                     const bf16_t* __restrict__ dout, bf16_t* __restrict__ dU,
                                                                     float* __restrict__ dlam, int H, int N, float eps, float post,
                                                                     int dv) {
  __shared__ float part[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const float lm = lam[0];
  float dl = 0.f;
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
    float a[2][8], u1[2][8], g[2][8];
    long off0[2];
    bool ok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int t = ((blockIdx.x * 4 + it) * 2 + k) * 256 + threadIdx.x;
      const int nn = t / LPV, sl = t - nn * LPV;
      const bool act = 8 * sl < dv;
      ok[k] = nn < N && act;
      const int n = nn < N ? nn : 0, so = act ? 8 * sl : 0;
      off0[k] = ((((long)b * 2 * H) + 2 * h) * N + n) * (long)dv + so;
      ldv<8>(a[k], U + off0[k]);
      ldv<8>(u1[k], U + off0[k] + (long)N * dv);
      ldv<8>(g[k], dout + ((long)b * N + n) * (long)(H * dv) + (long)h * dv + so);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float ss = 0.f, ga = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[k][e] -= lm * u1[k][e];
        ss += ok[k] ? a[k][e] * a[k][e] : 0.f;
        ga += ok[k] ? g[k][e] * a[k][e] : 0.f;
      }
      ss = lpv_sum<LPV>(ss);
      ga = lpv_sum<LPV>(ga);
      const float r = rsqrtf(ss / dv + eps);
      const float kk = r * r * ga / dv;
      float d0[8], d1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float da = post * r * (g[k][e] - a[k][e] * kk);
        d0[e] = da;
        d1[e] = -lm * da;
        if (ok[k]) dl -= da * u1[k][e];
      }
      if (ok[k]) {
        stv<8>(dU + off0[k], d0);
        stv<8>(dU + off0[k] + (long)N * dv, d1);
      }
    }
  }
  dl = wave_sum(dl);
  if (lane == 0) part[wave] = dl;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dlam, part[0] + part[1] + part[2] + part[3]);
}

// ------------------------------------------------------------------------------------------------------------------
static inline int chunks_for(int n) {
  int ch = cdiv(n, 1024);
  return ch > 64 ? 64 : (ch < 1 ? 1 : ch);
}

template <typename T>
static int transpose_impl(const T* x, long sxb, T* y, long syb, int B, int R, int Cc, hipStream_t stream,
                          const T* add = nullptr) {
  if (B <= 0 || R <= 0 || Cc <= 0) return CENET_EINVAL;
  CENET_LAUNCH((transpose_kernel<T>), dim3(cdiv(Cc, 32), cdiv(R, 32), B), dim3(256), stream, x, sxb, y, syb, R, Cc, add);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
template <>
int transpose_impl<bf16_t>(const bf16_t* x, long sxb, bf16_t* y, long syb, int B, int R, int Cc, hipStream_t stream,
                           const bf16_t* add) {
  if (B <= 0 || R <= 0 || Cc <= 0) return CENET_EINVAL;
  if (((R | Cc) & 1) == 0 && ((sxb | syb) & 1) == 0 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)add) & 3) == 0))
    CENET_LAUNCH(transpose_bf16x2_kernel, dim3(cdiv(Cc, 64), cdiv(R, 64), B), dim3(256), stream, x, sxb, y, syb, R, Cc, add);
  else
    CENET_LAUNCH((transpose_kernel<bf16_t>), dim3(cdiv(Cc, 32), cdiv(R, 32), B), dim3(256), stream, x, sxb, y, syb, R, Cc, add);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(transpose, (const T* x, long sxb, T* y, long syb, int B, int R, int Cc, hipStream_t stream),
           (x, sxb, y, syb, B, R, Cc, stream))
// y = x^T + add (add laid out like y, batch stride syb): the backward of a layout change whose source has a second consumer
// (pvtv2.py:320-321: a stage's token output feeds the decoder as NCHW and the next stage's patch embedding as tokens)
template <typename T>
static int transpose_add_impl(const T* x, long sxb, T* y, long syb, const T* add, int B, int R, int Cc, hipStream_t stream) {
  return transpose_impl<T>(x, sxb, y, syb, B, R, Cc, stream, add);
}
CENET_TWIN(transpose_add, (const T* x, long sxb, T* y, long syb, const T* add, int B, int R, int Cc, hipStream_t stream),
           (x, sxb, y, syb, add, B, R, Cc, stream))

template <typename T>
static int copy_batched_impl(const T* x, long sxb, T* y, long syb, int B, long n, int accumulate, hipStream_t stream) {
  if (B <= 0 || n <= 0) return CENET_EINVAL;
  int vw = vec_width<T>(n, x, y);
  if (vw > 1 && ((sxb | syb) % vw) != 0) vw = (vw == 8 && ((sxb | syb) & 3) == 0) ? 4 : 1;
  long blocks = (n / vw + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const dim3 grid((unsigned)blocks, B);
  if (vw == 8) CENET_LAUNCH((copy_batched_kernel<T, 8>), grid, dim3(256), stream, x, sxb, y, syb, accumulate, n / 8);
  else if (vw == 4) CENET_LAUNCH((copy_batched_kernel<T, 4>), grid, dim3(256), stream, x, sxb, y, syb, accumulate, n / 4);
  else CENET_LAUNCH((copy_batched_kernel<T, 1>), grid, dim3(256), stream, x, sxb, y, syb, accumulate, n);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(copy_batched, (const T* x, long sxb, T* y, long syb, int B, long n, int accumulate, hipStream_t stream),
           (x, sxb, y, syb, B, n, accumulate, stream))

template <typename T>
static int cat_channels_impl(T* p0, T* p1, T* p2, T* p3, int c0, int c1, int c2, int c3, T* joined, int B, long HW, int split,
                             hipStream_t stream, const T* a0 = nullptr, const T* a1 = nullptr, const T* a2 = nullptr,
                             const T* a3 = nullptr) {
  T* ps[4] = {p0, p1, p2, p3};
  const T* as[4] = {a0, a1, a2, a3};
  if (!split && (a0 || a1 || a2 || a3)) return CENET_EINVAL;
  const int cs[4] = {c0, c1, c2, c3};
  if (!joined || B <= 0 || HW <= 0) return CENET_EINVAL;
  CatArgs<T> a;
  a.joined = joined;
  int parts = 0;
  long off = 0, nmax = 0;
  uintptr_t align = (uintptr_t)joined;
  long lens = 0;
  for (int j = 0; j < 4; ++j) {
    a.part[j] = nullptr;
    a.add[j] = nullptr;
    a.n[j] = a.off[j] = 0;
    if (cs[j] <= 0) continue;
    if (!ps[j] || parts != j) return CENET_EINVAL;  // parts are given front to back without holes
    a.part[j] = ps[j];
    a.add[j] = as[j];
    align |= (uintptr_t)as[j];
    a.n[j] = (long)cs[j] * HW;
    a.off[j] = off;
    off += a.n[j];
    align |= (uintptr_t)ps[j];
    lens |= a.n[j];
    if (a.n[j] > nmax) nmax = a.n[j];
    ++parts;
  }
  if (parts == 0) return CENET_EINVAL;
  a.sjb = off;
  const int vmax = 16 / (int)sizeof(T);  // 16-byte moves when every part length and pointer allows
  int vw = 1;
  if ((lens % vmax) == 0 && (align % 16) == 0) vw = vmax;
  else if ((lens % 4) == 0 && (align % (4 * sizeof(T))) == 0) vw = 4;
  long blocks = (nmax / vw + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const dim3 grid((unsigned)blocks, B, parts);
#define CAT_GO(V_)                                                                                         \
  do {                                                                                                     \
    if (split) CENET_LAUNCH((cat_channels_kernel<T, V_, true>), grid, dim3(256), stream, a);               \
    else CENET_LAUNCH((cat_channels_kernel<T, V_, false>), grid, dim3(256), stream, a);                    \
  } while (0)
  if (vw == 8) CAT_GO(8);
  else if (vw == 4) CAT_GO(4);
  else CAT_GO(1);
#undef CAT_GO
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(cat_channels, (T* p0, T* p1, T* p2, T* p3, int c0, int c1, int c2, int c3, T* joined, int B, long HW, int split,
                          hipStream_t stream),
           (p0, p1, p2, p3, c0, c1, c2, c3, joined, B, HW, split, stream))
// the split with addends: part j = its channel slice of `joined` + a_j (a_j == NULL: the slice alone) — the backward of a channel
// concat whose inputs have further consumers (dseb.py:156: `dec` also feeds the level's residual add, `skip` the mixer's): the
// gradients that reached them by those paths are added here instead of by an elementwise add each
template <typename T>
static int split_channels_add_impl(T* p0, T* p1, T* p2, T* p3, const T* a0, const T* a1, const T* a2, const T* a3, int c0, int c1,
                                   int c2, int c3, const T* joined, int B, long HW, hipStream_t stream) {
  return cat_channels_impl<T>(p0, p1, p2, p3, c0, c1, c2, c3, const_cast<T*>(joined), B, HW, 1, stream, a0, a1, a2, a3);
}
CENET_TWIN(split_channels_add, (T* p0, T* p1, T* p2, T* p3, const T* a0, const T* a1, const T* a2, const T* a3, int c0, int c1,
                                int c2, int c3, const T* joined, int B, long HW, hipStream_t stream),
           (p0, p1, p2, p3, a0, a1, a2, a3, c0, c1, c2, c3, joined, B, HW, stream))

template <typename T>
static int patch_tok_impl(const T* src, T* dst, int B, int Ho, int Wo, int C, int S, int inverse, hipStream_t stream) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (S != 2 && S != 4 && S != 8)) return CENET_EINVAL;
  const long total = (long)B * Ho * Wo * C;
  const bool rows = (long)B * Ho <= 65535 && (long)Wo * C < (1L << 20);
#define PATCH_GO(S_, INV_)                                                                                                \
  do {                                                                                                                    \
    if (rows)                                                                                                             \
      CENET_LAUNCH((patch_tok_rows_kernel<T, S_, INV_>), dim3(cdiv((long)Wo * C, 256), B * Ho), dim3(256), stream, src, dst, C, Ho, \
                   Wo);                                                                                                   \
    else                                                                                                                  \
      CENET_LAUNCH((patch_tok_kernel<T, S_, INV_>), EW_GRID(total), dim3(256), stream, src, dst, C, Ho, Wo, total);        \
  } while (0)
  if (S == 2) {
    if (inverse) PATCH_GO(2, true); else PATCH_GO(2, false);
  } else if (S == 4) {
    if (inverse) PATCH_GO(4, true); else PATCH_GO(4, false);
  } else {
    if (inverse) PATCH_GO(8, true); else PATCH_GO(8, false);
  }
#undef PATCH_GO
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(patch_tok, (const T* src, T* dst, int B, int Ho, int Wo, int C, int S, int inverse, hipStream_t stream),
           (src, dst, B, Ho, Wo, C, S, inverse, stream))

template <typename T>
static int im2col_tok_impl(const T* src, T* dst, int B, int H, int W, int C, int K, int stride, int pad, int inverse,
                           hipStream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || K != 3 || stride < 1 || pad < 0) return CENET_EINVAL;
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return CENET_EINVAL;
  if (inverse && (long)B * H <= 65535 && (long)W * C < (1L << 20)) {
    const dim3 grid(cdiv((long)W * C, 256), B * H);
    if (stride == 2) CENET_LAUNCH((col2im_tok_rows_kernel<T, 3, true>), grid, dim3(256), stream, src, dst, C, H, W, Ho, Wo, stride, pad);
    else CENET_LAUNCH((col2im_tok_rows_kernel<T, 3, false>), grid, dim3(256), stream, src, dst, C, H, W, Ho, Wo, stride, pad);
  } else if (inverse) {
    const long total = (long)B * H * W * C;
    CENET_LAUNCH((col2im_tok_kernel<T, 3>), EW_GRID(total), dim3(256), stream, src, dst, C, H, W, Ho, Wo, stride, pad, total);
  } else if ((long)B * Ho <= 65535 && (long)Wo * C < (1L << 20)) {
    CENET_LAUNCH((im2col_tok_rows_kernel<T, 3>), dim3(cdiv((long)Wo * C, 256), B * Ho), dim3(256), stream, src, dst, C, H, W, Ho, Wo,
                 stride, pad);
  } else {
    const long total = (long)B * Ho * Wo * C;
    CENET_LAUNCH((im2col_tok_kernel<T, 3>), EW_GRID(total), dim3(256), stream, src, dst, C, H, W, Ho, Wo, stride, pad, total);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(im2col_tok, (const T* src, T* dst, int B, int H, int W, int C, int K, int stride, int pad, int inverse,
                        hipStream_t stream), (src, dst, B, H, W, C, K, stride, pad, inverse, stream))

template <typename T>
static int scale_batch_impl(const T* x, const float* s, T* y, int B, long n, hipStream_t stream) {
  if (B <= 0 || n <= 0) return CENET_EINVAL;
  const int vw = vec_width<T>(n, x, y);
  long blocks = (n / vw + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const dim3 grid((unsigned)blocks, B);
  if (vw == 8) CENET_LAUNCH((scale_batch_kernel<T, 8>), grid, dim3(256), stream, x, s, y, n / 8);
  else if (vw == 4) CENET_LAUNCH((scale_batch_kernel<T, 4>), grid, dim3(256), stream, x, s, y, n / 4);
  else CENET_LAUNCH((scale_batch_kernel<T, 1>), grid, dim3(256), stream, x, s, y, n);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(scale_batch, (const T* x, const float* s, T* y, int B, long n, hipStream_t stream), (x, s, y, B, n, stream))

template <typename T>
static int act_fwd_impl(const T* x, T* y, long n, int act, float slope, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  EW_LAUNCH_V(act_fwd_kernel, n, vec_width<T>(n, x, y), x, y, act, slope);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(act_fwd, (const T* x, T* y, long n, int act, float slope, hipStream_t stream), (x, y, n, act, slope, stream))

template <typename T>
static int act_bwd_impl(const T* pre, const T* dy, T* dx, long n, int act, float slope, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  EW_LAUNCH_V(act_bwd_kernel, n, vec_width<T>(n, pre, dy, dx), pre, dy, dx, act, slope);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(act_bwd, (const T* pre, const T* dy, T* dx, long n, int act, float slope, hipStream_t stream),
           (pre, dy, dx, n, act, slope, stream))

template <typename T>
static int silu_mul_fwd_impl(const T* a, const T* b, T* y, long n, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  EW_LAUNCH_V(silu_mul_fwd_kernel, n, vec_width<T>(n, a, b, y), a, b, y);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(silu_mul_fwd, (const T* a, const T* b, T* y, long n, hipStream_t stream), (a, b, y, n, stream))

template <typename T>
static int silu_mul_bwd_impl(const T* a, const T* b, const T* dy, T* da, T* db, long n, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  int vw = vec_width<T>(n, a, b, dy, da, db);
  if (vw == 8) vw = 4;  // five streams of 8 would not fit the register budget of a grid-stride loop comfortably
  EW_LAUNCH_V(silu_mul_bwd_kernel, n, vw, a, b, dy, da, db);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(silu_mul_bwd, (const T* a, const T* b, const T* dy, T* da, T* db, long n, hipStream_t stream),
           (a, b, dy, da, db, n, stream))

template <typename T>
static int mix_fwd_impl(const T* x, const T* p, const float* w, T* z, long n, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  EW_LAUNCH_V(mix_fwd_kernel, n, vec_width<T>(n, x, p, z), x, p, w, z);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(mix_fwd, (const T* x, const T* p, const float* w, T* z, long n, hipStream_t stream), (x, p, w, z, n, stream))

template <typename T>
static int mix_bwd_acc_impl(const T* x, const T* p, const float* w, const T* dz, T* dx, T* dp, float* dw_acc, long n,
                            hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  int vw = vec_width<T>(n, x, p, dz, dx, dp);
  if (vw == 8) vw = 4;
  if (sizeof(T) == 4) vw = 1;  // fp32 keeps the one-element-per-thread walk: the summation order of dw in parity mode is pinned
  long blocks = (n / vw + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (vw == 4) CENET_LAUNCH((mix_bwd_kernel<T, 4>), dim3((unsigned)blocks), dim3(256), stream, x, p, w, dz, dx, dp, dw_acc, n / 4);
  else CENET_LAUNCH((mix_bwd_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), stream, x, p, w, dz, dx, dp, dw_acc, n);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(mix_bwd_acc, (const T* x, const T* p, const float* w, const T* dz, T* dx, T* dp, float* dw_acc, long n,
                         hipStream_t stream), (x, p, w, dz, dx, dp, dw_acc, n, stream))

template <typename T>
static int scale_residual_fwd_impl(const T* x, const T* y, const float* ls, T* out, int B, int C, int HW, hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  const int vw = vec_width<T>(HW, x, y, out);
  const dim3 grid(B * C, chunks_for(HW / vw));
  if (vw == 8) CENET_LAUNCH((scale_residual_fwd_kernel<T, 8>), grid, dim3(256), stream, x, y, ls, out, C, HW / 8);
  else if (vw == 4) CENET_LAUNCH((scale_residual_fwd_kernel<T, 4>), grid, dim3(256), stream, x, y, ls, out, C, HW / 4);
  else CENET_LAUNCH((scale_residual_fwd_kernel<T, 1>), grid, dim3(256), stream, x, y, ls, out, C, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(scale_residual_fwd, (const T* x, const T* y, const float* ls, T* out, int B, int C, int HW, hipStream_t stream),
           (x, y, ls, out, B, C, HW, stream))

template <typename T>
static int scale_chan_impl(const T* g, const float* ls, T* out, int B, int C, int HW, hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  const int vw = vec_width<T>(HW, g, out);
  const dim3 grid(B * C, chunks_for(HW / vw));
  if (vw == 8) CENET_LAUNCH((scale_chan_kernel<T, 8>), grid, dim3(256), stream, g, ls, out, C, HW / 8);
  else if (vw == 4) CENET_LAUNCH((scale_chan_kernel<T, 4>), grid, dim3(256), stream, g, ls, out, C, HW / 4);
  else CENET_LAUNCH((scale_chan_kernel<T, 1>), grid, dim3(256), stream, g, ls, out, C, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(scale_chan, (const T* g, const float* ls, T* out, int B, int C, int HW, hipStream_t stream),
           (g, ls, out, B, C, HW, stream))

template <typename T>
static int chan_dot_acc_impl(const T* a, long sab, const T* b, long sbb, float* out_acc, int B, int C, int HW,
                             hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  long total = (long)B * HW;
  long want = 1024 / C, maxs = (total + 2047) / 2048;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  if ((HW & 3) == 0 && ((sab | sbb) & 3) == 0 && quad_aligned<T>(a) && quad_aligned<T>(b)) {
    long w4 = 2048 / C, m4 = (total / 4 + 1023) / 1024;
    if (w4 > m4) w4 = m4;
    if (w4 < 1) w4 = 1;
    if (w4 > 256) w4 = 256;
    CENET_LAUNCH((chan_dot_v4_kernel<T>), dim3(C, (unsigned)w4), dim3(256), stream, a, sab, b, sbb, out_acc, B, HW);
  } else {
    CENET_LAUNCH((chan_dot_kernel<T>), dim3(C, (unsigned)want), dim3(256), stream, a, sab, b, sbb, out_acc, B, HW);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(chan_dot_acc, (const T* a, long sab, const T* b, long sbb, float* out_acc, int B, int C, int HW, hipStream_t stream),
           (a, sab, b, sbb, out_acc, B, C, HW, stream))

template <typename T>
static int col_sum_acc_impl(const T* a, float* out_acc, long R, int C, hipStream_t stream) {
  if (R <= 0 || C <= 0) return CENET_EINVAL;
  if ((C & 3) == 0 && quad_aligned<T>(a)) {
    // ~2048 workgroups, each at least 64 rows deep — but every workgroup ends with one float atomic per column, and atomics onto the
    // same cache line serialise: with few columns (C <= 128: 2 - 4 lines) 1 568 workgroups spent most of a 40 us launch queueing on
    // them (12.8 MB of data = 3 us); ~256 workgroups there
    const int quads = C / 4;
    const int tpr = quads >= 64 ? 64 : (quads >= 32 ? 32 : 16);
    const int ctiles = cdiv(quads, tpr);
    static const int wg_small = getenv("CENET_COLSUM_WGS") ? atoi(getenv("CENET_COLSUM_WGS")) : 256;  // measurement aid
    const long target = C <= 128 ? wg_small : 2048;
    long rpb = (R * ctiles + target - 1) / target;
    if (rpb < 64) rpb = 64;
    rpb = (rpb + 15) & ~15L;
    dim3 grid(ctiles, (unsigned)((R + rpb - 1) / rpb));
    if (tpr == 64) CENET_LAUNCH((col_sum_v4_kernel<T, 64>), grid, dim3(256), stream, a, out_acc, R, C, (int)rpb);
    else if (tpr == 32) CENET_LAUNCH((col_sum_v4_kernel<T, 32>), grid, dim3(256), stream, a, out_acc, R, C, (int)rpb);
    else CENET_LAUNCH((col_sum_v4_kernel<T, 16>), grid, dim3(256), stream, a, out_acc, R, C, (int)rpb);
  } else {
    CENET_LAUNCH((col_sum_kernel<T>), dim3(cdiv(C, 64), (unsigned)((R + CS_ROWS - 1) / CS_ROWS)), dim3(256), stream, a, out_acc,
                 R, C);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(col_sum_acc, (const T* a, float* out_acc, long R, int C, hipStream_t stream), (a, out_acc, R, C, stream))

template <typename T>
static int add_act_fwd_impl(const T* a, const T* b, T* out, long n, int act, float slope, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  EW_LAUNCH_V(add_act_fwd_kernel, n, vec_width<T>(n, a, b, out), a, b, out, act, slope);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(add_act_fwd, (const T* a, const T* b, T* out, long n, int act, float slope, hipStream_t stream),
           (a, b, out, n, act, slope, stream))

template <typename T>
static int lrelu_bwd_from_out_impl(const T* out, const T* dy, T* dx, long n, float slope, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  EW_LAUNCH_V(lrelu_bwd_from_out_kernel, n, vec_width<T>(n, out, dy, dx), out, dy, dx, slope);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(lrelu_bwd_from_out, (const T* out, const T* dy, T* dx, long n, float slope, hipStream_t stream),
           (out, dy, dx, n, slope, stream))

template <typename T>
static int dseb_combine_fwd_impl(const T* y, const T* r0, const T* r1, const T* r2, int n, const float* w, const T* diff,
                                 float ycoef, T* z, int B, int C, int HW, hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || n < 2 || n > 3) return CENET_EINVAL;
  DsebArgs<T> a;
  memset(&a, 0, sizeof(a));
  a.y = y; a.r[0] = r0; a.r[1] = r1; a.r[2] = r2; a.w = w; a.diff = diff; a.z = z; a.n = n; a.C = C; a.ycoef = ycoef;
  const int vw = vec_width<T>(HW, y, r0, r1, r2, diff, z) >= 4 ? 4 : 1;
  a.HW = HW / vw;
  if (vw == 4) CENET_LAUNCH((dseb_combine_fwd_kernel<T, 4>), dim3(B * C, chunks_for(a.HW)), dim3(256), stream, a);
  else CENET_LAUNCH((dseb_combine_fwd_kernel<T, 1>), dim3(B * C, chunks_for(a.HW)), dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(dseb_combine_fwd, (const T* y, const T* r0, const T* r1, const T* r2, int n, const float* w, const T* diff,
                              float ycoef, T* z, int B, int C, int HW, hipStream_t stream),
           (y, r0, r1, r2, n, w, diff, ycoef, z, B, C, HW, stream))

template <typename T>
static int dseb_combine_bwd_acc_impl(const T* y, const T* r0, const T* r1, const T* r2, int n, const float* w, const T* diff,
                                     float ycoef, const T* dz, T* dy, T* dr0, T* dr1, T* dr2, T* ddiff, float* dw_acc, int B,
                                     int C, int HW, hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || n < 2 || n > 3) return CENET_EINVAL;
  DsebArgs<T> a;
  memset(&a, 0, sizeof(a));
  a.y = y; a.r[0] = r0; a.r[1] = r1; a.r[2] = r2; a.w = w; a.diff = diff; a.n = n; a.C = C; a.ycoef = ycoef;
  a.dz = dz; a.dy = dy; a.dr[0] = dr0; a.dr[1] = dr1; a.dr[2] = dr2; a.ddiff = ddiff; a.dw = dw_acc;
  const uintptr_t m = (uintptr_t)y | (uintptr_t)r0 | (uintptr_t)r1 | (uintptr_t)r2 | (uintptr_t)diff | (uintptr_t)dz |
                      (uintptr_t)dy | (uintptr_t)dr0 | (uintptr_t)dr1 | (uintptr_t)dr2 | (uintptr_t)ddiff;
  const int vw = ((HW & 3) == 0 && (m & (4 * sizeof(T) - 1)) == 0) ? 4 : 1;
  a.HW = HW / vw;
  if (vw == 4) CENET_LAUNCH((dseb_combine_bwd_kernel<T, 4>), dim3(B * C, chunks_for(a.HW)), dim3(256), stream, a);
  else CENET_LAUNCH((dseb_combine_bwd_kernel<T, 1>), dim3(B * C, chunks_for(a.HW)), dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(dseb_combine_bwd_acc, (const T* y, const T* r0, const T* r1, const T* r2, int n, const float* w, const T* diff,
                                  float ycoef, const T* dz, T* dy, T* dr0, T* dr1, T* dr2, T* ddiff, float* dw_acc, int B, int C,
                                  int HW, hipStream_t stream),
           (y, r0, r1, r2, n, w, diff, ycoef, dz, dy, dr0, dr1, dr2, ddiff, dw_acc, B, C, HW, stream))

extern "C" int cenet_diffattn_lambda_fwd_f32(const float* q1, const float* k1, const float* q2, const float* k2,
                                             float lambda_init, float* lam3, int hd, hipStream_t stream) {
  if (hd <= 0) return CENET_EINVAL;
  CENET_LAUNCH(diffattn_lambda_fwd_kernel, dim3(1), dim3(64), stream, q1, k1, q2, k2, lambda_init, lam3, hd);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_diffattn_lambda_bwd_acc_f32(const float* q1, const float* k1, const float* q2, const float* k2,
                                                 const float* lam3, const float* dlam, float* dq1, float* dk1, float* dq2,
                                                 float* dk2, int hd, hipStream_t stream) {
  if (hd <= 0) return CENET_EINVAL;
  CENET_LAUNCH(diffattn_lambda_bwd_kernel, dim3(1), dim3(64), stream, q1, k1, q2, k2, lam3, dlam, dq1, dk1, dq2, dk2, hd);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
template <typename T>
static int diffattn_combine_fwd_impl(const T* U, const float* lam3, T* out, int B, int H, int N, int dv, float eps, float post,
                                     hipStream_t stream) {
  if (B <= 0 || H <= 0 || N <= 0 || dv <= 0) return CENET_EINVAL;
  long nvec = (long)B * H * N;
  if (sizeof(T) == 2 && (dv & 7) == 0 && dv <= 256 && ((((uintptr_t)U | (uintptr_t)out) & 15) == 0) && (long)B * H <= 65535) {
    const int lpv = (dv == 16 || dv == 32 || dv == 64) ? dv / 8 : 32;
    const dim3 grid(cdiv((long)N * lpv, 256), B * H);
    if (lpv == 2) CENET_LAUNCH((diffattn_combine_fwd_v8_kernel<2>), grid, dim3(256), stream, (const bf16_t*)U, lam3, (bf16_t*)out, H, N, eps, post, dv);
    else if (lpv == 4) CENET_LAUNCH((diffattn_combine_fwd_v8_kernel<4>), grid, dim3(256), stream, (const bf16_t*)U, lam3, (bf16_t*)out, H, N, eps, post, dv);
    else if (lpv == 32) CENET_LAUNCH((diffattn_combine_fwd_v8_kernel<32>), grid, dim3(256), stream, (const bf16_t*)U, lam3, (bf16_t*)out, H, N, eps, post, dv);
    else CENET_LAUNCH((diffattn_combine_fwd_v8_kernel<8>), grid, dim3(256), stream, (const bf16_t*)U, lam3, (bf16_t*)out, H, N, eps, post, dv);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  const int sub = dv <= 16 ? 16 : (dv <= 32 ? 32 : 64);
  const long per_block = 4L * (64 / sub);
  CENET_LAUNCH((diffattn_combine_fwd_kernel<T>), dim3((unsigned)((nvec + per_block - 1) / per_block)), dim3(256), stream, U, lam3,
               out, H, N, dv, eps, post, nvec, sub);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(diffattn_combine_fwd, (const T* U, const float* lam3, T* out, int B, int H, int N, int dv, float eps, float post,
                                  hipStream_t stream), (U, lam3, out, B, H, N, dv, eps, post, stream))

template <typename T>
static int diffattn_combine_bwd_acc_impl(const T* U, const float* lam3, const T* dout, T* dU, float* dlam_acc, int B, int H,
                                         int N, int dv, float eps, float post, hipStream_t stream) {
  if (B <= 0 || H <= 0 || N <= 0 || dv <= 0) return CENET_EINVAL;
  long nvec = (long)B * H * N;
  if (sizeof(T) == 2 && (dv & 7) == 0 && dv <= 256 &&
      ((((uintptr_t)U | (uintptr_t)dout | (uintptr_t)dU) & 15) == 0) && (long)B * H <= 65535) {
    const int lpv = (dv == 16 || dv == 32 || dv == 64) ? dv / 8 : 32;
    const dim3 grid(cdiv((long)N * lpv, 2048), B * H);
    if (lpv == 2) CENET_LAUNCH((diffattn_combine_bwd_v8_kernel<2>), grid, dim3(256), stream, (const bf16_t*)U, lam3, (const bf16_t*)dout, (bf16_t*)dU, dlam_acc, H, N, eps, post, dv);
    else if (lpv == 4) CENET_LAUNCH((diffattn_combine_bwd_v8_kernel<4>), grid, dim3(256), stream, (const bf16_t*)U, lam3, (const bf16_t*)dout, (bf16_t*)dU, dlam_acc, H, N, eps, post, dv);
    else if (lpv == 32) CENET_LAUNCH((diffattn_combine_bwd_v8_kernel<32>), grid, dim3(256), stream, (const bf16_t*)U, lam3, (const bf16_t*)dout, (bf16_t*)dU, dlam_acc, H, N, eps, post, dv);
    else CENET_LAUNCH((diffattn_combine_bwd_v8_kernel<8>), grid, dim3(256), stream, (const bf16_t*)U, lam3, (const bf16_t*)dout, (bf16_t*)dU, dlam_acc, H, N, eps, post, dv);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  const int sub = dv <= 16 ? 16 : (dv <= 32 ? 32 : 64);
  const long per_block = 4L * (64 / sub);
  long blocks = (nvec + per_block - 1) / per_block;
  if (blocks > 2048) blocks = 2048;
  CENET_LAUNCH((diffattn_combine_bwd_kernel<T>), dim3((unsigned)blocks), dim3(256), stream, U, lam3, dout, dU, dlam_acc, H, N, dv,
               eps, post, nvec, sub);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(diffattn_combine_bwd_acc, (const T* U, const float* lam3, const T* dout, T* dU, float* dlam_acc, int B, int H, int N,
                                      int dv, float eps, float post, hipStream_t stream),
           (U, lam3, dout, dU, dlam_acc, B, H, N, dv, eps, post, stream))
