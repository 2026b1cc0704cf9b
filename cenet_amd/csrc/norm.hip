// norm.hip — LayerNorm (token layout) and train-mode BatchNorm (NCHW planes) forward/backward.
//   LayerNorm : pvtv2.py:117,124,166,69,221-245 (eps 1e-6 in blocks, 1e-5 in patch-embed / sr norm)
//   BatchNorm : cfam.py:22-32, blocks.py:151,161,212,307, nlb.py:81, unet.py:175-197, cfam.py:92,250
// Kernels are templates over the activation storage type T (float / bf16_t, common.h); gamma / beta / statistics / parameter
// gradients are fp32.  All statistics in fp32; BN batch statistics use per-channel shifted sums (shift = first element) so that
// E[x^2]-E[x]^2 does not cancel; partial sums from several workgroups per channel meet in float atomics.
#include "common.h"
#include <cstdlib>
#include "../../include/cenet_hip.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, 4 rows per 256-thread workgroup.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, T* __restrict__ y,
                                                           float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                           int C, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;  // wave-uniform
  const T* xr = x + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += ldf(xr + c);
  const float mu = wave_sum(s) / C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) {
    float d = ldf(xr + c) - mu;
    v += d * d;
  }
  const float rs = rsqrtf(wave_sum(v) / C + eps);
  T* yr = y + row * C;
  for (int c = lane; c < C; c += 64) stf(yr + c, (ldf(xr + c) - mu) * rs * gamma[c] + beta[c]);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// LayerNorm backward.  MAXJ = columns per lane (C <= 64*MAXJ); each wave walks its rows RPI at a time (independent
// load -> reduce -> store chains in flight together), column partials of dgamma/dbeta stay in registers and meet in LDS.
#define LN_ROWS_PER_BLOCK 64
template <typename T, int MAXJ, int RPI>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, T* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int rows, int C, const T* __restrict__ dx_add) {
  __shared__ float red[2][4][MAXJ * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float dg[MAXJ], db[MAXJ], gm[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    dg[j] = 0.f;
    db[j] = 0.f;
    int c = lane + 64 * j;
    gm[j] = c < C ? gamma[c] : 0.f;
  }
  const long r0 = (long)blockIdx.x * LN_ROWS_PER_BLOCK;
  for (int i = wave * RPI; i < LN_ROWS_PER_BLOCK; i += 4 * RPI) {
    float xh[RPI][MAXJ], g[RPI][MAXJ], s1[RPI], s2[RPI], rs[RPI];
#pragma unroll
    for (int q = 0; q < RPI; ++q) {
      const long row = r0 + i + q;
      const bool rok = row < rows;  // wave-uniform
      const float mu = rok ? mean[row] : 0.f;
      rs[q] = rok ? rstd[row] : 0.f;
      s1[q] = 0.f;
      s2[q] = 0.f;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        int c = lane + 64 * j;
        if (rok && c < C) {
          xh[q][j] = (ldf(x + row * C + c) - mu) * rs[q];
          g[q][j] = ldf(dy + row * C + c);
        } else {
          xh[q][j] = 0.f;
          g[q][j] = 0.f;
        }
        float gg = g[q][j] * gm[j];
        s1[q] += gg;
        s2[q] += gg * xh[q][j];
        dg[j] += g[q][j] * xh[q][j];
        db[j] += g[q][j];
      }
    }
#pragma unroll
    for (int q = 0; q < RPI; ++q) {
      s1[q] = wave_sum(s1[q]) / C;
      s2[q] = wave_sum(s2[q]) / C;
    }
#pragma unroll
    for (int q = 0; q < RPI; ++q) {
      const long row = r0 + i + q;
      if (row < rows) {
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
          int c = lane + 64 * j;
          if (c < C)
            stf(dx + row * C + c,
                rs[q] * (g[q][j] * gm[j] - s1[q] - xh[q][j] * s2[q]) + (dx_add ? ldf(dx_add + row * C + c) : 0.f));
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    red[0][wave][j * 64 + lane] = dg[j];
    red[1][wave][j * 64 + lane] = db[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    atomicAdd(&dgamma[c], a);
    atomicAdd(&dbeta[c], b);
  }
}

// ---- 16-byte LayerNorm kernels (C % 4 == 0, 16-byte aligned rows) -------------------------------------------------
// A row is spread over LPR lanes (16 / 32 / 64), 4 consecutive columns per lane and pass, NP passes; a wave holds
// 64 / LPR rows at once and a 256-thread workgroup 4 * 64 / LPR rows per step.  Row statistics are LPR-lane shuffles.
template <int LPR>
__device__ __forceinline__ float subrow_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T, int LPR, int NP>
__global__ __launch_bounds__(256) void ln_fwd_v4_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, T* __restrict__ y,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int rows, int C,
                                                       float eps, int steps) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / LPR, l = lane % LPR;
  f4 gm[NP], bt[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int c = 4 * (l + LPR * p);
    if (c < C) {
      gm[p] = ld4(gamma + c);
      bt[p] = ld4(beta + c);
    }
  }
  const float invC = 1.f / C;
  for (int it = 0; it < steps; ++it) {
    const long row = ((long)blockIdx.x * steps + it) * (4 * RPW) + wave * RPW + sub;
    const bool rok = row < rows;
    f4 v[NP];
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c = 4 * (l + LPR * p);
      if (rok && c < C) {
        v[p] = ld4(x + row * C + c);
        s += (v[p].v[0] + v[p].v[1]) + (v[p].v[2] + v[p].v[3]);
      } else {
        v[p].v[0] = v[p].v[1] = v[p].v[2] = v[p].v[3] = 0.f;
      }
    }
    const float mu = subrow_sum<LPR>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c = 4 * (l + LPR * p);
      if (c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[p].v[e] - mu;
          q += d * d;
        }
      }
    }
    const float rs = rsqrtf(subrow_sum<LPR>(q) * invC + eps);
    if (rok) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int c = 4 * (l + LPR * p);
        if (c < C) {
          f4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o.v[e] = (v[p].v[e] - mu) * rs * gm[p].v[e] + bt[p].v[e];
          st4(y + row * C + c, o);
        }
      }
      if (l == 0) {
        mean[row] = mu;
        rstd[row] = rs;
      }
    }
  }
}

template <typename T, int LPR, int NP>
__global__ __launch_bounds__(256) void ln_bwd_v4_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, T* __restrict__ dx,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C,
                                                       int steps, const T* __restrict__ dx_add) {
  constexpr int RPW = 64 / LPR;
  __shared__ float red[2][4][NP * LPR * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / LPR, l = lane % LPR;
  f4 gm[NP], dg[NP], db[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int c = 4 * (l + LPR * p);
#pragma unroll
    for (int e = 0; e < 4; ++e) dg[p].v[e] = db[p].v[e] = gm[p].v[e] = 0.f;
    if (c < C) gm[p] = ld4(gamma + c);
  }
  const float invC = 1.f / C;
#pragma unroll 2
  for (int it = 0; it < steps; ++it) {
    const long row = ((long)blockIdx.x * steps + it) * (4 * RPW) + wave * RPW + sub;
    const bool rok = row < rows;
    const float mu = rok ? mean[row] : 0.f, rs = rok ? rstd[row] : 0.f;
    f4 xh[NP], g[NP];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c = 4 * (l + LPR * p);
      if (rok && c < C) {
        xh[p] = ld4(x + row * C + c);
        g[p] = ld4(dy + row * C + c);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) xh[p].v[e] = g[p].v[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[p].v[e] = (rok && c < C) ? (xh[p].v[e] - mu) * rs : 0.f;
        const float gg = g[p].v[e] * gm[p].v[e];
        s1 += gg;
        s2 += gg * xh[p].v[e];
        dg[p].v[e] += g[p].v[e] * xh[p].v[e];
        db[p].v[e] += g[p].v[e];
      }
    }
    s1 = subrow_sum<LPR>(s1) * invC;
    s2 = subrow_sum<LPR>(s2) * invC;
    if (rok) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int c = 4 * (l + LPR * p);
        if (c < C) {
          f4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o.v[e] = rs * (g[p].v[e] * gm[p].v[e] - s1 - xh[p].v[e] * s2);
          if (dx_add) {  // gradient arriving through the residual path that by-passed the LayerNorm
            const f4 ad = ld4(dx_add + row * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) o.v[e] += ad.v[e];
          }
          st4(dx + row * C + c, o);
        }
      }
    }
  }
  // column partials: rows held by the same wave (lanes l, l + LPR, ...), then the four waves through LDS
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        dg[p].v[e] += __shfl_xor(dg[p].v[e], o);
        db[p].v[e] += __shfl_xor(db[p].v[e], o);
      }
      if (sub == 0) {
        red[0][wave][(p * LPR + l) * 4 + e] = dg[p].v[e];
        red[1][wave][(p * LPR + l) * 4 + e] = db[p].v[e];
      }
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {  // column c sits at quad c/4 = l + LPR*p -> index (p*LPR + l)*4 + e == c
    atomicAdd(&dgamma[c], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(&dbeta[c], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// bf16 rows, 8 channels (16 bytes) per lane: half the load / store instructions and twice the bytes in flight per lane of the
// quad form above (the backward pass moves 3 - 4 tensors and ran at 1.3 - 1.7 TB/s).  Same structure and reductions.
template <int LPR, int NP>
__global__ __launch_bounds__(256) void ln_bwd_v8_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C,
                                                       int steps, const bf16_t* __restrict__ dx_add, float* __restrict__ part,
                                                       const float* __restrict__ bscale, int rows_per_sample,
                                                       bf16_t* __restrict__ dxs) {
  constexpr int RPW = 64 / LPR;
  __shared__ float red[2][4][NP * LPR * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / LPR, l = lane % LPR;
  float gm[NP][8], dg[NP][8], db[NP][8];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int c = 8 * (l + LPR * p);
#pragma unroll
    for (int e = 0; e < 8; ++e) dg[p][e] = db[p][e] = gm[p][e] = 0.f;
    if (c < C) {
      const f4 a = ld4(gamma + c), b = ld4(gamma + c + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) gm[p][e] = a.v[e], gm[p][4 + e] = b.v[e];
    }
  }
  const float invC = 1.f / C;
#pragma unroll 2
  for (int it = 0; it < steps; ++it) {
    const long row = ((long)blockIdx.x * steps + it) * (4 * RPW) + wave * RPW + sub;
    const bool rok = row < rows;
    const float mu = rok ? mean[row] : 0.f, rs = rok ? rstd[row] : 0.f;
    float xh[NP][8], g[NP][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c = 8 * (l + LPR * p);
      const bool ok = rok && c < C;
      if (ok) {
        ldv<8>(xh[p], x + row * C + c);
        ldv<8>(g[p], dy + row * C + c);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[p][e] = ok ? (xh[p][e] - mu) * rs : 0.f;
        g[p][e] = ok ? g[p][e] : 0.f;
        const float gg = g[p][e] * gm[p][e];
        s1 += gg;
        s2 += gg * xh[p][e];
        dg[p][e] += g[p][e] * xh[p][e];
        db[p][e] += g[p][e];
      }
    }
    s1 = subrow_sum<LPR>(s1) * invC;
    s2 = subrow_sum<LPR>(s2) * invC;
    if (rok) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int c = 8 * (l + LPR * p);
        if (c < C) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rs * (g[p][e] * gm[p][e] - s1 - xh[p][e] * s2);
          if (dx_add) {  // gradient arriving through the residual path that by-passed the LayerNorm
            float ad[8];
            ldv<8>(ad, dx_add + row * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += ad[e];
          }
          stv<8>(dx + row * C + c, o);
          if (dxs) {  // the same rows times the per-sample (stochastic-depth) scale of the branch that produced this tensor
            const float sc = bscale[row / rows_per_sample];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = sc * cenet_bf2f(cenet_f2bf(o[e]));
            stv<8>(dxs + row * C + c, o);
          }
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        dg[p][e] += __shfl_xor(dg[p][e], o);
        db[p][e] += __shfl_xor(db[p][e], o);
      }
      if (sub == 0) {
        red[0][wave][(p * LPR + l) * 8 + e] = dg[p][e];
        red[1][wave][(p * LPR + l) * 8 + e] = db[p][e];
      }
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {  // column c sits at octet c/8 = l + LPR*p -> index (p*LPR + l)*8 + e == c
    const float vg = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float vb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    if (part) {  // deferred: this workgroup's row of the partial buffer (folded later by ln_fold_group_kernel, no atomics)
      part[(long)blockIdx.x * 2 * C + c] = vg;
      part[(long)blockIdx.x * 2 * C + C + c] = vb;
    } else {
      atomicAdd(&dgamma[c], vg);
      atomicAdd(&dbeta[c], vb);
    }
  }
}

// ---- LayerNorm of a split-K accumulator (round 5): the spatial-reduction conv of pvtv2.py:93-95 runs as a split-K GEMM into a
// zero-at-rest fp32 accumulator (ops._ZeroWs); instead of cast_clear_bias (-> bf16 rows) followed by the LayerNorm launch, ONE
// kernel reads the accumulator, adds the conv bias, stores the rounded rows xpre (the LayerNorm backward reads them), normalises
// THE ROUNDED VALUES (the same numbers the two-launch form normalises), stores y, mean, rstd and leaves the accumulator zero.
// One wave per row; C <= 512, C % 4 == 0.
__global__ __launch_bounds__(256) void ln_fwd_acc_kernel(float* __restrict__ acc, const float* __restrict__ bias,
                                                        bf16_t* __restrict__ xpre, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                        float* __restrict__ mean, float* __restrict__ rstd, int rows, int C,
                                                        float eps) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;  // (wave-uniform)
  float* ar = acc + (long)row * C;
  float v[2][4];
  const int nq = C >> 2;  // quads per row (<= 128: two per lane)
  float s1 = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int q = lane + 64 * k;
    const bool ok = q < nq;
    const int qq = ok ? q : 0;
    const f4 a = ld4(ar + 4 * qq);
    const f4 b = bias ? ld4(bias + 4 * qq) : f4{{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[k][e] = ok ? cenet_bf2f(cenet_f2bf(a.v[e] + b.v[e])) : 0.f;
      s1 += v[k][e];
    }
    if (ok) {
      st4(ar + 4 * q, f4{{0.f, 0.f, 0.f, 0.f}});
      st4v(xpre + (long)row * C + 4 * q, v[k]);
    }
  }
  const float mu = wave_sum(s1) / (float)C;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (lane + 64 * k < nq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s2 += (v[k][e] - mu) * (v[k][e] - mu);
    }
  const float rs = rsqrtf(wave_sum(s2) / (float)C + eps);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int q = lane + 64 * k;
    if (q < nq) {
      const f4 g = ld4(gamma + 4 * q), bt = ld4(beta + 4 * q);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[k][e] - mu) * rs * g.v[e] + bt.v[e];
      st4v(y + (long)row * C + 4 * q, o);
    }
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}
extern "C" int cenet_layernorm_fwd_acc_bf16(float* acc, const float* bias, bf16_t* xpre, const float* gamma, const float* beta,
                                            bf16_t* y, float* mean, float* rstd, int rows, int C, float eps, hipStream_t stream) {
  if (!acc || !xpre || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || C <= 0) return CENET_EINVAL;
  if ((C & 3) != 0 || C > 512 ||
      ((((uintptr_t)acc | (uintptr_t)bias | (uintptr_t)gamma | (uintptr_t)beta) & 15) != 0) ||
      ((((uintptr_t)xpre | (uintptr_t)y) & 7) != 0))
    return CENET_EUNSUPPORTED;
  CENET_LAUNCH(ln_fwd_acc_kernel, dim3(cdiv(rows, 4)), dim3(256), stream, acc, bias, xpre, gamma, beta, y, mean, rstd, rows, C, eps);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

// ---- deferred fold of LayerNorm affine-gradient partials (round 4) ----------------------------------------------------------
// With float atomics every LayerNorm backward ended in 2 C same-address adds per workgroup, which capped the grid at ~256
// workgroups x 6 - 8 dependent steps (14 - 18 us per call on 4 - 13 MB of data).  Writing one partial ROW per workgroup instead
// lets the grid be ~1 000 single-step workgroups; the rows of ALL LayerNorms of a backward segment are folded by ONE launch.
struct LnFoldDesc {
  const float* part;  // [nrows][2 C]
  float* dg;          // [C] +=
  float* db;          // [C] +=
  int nrows, C;
};
#define LN_FOLD_MAX 48
struct LnFoldArgs {
  LnFoldDesc d[LN_FOLD_MAX];
};
// grid (chunks of 32 columns, layers), 1024 threads = 32 columns x 32 row lanes
__global__ __launch_bounds__(1024) void ln_fold_group_kernel(LnFoldArgs a) {
  __shared__ float red[32][33];
  const LnFoldDesc d = a.d[blockIdx.y];
  const int C2 = 2 * d.C;
  const int col = blockIdx.x * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
  if (blockIdx.x * 32 >= C2) return;
  float s = 0.f;
  if (col < C2) {
    // eight independent loads in flight per thread (one dependent add per ~1 us round trip otherwise: 56 us for 47 MB)
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r = rl;
    for (; r + 7 * 32 < d.nrows; r += 8 * 32) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a8[k] += d.part[(long)(r + 32 * k) * C2 + col];
    }
    for (; r < d.nrows; r += 32) a8[0] += d.part[(long)r * C2 + col];
    s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  }
  red[rl][threadIdx.x & 31] = s;
  __syncthreads();
  if (rl == 0 && col < C2) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[i][threadIdx.x & 31];
    if (col < d.C) d.dg[col] += t;
    else d.db[col - d.C] += t;
  }
}

static inline int ln_steps(int rows, int rows_per_step, long target);
/* rows of the partial buffer cenet_layernorm_bwd_add_part_bf16 writes for (rows, C): [that many][2 C] floats */
extern "C" int cenet_layernorm_bwd_part_rows(int rows, int C) {
  const int lpr = C <= 64 ? 8 : C <= 128 ? 16 : C <= 256 ? 32 : 64;
  const int rps = 4 * 64 / lpr, st = ln_steps(rows, rps, 1024);
  return cdiv(rows, rps * st);
}

/* LayerNorm backward (+ residual gradient dx_add) on bf16 rows whose affine gradients go to a partial buffer instead of float
 * atomics: part[cenet_layernorm_bwd_part_rows(rows, C)][2 C]; fold with cenet_ln_fold_group. */
/* ... and optionally dxs = bscale[row / rows_per_sample] * dx (bf16, rounded from the stored dx): the tensor this LayerNorm read was
 * x + bscale_b * branch(...) (pvtv2.py:141-149 DropPath), so the branch's backward wants the scaled gradient — written here by the
 * kernel that produces dx instead of a scale pass of its own.  bscale == NULL / dxs == NULL: plain. */
extern "C" int cenet_layernorm_bwd_add_part_scaled_bf16(const bf16_t* dy, const bf16_t* x, const float* gamma, const float* mean,
                                                        const float* rstd, const bf16_t* dx_add, bf16_t* dx, float* part,
                                                        const float* bscale, int rows_per_sample, bf16_t* dxs, int rows, int C,
                                                        hipStream_t stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !part || rows <= 0 || C <= 0) return CENET_EINVAL;
  if ((bscale != nullptr) != (dxs != nullptr) || (bscale && rows_per_sample <= 0)) return CENET_EINVAL;
  if (C > 512 || (C & 7) != 0 ||
      ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dx_add | (uintptr_t)gamma | (uintptr_t)dxs) & 15) != 0))
    return CENET_EUNSUPPORTED;
#define CENET_LNP8(LPRv)                                                                                               \
  {                                                                                                                   \
    const int rps = 4 * 64 / LPRv, st = ln_steps(rows, rps, 1024);                                                    \
    CENET_LAUNCH((ln_bwd_v8_kernel<LPRv, 1>), dim3(cdiv(rows, rps * st)), dim3(256), stream, dy, x, gamma, mean, rstd, dx, \
                 (float*)nullptr, (float*)nullptr, rows, C, st, dx_add, part, bscale, rows_per_sample, dxs);            \
  }
  if (C <= 64) CENET_LNP8(8)
  else if (C <= 128) CENET_LNP8(16)
  else if (C <= 256) CENET_LNP8(32)
  else CENET_LNP8(64)
#undef CENET_LNP8
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_layernorm_bwd_add_part_bf16(const bf16_t* dy, const bf16_t* x, const float* gamma, const float* mean,
                                                 const float* rstd, const bf16_t* dx_add, bf16_t* dx, float* part, int rows, int C,
                                                 hipStream_t stream) {
  return cenet_layernorm_bwd_add_part_scaled_bf16(dy, x, gamma, mean, rstd, dx_add, dx, part, nullptr, 1, nullptr, rows, C, stream);
}

/* n <= 48 per launch (more: several launches): dg / db += column sums of the partial buffers.
 * The fold adds with plain read-modify-writes, one workgroup row per item, so two items of ONE launch must not share a
 * destination (a LayerNorm run twice before one backward pass, a shared norm module): an item whose dg / db is already in the
 * launch being assembled closes that launch and opens the next — launches of one stream run in order, nothing is lost. */
extern "C" int cenet_ln_fold_group(const void* const* part, float* const* dg, float* const* db, const int* nrows, const int* C,
                                   int n, hipStream_t stream) {
  if (n < 0 || (n > 0 && (!part || !dg || !db || !nrows || !C))) return CENET_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!part[i] || !dg[i] || !db[i] || nrows[i] <= 0 || C[i] <= 0) return CENET_EINVAL;
  int i = 0;
  while (i < n) {
    LnFoldArgs a;
    int m = 0, cmax = 0;
    for (; i < n && m < LN_FOLD_MAX; ++i) {
      bool dup = false;
      for (int j = 0; j < m && !dup; ++j)
        dup = a.d[j].dg == dg[i] || a.d[j].db == db[i] || a.d[j].dg == db[i] || a.d[j].db == dg[i];
      if (dup) break;
      a.d[m].part = (const float*)part[i]; a.d[m].dg = dg[i]; a.d[m].db = db[i];
      a.d[m].nrows = nrows[i]; a.d[m].C = C[i];
      if (C[i] > cmax) cmax = C[i];
      ++m;
    }
    CENET_LAUNCH(ln_fold_group_kernel, dim3(cdiv(2 * cmax, 32), m), dim3(1024), stream, a);
    CENET_CHECK_LAUNCH();
  }
  return CENET_OK;
}

// activation pointers a, b (quads of T) and parameter pointers c, d (quads of float)
template <typename T>
static inline bool ln_v4_ok(const void* a, const void* b, const void* c, const void* d, int C) {
  return (C & 3) == 0 && C <= 512 && ((((uintptr_t)a | (uintptr_t)b) & (4 * sizeof(T) - 1)) == 0) &&
         ((((uintptr_t)c | (uintptr_t)d) & 15) == 0);
}
// rows per workgroup step = 4 * 64 / LPR; steps per workgroup chosen to keep >= ~1024 workgroups while amortising the
// per-workgroup column reduction of the backward pass
// `target`: workgroups wanted.  Forward: ~1024 (latency hiding).  Backward: ~256 — every workgroup ends with 2*C float atomics
// onto the SAME 2*C addresses, and same-row atomics run at ~0.09 TB/s (MI355X_MICROARCH.md, Global float atomics): at C = 320
// and 1568 workgroups that was 4 MB of contended adds = 44 us per LayerNorm; measured 31 -> 17 us per call with 256.
static inline int ln_steps(int rows, int rows_per_step, long target) {
  long st = rows / ((long)rows_per_step * target);
  if (st < 1) st = 1;
  if (st > 8) st = 8;
  return (int)st;
}

template <typename T>
static int layernorm_fwd_impl(const T* x, const float* gamma, const float* beta, T* y, float* mean, float* rstd, int rows, int C,
                              float eps, hipStream_t stream) {
  if (rows <= 0 || C <= 0) return CENET_EINVAL;
  if (ln_v4_ok<T>(x, y, gamma, beta, C)) {
#define CENET_LNF(LPRv, NPv)                                                                                          \
  {                                                                                                                   \
    const int rps = 4 * 64 / LPRv, st = ln_steps(rows, rps, 1024);                                                          \
    CENET_LAUNCH((ln_fwd_v4_kernel<T, LPRv, NPv>), dim3(cdiv(rows, rps * st)), dim3(256), stream, x, gamma, beta, y, mean, rstd, \
                 rows, C, eps, st);                                                                                   \
  }
    if (C <= 64) CENET_LNF(16, 1)
    else if (C <= 128) CENET_LNF(32, 1)
    else if (C <= 256) CENET_LNF(64, 1)
    else CENET_LNF(64, 2)
#undef CENET_LNF
  } else {
    CENET_LAUNCH((layernorm_fwd_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), stream, x, gamma, beta, y, mean, rstd, rows, C, eps);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(layernorm_fwd, (const T* x, const float* gamma, const float* beta, T* y, float* mean, float* rstd, int rows, int C,
                           float eps, hipStream_t stream), (x, gamma, beta, y, mean, rstd, rows, C, eps, stream))

template <typename T>
static int layernorm_bwd_add_acc_impl(const T* dy, const T* x, const float* gamma, const float* mean, const float* rstd,
                                      const T* dx_add, T* dx, float* dgamma_acc, float* dbeta_acc, int rows, int C,
                                      hipStream_t stream) {
  if (rows <= 0 || C <= 0) return CENET_EINVAL;
  if (C > 512) return CENET_EUNSUPPORTED;
  if (sizeof(T) == 2 && (C & 7) == 0 &&
      ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dx_add | (uintptr_t)gamma) & 15) == 0)) {
#define CENET_LNB8(LPRv, NPv)                                                                                         \
  {                                                                                                                   \
    const int rps = 4 * 64 / LPRv, st = ln_steps(rows, rps, 256);                                                     \
    CENET_LAUNCH((ln_bwd_v8_kernel<LPRv, NPv>), dim3(cdiv(rows, rps * st)), dim3(256), stream, (const bf16_t*)dy,      \
                 (const bf16_t*)x, gamma, mean, rstd, (bf16_t*)dx, dgamma_acc, dbeta_acc, rows, C, st, (const bf16_t*)dx_add, (float*)nullptr, \
                 (const float*)nullptr, 1, (bf16_t*)nullptr); \
  }
    if (C <= 64) CENET_LNB8(8, 1)
    else if (C <= 128) CENET_LNB8(16, 1)
    else if (C <= 256) CENET_LNB8(32, 1)
    else CENET_LNB8(64, 1)
#undef CENET_LNB8
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (ln_v4_ok<T>(dy, x, gamma, gamma, C) && quad_aligned<T>(dx) && quad_aligned<T>(dx_add)) {
#define CENET_LNB(LPRv, NPv)                                                                                          \
  {                                                                                                                   \
    const int rps = 4 * 64 / LPRv, st = ln_steps(rows, rps, sizeof(T) == 2 ? 256 : 1024);                             \
    CENET_LAUNCH((ln_bwd_v4_kernel<T, LPRv, NPv>), dim3(cdiv(rows, rps * st)), dim3(256), stream, dy, x, gamma, mean, rstd, dx, \
                 dgamma_acc, dbeta_acc, rows, C, st, dx_add);                                                         \
  }
    if (C <= 64) CENET_LNB(16, 1)
    else if (C <= 128) CENET_LNB(32, 1)
    else if (C <= 256) CENET_LNB(64, 1)
    else CENET_LNB(64, 2)
#undef CENET_LNB
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  dim3 grid(cdiv(rows, LN_ROWS_PER_BLOCK));
#define CENET_LN(MJ, RP)                                                                                            \
  CENET_LAUNCH((layernorm_bwd_kernel<T, MJ, RP>), grid, dim3(256), stream, dy, x, gamma, mean, rstd, dx, dgamma_acc, dbeta_acc, \
               rows, C, dx_add)
  if (C <= 64) { CENET_LN(1, 4); }
  else if (C <= 128) { CENET_LN(2, 4); }
  else if (C <= 256) { CENET_LN(4, 2); }
  else { CENET_LN(8, 1); }
#undef CENET_LN
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

CENET_TWIN(layernorm_bwd_add_acc, (const T* dy, const T* x, const float* gamma, const float* mean, const float* rstd,
                                   const T* dx_add, T* dx, float* dgamma_acc, float* dbeta_acc, int rows, int C,
                                   hipStream_t stream),
           (dy, x, gamma, mean, rstd, dx_add, dx, dgamma_acc, dbeta_acc, rows, C, stream))
template <typename T>
static int layernorm_bwd_acc_impl(const T* dy, const T* x, const float* gamma, const float* mean, const float* rstd, T* dx,
                                  float* dgamma_acc, float* dbeta_acc, int rows, int C, hipStream_t stream) {
  return layernorm_bwd_add_acc_impl<T>(dy, x, gamma, mean, rstd, nullptr, dx, dgamma_acc, dbeta_acc, rows, C, stream);
}
CENET_TWIN(layernorm_bwd_acc, (const T* dy, const T* x, const float* gamma, const float* mean, const float* rstd, T* dx,
                               float* dgamma_acc, float* dbeta_acc, int rows, int C, hipStream_t stream),
           (dy, x, gamma, mean, rstd, dx, dgamma_acc, dbeta_acc, rows, C, stream))

// ------------------------------------------------------------------------------------------------
// BatchNorm (training): statistics over (B, HW) per channel of x[b*sb + c*HW + p].
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, long sb, int B, int HW,
                                                        float* __restrict__ ws) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const T* xc = x + (long)c * HW;
  const float shift = ldf(xc);
  const long total = (long)B * HW;
  float s1 = 0.f, s2 = 0.f;
  for (int w__ = blockIdx.y; w__ < B * ((HW + 1023) / 1024); w__ += gridDim.y)  // (image, 1024-pixel chunk) items: no per-element division
  for (int b = w__ / ((HW + 1023) / 1024), p = (w__ - b * ((HW + 1023) / 1024)) * 1024 + threadIdx.x,
           pend__ = ((w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 < HW) ? (w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 : HW;
       p < pend__; p += 256) {
    float v = ldf(xc + (long)b * sb + p) - shift;
    s1 += v;
    s2 += v * v;
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {  // one slot per (channel, split): no zero-fill, no atomics; the consumer adds the gridDim.y slots
    ws[((long)c) * gridDim.y + blockIdx.y] = s1;
    ws[((long)gridDim.x + c) * gridDim.y + blockIdx.y] = s2;
  }
}

template <typename T>
__global__ void bn_finalize_kernel(const T* __restrict__ x, int HW, const float* __restrict__ ws, int C, int S, float n,
                                   float* __restrict__ mean, float* __restrict__ var, float* running_mean,
                                   float* running_var, float momentum, long* nbt, int nbt_n) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float shift = ldf(x + (long)c * HW);
    float a1 = 0.f, a2 = 0.f;
    for (int i = 0; i < S; ++i) {
      a1 += ws[(long)c * S + i];
      a2 += ws[((long)C + c) * S + i];
    }
    float m = a1 / n;
    float v = a2 / n - m * m;
    if (v < 0.f) v = 0.f;
    float mu = shift + m;
    mean[c] = mu;
    var[c] = v;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * v * (n / (n - 1.f));
    }
  }
  if (nbt && c < nbt_n) nbt[c] += 1;
}

// Train-mode forward in two launches instead of three: when `fin.ws` is set, the normalisation pass derives the batch mean /
// variance of its channel from the partial sums itself (what bn_finalize_kernel does), and the workgroup (thread) that handles
// the first elements of image 0's plane publishes mean / var for the backward pass and updates the running statistics.
struct BnFin {
  const float* ws;  // partial sums [2][C][S] of the shifted values (nullptr: mean / var are given)
  int S;
  float n;          // B * HW
  float *mean, *var, *rmean, *rvar;
  float momentum;
  long* nbt;
  int nbt_n;        // counters at nbt[0 .. nbt_n): one per nn.BatchNorm2d whose channels this launch covers (merged launches)
};
__device__ __forceinline__ void bn_fin_publish(const BnFin& f, int c, float mu, float v) {
  f.mean[c] = mu;
  f.var[c] = v;
  if (f.rmean) {
    f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * mu;
    f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * v * (f.n / (f.n - 1.f));
  }
  if (f.nbt && c < (f.nbt_n > 1 ? f.nbt_n : 1)) f.nbt[c] += 1;
}
// plane-per-workgroup kernels: every thread of the workgroup calls (block_sum inside)
template <typename T>
__device__ __forceinline__ void bn_fin_plane(const BnFin& f, const T* x, int C, int HW, int c, bool writer, float* red, float& mu,
                                             float& v) {
  float p1 = 0.f, p2 = 0.f;
  for (int i = threadIdx.x; i < f.S; i += blockDim.x) {
    p1 += f.ws[(long)c * f.S + i];
    p2 += f.ws[((long)C + c) * f.S + i];
  }
  const float a1 = block_sum(p1, red), a2 = block_sum(p2, red);
  const float m = a1 / f.n;
  v = a2 / f.n - m * m;
  if (v < 0.f) v = 0.f;
  mu = ldf(x + (long)c * HW) + m;
  if (writer && threadIdx.x == 0) bn_fin_publish(f, c, mu, v);
}

// y = act(gamma*(x-mean)*rsqrt(var+eps)+beta); grid (B*C, chunks)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, long sxb, T* __restrict__ y, long syb,
                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                      float eps, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int act, float slope, int C, int HW,
                                                      BnFin fin) {
  __shared__ float red[16];
  const int bc = blockIdx.x;
  const int b = bc / C, c = bc - b * C;
  float mu, vr;
  if (fin.ws) bn_fin_plane<T>(fin, x, C, HW, c, b == 0 && blockIdx.y == 0, red, mu, vr);
  else mu = mean[c], vr = var[c];
  const float sc = gamma[c] * rsqrtf(vr + eps);
  const float sh = beta[c] - mu * sc;
  const T* xp = x + (long)b * sxb + (long)c * HW;
  T* yp = y + (long)b * syb + (long)c * HW;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < HW; p += gridDim.y * 256) stf(yp + p, act_fwd(act, ldf(xp + p) * sc + sh, slope));
}

// partial sums of g' = dy*act'(pre) and g'*xhat per channel
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ dy, long sgb,
                                                            const T* __restrict__ x, long sxb,
                                                            const float* __restrict__ mean, const float* __restrict__ var,
                                                            float eps, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int act, float slope, int B,
                                                            int HW, float* __restrict__ ws) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const float rs = rsqrtf(var[c] + eps), mu = mean[c], gm = gamma[c], bt = beta[c];
  const long total = (long)B * HW;
  float s1 = 0.f, s2 = 0.f;
  for (int w__ = blockIdx.y; w__ < B * ((HW + 1023) / 1024); w__ += gridDim.y)  // (image, 1024-pixel chunk) items: no per-element division
  for (int b = w__ / ((HW + 1023) / 1024), p = (w__ - b * ((HW + 1023) / 1024)) * 1024 + threadIdx.x,
           pend__ = ((w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 < HW) ? (w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 : HW;
       p < pend__; p += 256) {
    float xh = (ldf(x + (long)b * sxb + (long)c * HW + p) - mu) * rs;
    float g = ldf(dy + (long)b * sgb + (long)c * HW + p);
    if (act != ACT_NONE) g *= act_bwd(act, xh * gm + bt, slope);
    s1 += g;
    s2 += g * xh;
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {  // one slot per (channel, split): no zero-fill, no atomics; the consumer adds the gridDim.y slots
    ws[((long)c) * gridDim.y + blockIdx.y] = s1;
    ws[((long)gridDim.x + c) * gridDim.y + blockIdx.y] = s2;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, long sgb, const T* __restrict__ x,
                                                          long sxb, T* __restrict__ dx, long sdb,
                                                          const float* __restrict__ mean, const float* __restrict__ var,
                                                          float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int act, float slope, int C, int HW,
                                                          float n, const float* __restrict__ ws, int S, float* dgamma,
                                                          float* dbeta, const T* __restrict__ dx_add, long sab) {
  const int bc = blockIdx.x;
  const int b = bc / C, c = bc - b * C;
  const float rs = rsqrtf(var[c] + eps), mu = mean[c], gm = gamma[c], bt = beta[c];
  __shared__ float red[16];
  float p1 = 0.f, p2 = 0.f;
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    p1 += ws[(long)c * S + i];
    p2 += ws[((long)C + c) * S + i];
  }
  const float s1 = block_sum(p1, red), s2 = block_sum(p2, red);
  const float m1 = s1 / n, m2 = s2 / n;
  const T* xp = x + (long)b * sxb + (long)c * HW;
  const T* gp = dy + (long)b * sgb + (long)c * HW;
  const T* ap = dx_add ? dx_add + (long)b * sab + (long)c * HW : nullptr;  // gradient that by-passed the BatchNorm (residual path)
  T* dp = dx + (long)b * sdb + (long)c * HW;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < HW; p += gridDim.y * 256) {
    float xh = (ldf(xp + p) - mu) * rs;
    float g = ldf(gp + p);
    if (act != ACT_NONE) g *= act_bwd(act, xh * gm + bt, slope);
    stf(dp + p, gm * rs * (g - m1 - xh * m2) + (ap ? ldf(ap + p) : 0.f));
  }
  if (b == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    atomicAdd(&dgamma[c], s2);
    atomicAdd(&dbeta[c], s1);
  }
}

// ---- 16-byte BatchNorm kernels (HW % 4 == 0, 16-byte aligned planes) -----------------------------------------------------
// Reductions walk the channel's B*HW/4 quads flat (grid-stride, four independent 16-byte loads in flight per thread).
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_v4_kernel(const T* __restrict__ x, long sb, int B, int HW,
                                                           float* __restrict__ ws) {
  __shared__ float red[16];
  const int c = blockIdx.x, nq = HW >> 2;
  const T* xc = x + (long)c * HW;
  const float shift = ldf(xc);
  const int total = B * nq, stride = gridDim.y * 256;
  const float inv_nq = cenet_inv_small(nq, total);
  float s1 = 0.f, s2 = 0.f;
  for (int q0 = blockIdx.y * 256 + threadIdx.x; q0 < total; q0 += 4 * stride) {
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + u * stride;
      if (q < total) {
        const int b = cenet_div_small(q, nq, inv_nq), qi = q - b * nq;
        ld4v(v[u], xc + (long)b * sb + 4 * qi);
      } else {
        v[u][0] = v[u][1] = v[u][2] = v[u][3] = shift;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[u][e] - shift;
        s1 += d;
        s2 += d * d;
      }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {  // one slot per (channel, split): no zero-fill, no atomics; the consumer adds the gridDim.y slots
    ws[((long)c) * gridDim.y + blockIdx.y] = s1;
    ws[((long)gridDim.x + c) * gridDim.y + blockIdx.y] = s2;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_v4_kernel(const T* __restrict__ x, long sxb, T* __restrict__ y, long syb,
                                                         const float* __restrict__ mean, const float* __restrict__ var,
                                                         float eps, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int act, float slope, int C, int HW,
                                                         BnFin fin) {
  __shared__ float red[16];
  const int bc = blockIdx.x;
  const int b = bc / C, c = bc - b * C;
  float mu, vr;
  if (fin.ws) bn_fin_plane<T>(fin, x, C, HW, c, b == 0 && blockIdx.y == 0, red, mu, vr);
  else mu = mean[c], vr = var[c];
  const float sc = gamma[c] * rsqrtf(vr + eps);
  const float sh = beta[c] - mu * sc;
  const T* xp = x + (long)b * sxb + (long)c * HW;
  T* yp = y + (long)b * syb + (long)c * HW;
  const int nq = HW >> 2;
  for (int q = blockIdx.y * blockDim.x + threadIdx.x; q < nq; q += gridDim.y * blockDim.x) {
    float v[4];
    ld4v(v, xp + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_fwd(act, v[e] * sc + sh, slope);
    st4v(yp + 4 * q, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_v4_kernel(const T* __restrict__ dy, long sgb,
                                                               const T* __restrict__ x, long sxb,
                                                               const float* __restrict__ mean, const float* __restrict__ var,
                                                               float eps, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int act, float slope, int B,
                                                               int HW, float* __restrict__ ws) {
  __shared__ float red[16];
  const int c = blockIdx.x, nq = HW >> 2;
  const float rs = rsqrtf(var[c] + eps), mu = mean[c], gm = gamma[c], bt = beta[c];
  const int total = B * nq, stride = gridDim.y * 256;
  const float inv_nq = cenet_inv_small(nq, total);
  float s1 = 0.f, s2 = 0.f;
  for (int q0 = blockIdx.y * 256 + threadIdx.x; q0 < total; q0 += 2 * stride) {
    float xv[2][4], gv[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = q0 + u * stride;
      if (q < total) {
        const int b = cenet_div_small(q, nq, inv_nq), qi = q - b * nq;
        ld4v(xv[u], x + (long)b * sxb + (long)c * HW + 4 * qi);
        ld4v(gv[u], dy + (long)b * sgb + (long)c * HW + 4 * qi);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xv[u][e] = mu;
          gv[u][e] = 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[u][e] - mu) * rs;
        float g = gv[u][e];
        if (act != ACT_NONE) g *= act_bwd(act, xh * gm + bt, slope);
        s1 += g;
        s2 += g * xh;
      }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {  // one slot per (channel, split): no zero-fill, no atomics; the consumer adds the gridDim.y slots
    ws[((long)c) * gridDim.y + blockIdx.y] = s1;
    ws[((long)gridDim.x + c) * gridDim.y + blockIdx.y] = s2;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_v4_kernel(const T* __restrict__ dy, long sgb, const T* __restrict__ x,
                                                             long sxb, T* __restrict__ dx, long sdb,
                                                             const float* __restrict__ mean, const float* __restrict__ var,
                                                             float eps, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int act, float slope, int C,
                                                             int HW, float n, const float* __restrict__ ws, int S, float* dgamma,
                                                             float* dbeta, const T* __restrict__ dx_add, long sab) {
  const int bc = blockIdx.x;
  const int b = bc / C, c = bc - b * C;
  const float rs = rsqrtf(var[c] + eps), mu = mean[c], gm = gamma[c], bt = beta[c];
  __shared__ float red[16];
  float p1 = 0.f, p2 = 0.f;
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    p1 += ws[(long)c * S + i];
    p2 += ws[((long)C + c) * S + i];
  }
  const float s1 = block_sum(p1, red), s2 = block_sum(p2, red);
  const float m1 = s1 / n, m2 = s2 / n;
  const T* xp = x + (long)b * sxb + (long)c * HW;
  const T* gp = dy + (long)b * sgb + (long)c * HW;
  T* dp = dx + (long)b * sdb + (long)c * HW;
  const T* ap = dx_add ? dx_add + (long)b * sab + (long)c * HW : nullptr;
  const int nq = HW >> 2;
  for (int q = blockIdx.y * blockDim.x + threadIdx.x; q < nq; q += gridDim.y * blockDim.x) {
    float xv[4], gv[4], av[4] = {0.f, 0.f, 0.f, 0.f};
    ld4v(xv, xp + 4 * q);
    ld4v(gv, gp + 4 * q);
    if (ap) ld4v(av, ap + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xv[e] - mu) * rs;
      float g = gv[e];
      if (act != ACT_NONE) g *= act_bwd(act, xh * gm + bt, slope);
      gv[e] = gm * rs * (g - m1 - xh * m2) + av[e];
    }
    st4v(dp + 4 * q, gv);
  }
  if (b == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    atomicAdd(&dgamma[c], s2);
    atomicAdd(&dbeta[c], s1);
  }
}

// ---- small planes (HW <= 1024, contiguous [B, C, HW] tensors, bf16) --------------------------------------------------------
// The plane-per-workgroup kernels above launch B*C workgroups of which most threads have nothing to do (7x7: 49 of 256
// lanes; 16384 workgroups each re-reading the channel sums and running two barriers) — 25 - 36 us per BatchNorm pass at the
// 7x7 and 14x14 decoder levels.  Flat walks instead: the element-wise passes take V consecutive elements of the whole tensor
// per thread and look their channel up by division; the per-channel reductions walk the channel's B*HW elements flat.
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_apply_flat_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ var,
                                                           float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int act, float slope, int C, int HWv,
                                                           long nvec, BnFin fin) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  // (flat tensors are the small decoder levels: i < 2^20, no 64-bit division — ~150 instructions — in front of four loads)
  const int plane = nvec < (1L << 20) ? cenet_div_small((int)i, HWv, 1.f / (float)HWv) : (int)(i / HWv);
  const int c = plane - cenet_div_small(plane, C, plane < (1 << 20) ? 1.f / (float)C : 0.f) * C;
  float mu, vr;
  if (fin.ws) {
    float a1 = 0.f, a2 = 0.f;
    for (int k = 0; k < fin.S; ++k) {
      a1 += fin.ws[(long)c * fin.S + k];
      a2 += fin.ws[((long)C + c) * fin.S + k];
    }
    const float m = a1 / fin.n;
    vr = a2 / fin.n - m * m;
    if (vr < 0.f) vr = 0.f;
    mu = ldf(x + (long)c * HWv * V) + m;
    if (plane < C && i == (long)plane * HWv) bn_fin_publish(fin, c, mu, vr);
  } else {
    mu = mean[c], vr = var[c];
  }
  const float sc = gamma[c] * rsqrtf(vr + eps);
  const float sh = beta[c] - mu * sc;
  float v[V];
  ldv<V>(v, x + i * V);
#pragma unroll
  for (int e = 0; e < V; ++e) v[e] = act_fwd(act, v[e] * sc + sh, slope);
  stv<V>(y + i * V, v);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_flat_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                               T* __restrict__ dx, const float* __restrict__ mean,
                                                               const float* __restrict__ var, float eps,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               int act, float slope, int C, int HWv, long nvec, float n,
                                                               const float* __restrict__ ws, int S, float* dgamma, float* dbeta,
                                                               const T* __restrict__ dx_add) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  // (flat tensors are the small decoder levels: i < 2^20, no 64-bit division — ~150 instructions — in front of four loads)
  const int plane = nvec < (1L << 20) ? cenet_div_small((int)i, HWv, 1.f / (float)HWv) : (int)(i / HWv);
  const int c = plane - cenet_div_small(plane, C, plane < (1 << 20) ? 1.f / (float)C : 0.f) * C;
  const float rs = rsqrtf(var[c] + eps), mu = mean[c], gm = gamma[c], bt = beta[c];
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < S; ++k) {
    s1 += ws[(long)c * S + k];
    s2 += ws[((long)C + c) * S + k];
  }
  const float m1 = s1 / n, m2 = s2 / n;
  float xv[V], gv[V], av[V];
  ldv<V>(xv, x + i * V);
  ldv<V>(gv, dy + i * V);
#pragma unroll
  for (int e = 0; e < V; ++e) av[e] = 0.f;
  if (dx_add) ldv<V>(av, dx_add + i * V);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float xh = (xv[e] - mu) * rs;
    float g = gv[e];
    if (act != ACT_NONE) g *= act_bwd(act, xh * gm + bt, slope);
    gv[e] = gm * rs * (g - m1 - xh * m2) + av[e];
  }
  stv<V>(dx + i * V, gv);
  if (plane < C && i == (long)plane * HWv) {  // first vector of image 0's plane of channel c
    atomicAdd(&dgamma[c], s2);
    atomicAdd(&dbeta[c], s1);
  }
}

// per-channel partial sums over a flat walk of the channel's B*HW elements (grid (C, S))
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_flat_kernel(const T* __restrict__ x, long sb, int B, int HW,
                                                             float* __restrict__ ws) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const int total = B * HW;
  const float shift = ldf(x + (long)c * HW);
  float s1 = 0.f, s2 = 0.f;
  for (int e = blockIdx.y * 256 + threadIdx.x; e < total; e += gridDim.y * 256) {
    const int b = e / HW, p = e - b * HW;
    const float d = ldf(x + (long)b * sb + (long)c * HW + p) - shift;
    s1 += d;
    s2 += d * d;
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    ws[((long)c) * gridDim.y + blockIdx.y] = s1;
    ws[((long)gridDim.x + c) * gridDim.y + blockIdx.y] = s2;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_flat_kernel(const T* __restrict__ dy, long sgb, const T* __restrict__ x,
                                                                 long sxb, const float* __restrict__ mean,
                                                                 const float* __restrict__ var, float eps,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 int act, float slope, int B, int HW, float* __restrict__ ws) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const float rs = rsqrtf(var[c] + eps), mu = mean[c], gm = gamma[c], bt = beta[c];
  const int total = B * HW;
  float s1 = 0.f, s2 = 0.f;
  for (int e = blockIdx.y * 256 + threadIdx.x; e < total; e += gridDim.y * 256) {
    const int b = e / HW, p = e - b * HW;
    const float xh = (ldf(x + (long)b * sxb + (long)c * HW + p) - mu) * rs;
    float g = ldf(dy + (long)b * sgb + (long)c * HW + p);
    if (act != ACT_NONE) g *= act_bwd(act, xh * gm + bt, slope);
    s1 += g;
    s2 += g * xh;
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    ws[((long)c) * gridDim.y + blockIdx.y] = s1;
    ws[((long)gridDim.x + c) * gridDim.y + blockIdx.y] = s2;
  }
}
// ---- one kernel per pass for channels whose whole batch fits a workgroup's registers (bf16, B*HW <= 2048, e.g. the 7x7
// decoder level at B = 32): workgroup = channel; its 256 threads hold up to 8 elements each, so the statistics, the
// normalisation (or the whole backward formula) and the write happen without a second launch or a partial-sum round trip.
#define BN1K_EPT 8   // (measured: 32 elements per thread — 14x14 maps at B = 32 — ran slower than the two-kernel flat form)
__global__ __launch_bounds__(256) void bn_train_fwd_1k_kernel(const bf16_t* __restrict__ x, long sxb, bf16_t* __restrict__ y, long syb,
                                                             float eps, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int act, float slope, int B, int HW,
                                                             BnFin fin) {
  __shared__ float red[16];
  const int c = blockIdx.x, total = B * HW;
  float v[BN1K_EPT];
  float s1 = 0.f;
#pragma unroll
  for (int k = 0; k < BN1K_EPT; ++k) {
    const int e = threadIdx.x + 256 * k;
    v[k] = 0.f;
    if (e < total) {
      const int b = e / HW, p = e - b * HW;
      v[k] = cenet_bf2f(x[(long)b * sxb + (long)c * HW + p]);
      s1 += v[k];
    }
  }
  const float n = (float)total;
  const float mu = block_sum(s1, red) / n;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < BN1K_EPT; ++k)
    if (threadIdx.x + 256 * k < total) s2 += (v[k] - mu) * (v[k] - mu);
  const float vr = block_sum(s2, red) / n;
  if (threadIdx.x == 0) bn_fin_publish(fin, c, mu, vr);
  const float sc = gamma[c] * rsqrtf(vr + eps), sh = beta[c] - mu * sc;
#pragma unroll
  for (int k = 0; k < BN1K_EPT; ++k) {
    const int e = threadIdx.x + 256 * k;
    if (e < total) {
      const int b = e / HW, p = e - b * HW;
      stf(y + (long)b * syb + (long)c * HW + p, act_fwd(act, v[k] * sc + sh, slope));
    }
  }
}

__global__ __launch_bounds__(256) void bn_bwd_1k_kernel(const bf16_t* __restrict__ dy, long sgb, const bf16_t* __restrict__ x,
                                                       long sxb, bf16_t* __restrict__ dx, long sdb,
                                                       const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                       float slope, int B, int HW, float* dgamma, float* dbeta,
                                                       const bf16_t* __restrict__ dx_add, long sab) {
  __shared__ float red[16];
  const int c = blockIdx.x, total = B * HW;
  const float rs = rsqrtf(var[c] + eps), mu = mean[c], gm = gamma[c], bt = beta[c];
  float xh[BN1K_EPT], g[BN1K_EPT];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < BN1K_EPT; ++k) {
    const int e = threadIdx.x + 256 * k;
    xh[k] = g[k] = 0.f;
    if (e < total) {
      const int b = e / HW, p = e - b * HW;
      xh[k] = (cenet_bf2f(x[(long)b * sxb + (long)c * HW + p]) - mu) * rs;
      float gg = cenet_bf2f(dy[(long)b * sgb + (long)c * HW + p]);
      if (act != ACT_NONE) gg *= act_bwd(act, xh[k] * gm + bt, slope);
      g[k] = gg;
      s1 += gg;
      s2 += gg * xh[k];
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  const float n = (float)total, m1 = s1 / n, m2 = s2 / n;
#pragma unroll
  for (int k = 0; k < BN1K_EPT; ++k) {
    const int e = threadIdx.x + 256 * k;
    if (e < total) {
      const int b = e / HW, p = e - b * HW;
      stf(dx + (long)b * sdb + (long)c * HW + p,
          gm * rs * (g[k] - m1 - xh[k] * m2) + (dx_add ? cenet_bf2f(dx_add[(long)b * sab + (long)c * HW + p]) : 0.f));
    }
  }
  if (threadIdx.x == 0) {
    atomicAdd(&dgamma[c], s2);
    atomicAdd(&dbeta[c], s1);
  }
}

// ---- BatchNorm1d on a [B, C] fp32 matrix, B <= 64 (the CCU gate, cfam.py:251-264: one value per image and channel) ------------------
// The plane kernels treat this as C planes of B strided single elements: a 32-iteration serial loop per workgroup in the partial
// passes and B * C workgroups of one busy thread in the apply passes — 5 launches and ~55 us per CCU for a 64 KB tensor.  Here the
// B values of a channel live in the registers of four threads (coalesced over the channels of a row), and a pass is one launch.
#define BN1D_MAXB 64
// workgroup = 64 channels x 4 row groups (thread (cx, gy) owns rows gy, gy + 4, ... of channel c: <= 16 values in registers, the loads
// of a wave are 256-byte row segments); the four groups meet in LDS
__global__ __launch_bounds__(256) void bn1d_train_fwd_kernel(const float* __restrict__ z, float* __restrict__ zn,
                                                            float* __restrict__ mean, float* __restrict__ var, float* rmean,
                                                            float* rvar, float momentum, long* nbt, float eps,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int B,
                                                            int C) {
  __shared__ float red[2][4][64];
  const int cx = threadIdx.x & 63, gy = threadIdx.x >> 6, c = blockIdx.x * 64 + cx;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
  const bool ok = c < C;
  float v[BN1D_MAXB / 4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < BN1D_MAXB / 4; ++i) {
    const int b = gy + 4 * i;
    v[i] = (ok && b < B) ? z[(long)b * C + c] : 0.f;
    s += v[i];
  }
  red[0][gy][cx] = s;
  __syncthreads();
  const float mu = (red[0][0][cx] + red[0][1][cx] + red[0][2][cx] + red[0][3][cx]) / B;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < BN1D_MAXB / 4; ++i)
    if (gy + 4 * i < B) q += (v[i] - mu) * (v[i] - mu);
  red[1][gy][cx] = q;
  __syncthreads();
  const float va = (red[1][0][cx] + red[1][1][cx] + red[1][2][cx] + red[1][3][cx]) / B;
  if (!ok) return;
  if (gy == 0) {
    mean[c] = mu;
    var[c] = va;
    if (rmean) {
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * va * ((float)B / (float)(B - 1));
    }
  }
  const float a = gamma[c] * rsqrtf(va + eps), sh = beta[c] - a * mu;
#pragma unroll
  for (int i = 0; i < BN1D_MAXB / 4; ++i) {
    const int b = gy + 4 * i;
    if (b < B) zn[(long)b * C + c] = a * v[i] + sh;
  }
}
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z, float* __restrict__ dz,
                                                      const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                      const float* __restrict__ gamma, float* dgamma, float* dbeta, int B, int C) {
  __shared__ float red[2][4][64];
  const int cx = threadIdx.x & 63, gy = threadIdx.x >> 6, c = blockIdx.x * 64 + cx;
  const bool ok = c < C;
  const float rs = ok ? rsqrtf(var[c] + eps) : 0.f, mu = ok ? mean[c] : 0.f, gm = ok ? gamma[c] : 0.f;
  float g[BN1D_MAXB / 4], xh[BN1D_MAXB / 4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < BN1D_MAXB / 4; ++i) {
    const int b = gy + 4 * i;
    const bool in = ok && b < B;
    g[i] = in ? dy[(long)b * C + c] : 0.f;
    xh[i] = in ? (z[(long)b * C + c] - mu) * rs : 0.f;
    s1 += g[i];
    s2 += g[i] * xh[i];
  }
  red[0][gy][cx] = s1;
  red[1][gy][cx] = s2;
  __syncthreads();
  s1 = red[0][0][cx] + red[0][1][cx] + red[0][2][cx] + red[0][3][cx];
  s2 = red[1][0][cx] + red[1][1][cx] + red[1][2][cx] + red[1][3][cx];
  if (!ok) return;
  const float m1 = s1 / B, m2 = s2 / B;
#pragma unroll
  for (int i = 0; i < BN1D_MAXB / 4; ++i) {
    const int b = gy + 4 * i;
    if (b < B) dz[(long)b * C + c] = gm * rs * (g[i] - m1 - xh[i] * m2);
  }
  if (gy == 0) {  // (this thread is the channel's only writer)
    if (dgamma) dgamma[c] += s2;
    if (dbeta) dbeta[c] += s1;
  }
}
extern "C" int cenet_bn1d_supported(int B) { return B >= 2 && B <= BN1D_MAXB; }
/* train-mode BatchNorm1d of z [B, C] (fp32, 2 <= B <= 64): zn = gamma (z - mean) / sqrt(var + eps) + beta, batch mean / biased
 * variance written, running statistics (unbiased variance) and the batch counter updated (NULL: not wanted) — one launch */
extern "C" int cenet_bn1d_train_fwd_f32(const float* z, float* zn, float* mean, float* var, float* running_mean, float* running_var,
                                        float momentum, long* num_batches_tracked, float eps, const float* gamma, const float* beta,
                                        int B, int C, hipStream_t stream) {
  if (!z || !zn || !mean || !var || !gamma || !beta || C <= 0 || (running_mean != nullptr) != (running_var != nullptr))
    return CENET_EINVAL;
  if (!cenet_bn1d_supported(B)) return CENET_EUNSUPPORTED;
  CENET_LAUNCH(bn1d_train_fwd_kernel, dim3(cdiv(C, 64)), dim3(256), stream, z, zn, mean, var, running_mean, running_var, momentum,
               num_batches_tracked, eps, gamma, beta, B, C);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
/* its backward: dz written; dgamma / dbeta ADDED into (NULL: not wanted) — one launch */
extern "C" int cenet_bn1d_bwd_acc_f32(const float* dy, const float* z, float* dz, const float* mean, const float* var, float eps,
                                      const float* gamma, float* dgamma_acc, float* dbeta_acc, int B, int C, hipStream_t stream) {
  if (!dy || !z || !dz || !mean || !var || !gamma || C <= 0) return CENET_EINVAL;
  if (!cenet_bn1d_supported(B)) return CENET_EUNSUPPORTED;
  CENET_LAUNCH(bn1d_bwd_kernel, dim3(cdiv(C, 64)), dim3(256), stream, dy, z, dz, mean, var, eps, gamma, dgamma_acc, dbeta_acc, B, C);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

// the flat forms apply: bf16 tensors (the fp32 parity mode keeps its summation order), small planes, no batch gaps
template <typename T>
static inline bool bn_flat_ok(int C, int HW, long s0, long s1, long s2) {
  const long cs = (long)C * HW;
  return sizeof(T) == 2 && HW <= 1024 && s0 == cs && s1 == cs && s2 == cs;
}

template <typename T>
static inline bool bn_v4_ok(int HW, const void* p0, long s0, const void* p1, long s1, const void* p2, long s2) {
  return (HW & 3) == 0 && ((s0 | s1 | s2) & 3) == 0 &&
         ((((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & (4 * sizeof(T) - 1)) == 0);
}
// threads per plane workgroup and plane chunks of the element-wise passes (nq = quads per plane)
static inline void bn_plane_launch(int nq, int* threads, int* chunks) {
  *threads = nq <= 64 ? 64 : (nq <= 128 ? 128 : 256);
  int ch = cdiv(nq, *threads * 2);
  if (ch > 16) ch = 16;
  if (ch < 1) ch = 1;
  *chunks = ch;
}
// workgroups per channel of the reductions: ~2048 in all, each at least 1024 quads deep
static inline int bn_splits_v4(int C, long quads) {
  long want = 2048 / (C > 0 ? C : 1);
  long maxs = (quads + 1023) / 1024;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  return (int)want;
}

static inline int bn_splits(int C, long total) {
  long want = 1024 / (C > 0 ? C : 1);
  long maxs = (total + 2047) / 2048;
  if (want < 1) want = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  return (int)want;
}

template <typename T>
static int bn_partial_launch(const T* x, long sb, int B, int C, int HW, float* ws, hipStream_t stream) {
  const long total = (long)B * HW;
  int S;
  if (bn_v4_ok<T>(HW, x, sb, x, sb, x, sb)) {
    S = bn_splits_v4(C, total / 4);
    CENET_LAUNCH((bn_partial_v4_kernel<T>), dim3(C, S), dim3(256), stream, x, sb, B, HW, ws);
  } else if (sizeof(T) == 2 && HW <= 1024) {
    S = bn_splits(C, total);
    CENET_LAUNCH((bn_partial_flat_kernel<T>), dim3(C, S), dim3(256), stream, x, sb, B, HW, ws);
  } else {
    S = bn_splits(C, total);
    CENET_LAUNCH((bn_partial_kernel<T>), dim3(C, S), dim3(256), stream, x, sb, B, HW, ws);
  }
  return S;
}

template <typename T>
static int bn_stats_impl(const T* x, long sb, int B, int C, int HW, float* ws, float* mean, float* var, float* running_mean,
                         float* running_var, float momentum, long* num_batches_tracked, hipStream_t stream, int nbt_count = 1) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  const long total = (long)B * HW;
  int S;
  if (bn_v4_ok<T>(HW, x, sb, x, sb, x, sb)) {
    S = bn_splits_v4(C, total / 4);
    CENET_LAUNCH((bn_partial_v4_kernel<T>), dim3(C, S), dim3(256), stream, x, sb, B, HW, ws);
  } else if (sizeof(T) == 2 && HW <= 1024) {
    S = bn_splits(C, total);
    CENET_LAUNCH((bn_partial_flat_kernel<T>), dim3(C, S), dim3(256), stream, x, sb, B, HW, ws);
  } else {
    S = bn_splits(C, total);
    CENET_LAUNCH((bn_partial_kernel<T>), dim3(C, S), dim3(256), stream, x, sb, B, HW, ws);
  }
  CENET_LAUNCH((bn_finalize_kernel<T>), dim3(cdiv(C, 64)), dim3(64), stream, x, HW, (const float*)ws, C, S, (float)total, mean,
               var, running_mean, running_var, momentum, num_batches_tracked, nbt_count);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(bn_stats, (const T* x, long sb, int B, int C, int HW, float* ws, float* mean, float* var, float* running_mean,
                      float* running_var, float momentum, long* num_batches_tracked, hipStream_t stream),
           (x, sb, B, C, HW, ws, mean, var, running_mean, running_var, momentum, num_batches_tracked, stream))

template <typename T>
static int bn_apply_launch(const T* x, long sxb, T* y, long syb, const float* mean, const float* var, float eps,
                           const float* gamma, const float* beta, int act, float slope, int B, int C, int HW, const BnFin& fin,
                           hipStream_t stream) {
  if (bn_flat_ok<T>(C, HW, sxb, syb, sxb)) {
    if (bn_v4_ok<T>(HW, x, sxb, y, syb, x, sxb)) {
      const long nvec = (long)B * C * HW / 4;
      CENET_LAUNCH((bn_apply_flat_kernel<T, 4>), dim3(cdiv(nvec, 256)), dim3(256), stream, x, y, mean, var, eps, gamma, beta, act,
                   slope, C, HW / 4, nvec, fin);
    } else {
      const long nvec = (long)B * C * HW;
      CENET_LAUNCH((bn_apply_flat_kernel<T, 1>), dim3(cdiv(nvec, 256)), dim3(256), stream, x, y, mean, var, eps, gamma, beta, act,
                   slope, C, HW, nvec, fin);
    }
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (bn_v4_ok<T>(HW, x, sxb, y, syb, x, sxb)) {
    int threads, chunks;
    bn_plane_launch(HW / 4, &threads, &chunks);
    CENET_LAUNCH((bn_apply_v4_kernel<T>), dim3(B * C, chunks), dim3(threads), stream, x, sxb, y, syb, mean, var, eps, gamma,
                 beta, act, slope, C, HW, fin);
  } else {
    int chunks = cdiv(HW, 1024);
    if (chunks > 64) chunks = 64;
    CENET_LAUNCH((bn_apply_kernel<T>), dim3(B * C, chunks), dim3(256), stream, x, sxb, y, syb, mean, var, eps, gamma, beta, act,
                 slope, C, HW, fin);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

template <typename T>
static int bn_apply_impl(const T* x, long sxb, T* y, long syb, const float* mean, const float* var, float eps,
                         const float* gamma, const float* beta, int act, float slope, int B, int C, int HW,
                         hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  BnFin fin;
  fin.ws = nullptr; fin.S = 0; fin.n = 0.f; fin.mean = fin.var = fin.rmean = fin.rvar = nullptr; fin.momentum = 0.f; fin.nbt = nullptr; fin.nbt_n = 1;
  return bn_apply_launch<T>(x, sxb, y, syb, mean, var, eps, gamma, beta, act, slope, B, C, HW, fin, stream);
}

// train-mode forward: partial sums, then ONE pass that finalises the statistics, normalises, and publishes mean / var /
// running statistics (bf16 tensors; fp32 tensors keep the three-launch form and its summation order)
template <typename T>
static int bn_train_fwd_impl(const T* x, long sxb, T* y, long syb, float* ws, float* mean, float* var, float* running_mean,
                             float* running_var, float momentum, long* num_batches_tracked, float eps, const float* gamma,
                             const float* beta, int act, float slope, int B, int C, int HW, int nbt_count, hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || nbt_count < 1 || nbt_count > C) return CENET_EINVAL;
  if (sizeof(T) != 2) {
    const int rc = bn_stats_impl<T>(x, sxb, B, C, HW, ws, mean, var, running_mean, running_var, momentum, num_batches_tracked,
                                    stream, nbt_count);
    if (rc != CENET_OK) return rc;
    return bn_apply_impl<T>(x, sxb, y, syb, mean, var, eps, gamma, beta, act, slope, B, C, HW, stream);
  }
  if ((long)B * HW <= 256 * BN1K_EPT && (long)B * HW >= 2) {
    BnFin f1;
    f1.ws = nullptr; f1.S = 0; f1.n = (float)((long)B * HW); f1.mean = mean; f1.var = var; f1.rmean = running_mean;
    f1.rvar = running_var; f1.momentum = momentum; f1.nbt = num_batches_tracked; f1.nbt_n = nbt_count;
    CENET_LAUNCH(bn_train_fwd_1k_kernel, dim3(C), dim3(256), stream, (const bf16_t*)x, sxb, (bf16_t*)y, syb, eps, gamma, beta, act,
                 slope, B, HW, f1);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  BnFin fin;
  fin.S = bn_partial_launch<T>(x, sxb, B, C, HW, ws, stream);
  fin.ws = ws; fin.n = (float)((long)B * HW); fin.mean = mean; fin.var = var; fin.rmean = running_mean; fin.rvar = running_var;
  fin.momentum = momentum; fin.nbt = num_batches_tracked; fin.nbt_n = nbt_count;
  return bn_apply_launch<T>(x, sxb, y, syb, mean, var, eps, gamma, beta, act, slope, B, C, HW, fin, stream);
}
CENET_TWIN(bn_train_fwd, (const T* x, long sxb, T* y, long syb, float* ws, float* mean, float* var, float* running_mean,
                          float* running_var, float momentum, long* num_batches_tracked, float eps, const float* gamma,
                          const float* beta, int act, float slope, int B, int C, int HW, int nbt_count, hipStream_t stream),
           (x, sxb, y, syb, ws, mean, var, running_mean, running_var, momentum, num_batches_tracked, eps, gamma, beta, act, slope, B,
            C, HW, nbt_count, stream))
CENET_TWIN(bn_apply, (const T* x, long sxb, T* y, long syb, const float* mean, const float* var, float eps, const float* gamma,
                      const float* beta, int act, float slope, int B, int C, int HW, hipStream_t stream),
           (x, sxb, y, syb, mean, var, eps, gamma, beta, act, slope, B, C, HW, stream))

template <typename T>
static int bn_bwd_acc_impl(const T* dy, long sgb, const T* x, long sxb, T* dx, long sdb, const float* mean, const float* var,
                           float eps, const float* gamma, const float* beta, int act, float slope, int B, int C, int HW,
                           float* ws, float* dgamma_acc, float* dbeta_acc, hipStream_t stream, const T* dx_add = nullptr,
                           long sab = 0) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  const long total = (long)B * HW;
  if (sizeof(T) == 2 && total <= 256 * BN1K_EPT) {
    CENET_LAUNCH(bn_bwd_1k_kernel, dim3(C), dim3(256), stream, (const bf16_t*)dy, sgb, (const bf16_t*)x, sxb, (bf16_t*)dx, sdb, mean,
                 var, eps, gamma, beta, act, slope, B, HW, dgamma_acc, dbeta_acc, (const bf16_t*)dx_add, sab);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (bn_flat_ok<T>(C, HW, sgb, sxb, sdb) && (!dx_add || sab == (long)C * HW)) {
    const bool v4 = bn_v4_ok<T>(HW, dy, sgb, x, sxb, dx, sdb) && bn_v4_ok<T>(HW, dx_add, sab, x, sxb, dx, sdb);
    const int S = v4 ? bn_splits_v4(C, total / 4) : bn_splits(C, total);
    if (v4)
      CENET_LAUNCH((bn_bwd_partial_v4_kernel<T>), dim3(C, S), dim3(256), stream, dy, sgb, x, sxb, mean, var, eps, gamma, beta, act,
                   slope, B, HW, ws);
    else
      CENET_LAUNCH((bn_bwd_partial_flat_kernel<T>), dim3(C, S), dim3(256), stream, dy, sgb, x, sxb, mean, var, eps, gamma, beta,
                   act, slope, B, HW, ws);
    if (v4) {
      const long nvec = (long)B * C * HW / 4;
      CENET_LAUNCH((bn_bwd_apply_flat_kernel<T, 4>), dim3(cdiv(nvec, 256)), dim3(256), stream, dy, x, dx, mean, var, eps, gamma,
                   beta, act, slope, C, HW / 4, nvec, (float)total, (const float*)ws, S, dgamma_acc, dbeta_acc, dx_add);
    } else {
      const long nvec = (long)B * C * HW;
      CENET_LAUNCH((bn_bwd_apply_flat_kernel<T, 1>), dim3(cdiv(nvec, 256)), dim3(256), stream, dy, x, dx, mean, var, eps, gamma,
                   beta, act, slope, C, HW, nvec, (float)total, (const float*)ws, S, dgamma_acc, dbeta_acc, dx_add);
    }
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (bn_v4_ok<T>(HW, dy, sgb, x, sxb, dx, sdb) && bn_v4_ok<T>(HW, dx_add, sab, x, sxb, dx, sdb)) {
    const int S = bn_splits_v4(C, total / 4);
    CENET_LAUNCH((bn_bwd_partial_v4_kernel<T>), dim3(C, S), dim3(256), stream, dy, sgb, x, sxb, mean, var, eps, gamma, beta, act,
                 slope, B, HW, ws);
    int threads, chunks;
    bn_plane_launch(HW / 4, &threads, &chunks);
    CENET_LAUNCH((bn_bwd_apply_v4_kernel<T>), dim3(B * C, chunks), dim3(threads), stream, dy, sgb, x, sxb, dx, sdb, mean, var, eps,
                 gamma, beta, act, slope, C, HW, (float)total, (const float*)ws, S, dgamma_acc, dbeta_acc, dx_add, sab);
  } else {
    const int S = bn_splits(C, total);
    CENET_LAUNCH((bn_bwd_partial_kernel<T>), dim3(C, S), dim3(256), stream, dy, sgb, x, sxb, mean, var, eps, gamma, beta, act, slope,
                 B, HW, ws);
    int chunks = cdiv(HW, 1024);
    if (chunks > 64) chunks = 64;
    CENET_LAUNCH((bn_bwd_apply_kernel<T>), dim3(B * C, chunks), dim3(256), stream, dy, sgb, x, sxb, dx, sdb, mean, var, eps, gamma,
                 beta, act, slope, C, HW, (float)total, (const float*)ws, S, dgamma_acc, dbeta_acc, dx_add, sab);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(bn_bwd_acc, (const T* dy, long sgb, const T* x, long sxb, T* dx, long sdb, const float* mean, const float* var,
                        float eps, const float* gamma, const float* beta, int act, float slope, int B, int C, int HW, float* ws,
                        float* dgamma_acc, float* dbeta_acc, hipStream_t stream),
           (dy, sgb, x, sxb, dx, sdb, mean, var, eps, gamma, beta, act, slope, B, C, HW, ws, dgamma_acc, dbeta_acc, stream))
// ... + dx_add (batch stride sab): the gradient of a residual connection that by-passed this BatchNorm (cfam.py:365-374: x +
// ls * branch(BN(x))) added by the kernel that writes dx, instead of an aten::add behind it
template <typename T>
static int bn_bwd_add_acc_impl(const T* dy, long sgb, const T* x, long sxb, T* dx, long sdb, const T* dx_add, long sab,
                               const float* mean, const float* var, float eps, const float* gamma, const float* beta, int act,
                               float slope, int B, int C, int HW, float* ws, float* dgamma_acc, float* dbeta_acc,
                               hipStream_t stream) {
  return bn_bwd_acc_impl<T>(dy, sgb, x, sxb, dx, sdb, mean, var, eps, gamma, beta, act, slope, B, C, HW, ws, dgamma_acc, dbeta_acc,
                            stream, dx_add, sab);
}
CENET_TWIN(bn_bwd_add_acc, (const T* dy, long sgb, const T* x, long sxb, T* dx, long sdb, const T* dx_add, long sab,
                            const float* mean, const float* var, float eps, const float* gamma, const float* beta, int act,
                            float slope, int B, int C, int HW, float* ws, float* dgamma_acc, float* dbeta_acc, hipStream_t stream),
           (dy, sgb, x, sxb, dx, sdb, dx_add, sab, mean, var, eps, gamma, beta, act, slope, B, C, HW, ws, dgamma_acc, dbeta_acc,
            stream))
