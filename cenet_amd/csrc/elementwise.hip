// elementwise.hip — HBM-bound glue kernels of the CENet path (layout changes, gates, residual mixes, reducers).
//   token <-> NCHW transposes                 pvtv2.py:320-321, 93, 366-368
//   SiLU(g)*SiLU(v)                           cfam.py:302
//   (1-w)x + w p                              nlb.py:147
//   x + layer_scale * y                       cfam.py:368-373
//   add (+LeakyReLU)                          unet.py:212-213, decoders.py:96,100,104
//   DSEB combine: (FEA(y)+y) + diff*y         dseb.py:40-50,63-76,156-163
//   differential-attention combine + RMSNorm  multihead_diffattn.py:112-123, rms_norm.py:15-22
//   per-channel / per-column gradient reducers (bias, layer-scale, FEA weight gradients)
#include "common.h"
#include "../../include/cenet_hip.h"

#define EW_GRID(total) dim3((unsigned)((((total) + 255) / 256) > 8192 ? 8192 : (((total) + 255) / 256)))

// ---- batched 2-D transpose: y[b][j][i] = x[b][i][j], x: [R x Cc] per batch -------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, long sxb, float* __restrict__ y, long syb,
                                                       int R, int Cc) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* xb = x + (long)b * sxb;
  float* yb = y + (long)b * syb;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < Cc) ? xb[(long)r * Cc + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < Cc) yb[(long)c * R + r] = tile[tx][ty + 8 * i];
  }
}

// ---- strided batch copy: y[b*syb + i] = x[b*sxb + i], i < n (channel-slice concat / split) ----------------------
__global__ __launch_bounds__(256) void copy_batched_kernel(const float* __restrict__ x, long sxb, float* __restrict__ y,
                                                          long syb, long n, int accumulate) {
  const int b = blockIdx.y;
  const float* xb = x + (long)b * sxb;
  float* yb = y + (long)b * syb;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    yb[i] = accumulate ? yb[i] + xb[i] : xb[i];
}

// ---- space-to-depth of a token map: the kernel == stride spatial-reduction conv of pvtv2.py:93-95 reads each input
// pixel exactly once, so gathering the s x s patches into rows turns it into a dense GEMM with both operands k-contiguous.
// tok [B, Ho*S, Wo*S, C] <-> patch [B*Ho*Wo, C*S*S], k = (c, ky, kx) as in the conv weight [Cout, C, S, S].
// One thread moves the S*S values of one (patch, channel): token side coalesced over c, patch side S contiguous floats.
template <int S, bool INV>
__global__ __launch_bounds__(256) void patch_tok_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int Ho,
                                                       int Wo, long total) {
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
      float v[S];
      if (INV) {
        const float* p = src + pbase + ky * S;
        if (S == 2) {
          const float2 q = *reinterpret_cast<const float2*>(p);
          v[0] = q.x, v[1] = q.y;
        } else {
#pragma unroll
          for (int j = 0; j < S; j += 4) {
            const float4 q = *reinterpret_cast<const float4*>(p + j);
            v[j] = q.x, v[j + 1] = q.y, v[j + 2] = q.z, v[j + 3] = q.w;
          }
        }
#pragma unroll
        for (int kx = 0; kx < S; ++kx) dst[tbase + (ky * Wd + kx) * C] = v[kx];
      } else {
#pragma unroll
        for (int kx = 0; kx < S; ++kx) v[kx] = src[tbase + (ky * Wd + kx) * C];
        float* p = dst + pbase + ky * S;
        if (S == 2) {
          float2 q;
          q.x = v[0], q.y = v[1];
          *reinterpret_cast<float2*>(p) = q;
        } else {
#pragma unroll
          for (int j = 0; j < S; j += 4) *reinterpret_cast<float4*>(p + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
        }
      }
    }
  }
}

// ---- y[b, i] = s[b] * x[b, i] ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_batch_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                         float* __restrict__ y, long n) {
  const int b = blockIdx.y;
  const float sv = s[b];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[b * n + i] = sv * x[b * n + i];
}

// ---- dx = dy * act'(pre) -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dy,
                                                     float* __restrict__ dx, long n, int act, float slope) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    dx[i] = dy[i] * act_bwd(act, pre[i], slope);
}
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act,
                                                     float slope) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = act_fwd(act, x[i], slope);
}

// ---- y = silu(a)*silu(b) -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void silu_mul_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = act_fwd(ACT_SILU, a[i], 0.f) * act_fwd(ACT_SILU, b[i], 0.f);
}
__global__ __launch_bounds__(256) void silu_mul_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ dy, float* __restrict__ da,
                                                          float* __restrict__ db, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float g = dy[i], av = a[i], bv = b[i];
    da[i] = g * act_fwd(ACT_SILU, bv, 0.f) * act_bwd(ACT_SILU, av, 0.f);
    db[i] = g * act_fwd(ACT_SILU, av, 0.f) * act_bwd(ACT_SILU, bv, 0.f);
  }
}

// ---- z = (1-w)x + w p ; w is a device scalar ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void mix_fwd_kernel(const float* __restrict__ x, const float* __restrict__ p,
                                                     const float* __restrict__ w, float* __restrict__ z, long n) {
  const float wv = w[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) z[i] = (1.f - wv) * x[i] + wv * p[i];
}
__global__ __launch_bounds__(256) void mix_bwd_kernel(const float* __restrict__ x, const float* __restrict__ p,
                                                     const float* __restrict__ w, const float* __restrict__ dz,
                                                     float* __restrict__ dx, float* __restrict__ dp, float* __restrict__ dw,
                                                     long n) {
  __shared__ float red[16];
  const float wv = w[0];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float g = dz[i];
    dx[i] = (1.f - wv) * g;
    dp[i] = wv * g;
    s += g * (p[i] - x[i]);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(dw, s);
}

// ---- out = x + ls[c]*y (NCHW) ; dy = ls[c]*g -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_residual_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ ls, float* __restrict__ out, int C,
                                                                int HW) {
  const int bc = blockIdx.x, c = bc % C;
  const float s = ls[c];
  const long base = (long)bc * HW;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < HW; p += gridDim.y * 256) out[base + p] = x[base + p] + s * y[base + p];
}
__global__ __launch_bounds__(256) void scale_chan_kernel(const float* __restrict__ g, const float* __restrict__ ls,
                                                        float* __restrict__ out, int C, int HW) {
  const int bc = blockIdx.x, c = bc % C;
  const float s = ls[c];
  const long base = (long)bc * HW;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < HW; p += gridDim.y * 256) out[base + p] = s * g[base + p];
}

// ---- out[c] += sum_{b,p} a[b,c,p] * (bb ? bb[b,c,p] : 1)  (grid C x splits) -------------------------------------
__global__ __launch_bounds__(256) void chan_dot_kernel(const float* __restrict__ a, long sab, const float* __restrict__ bb,
                                                      long sbb, float* __restrict__ out, int B, int HW) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const long total = (long)B * HW;
  float s = 0.f;
  for (int w__ = blockIdx.y; w__ < B * ((HW + 1023) / 1024); w__ += gridDim.y)  // (image, 1024-pixel chunk) items: no per-element division
  for (int b = w__ / ((HW + 1023) / 1024), p = (w__ - b * ((HW + 1023) / 1024)) * 1024 + threadIdx.x,
           pend__ = ((w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 < HW) ? (w__ - b * ((HW + 1023) / 1024)) * 1024 + 1024 : HW;
       p < pend__; p += 256) {
    const float av = a[(long)b * sab + (long)c * HW + p];
    s += bb ? av * bb[(long)b * sbb + (long)c * HW + p] : av;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(&out[c], s);
}

// 16-byte form (HW % 4 == 0): flat grid-stride walk over the channel's B*HW/4 quads, two independent loads in flight
__global__ __launch_bounds__(256) void chan_dot_v4_kernel(const float* __restrict__ a, long sab, const float* __restrict__ bb,
                                                         long sbb, float* __restrict__ out, int B, int HW) {
  __shared__ float red[16];
  const int c = blockIdx.x, nq = HW >> 2, total = B * nq, stride = gridDim.y * 256;
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
        const int b = q / nq, qi = q - b * nq;
        memcpy(av[u], a + (long)b * sab + (long)c * HW + 4 * qi, 16);
        if (bb) memcpy(bv[u], bb + (long)b * sbb + (long)c * HW + 4 * qi, 16);
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
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ a, float* __restrict__ out, long R, int C) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long r0 = (long)blockIdx.y * CS_ROWS;
  float s = 0.f;
  if (c < C)
    for (long r = r0 + rl; r < r0 + CS_ROWS && r < R; r += 4) s += a[r * C + c];
  part[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) atomicAdd(&out[c], part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl]);
}
// 16-byte form (C % 4 == 0): a thread owns one column quad, TPR threads cover a row slab of 4*TPR columns and the other
// 256/TPR thread groups take interleaved rows, 4 independent 16-byte loads in flight each
template <int TPR>
__global__ __launch_bounds__(256) void col_sum_v4_kernel(const float* __restrict__ a, float* __restrict__ out, long R, int C,
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
      for (int u = 0; u < 4; ++u) memcpy(v[u], a + (r + u * RL) * C + c, 16);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += v[u][e];
    }
    for (; r < r1; r += RL) {
      float v[4];
      memcpy(v, a + r * C + c, 16);
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
__global__ __launch_bounds__(256) void add_act_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, long n, int act, float slope) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = act_fwd(act, a[i] + b[i], slope);
}
__global__ __launch_bounds__(256) void lrelu_bwd_from_out_kernel(const float* __restrict__ out, const float* __restrict__ dy,
                                                                float* __restrict__ dx, long n, float slope) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    dx[i] = out[i] > 0.f ? dy[i] : dy[i] * slope;
}

// ---- DSEB combine ---------------------------------------------------------------------------------------------------
// z = 2y + w[c]*edge + diff*y,  edge = (1/m) sum_{i<j} | e_i - e_j |,  e_s = | y - r_s |  (r_s == nullptr: scale 1.0 -> e_s = 0)
struct DsebArgs {
  const float* y;
  const float* r[3];
  const float* w;
  const float* diff;
  float* z;
  // backward
  const float* dz;
  float* dy;
  float* dr[3];
  float* ddiff;
  float* dw;
  int n, C, HW;
  float ycoef;
};
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void dseb_combine_fwd_kernel(DsebArgs a) {
  const int bc = blockIdx.x, c = bc % a.C;
  const float wc = a.w[c];
  const long base = (long)bc * a.HW;
  const float inv_m = 1.f / (float)(a.n * (a.n - 1) / 2);
  for (int p = blockIdx.y * 256 + threadIdx.x; p < a.HW; p += gridDim.y * 256) {
    const float yv = a.y[base + p];
    float e[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) e[s] = (s < a.n && a.r[s]) ? fabsf(yv - a.r[s][base + p]) : 0.f;
    float edge = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i + 1; j < 3; ++j)
        if (j < a.n) edge += fabsf(e[i] - e[j]);
    edge *= inv_m;
    a.z[base + p] = a.ycoef * yv + wc * edge + (a.diff ? a.diff[base + p] * yv : 0.f);
  }
}
__global__ __launch_bounds__(256) void dseb_combine_bwd_kernel(DsebArgs a) {
  __shared__ float red[16];
  const int bc = blockIdx.x, c = bc % a.C;
  const float wc = a.w[c];
  const long base = (long)bc * a.HW;
  const float inv_m = 1.f / (float)(a.n * (a.n - 1) / 2);
  float dws = 0.f;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < a.HW; p += gridDim.y * 256) {
    const float yv = a.y[base + p], g = a.dz[base + p], df = a.diff ? a.diff[base + p] : 0.f;
    float e[3], sy[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (s < a.n && a.r[s]) {
        const float d = yv - a.r[s][base + p];
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
      if (s < a.n && a.r[s]) {
        const float t = wc * g * inv_m * de[s] * sy[s];
        dyv += t;
        a.dr[s][base + p] = -t;
      }
    }
    a.dy[base + p] = dyv;
    if (a.ddiff) a.ddiff[base + p] = g * yv;
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
__global__ __launch_bounds__(256) void diffattn_combine_fwd_kernel(const float* __restrict__ U, const float* __restrict__ lam,
                                                                  float* __restrict__ out, int H, int N, int dv, float eps,
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
  const float* u0 = U + (((b * 2 * H) + 2 * h) * N + n) * (long)dv;
  const float* u1 = u0 + (long)N * dv;
  float ss = 0.f;
  if (ok)
    for (int d = sl; d < dv; d += sub) {
      const float av = u0[d] - lm * u1[d];
      ss += av * av;
    }
  const float r = rsqrtf(subwave_sum(ss, sub) / dv + eps) * post;
  if (ok) {
    float* o = out + (b * N + n) * (long)(H * dv) + (long)h * dv;
    for (int d = sl; d < dv; d += sub) o[d] = (u0[d] - lm * u1[d]) * r;
  }
}
__global__ __launch_bounds__(256) void diffattn_combine_bwd_kernel(const float* __restrict__ U, const float* __restrict__ lam,
                                                                  const float* __restrict__ dout, float* __restrict__ dU,
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
    const float* u0 = U + off0;
    const float* u1 = u0 + (long)N * dv;
    const float* g = dout + (b * N + n) * (long)(H * dv) + (long)h * dv;
    float ss = 0.f, ga = 0.f;
    if (ok)
      for (int d = sl; d < dv; d += sub) {
        const float av = u0[d] - lm * u1[d];
        ss += av * av;
        ga += g[d] * av;
      }
    ss = subwave_sum(ss, sub);
    ga = subwave_sum(ga, sub);
    const float r = rsqrtf(ss / dv + eps);
    const float k = r * r * ga / dv;
    if (ok) {
      float* d0 = dU + off0;
      float* d1 = d0 + (long)N * dv;
      for (int d = sl; d < dv; d += sub) {
        const float av = u0[d] - lm * u1[d];
        const float da = post * r * (g[d] - av * k);
        d0[d] = da;
        d1[d] = -lm * da;
        dl -= da * u1[d];
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

extern "C" int cenet_transpose_f32(const float* x, long sxb, float* y, long syb, int B, int R, int Cc, hipStream_t stream) {
  if (B <= 0 || R <= 0 || Cc <= 0) return CENET_EINVAL;
  CENET_LAUNCH(transpose_kernel, dim3(cdiv(Cc, 32), cdiv(R, 32), B), dim3(256), stream, x, sxb, y, syb, R, Cc);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_copy_batched_f32(const float* x, long sxb, float* y, long syb, int B, long n, int accumulate,
                                      hipStream_t stream) {
  if (B <= 0 || n <= 0) return CENET_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  CENET_LAUNCH(copy_batched_kernel, dim3((unsigned)blocks, B), dim3(256), stream, x, sxb, y, syb, n, accumulate);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_patch_tok_f32(const float* src, float* dst, int B, int Ho, int Wo, int C, int S, int inverse,
                                   hipStream_t stream) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (S != 2 && S != 4 && S != 8)) return CENET_EINVAL;
  const long total = (long)B * Ho * Wo * C;
#define PATCH_GO(S_, INV_) \
  CENET_LAUNCH((patch_tok_kernel<S_, INV_>), EW_GRID(total), dim3(256), stream, src, dst, C, Ho, Wo, total)
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
extern "C" int cenet_scale_batch_f32(const float* x, const float* s, float* y, int B, long n, hipStream_t stream) {
  if (B <= 0 || n <= 0) return CENET_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  CENET_LAUNCH(scale_batch_kernel, dim3((unsigned)blocks, B), dim3(256), stream, x, s, y, n);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_act_fwd_f32(const float* x, float* y, long n, int act, float slope, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  CENET_LAUNCH(act_fwd_kernel, EW_GRID(n), dim3(256), stream, x, y, n, act, slope);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_act_bwd_f32(const float* pre, const float* dy, float* dx, long n, int act, float slope,
                                 hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  CENET_LAUNCH(act_bwd_kernel, EW_GRID(n), dim3(256), stream, pre, dy, dx, n, act, slope);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_silu_mul_fwd_f32(const float* a, const float* b, float* y, long n, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  CENET_LAUNCH(silu_mul_fwd_kernel, EW_GRID(n), dim3(256), stream, a, b, y, n);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_silu_mul_bwd_f32(const float* a, const float* b, const float* dy, float* da, float* db, long n,
                                      hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  CENET_LAUNCH(silu_mul_bwd_kernel, EW_GRID(n), dim3(256), stream, a, b, dy, da, db, n);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_mix_fwd_f32(const float* x, const float* p, const float* w, float* z, long n, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  CENET_LAUNCH(mix_fwd_kernel, EW_GRID(n), dim3(256), stream, x, p, w, z, n);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_mix_bwd_acc_f32(const float* x, const float* p, const float* w, const float* dz, float* dx, float* dp,
                                     float* dw_acc, long n, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  CENET_LAUNCH(mix_bwd_kernel, dim3((unsigned)blocks), dim3(256), stream, x, p, w, dz, dx, dp, dw_acc, n);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_scale_residual_fwd_f32(const float* x, const float* y, const float* ls, float* out, int B, int C, int HW,
                                            hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  CENET_LAUNCH(scale_residual_fwd_kernel, dim3(B * C, chunks_for(HW)), dim3(256), stream, x, y, ls, out, C, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_scale_chan_f32(const float* g, const float* ls, float* out, int B, int C, int HW, hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  CENET_LAUNCH(scale_chan_kernel, dim3(B * C, chunks_for(HW)), dim3(256), stream, g, ls, out, C, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_chan_dot_acc_f32(const float* a, long sab, const float* b, long sbb, float* out_acc, int B, int C, int HW,
                                      hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  long total = (long)B * HW;
  long want = 1024 / C, maxs = (total + 2047) / 2048;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  if ((HW & 3) == 0 && ((sab | sbb) & 3) == 0 && ((((uintptr_t)a | (uintptr_t)b) & 15) == 0)) {
    long w4 = 2048 / C, m4 = (total / 4 + 1023) / 1024;
    if (w4 > m4) w4 = m4;
    if (w4 < 1) w4 = 1;
    if (w4 > 256) w4 = 256;
    CENET_LAUNCH(chan_dot_v4_kernel, dim3(C, (unsigned)w4), dim3(256), stream, a, sab, b, sbb, out_acc, B, HW);
  } else {
    CENET_LAUNCH(chan_dot_kernel, dim3(C, (unsigned)want), dim3(256), stream, a, sab, b, sbb, out_acc, B, HW);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_col_sum_acc_f32(const float* a, float* out_acc, long R, int C, hipStream_t stream) {
  if (R <= 0 || C <= 0) return CENET_EINVAL;
  if ((C & 3) == 0 && (((uintptr_t)a) & 15) == 0) {
    // ~2048 workgroups, each at least 64 rows deep
    const int quads = C / 4;
    const int tpr = quads >= 64 ? 64 : (quads >= 32 ? 32 : 16);
    const int ctiles = cdiv(quads, tpr);
    long rpb = (R * ctiles + 2047) / 2048;
    if (rpb < 64) rpb = 64;
    rpb = (rpb + 15) & ~15L;
    dim3 grid(ctiles, (unsigned)((R + rpb - 1) / rpb));
    if (tpr == 64) CENET_LAUNCH((col_sum_v4_kernel<64>), grid, dim3(256), stream, a, out_acc, R, C, (int)rpb);
    else if (tpr == 32) CENET_LAUNCH((col_sum_v4_kernel<32>), grid, dim3(256), stream, a, out_acc, R, C, (int)rpb);
    else CENET_LAUNCH((col_sum_v4_kernel<16>), grid, dim3(256), stream, a, out_acc, R, C, (int)rpb);
  } else {
    CENET_LAUNCH(col_sum_kernel, dim3(cdiv(C, 64), (unsigned)((R + CS_ROWS - 1) / CS_ROWS)), dim3(256), stream, a, out_acc, R, C);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_add_act_fwd_f32(const float* a, const float* b, float* out, long n, int act, float slope,
                                     hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  CENET_LAUNCH(add_act_fwd_kernel, EW_GRID(n), dim3(256), stream, a, b, out, n, act, slope);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_lrelu_bwd_from_out_f32(const float* out, const float* dy, float* dx, long n, float slope,
                                            hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  CENET_LAUNCH(lrelu_bwd_from_out_kernel, EW_GRID(n), dim3(256), stream, out, dy, dx, n, slope);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_dseb_combine_fwd_f32(const float* y, const float* r0, const float* r1, const float* r2, int n,
                                          const float* w, const float* diff, float ycoef, float* z, int B, int C,
                                          int HW, hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || n < 2 || n > 3) return CENET_EINVAL;
  DsebArgs a;
  memset(&a, 0, sizeof(a));
  a.y = y; a.r[0] = r0; a.r[1] = r1; a.r[2] = r2; a.w = w; a.diff = diff; a.z = z; a.n = n; a.C = C; a.HW = HW; a.ycoef = ycoef;
  CENET_LAUNCH(dseb_combine_fwd_kernel, dim3(B * C, chunks_for(HW)), dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_dseb_combine_bwd_acc_f32(const float* y, const float* r0, const float* r1, const float* r2, int n,
                                              const float* w, const float* diff, float ycoef, const float* dz, float* dy,
                                              float* dr0, float* dr1, float* dr2, float* ddiff, float* dw_acc, int B, int C,
                                              int HW, hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || n < 2 || n > 3) return CENET_EINVAL;
  DsebArgs a;
  memset(&a, 0, sizeof(a));
  a.y = y; a.r[0] = r0; a.r[1] = r1; a.r[2] = r2; a.w = w; a.diff = diff; a.n = n; a.C = C; a.HW = HW; a.ycoef = ycoef;
  a.dz = dz; a.dy = dy; a.dr[0] = dr0; a.dr[1] = dr1; a.dr[2] = dr2; a.ddiff = ddiff; a.dw = dw_acc;
  CENET_LAUNCH(dseb_combine_bwd_kernel, dim3(B * C, chunks_for(HW)), dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
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
extern "C" int cenet_diffattn_combine_fwd_f32(const float* U, const float* lam3, float* out, int B, int H, int N, int dv,
                                              float eps, float post, hipStream_t stream) {
  if (B <= 0 || H <= 0 || N <= 0 || dv <= 0) return CENET_EINVAL;
  long nvec = (long)B * H * N;
  const int sub = dv <= 16 ? 16 : (dv <= 32 ? 32 : 64);
  const long per_block = 4L * (64 / sub);
  CENET_LAUNCH(diffattn_combine_fwd_kernel, dim3((unsigned)((nvec + per_block - 1) / per_block)), dim3(256), stream, U, lam3, out,
               H, N, dv, eps, post, nvec, sub);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_diffattn_combine_bwd_acc_f32(const float* U, const float* lam3, const float* dout, float* dU,
                                                  float* dlam_acc, int B, int H, int N, int dv, float eps, float post,
                                                  hipStream_t stream) {
  if (B <= 0 || H <= 0 || N <= 0 || dv <= 0) return CENET_EINVAL;
  long nvec = (long)B * H * N;
  const int sub = dv <= 16 ? 16 : (dv <= 32 ? 32 : 64);
  const long per_block = 4L * (64 / sub);
  long blocks = (nvec + per_block - 1) / per_block;
  if (blocks > 2048) blocks = 2048;
  CENET_LAUNCH(diffattn_combine_bwd_kernel, dim3((unsigned)blocks), dim3(256), stream, U, lam3, dout, dU, dlam_acc, H, N, dv, eps,
               post, nvec, sub);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
