// stats.hip — the two "style statistics" gates of the CFAM decoder (HBM-bound full-tensor reductions).
//   CCU  : per-(b,c) plane [max, mean, std(biased)] -> grouped conv1d k3 -> ReLU -> k1 -> (BN1d) -> sigmoid gate
//          cfam.py:251-264
//   SRM  : per-pixel channel [max, mean, std(unbiased)] -> 1x1 + 3x3 conv (3->1) -> GELU -> BN(1) -> sigmoid gate
//          cfam.py:93-101
// The BN steps in the middle reuse the generic BatchNorm kernels of norm.hip; the kernels here do the
// reductions, the tiny per-channel MLP / 3->1 conv, and the scatter of the statistic gradients back onto x.
// Templates over the activation storage type T (x, y, dy, dx: float or bf16_t); the statistic maps (u, z, f, du, df: 1/HW or
// 1/C of the activation) are always fp32.  Plane kernels take a vector width V (4 when HW % 4 == 0 and the planes are
// quad-aligned, else 1); HWv = HW / V.
#include "common.h"
#include <cstdlib>
#include "../../include/cenet_hip.h"

// ---------------------------------------------------------------------------------------------------------------
// CCU forward: grid (B*C); u[bc*3+{0,1,2}] = max, mean, std ; amax[bc] = argmax ; z[bc] = fc2(relu(fc1(u)))
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void ccu_stats_fwd_kernel(const T* __restrict__ x, const float* __restrict__ fc1,
                                                           const float* __restrict__ fc2, float* __restrict__ u,
                                                           int* __restrict__ amax, float* __restrict__ z, int C, int HWv) {
  __shared__ float red[16];
  __shared__ float rv[4];
  __shared__ int ri[4];
  const int bc = blockIdx.x, c = bc % C, HW = HWv * V;
  const T* xp = x + (long)bc * HW;
  float s = 0.f, mx = -3.4e38f;
  int mi = 0;
  for (int p = threadIdx.x; p < HWv; p += 256) {
    float v[V];
    ldv<V>(v, xp + p * V);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      s += v[e];
      if (v[e] > mx) {
        mx = v[e];
        mi = p * V + e;
      }
    }
  }
  const float mean = block_sum(s, red) / HW;
  float q = 0.f;
  for (int p = threadIdx.x; p < HWv; p += 256) {
    float v[V];
    ldv<V>(v, xp + p * V);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float d = v[e] - mean;
      q += d * d;
    }
  }
  const float var = block_sum(q, red) / HW;
  // arg-max: wave reduce (value, smallest index on ties), then across waves
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(mx, o);
    int oi = __shfl_xor(mi, o);
    if (ov > mx || (ov == mx && oi < mi)) {
      mx = ov;
      mi = oi;
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) {
    rv[wave] = mx;
    ri[wave] = mi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (rv[w] > mx || (rv[w] == mx && ri[w] < mi)) {
        mx = rv[w];
        mi = ri[w];
      }
    const float sd = sqrtf(var);
    u[bc * 3 + 0] = mx;
    u[bc * 3 + 1] = mean;
    u[bc * 3 + 2] = sd;
    amax[bc] = mi;
    float zz = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* w1 = fc1 + (c * 3 + j) * 3;
      float hdn = w1[0] * mx + w1[1] * mean + w1[2] * sd;
      if (hdn > 0.f) zz += fc2[c * 3 + j] * hdn;
    }
    z[bc] = zz;
  }
}

// The same statistics for SMALL planes (HWv <= 64 vector elements: 7x7 as 49 scalars, 14x14 as 49 quads), bf16 mode: one WAVE
// per plane, four planes per workgroup, every reduction a shuffle tree and no workgroup barrier — the 256-thread form left
// 49 lanes busy behind three barriers per plane (34 us for 32 x 512 planes of 7x7).
template <typename T, int V>
__global__ __launch_bounds__(256) void ccu_stats_fwd_wave_kernel(const T* __restrict__ x, const float* __restrict__ fc1,
                                                                const float* __restrict__ fc2, float* __restrict__ u,
                                                                int* __restrict__ amax, float* __restrict__ z, int C, int HWv,
                                                                int BC) {
  const int lane = threadIdx.x & 63, bc = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bc >= BC) return;  // (wave-uniform)
  const int c = bc % C, HW = HWv * V;
  const T* xp = x + (long)bc * HW;
  float v[V];
#pragma unroll
  for (int e = 0; e < V; ++e) v[e] = 0.f;
  if (lane < HWv) ldv<V>(v, xp + lane * V);
  float s = 0.f, mx = -3.4e38f;
  int mi = 0;
  if (lane < HWv) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      s += v[e];
      if (v[e] > mx) {
        mx = v[e];
        mi = lane * V + e;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / HW;
  float q = 0.f;
  if (lane < HWv) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float d = v[e] - mean;
      q += d * d;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    q += __shfl_xor(q, o);
    const float ov = __shfl_xor(mx, o);
    const int oi = __shfl_xor(mi, o);
    if (ov > mx || (ov == mx && oi < mi)) {
      mx = ov;
      mi = oi;
    }
  }
  if (lane == 0) {
    const float sd = sqrtf(q / HW);
    u[bc * 3 + 0] = mx;
    u[bc * 3 + 1] = mean;
    u[bc * 3 + 2] = sd;
    amax[bc] = mi;
    float zz = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* w1 = fc1 + (c * 3 + j) * 3;
      const float hdn = w1[0] * mx + w1[1] * mean + w1[2] * sd;
      if (hdn > 0.f) zz += fc2[c * 3 + j] * hdn;
    }
    z[bc] = zz;
  }
}

// y = x * sigmoid(g[bc])
template <typename T, int V>
__global__ __launch_bounds__(256) void gate_chan_fwd_kernel(const T* __restrict__ x, const float* __restrict__ g,
                                                           T* __restrict__ y, int HWv) {
  const int bc = blockIdx.x;
  const float s = sigmoid_f(g[bc]);
  const T* xp = x + (long)bc * HWv * V;
  T* yp = y + (long)bc * HWv * V;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < HWv; p += gridDim.y * 256) {
    float v[V];
    ldv<V>(v, xp + p * V);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = v[e] * s;
    stv<V>(yp + p * V, v);
  }
}

// dg[bc] = sigmoid'(g) * sum_p dy*x    (grid B*C)
template <typename T, int V>
__global__ __launch_bounds__(256) void gate_chan_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                  const float* __restrict__ g, float* __restrict__ dg,
                                                                  int HWv) {
  __shared__ float red[16];
  const int bc = blockIdx.x;
  const T* xp = x + (long)bc * HWv * V;
  const T* gp = dy + (long)bc * HWv * V;
  float s = 0.f;
  for (int p = threadIdx.x; p < HWv; p += 256) {
    float xv[V], gv[V];
    ldv<V>(xv, xp + p * V);
    ldv<V>(gv, gp + p * V);
#pragma unroll
    for (int e = 0; e < V; ++e) s += xv[e] * gv[e];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    float sg = sigmoid_f(g[bc]);
    dg[bc] = s * sg * (1.f - sg);
  }
}

// CCU backward apply (grid B*C): from dz[bc] (grad of the pre-BN MLP output) rebuild du through the MLP, accumulate
// fc1/fc2 gradients, and write dx = dy*sigmoid(g) + du_max*[p==amax] + du_mean/HW + du_std*(x-mean)/(HW*std)
template <typename T, int V>
__global__ __launch_bounds__(256) void ccu_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const float* __restrict__ g, const float* __restrict__ dz,
                                                           const float* __restrict__ u, const int* __restrict__ amax,
                                                           const float* __restrict__ fc1, const float* __restrict__ fc2,
                                                           float* __restrict__ dfc1, float* __restrict__ dfc2,
                                                           T* __restrict__ dx, int C, int HWv, const T* __restrict__ dx_add) {
  __shared__ float du_s[3];
  const int bc = blockIdx.x, c = bc % C, HW = HWv * V;
  if (threadIdx.x == 0) {
    const float mx = u[bc * 3], mean = u[bc * 3 + 1], sd = u[bc * 3 + 2];
    const float gz = dz[bc];
    float du[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* w1 = fc1 + (c * 3 + j) * 3;
      const float hdn = w1[0] * mx + w1[1] * mean + w1[2] * sd;
      const float w2 = fc2[c * 3 + j];
      if (hdn > 0.f) {
        atomicAdd(&dfc2[c * 3 + j], gz * hdn);
        const float gh = gz * w2;
        atomicAdd(&dfc1[(c * 3 + j) * 3 + 0], gh * mx);
        atomicAdd(&dfc1[(c * 3 + j) * 3 + 1], gh * mean);
        atomicAdd(&dfc1[(c * 3 + j) * 3 + 2], gh * sd);
        du[0] += gh * w1[0];
        du[1] += gh * w1[1];
        du[2] += gh * w1[2];
      }
    }
    du_s[0] = du[0];
    du_s[1] = du[1];
    du_s[2] = du[2];
  }
  __syncthreads();
  const float sg = sigmoid_f(g[bc]);
  const float mean = u[bc * 3 + 1], sd = u[bc * 3 + 2];
  const int am = amax[bc];
  // std backward at std == 0 (constant plane, e.g. a 1x1 map): aten::std_backward masks the 0/0 to 0
  const float dmax = du_s[0], dmean = du_s[1] / HW, dstd = (sd > 0.f) ? du_s[2] / (HW * sd) : 0.f;
  const T* xp = x + (long)bc * HW;
  const T* gp = dy + (long)bc * HW;
  T* dp = dx + (long)bc * HW;
  const T* ap = dx_add ? dx_add + (long)bc * HW : nullptr;  // gradient of x's other consumer (cfam.py:298-303: the shortcut)
  for (int p = threadIdx.x; p < HWv; p += 256) {
    float xv[V], gv[V], av[V];
    ldv<V>(xv, xp + p * V);
    ldv<V>(gv, gp + p * V);
#pragma unroll
    for (int e = 0; e < V; ++e) av[e] = 0.f;
    if (ap) ldv<V>(av, ap + p * V);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float v = gv[e] * sg + dmean + dstd * (xv[e] - mean);
      if (p * V + e == am) v += dmax;
      gv[e] = v + av[e];
    }
    stv<V>(dp + p * V, gv);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// SRM: channel statistics per pixel. x [B, C, HW]; u [B, 3, HW]; amax [B, HW]
// ---------------------------------------------------------------------------------------------------------------
// SRM channel statistics, single pass: workgroup = 64 pixels x 4 channel groups (lanes along pixels: every load is a
// contiguous 256-byte row segment), shifted sums (shift = channel 0) for mean / unbiased std, groups meet in LDS.
// NG channel groups = NG waves: 4, or 16 for small maps in bf16 mode (7x7: one workgroup per image walked 2048 channels with
// four groups: 144 us for 6 MB)
template <typename T, int NG>
__global__ __launch_bounds__(64 * NG) void srm_stats_fwd_kernel(const T* __restrict__ x, float* __restrict__ u,
                                                               int* __restrict__ amax, int C, int HW) {
  __shared__ float s1_s[NG][64], s2_s[NG][64], mx_s[NG][64];
  __shared__ int mi_s[NG][64];
  const int b = blockIdx.y;
  const int pl = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + pl;
  const bool ok = p < HW;
  const T* xb = x + (long)b * C * HW + (ok ? p : 0);
  const float shift = ok ? ldf(xb) : 0.f;
  float s1 = 0.f, s2 = 0.f, mx = -3.4e38f;
  int mi = 0;
  if (ok) {
    // eight channel rows in flight per thread: at 7x7 / 14x14 a workgroup walks 80 - 128 rows per thread and the launch has
    // only 32 - 128 workgroups, so the walk is a latency chain unless the loads are issued ahead of the (ordered) compares
    int c = cg;
    for (; c + 7 * NG < C; c += 8 * NG) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ldf(xb + (long)(c + j * NG) * HW);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[j] - shift;
        s1 += d;
        s2 += d * d;
        if (v[j] > mx) {
          mx = v[j];
          mi = c + j * NG;
        }
      }
    }
    for (; c < C; c += NG) {
      const float v = ldf(xb + (long)c * HW);
      const float d = v - shift;
      s1 += d;
      s2 += d * d;
      if (v > mx) {
        mx = v;
        mi = c;
      }
    }
  }
  s1_s[cg][pl] = s1;
  s2_s[cg][pl] = s2;
  mx_s[cg][pl] = mx;
  mi_s[cg][pl] = mi;
  __syncthreads();
  if (cg == 0 && ok) {
    float a1 = 0.f, a2 = 0.f, bm = mx_s[0][pl];
    int bi = mi_s[0][pl];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      a1 += s1_s[g][pl];
      a2 += s2_s[g][pl];
      const float m = mx_s[g][pl];
      const int i = mi_s[g][pl];
      if (m > bm || (m == bm && i < bi)) {  // first maximal channel, like torch.max
        bm = m;
        bi = i;
      }
    }
    const float mean_d = a1 / C;
    float var = (a2 - a1 * mean_d) / (C - 1);
    if (var < 0.f) var = 0.f;
    float* ub = u + (long)b * 3 * HW + p;
    ub[0] = bm;
    ub[HW] = shift + mean_d;
    ub[2 * HW] = sqrtf(var);
    amax[(long)b * HW + p] = bi;
  }
}

// f[b,p] = pwc(u) + dwc(u): 3->1 1x1 plus 3->1 3x3 (pad 1), no bias
__global__ __launch_bounds__(256) void srm_conv_fwd_kernel(const float* __restrict__ u, const float* __restrict__ pwc,
                                                          const float* __restrict__ dwc, float* __restrict__ f, int H, int W) {
  const int b = blockIdx.y, HW = H * W;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int py = p / W, px = p - py * W;
  const float* ub = u + (long)b * 3 * HW;
  float acc = 0.f;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    acc += pwc[ch] * ub[ch * HW + p];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = py + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = px + kx - 1;
        if (ix < 0 || ix >= W) continue;
        acc += dwc[ch * 9 + ky * 3 + kx] * ub[ch * HW + iy * W + ix];
      }
    }
  }
  f[(long)b * HW + p] = acc;
}

// du[b,ch,p] from df[b,p]; weight grads (3 + 27): per-thread partial sums over the workgroup's pixel chunks, ONE set of 30
// sums and 30 float atomics per workgroup.  (All 30 gradients live in one cache line and same-line atomics serialise at
// ~12 ns each: with a workgroup per 256 pixels the 56x56 level issued 12 480 of them = 150 us.  Round 4: the 30 sums meet in ONE
// pass through LDS instead of 30 block_sum calls = 60 barriers, and the grid is (min(8, pixels / 256), B): 38 -> 19 us per call.)
__global__ __launch_bounds__(256) void srm_conv_bwd_kernel(const float* __restrict__ u, const float* __restrict__ df,
                                                          const float* __restrict__ pwc, const float* __restrict__ dwc,
                                                          float* __restrict__ du, float* __restrict__ dpwc,
                                                          float* __restrict__ ddwc, int H, int W) {
  __shared__ float red[16];
  const int b = blockIdx.y, HW = H * W;
  const float* ub = u + (long)b * 3 * HW;
  const float* gb = df + (long)b * HW;
  float wsum[3][10];  // [channel][9 taps, pointwise]
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int t = 0; t < 10; ++t) wsum[ch][t] = 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const int py = p / W, px = p - py * W;
    const float g = gb[p];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float acc = pwc[ch] * g;  // data gradient (correlation with flipped taps)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int t = ky * 3 + kx;
          // du[p] += dwc[t] * df[p - off_t]  ;  ddwc[t] += df[p] * u[p + off_t]
          const int qy = py - (ky - 1), qx = px - (kx - 1);
          if (qy >= 0 && qy < H && qx >= 0 && qx < W) acc += dwc[ch * 9 + t] * gb[qy * W + qx];
          const int iy = py + ky - 1, ix = px + kx - 1;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) wsum[ch][t] += g * ub[ch * HW + iy * W + ix];
        }
      du[(long)b * 3 * HW + ch * HW + p] = acc;
      wsum[ch][9] += g * ub[ch * HW + p];
    }
  }
  // the 30 sums: wave shuffles, then ONE pass through LDS (30 block_sum calls were 60 barriers per workgroup)
  __shared__ float part[4][30];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      const float v = wave_sum(wsum[ch][t]);
      if (lane == 0) part[wave][ch * 10 + t] = v;
    }
  __syncthreads();
  if (threadIdx.x < 30) {
    const float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    const int ch = threadIdx.x / 10, t = threadIdx.x - ch * 10;
    if (t == 9) atomicAdd(&dpwc[ch], v);
    else atomicAdd(&ddwc[ch * 9 + t], v);
  }
}

// y[b,c,p] = x[b,c,p] * sigmoid(f[b,p]); grid (B*C planes, chunks)
template <typename T, int V>
__global__ __launch_bounds__(256) void gate_pix_fwd_kernel(const T* __restrict__ x, const float* __restrict__ f,
                                                          T* __restrict__ y, int C, int HWv) {
  const int bc = blockIdx.x, b = bc / C;
  const float* fb = f + (long)b * HWv * V;
  const long base = (long)bc * HWv * V;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < HWv; p += gridDim.y * 256) {
    float xv[V], fv[V];
    ldv<V>(xv, x + base + p * V);
    ldv<V>(fv, fb + p * V);
#pragma unroll
    for (int e = 0; e < V; ++e) xv[e] = xv[e] * sigmoid_f(fv[e]);
    stv<V>(y + base + p * V, xv);
  }
}

// 16-byte forms (HW % 4 == 0): workgroup = 64 pixels x 16 channel groups; a thread owns 4 consecutive pixels of every 16th
// channel (16 lanes cover a contiguous 256-byte row segment), 4 independent 16-byte loads in flight, groups meet in LDS
template <typename T>
__global__ __launch_bounds__(256) void srm_stats_fwd_v4_kernel(const T* __restrict__ x, float* __restrict__ u,
                                                              int* __restrict__ amax, int C, int HW) {
  __shared__ float s1_s[16][64], s2_s[16][64], mx_s[16][64];
  __shared__ int mi_s[16][64];
  const int b = blockIdx.y;
  const int pq = threadIdx.x & 15, cg = threadIdx.x >> 4;
  const int p = blockIdx.x * 64 + 4 * pq;
  const bool ok = p < HW;
  const T* xb = x + (long)b * C * HW + (ok ? p : 0);
  float shift[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok) ld4v(shift, xb);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, mx[4] = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
  int mi[4] = {0, 0, 0, 0};
  if (ok) {
#pragma unroll 4
    for (int c = cg; c < C; c += 16) {
      float v[4];
      ld4v(v, xb + (long)c * HW);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[e] - shift[e];
        s1[e] += d;
        s2[e] += d * d;
        if (v[e] > mx[e]) {
          mx[e] = v[e];
          mi[e] = c;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s1_s[cg][4 * pq + e] = s1[e];
    s2_s[cg][4 * pq + e] = s2[e];
    mx_s[cg][4 * pq + e] = mx[e];
    mi_s[cg][4 * pq + e] = mi[e];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int pl = threadIdx.x, pp = blockIdx.x * 64 + pl;
    if (pp < HW) {
      float a1 = 0.f, a2 = 0.f, bm = mx_s[0][pl];
      int bi = mi_s[0][pl];
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        a1 += s1_s[g][pl];
        a2 += s2_s[g][pl];
        const float m = mx_s[g][pl];
        const int i = mi_s[g][pl];
        if (m > bm || (m == bm && i < bi)) {  // first maximal channel, like torch.max
          bm = m;
          bi = i;
        }
      }
      const float sh = ldf(x + (long)b * C * HW + pp);
      const float mean_d = a1 / C;
      float var = (a2 - a1 * mean_d) / (C - 1);
      if (var < 0.f) var = 0.f;
      float* ub = u + (long)b * 3 * HW + pp;
      ub[0] = bm;
      ub[HW] = sh + mean_d;
      ub[2 * HW] = sqrtf(var);
      amax[(long)b * HW + pp] = bi;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gate_pix_bwd_reduce_v4_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                    const float* __restrict__ f, float* __restrict__ df, int C,
                                                                    int HW) {
  __shared__ float part[16][64];
  const int b = blockIdx.y;
  const int pq = threadIdx.x & 15, cg = threadIdx.x >> 4;
  const int p = blockIdx.x * 64 + 4 * pq;
  const bool ok = p < HW;
  const long base = (long)b * C * HW + (ok ? p : 0);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  // gridDim.z > 1 (bf16 tensors, small maps): the channel range is split over workgroups that add into a zero-filled df
  const int cper = (C + gridDim.z - 1) / gridDim.z, cbeg = blockIdx.z * cper, cend = cbeg + cper < C ? cbeg + cper : C;
  if (ok) {
#pragma unroll 4
    for (int c = cbeg + cg; c < cend; c += 16) {
      float xv[4], gv[4];
      ld4v(xv, x + base + (long)c * HW);
      ld4v(gv, dy + base + (long)c * HW);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += xv[e] * gv[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[cg][4 * pq + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int pl = threadIdx.x, pp = blockIdx.x * 64 + pl;
    if (pp < HW) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += part[g][pl];
      const float sg = sigmoid_f(f[(long)b * HW + pp]);
      if (gridDim.z > 1) atomicAdd(&df[(long)b * HW + pp], t * sg * (1.f - sg));
      else df[(long)b * HW + pp] = t * sg * (1.f - sg);
    }
  }
}

// df[b,p] = sigmoid'(f) * sum_c dy*x ; workgroup = 64 pixels x 4 channel groups
template <typename T>
__global__ __launch_bounds__(256) void gate_pix_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                 const float* __restrict__ f, float* __restrict__ df, int C,
                                                                 int HW) {
  __shared__ float part[4][64];
  const int b = blockIdx.y;
  const int pl = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + pl;
  const bool ok = p < HW;
  const long base = (long)b * C * HW + (ok ? p : 0);
  float s = 0.f;
  const int cper = (C + gridDim.z - 1) / gridDim.z, cbeg = blockIdx.z * cper, cend = cbeg + cper < C ? cbeg + cper : C;
  if (ok)
    for (int c = cbeg + cg; c < cend; c += 4) s += ldf(x + base + (long)c * HW) * ldf(dy + base + (long)c * HW);
  part[cg][pl] = s;
  __syncthreads();
  if (cg == 0 && ok) {
    const float t = part[0][pl] + part[1][pl] + part[2][pl] + part[3][pl];
    const float sg = sigmoid_f(f[(long)b * HW + p]);
    if (gridDim.z > 1) atomicAdd(&df[(long)b * HW + p], t * sg * (1.f - sg));
    else df[(long)b * HW + p] = t * sg * (1.f - sg);
  }
}

// dx = dy*sigmoid(f) + du_max*[c==amax] + du_mean/C + du_std*(x-mean)/((C-1)*std); grid (B*C planes, chunks)
template <typename T, int V>
__global__ __launch_bounds__(256) void srm_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const float* __restrict__ f, const float* __restrict__ u,
                                                           const float* __restrict__ du, const int* __restrict__ amax,
                                                           T* __restrict__ dx, int C, int HWv) {
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C, HW = HWv * V;
  const float* fb = f + (long)b * HW;
  const float* ub = u + (long)b * 3 * HW;
  const float* db = du + (long)b * 3 * HW;
  const int* ab = amax + (long)b * HW;
  const long base = (long)bc * HW;
  for (int pv = blockIdx.y * 256 + threadIdx.x; pv < HWv; pv += gridDim.y * 256) {
    float xv[V], gv[V];
    ldv<V>(xv, x + base + pv * V);
    ldv<V>(gv, dy + base + pv * V);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int p = pv * V + e;
      const float sg = sigmoid_f(fb[p]);
      const float mean = ub[HW + p], sd = ub[2 * HW + p];
      float v = gv[e] * sg + db[HW + p] / C + ((sd > 0.f) ? db[2 * HW + p] / ((C - 1) * sd) * (xv[e] - mean) : 0.f);
      if (c == ab[p]) v += db[p];
      gv[e] = v;
    }
    stv<V>(dx + base + pv * V, gv);
  }
}

// bf16 form: a thread owns V consecutive pixels and walks a slice of the channels, so the per-PIXEL terms (sigmoid of the gate,
// mean / std gradients, arg-max channel) are loaded and evaluated once instead of once per element — the plane-per-workgroup
// form above spends 7 scalar loads, an exponential and two divisions on every element (156 us for 154 MB at 56x56x256).
// grid (pixel chunks, B, channel slices)
template <typename T, int V>
__global__ __launch_bounds__(256) void srm_bwd_apply_pix_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                               const float* __restrict__ f, const float* __restrict__ u,
                                                               const float* __restrict__ du, const int* __restrict__ amax,
                                                               T* __restrict__ dx, int C, int HWv) {
  const int pv = blockIdx.x * 256 + threadIdx.x;
  if (pv >= HWv) return;
  const int b = blockIdx.y, HW = HWv * V;
  const int cper = (C + gridDim.z - 1) / gridDim.z, c0 = blockIdx.z * cper, c1 = c0 + cper < C ? c0 + cper : C;
  const float* ub = u + (long)b * 3 * HW;
  const float* db = du + (long)b * 3 * HW;
  float sg[V], a1[V], kk[V], mean[V], dmax[V];
  int am[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const int p = pv * V + e;
    sg[e] = sigmoid_f(f[(long)b * HW + p]);
    mean[e] = ub[HW + p];
    const float sd = ub[2 * HW + p];
    a1[e] = db[HW + p] / C;
    kk[e] = sd > 0.f ? db[2 * HW + p] / ((C - 1) * sd) : 0.f;
    dmax[e] = db[p];
    am[e] = amax[(long)b * HW + p];
  }
  const long base = (long)b * C * HW + (long)pv * V;
#pragma unroll 4
  for (int c = c0; c < c1; ++c) {
    float xv[V], gv[V];
    ldv<V>(xv, x + base + (long)c * HW);
    ldv<V>(gv, dy + base + (long)c * HW);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float v = gv[e] * sg[e] + a1[e] + kk[e] * (xv[e] - mean[e]);
      if (c == am[e]) v += dmax[e];
      gv[e] = v;
    }
    stv<V>(dx + base + (long)c * HW, gv);
  }
}

static inline int chunks_for(int n) {
  int ch = cdiv(n, 1024);
  return ch > 64 ? 64 : (ch < 1 ? 1 : ch);
}

// plane vector width: 4 when the planes split into aligned quads, else 1
template <typename T>
static inline int plane_vw(int HW, const void* a, const void* b = nullptr, const void* c = nullptr) {
  const uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
  return ((HW & 3) == 0 && (m & (4 * sizeof(T) - 1)) == 0) ? 4 : 1;
}

template <typename T>
static int ccu_stats_fwd_impl(const T* x, const float* fc1, const float* fc2, float* u, int* amax, float* z, int B, int C, int HW,
                              hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  // fp32 keeps the one-element-per-thread walk: the summation order of the parity mode is pinned
  const int vw = sizeof(T) == 2 ? plane_vw<T>(HW, x) : 1;
  if (sizeof(T) == 2 && HW / vw <= 64) {  // small planes, bf16 mode: one wave per plane
    if (vw == 4) CENET_LAUNCH((ccu_stats_fwd_wave_kernel<T, 4>), dim3(cdiv(B * C, 4)), dim3(256), stream, x, fc1, fc2, u, amax, z, C, HW / 4, B * C);
    else CENET_LAUNCH((ccu_stats_fwd_wave_kernel<T, 1>), dim3(cdiv(B * C, 4)), dim3(256), stream, x, fc1, fc2, u, amax, z, C, HW, B * C);
  } else if (vw == 4) CENET_LAUNCH((ccu_stats_fwd_kernel<T, 4>), dim3(B * C), dim3(256), stream, x, fc1, fc2, u, amax, z, C, HW / 4);
  else CENET_LAUNCH((ccu_stats_fwd_kernel<T, 1>), dim3(B * C), dim3(256), stream, x, fc1, fc2, u, amax, z, C, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(ccu_stats_fwd, (const T* x, const float* fc1, const float* fc2, float* u, int* amax, float* z, int B, int C, int HW,
                           hipStream_t stream), (x, fc1, fc2, u, amax, z, B, C, HW, stream))

template <typename T>
static int gate_chan_fwd_impl(const T* x, const float* g, T* y, int BC, int HW, hipStream_t stream) {
  if (BC <= 0 || HW <= 0) return CENET_EINVAL;
  if (plane_vw<T>(HW, x, y) == 4) CENET_LAUNCH((gate_chan_fwd_kernel<T, 4>), dim3(BC, chunks_for(HW / 4)), dim3(256), stream, x, g, y, HW / 4);
  else CENET_LAUNCH((gate_chan_fwd_kernel<T, 1>), dim3(BC, chunks_for(HW)), dim3(256), stream, x, g, y, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(gate_chan_fwd, (const T* x, const float* g, T* y, int BC, int HW, hipStream_t stream), (x, g, y, BC, HW, stream))

template <typename T>
static int gate_chan_bwd_reduce_impl(const T* x, const T* dy, const float* g, float* dg, int BC, int HW, hipStream_t stream) {
  if (BC <= 0 || HW <= 0) return CENET_EINVAL;
  if (sizeof(T) == 2 && plane_vw<T>(HW, x, dy) == 4) CENET_LAUNCH((gate_chan_bwd_reduce_kernel<T, 4>), dim3(BC), dim3(256), stream, x, dy, g, dg, HW / 4);
  else CENET_LAUNCH((gate_chan_bwd_reduce_kernel<T, 1>), dim3(BC), dim3(256), stream, x, dy, g, dg, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(gate_chan_bwd_reduce, (const T* x, const T* dy, const float* g, float* dg, int BC, int HW, hipStream_t stream),
           (x, dy, g, dg, BC, HW, stream))

template <typename T>
static int ccu_bwd_apply_acc_impl(const T* x, const T* dy, const float* g, const float* dz, const float* u, const int* amax,
                                  const float* fc1, const float* fc2, float* dfc1_acc, float* dfc2_acc, T* dx, int B, int C,
                                  int HW, hipStream_t stream, const T* dx_add = nullptr) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  if (plane_vw<T>(HW, x, dy, dx) == 4 && (!dx_add || plane_vw<T>(HW, dx_add, dy, dx) == 4))
    CENET_LAUNCH((ccu_bwd_apply_kernel<T, 4>), dim3(B * C), dim3(256), stream, x, dy, g, dz, u, amax, fc1, fc2, dfc1_acc, dfc2_acc,
                 dx, C, HW / 4, dx_add);
  else
    CENET_LAUNCH((ccu_bwd_apply_kernel<T, 1>), dim3(B * C), dim3(256), stream, x, dy, g, dz, u, amax, fc1, fc2, dfc1_acc, dfc2_acc,
                 dx, C, HW, dx_add);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(ccu_bwd_apply_acc, (const T* x, const T* dy, const float* g, const float* dz, const float* u, const int* amax,
                               const float* fc1, const float* fc2, float* dfc1_acc, float* dfc2_acc, T* dx, int B, int C, int HW,
                               hipStream_t stream),
           (x, dy, g, dz, u, amax, fc1, fc2, dfc1_acc, dfc2_acc, dx, B, C, HW, stream))
// ... + dx_add (contiguous, like x): the gradient of x's other consumer (the MCA shortcut, cfam.py:298-303) added by this kernel
template <typename T>
static int ccu_bwd_apply_add_acc_impl(const T* x, const T* dy, const float* g, const float* dz, const float* u, const int* amax,
                                      const float* fc1, const float* fc2, float* dfc1_acc, float* dfc2_acc, T* dx, const T* dx_add,
                                      int B, int C, int HW, hipStream_t stream) {
  return ccu_bwd_apply_acc_impl<T>(x, dy, g, dz, u, amax, fc1, fc2, dfc1_acc, dfc2_acc, dx, B, C, HW, stream, dx_add);
}
CENET_TWIN(ccu_bwd_apply_add_acc, (const T* x, const T* dy, const float* g, const float* dz, const float* u, const int* amax,
                                   const float* fc1, const float* fc2, float* dfc1_acc, float* dfc2_acc, T* dx, const T* dx_add,
                                   int B, int C, int HW, hipStream_t stream),
           (x, dy, g, dz, u, amax, fc1, fc2, dfc1_acc, dfc2_acc, dx, dx_add, B, C, HW, stream))

template <typename T>
static int srm_stats_fwd_impl(const T* x, float* u, int* amax, int B, int C, int HW, hipStream_t stream) {
  if (B <= 0 || C <= 1 || HW <= 0) return CENET_EINVAL;
  if ((HW & 3) == 0 && quad_aligned<T>(x))
    CENET_LAUNCH((srm_stats_fwd_v4_kernel<T>), dim3(cdiv(HW, 64), B), dim3(256), stream, x, u, amax, C, HW);
  else if (sizeof(T) == 2 && (long)cdiv(HW, 64) * B < 256 && C >= 256)
    CENET_LAUNCH((srm_stats_fwd_kernel<T, 16>), dim3(cdiv(HW, 64), B), dim3(1024), stream, x, u, amax, C, HW);
  else
    CENET_LAUNCH((srm_stats_fwd_kernel<T, 4>), dim3(cdiv(HW, 64), B), dim3(256), stream, x, u, amax, C, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(srm_stats_fwd, (const T* x, float* u, int* amax, int B, int C, int HW, hipStream_t stream),
           (x, u, amax, B, C, HW, stream))

extern "C" int cenet_srm_conv_fwd_f32(const float* u, const float* pwc, const float* dwc, float* f, int B, int H, int W,
                                      hipStream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  CENET_LAUNCH(srm_conv_fwd_kernel, dim3(cdiv(H * W, 256), B), dim3(256), stream, u, pwc, dwc, f, H, W);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_srm_conv_bwd_acc_f32(const float* u, const float* df, const float* pwc, const float* dwc, float* du,
                                          float* dpwc_acc, float* ddwc_acc, int B, int H, int W, hipStream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  static const int gx = getenv("CENET_SRM_BWD_GX") ? atoi(getenv("CENET_SRM_BWD_GX")) : 0;  // measurement aid
  // measured at the four decoder levels of the ACDC preset, average us per call: gx = 1: 52, 2: 32, 4: 23, 8: 19, 12: 21, 16: 21, 32: 29
  int want = cdiv(H * W, 256);
  if (want > 8) want = 8;
  CENET_LAUNCH(srm_conv_bwd_kernel, dim3(gx > 0 ? gx : want, B), dim3(256), stream, u, df, pwc, dwc, du, dpwc_acc, ddwc_acc, H, W);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

template <typename T>
static int gate_pix_fwd_impl(const T* x, const float* f, T* y, int B, int C, int HW, hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  if (plane_vw<T>(HW, x, y) == 4 && (((uintptr_t)f) & 15) == 0)
    CENET_LAUNCH((gate_pix_fwd_kernel<T, 4>), dim3(B * C, chunks_for(HW / 4)), dim3(256), stream, x, f, y, C, HW / 4);
  else
    CENET_LAUNCH((gate_pix_fwd_kernel<T, 1>), dim3(B * C, chunks_for(HW)), dim3(256), stream, x, f, y, C, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(gate_pix_fwd, (const T* x, const float* f, T* y, int B, int C, int HW, hipStream_t stream), (x, f, y, B, C, HW, stream))

__global__ void stats_zero_kernel(float* p, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

template <typename T>
static int gate_pix_bwd_reduce_impl(const T* x, const T* dy, const float* f, float* df, int B, int C, int HW,
                                    hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  // small maps leave the chip idle (7x7: 32 workgroups walking 2048 channels each, 206 us): split the channel range over
  // grid.z and add into a zero-filled df.  bf16 tensors only: the fp32 (parity) mode keeps its fixed summation order.
  int cs = 1;
  if (sizeof(T) == 2) {
    const long wgs = (long)cdiv(HW, 64) * B;
    while (cs < 32 && wgs * cs < 512 && C / (cs * 2) >= 64) cs *= 2;
  }
  if (cs > 1) CENET_LAUNCH(stats_zero_kernel, dim3(cdiv(B * HW, 256)), dim3(256), stream, df, (long)B * HW);
  if ((HW & 3) == 0 && quad_aligned<T>(x) && quad_aligned<T>(dy))
    CENET_LAUNCH((gate_pix_bwd_reduce_v4_kernel<T>), dim3(cdiv(HW, 64), B, cs), dim3(256), stream, x, dy, f, df, C, HW);
  else
    CENET_LAUNCH((gate_pix_bwd_reduce_kernel<T>), dim3(cdiv(HW, 64), B, cs), dim3(256), stream, x, dy, f, df, C, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(gate_pix_bwd_reduce, (const T* x, const T* dy, const float* f, float* df, int B, int C, int HW, hipStream_t stream),
           (x, dy, f, df, B, C, HW, stream))

template <typename T>
static int srm_bwd_apply_impl(const T* x, const T* dy, const float* f, const float* u, const float* du, const int* amax, T* dx,
                              int B, int C, int HW, hipStream_t stream) {
  if (B <= 0 || C <= 1 || HW <= 0) return CENET_EINVAL;
  if (sizeof(T) == 2) {  // (fp32 keeps the plane form: same expression order as the parity goldens were checked with)
    const int V = plane_vw<T>(HW, x, dy, dx);
    const int HWv = HW / V, px = cdiv(HWv, 256);
    int cs = 1;
    while (cs < 64 && (long)px * B * cs < 1024 && C / (cs * 2) >= 8) cs *= 2;
    const dim3 grid(px, B, cs);
    if (V == 4) CENET_LAUNCH((srm_bwd_apply_pix_kernel<T, 4>), grid, dim3(256), stream, x, dy, f, u, du, amax, dx, C, HWv);
    else CENET_LAUNCH((srm_bwd_apply_pix_kernel<T, 1>), grid, dim3(256), stream, x, dy, f, u, du, amax, dx, C, HWv);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (plane_vw<T>(HW, x, dy, dx) == 4)
    CENET_LAUNCH((srm_bwd_apply_kernel<T, 4>), dim3(B * C, chunks_for(HW / 4)), dim3(256), stream, x, dy, f, u, du, amax, dx, C,
                 HW / 4);
  else
    CENET_LAUNCH((srm_bwd_apply_kernel<T, 1>), dim3(B * C, chunks_for(HW)), dim3(256), stream, x, dy, f, u, du, amax, dx, C, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(srm_bwd_apply, (const T* x, const T* dy, const float* f, const float* u, const float* du, const int* amax, T* dx,
                           int B, int C, int HW, hipStream_t stream), (x, dy, f, u, du, amax, dx, B, C, HW, stream))


// =====================================================================================================================
// SRM tail without its BatchNorm / activation launches (round 5).  cfam.py:93-101 after the channel statistics:
//   f = pwc(u) + dwc(u);  fa = GELU(f);  fb = BatchNorm_train(fa) (ONE channel over B*H*W);  y = x * sigmoid(fb)
// ran as conv, act, bn_stats, bn_apply, gate (5 launches on a [B, 1, H, W] map at every decoder level) and 6 backwards.
//   srm_conv_gelu_fwd   conv + GELU, and per workgroup (count, mean, M2) of its fa values        -> f, fa, part [G][3]
//   gate_pix_bn_fwd     folds the G partial triples (Chan's merge: exact in any order), normalises and gates in one pass;
//                       image-0-channel-0's workgroups publish mean / var / running statistics, the first channel of every
//                       image writes fb (the backward kernels read it)
//   srm_bn_bwd_part     per workgroup sum(dfb), sum(dfb * xhat)                                    -> part2 [G][2]
//   srm_conv_bn_bwd     folds part2, rebuilds df = BatchNormBackward(dfb) * GELU'(f) for its image in LDS, then the conv backward
// =====================================================================================================================
__global__ __launch_bounds__(256) void srm_conv_gelu_fwd_kernel(const float* __restrict__ u, const float* __restrict__ pwc,
                                                               const float* __restrict__ dwc, float* __restrict__ f,
                                                               float* __restrict__ fa, float* __restrict__ part, int H, int W) {
  __shared__ float red[16];
  const int b = blockIdx.y, HW = H * W;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool ok = p < HW;
  float a = 0.f;
  if (ok) {
    const int py = p / W, px = p - py * W;
    const float* ub = u + (long)b * 3 * HW;
    float acc = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      acc += pwc[ch] * ub[ch * HW + p];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = py + ky - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = px + kx - 1;
          if (ix < 0 || ix >= W) continue;
          acc += dwc[ch * 9 + ky * 3 + kx] * ub[ch * HW + iy * W + ix];
        }
      }
    }
    a = gelu_f(acc);
    f[(long)b * HW + p] = acc;
    fa[(long)b * HW + p] = a;
  }
  const int n = HW - blockIdx.x * 256 < 256 ? HW - blockIdx.x * 256 : 256;
  const float m = block_sum(ok ? a : 0.f, red) / n;
  const float m2 = block_sum(ok ? (a - m) * (a - m) : 0.f, red);
  if (threadIdx.x == 0) {
    float* q = part + 3 * ((long)b * gridDim.x + blockIdx.x);
    q[0] = (float)n;
    q[1] = m;
    q[2] = m2;
  }
}

// total (mean, biased variance) of G (count, mean, M2) triples.  Every WAVE folds for itself (same order in every wave, so every
// wave of the launch holds the same bits): no LDS, no barrier in front of the kernel's real work.
__device__ __forceinline__ void srm_fold(const float* __restrict__ part, int G, float& mean, float& var, float& cnt) {
  const int lane = threadIdx.x & 63;
  float n = 0.f, nm = 0.f;
  for (int i = lane; i < G; i += 64) {
    n += part[3 * i];
    nm += part[3 * i] * part[3 * i + 1];
  }
  n = wave_sum(n);
  nm = wave_sum(nm);
  mean = nm / n;
  float m2 = 0.f;
  for (int i = lane; i < G; i += 64) {
    const float d = part[3 * i + 1] - mean;
    m2 += part[3 * i + 2] + part[3 * i] * d * d;
  }
  var = wave_sum(m2) / n;
  cnt = n;
}

// y = x * sigmoid(a fa + c), a / c from the folded statistics.  grid (ceil(B*HWv / 256), channel slices): a thread owns V
// consecutive pixels of one image — the normalisation and the sigmoid are evaluated ONCE per pixel, not once per channel — and
// walks its channel slice with 4 independent loads in flight.  Slice 0 also stores fb (the backward kernels read it).
template <typename T, int V>
__global__ __launch_bounds__(256) void gate_pix_bn_fwd_kernel(const T* __restrict__ x, const float* __restrict__ fa,
                                                             const float* __restrict__ part, int G, float* __restrict__ fb,
                                                             T* __restrict__ y, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, float* mean_out,
                                                             float* var_out, float* rmean, float* rvar, float momentum, long* nbt,
                                                             int B, int C, int HWv) {
  float mean, var, cnt;
  srm_fold(part, G, mean, var, cnt);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    mean_out[0] = mean;
    var_out[0] = var;
    if (rmean) {
      rmean[0] = (1.f - momentum) * rmean[0] + momentum * mean;
      rvar[0] = (1.f - momentum) * rvar[0] + momentum * var * (cnt / (cnt - 1.f));
    }
    if (nbt) nbt[0] += 1;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * HWv) return;
  const float a = gamma[0] * rsqrtf(var + eps), c = beta[0] - mean * a;
  const int b = i / HWv, pv = i - b * HWv;
  float sg[V];
  ldv<V>(sg, fa + (long)i * V);
#pragma unroll
  for (int e = 0; e < V; ++e) sg[e] = a * sg[e] + c;
  if (blockIdx.y == 0) stv<V>(fb + (long)i * V, sg);
#pragma unroll
  for (int e = 0; e < V; ++e) sg[e] = sigmoid_f(sg[e]);
  const int per = (C + (int)gridDim.y - 1) / (int)gridDim.y, c0 = blockIdx.y * per, c1 = c0 + per < C ? c0 + per : C;
  const long HW = (long)HWv * V;
  const T* xp = x + ((long)b * C + c0) * HW + (long)pv * V;
  T* yp = y + ((long)b * C + c0) * HW + (long)pv * V;
  int ch = c0;
  for (; ch + 4 <= c1; ch += 4, xp += 4 * HW, yp += 4 * HW) {
    float xv[4][V];
#pragma unroll
    for (int k = 0; k < 4; ++k) ldv<V>(xv[k], xp + k * HW);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int e = 0; e < V; ++e) xv[k][e] *= sg[e];
      stv<V>(yp + k * HW, xv[k]);
    }
  }
  for (; ch < c1; ++ch, xp += HW, yp += HW) {
    float xv[V];
    ldv<V>(xv, xp);
#pragma unroll
    for (int e = 0; e < V; ++e) xv[e] *= sg[e];
    stv<V>(yp, xv);
  }
}

__global__ __launch_bounds__(256) void srm_bn_bwd_part_kernel(const float* __restrict__ dfb, const float* __restrict__ fa,
                                                             const float* __restrict__ mean, const float* __restrict__ var,
                                                             float eps, float* __restrict__ part2, int HW) {
  __shared__ float red[16];
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  float g = 0.f, gx = 0.f;
  if (p < HW) {
    g = dfb[(long)b * HW + p];
    gx = g * (fa[(long)b * HW + p] - mean[0]) * rsqrtf(var[0] + eps);
  }
  g = block_sum(g, red);
  gx = block_sum(gx, red);
  if (threadIdx.x == 0) {
    float* q = part2 + 2 * ((long)b * gridDim.x + blockIdx.x);
    q[0] = g;
    q[1] = gx;
  }
}

#define SRM_MAXHW 4096
// grid (gx, B): as srm_conv_bwd_kernel, with df built in LDS from dfb / fa / f and the folded BatchNorm sums
__global__ __launch_bounds__(256) void srm_conv_bn_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dfb,
                                                             const float* __restrict__ fa, const float* __restrict__ f,
                                                             const float* __restrict__ part2, int G2, float ntot,
                                                             const float* __restrict__ mean, const float* __restrict__ var,
                                                             float eps, const float* __restrict__ gamma,
                                                             const float* __restrict__ pwc, const float* __restrict__ dwc,
                                                             float* __restrict__ du, float* __restrict__ dpwc,
                                                             float* __restrict__ ddwc, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int H, int W) {
  __shared__ float red[16];
  __shared__ float gl[SRM_MAXHW];
  const int b = blockIdx.y, HW = H * W;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < G2; i += 256) {
    s1 += part2[2 * i];
    s2 += part2[2 * i + 1];
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
    dgamma[0] += s2;
    dbeta[0] += s1;
  }
  const float mu = mean[0], rs = rsqrtf(var[0] + eps), k0 = gamma[0] * rs, m1 = s1 / ntot, m2 = s2 / ntot;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const long q = (long)b * HW + p;
    const float xh = (fa[q] - mu) * rs;
    gl[p] = k0 * (dfb[q] - m1 - xh * m2) * gelu_grad_f(f[q]);
  }
  __syncthreads();
  const float* ub = u + (long)b * 3 * HW;
  float wsum[3][10];  // [channel][9 taps, pointwise]
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int t = 0; t < 10; ++t) wsum[ch][t] = 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const int py = p / W, px = p - py * W;
    const float g = gl[p];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float acc = pwc[ch] * g;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int t = ky * 3 + kx;
          const int qy = py - (ky - 1), qx = px - (kx - 1);
          if (qy >= 0 && qy < H && qx >= 0 && qx < W) acc += dwc[ch * 9 + t] * gl[qy * W + qx];
          const int iy = py + ky - 1, ix = px + kx - 1;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) wsum[ch][t] += g * ub[ch * HW + iy * W + ix];
        }
      du[(long)b * 3 * HW + ch * HW + p] = acc;
      wsum[ch][9] += g * ub[ch * HW + p];
    }
  }
  __shared__ float part[4][30];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      const float v = wave_sum(wsum[ch][t]);
      if (lane == 0) part[wave][ch * 10 + t] = v;
    }
  __syncthreads();
  if (threadIdx.x < 30) {
    const float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    const int ch = threadIdx.x / 10, t = threadIdx.x - ch * 10;
    if (t == 9) atomicAdd(&dpwc[ch], v);
    else atomicAdd(&ddwc[ch * 9 + t], v);
  }
}

// (B H W > 1: the running variance is updated with the unbiased factor n / (n - 1); a single value has no variance estimate and
// PyTorch itself refuses to train a BatchNorm on one value per channel)
extern "C" int cenet_srm_fused_supported(int B, int H, int W) {
  return B > 0 && H > 0 && W > 0 && (long)H * W <= SRM_MAXHW && (long)B * H * W > 1;
}
/* partial triples written by cenet_srm_conv_gelu_fwd_f32: B * ceil(H*W / 256) */
extern "C" int cenet_srm_parts(int B, int H, int W) { return B * cdiv(H * W, 256); }
extern "C" int cenet_srm_conv_gelu_fwd_f32(const float* u, const float* pwc, const float* dwc, float* f, float* fa, float* part,
                                           int B, int H, int W, hipStream_t stream) {
  if (!u || !pwc || !dwc || !f || !fa || !part || B <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  CENET_LAUNCH(srm_conv_gelu_fwd_kernel, dim3(cdiv(H * W, 256), B), dim3(256), stream, u, pwc, dwc, f, fa, part, H, W);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
template <typename T>
static int gate_pix_bn_fwd_impl(const T* x, const float* fa, const float* part, int G, float* fb, T* y, const float* gamma,
                                const float* beta, float eps, float* mean, float* var, float* rmean, float* rvar, float momentum,
                                long* nbt, int B, int C, int HW, hipStream_t stream) {
  if (!x || !fa || !part || !fb || !y || !gamma || !beta || !mean || !var || G <= 0 || B <= 0 || C <= 0 || HW <= 0) return CENET_EINVAL;
  const int V = (plane_vw<T>(HW, x, y) == 4 && (((uintptr_t)fa | (uintptr_t)fb) & 15) == 0) ? 4 : 1;
  const int px = cdiv(B * (HW / V), 256);
  int cs = 1;
  while (cs < 64 && (long)px * cs < 1024 && C / (cs * 2) >= 8) cs *= 2;
  if (V == 4)
    CENET_LAUNCH((gate_pix_bn_fwd_kernel<T, 4>), dim3(px, cs), dim3(256), stream, x, fa, part, G, fb, y, gamma, beta, eps, mean, var,
                 rmean, rvar, momentum, nbt, B, C, HW / 4);
  else
    CENET_LAUNCH((gate_pix_bn_fwd_kernel<T, 1>), dim3(px, cs), dim3(256), stream, x, fa, part, G, fb, y, gamma, beta, eps, mean, var,
                 rmean, rvar, momentum, nbt, B, C, HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(gate_pix_bn_fwd, (const T* x, const float* fa, const float* part, int G, float* fb, T* y, const float* gamma,
                             const float* beta, float eps, float* mean, float* var, float* running_mean, float* running_var,
                             float momentum, long* num_batches_tracked, int B, int C, int HW, hipStream_t stream),
           (x, fa, part, G, fb, y, gamma, beta, eps, mean, var, running_mean, running_var, momentum, num_batches_tracked, B, C, HW,
            stream))
extern "C" int cenet_srm_conv_bn_bwd_acc_f32(const float* u, const float* dfb, const float* fa, const float* f, const float* mean,
                                             const float* var, float eps, const float* gamma, const float* pwc, const float* dwc,
                                             float* part2_ws, float* du, float* dpwc_acc, float* ddwc_acc, float* dgamma_acc,
                                             float* dbeta_acc, int B, int H, int W, hipStream_t stream) {
  if (!u || !dfb || !fa || !f || !mean || !var || !gamma || !pwc || !dwc || !part2_ws || !du || !dpwc_acc || !ddwc_acc ||
      !dgamma_acc || !dbeta_acc || B <= 0 || H <= 0 || W <= 0)
    return CENET_EINVAL;
  if ((long)H * W > SRM_MAXHW) return CENET_EUNSUPPORTED;
  const int gx = cdiv(H * W, 256);
  CENET_LAUNCH(srm_bn_bwd_part_kernel, dim3(gx, B), dim3(256), stream, dfb, fa, mean, var, eps, part2_ws, H * W);
  int want = gx > 8 ? 8 : gx;
  CENET_LAUNCH(srm_conv_bn_bwd_kernel, dim3(want, B), dim3(256), stream, u, dfb, fa, f, (const float*)part2_ws, gx * B,
               (float)((long)B * H * W), mean, var, eps, gamma, pwc, dwc, du, dpwc_acc, ddwc_acc, dgamma_acc, dbeta_acc, H, W);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
