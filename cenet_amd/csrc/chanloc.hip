// chanloc.hip — CHANNEL-LOCAL fused chains of the decoder (round 5).
//
// BatchNorm couples the images of a batch, depthwise convs / nearest up-sampling / per-channel gates do not couple channels: a
// chain made of such steps only ever mixes the values of ONE channel.  One workgroup therefore owns one channel over the whole
// batch and runs the chain start to end — batch statistics are workgroup reductions (no second launch, no partial-sum round
// trip, no float atomics: the results do not depend on scheduling), intermediates live in LDS or are recomputed, and only the
// tensors the next channel-MIXING step (a 1x1 conv) needs are written.  At the 7x7 ... 28x28 decoder levels of a 224x224
// input a channel over a batch of 32 is 1.5 ... 25 K values: the launches this replaces were bound by their fill / drain
// latency, not by bytes (profiles/r04_*: ~120 dependent launches per level on <= 4 MB tensors).
//
//   eucb_{fwd,bwd}   blocks.py:297-321 EUCB up to its 1x1 conv: nearest x2 -> DW3x3 -> BatchNorm(train) -> LeakyReLU.
//                    The up-sampled tensor and the conv output are never stored (forward: 4 launches -> 1; backward: the
//                    BatchNorm backward, LeakyReLU mask, depthwise data + weight gradient and the x2 fold: 5 launches -> 1).
//
// Templates over the activation storage type T (float: parity mode, bf16_t: throughput mode); arithmetic in fp32.
#include "common.h"
#include <cstdlib>
#include "../../include/cenet_hip.h"

// elements of one channel over the batch (B * HW) up to which a workgroup owns a whole channel (28 x 28 planes at a batch of 32 and
// some headroom; beyond that the launch chains with partial sums over many workgroups win)
#define CENET_CHANLOC_MAX 32768

namespace {

// sum over the workgroup of N values at once (one barrier pair for all of them); result in every thread.  red: >= 16 * N floats
template <int N>
__device__ __forceinline__ void block_sum_n(float (&v)[N], float* red) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) red[w * N + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i * N + k];
    v[k] = t;
  }
}

// Strided walks without a division per element: thread t visits e = t, t + NT, ... of a [B][HW] (or [B][H][W]) index space and
// keeps (b, p) / (b, i, j) up to date with adds and carries (an integer division by a run-time value is ~40 instructions: more
// than the arithmetic of any element here).
struct Walk2 {
  int b, p, dB, dP, HW;
  __device__ __forceinline__ Walk2(int e0, int stride, int HW_) : HW(HW_) {
    b = e0 / HW_;
    p = e0 - b * HW_;
    dB = stride / HW_;
    dP = stride - dB * HW_;
  }
  __device__ __forceinline__ void next() {
    p += dP;
    b += dB;
    if (p >= HW) {
      p -= HW;
      ++b;
    }
  }
};
struct Walk3 {
  int b, i, j, dB, dI, dJ, H, W;
  __device__ __forceinline__ Walk3(int e0, int stride, int H_, int W_) : H(H_), W(W_) {
    const int HW = H_ * W_;
    b = e0 / HW;
    int p = e0 - b * HW;
    i = p / W_;
    j = p - i * W_;
    dB = stride / HW;
    p = stride - dB * HW;
    dI = p / W_;
    dJ = p - dI * W_;
  }
  __device__ __forceinline__ void next() {
    j += dJ;
    i += dI;
    b += dB;
    if (j >= W) {
      j -= W;
      ++i;
    }
    if (i >= H) {
      i -= H;
      ++b;
    }
  }
};

struct EucbArgs {
  const void* x;   // [B, C, H, W] (batch stride sxb)
  long sxb;
  void* y;         // forward: [B, C, 2H, 2W] (batch stride syb)       backward: dx [B, C, H, W] (batch stride syb)
  long syb;
  const void* g;   // backward: gradient of the LeakyReLU output [B, C, 2H, 2W] (batch stride sgb)
  long sgb;
  const float* w;  // [C][9]
  const float *gamma, *beta;
  float eps, slope;
  float *mean, *var;             // forward: written; backward: read
  float *rmean, *rvar;           // forward: running statistics (may be null)
  float momentum;
  long* nbt;
  float *dw, *dgamma, *dbeta;    // backward: ADDED into (this workgroup is the only writer of its channel's entries)
  int B, H, W;
};

// The 2x2 output quad of source pixel (i, j) reads the 3x3 source neighbourhood S (zero outside the plane: the conv pads the
// UP-SAMPLED grid, whose border pixels are copies of the source border, so "outside" coincides):
//   up rows 2i-1, 2i, 2i+1 (output row 2i)   -> source rows i-1, i, i      up rows 2i, 2i+1, 2i+2 (row 2i+1) -> i, i, i+1
// so output parity py touches only S rows {py, py + 1} (S row 0 = i - 1), with the taps that fall on one source row summed:
//   u[py][px] = sum_{a2, c2 in {0, 1}} q[py][px][a2][c2] * S[py + a2][px + c2],
//   q = sum of w[ky][kx] over KY(py, a2) x KX(px, c2),   KY(0, 0) = {0}, KY(0, 1) = {1, 2}, KY(1, 0) = {0, 1}, KY(1, 1) = {2}
// 16 FMAs per quad instead of 36.  The planes sit in LDS with a one-pixel ZERO border ([H + 2][W + 2]): no bounds tests.
__device__ __forceinline__ int eucb_a2(int par, int k) { return par == 0 ? (k == 0 ? 0 : 1) : (k == 2 ? 1 : 0); }

__device__ __forceinline__ void eucb_combine(const float (&w)[9], float (&q)[2][2][2][2]) {
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px) {
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) q[py][px][a2][c2] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) q[py][px][eucb_a2(py, ky)][eucb_a2(px, kx)] += w[ky * 3 + kx];
    }
}

// channel planes of the batch -> LDS [B][H + 2][W + 2] with a zero border
template <typename T>
__device__ __forceinline__ void eucb_stage(const T* x, long sxb, T* xs, int B, int H, int W, int NT) {
  const int PW = W + 2, tot = B * (H + 2) * PW;
  Walk3 k(threadIdx.x, NT, H + 2, PW);
  for (int e = threadIdx.x; e < tot; e += NT, k.next()) {
    const int yy = k.i - 1, xx = k.j - 1;
    T v;
    memset(&v, 0, sizeof(T));
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = x[(long)k.b * sxb + yy * W + xx];
    xs[e] = v;
  }
}
template <typename T>
__device__ __forceinline__ void eucb_load_nb(const T* xs, int b, int i, int j, int H, int W, float (&S)[3][3]) {
  const T* p = xs + ((long)b * (H + 2) + i) * (W + 2) + j;  // padded row i = source row i - 1
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) S[a][c] = ldf(p + a * (W + 2) + c);
}
// the four conv outputs of the quad: u[py][px]
__device__ __forceinline__ void eucb_quad(const float (&S)[3][3], const float (&q)[2][2][2][2], float (&u)[2][2]) {
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px)
      u[py][px] = q[py][px][0][0] * S[py][px] + q[py][px][0][1] * S[py][px + 1] + q[py][px][1][0] * S[py + 1][px] +
                  q[py][px][1][1] * S[py + 1][px + 1];
}

// grid = C, NT threads, SM bytes of LDS holding the channel's padded source planes as T
template <typename T, int NT, int SM>
__global__ __launch_bounds__(NT) void eucb_fwd_kernel(EucbArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  __shared__ float red[16 * 2];
  T* xs = (T*)smem;
  const int c = blockIdx.x, H = a.H, W = a.W, HW = H * W, items = a.B * HW;
  eucb_stage((const T*)a.x + (long)c * HW, a.sxb, xs, a.B, H, W, NT);
  float q[2][2][2][2];
  {
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = a.w[c * 9 + k];
    eucb_combine(w, q);
  }
  __syncthreads();
  // pass 1: batch statistics of the conv output (shifted by the value at the plane centre of image 0: fp32 sums of squares)
  float S[3][3], u[2][2];
  eucb_load_nb(xs, 0, H / 2, W / 2, H, W, S);
  eucb_quad(S, q, u);
  const float K = u[0][0];
  float s[2] = {0.f, 0.f};
  Walk3 wk(threadIdx.x, NT, H, W);
  for (int e = threadIdx.x; e < items; e += NT, wk.next()) {
    eucb_load_nb(xs, wk.b, wk.i, wk.j, H, W, S);
    eucb_quad(S, q, u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = u[k >> 1][k & 1] - K;
      s[0] += d;
      s[1] += d * d;
    }
  }
  block_sum_n<2>(s, red);
  const float n = 4.f * (float)items;
  const float m = s[0] / n;
  float var = s[1] / n - m * m;
  if (var < 0.f) var = 0.f;
  const float mu = K + m;
  if (threadIdx.x == 0) {
    a.mean[c] = mu;
    a.var[c] = var;
    if (a.rmean) {
      a.rmean[c] = (1.f - a.momentum) * a.rmean[c] + a.momentum * mu;
      a.rvar[c] = (1.f - a.momentum) * a.rvar[c] + a.momentum * var * (n / (n - 1.f));
    }
    if (a.nbt && c == 0) a.nbt[0] += 1;
  }
  const float sc = a.gamma[c] * rsqrtf(var + a.eps), sh = a.beta[c] - mu * sc;
  // pass 2: normalise + LeakyReLU + store (two pixels = one 4-byte / 8-byte store per output row)
  const int OW = 2 * W;
  T* y = (T*)a.y + (long)c * 4 * HW;
  wk = Walk3(threadIdx.x, NT, H, W);
  for (int e = threadIdx.x; e < items; e += NT, wk.next()) {
    eucb_load_nb(xs, wk.b, wk.i, wk.j, H, W, S);
    eucb_quad(S, q, u);
    T* yp = y + (long)wk.b * a.syb + (long)(2 * wk.i) * OW + 2 * wk.j;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float v0 = u[py][0] * sc + sh, v1 = u[py][1] * sc + sh;
      v0 = v0 > 0.f ? v0 : v0 * a.slope;
      v1 = v1 > 0.f ? v1 : v1 * a.slope;
      if (sizeof(T) == 2) {
        const unsigned pk = cenet_pack_bf2(v0, v1);
        memcpy(yp + py * OW, &pk, 4);
      } else {
        const float pr[2] = {v0, v1};
        memcpy(yp + py * OW, pr, 8);
      }
    }
  }
}

// backward.  LDS: the padded source planes as T, then G zero-bordered planes [2H + 2][2W + 2] of fp32 conv-output gradients (one
// group of G images at a time: the depthwise data gradient needs the neighbours of an output pixel's gradient)
template <typename T, int NT, int SM>
__global__ __launch_bounds__(NT) void eucb_bwd_kernel(EucbArgs a, int G) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  __shared__ float red[16 * 11];
  const int c = blockIdx.x, H = a.H, W = a.W, HW = H * W, items = a.B * HW, OW = 2 * W, OH = 2 * H;
  const int DW_ = OW + 2, DP = (OH + 2) * DW_;
  T* xs = (T*)smem;
  float* du = (float*)(smem + (((long)a.B * (H + 2) * (W + 2) * sizeof(T) + 15) & ~15L));
  eucb_stage((const T*)a.x + (long)c * HW, a.sxb, xs, a.B, H, W, NT);
  for (int e = threadIdx.x; e < G * DP; e += NT) du[e] = 0.f;  // (the borders stay zero: only interiors are rewritten)
  float w[9], q[2][2][2][2];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = a.w[c * 9 + k];
  eucb_combine(w, q);
  const float mu = a.mean[c], rs = rsqrtf(a.var[c] + a.eps), gm = a.gamma[c], bt = a.beta[c];
  __syncthreads();
  const T* g = (const T*)a.g + (long)c * 4 * HW;
  float S[3][3], u[2][2];
  // gradient of the quad's four conv outputs through LeakyReLU, and their normalised values
  auto quad_g = [&](int b, int i, int j, float (&gy)[2][2], float (&xh)[2][2]) __attribute__((always_inline)) {
    eucb_load_nb(xs, b, i, j, H, W, S);
    eucb_quad(S, q, u);
    const T* gp = g + (long)b * a.sgb + (long)(2 * i) * OW + 2 * j;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float g0, g1;
      if (sizeof(T) == 2) {
        unsigned pk;
        memcpy(&pk, gp + py * OW, 4);
        g0 = cenet_bf2f(pk & 0xFFFFu), g1 = cenet_bf2f(pk >> 16);
      } else {
        float pr[2];
        memcpy(pr, gp + py * OW, 8);
        g0 = pr[0], g1 = pr[1];
      }
      xh[py][0] = (u[py][0] - mu) * rs;
      xh[py][1] = (u[py][1] - mu) * rs;
      gy[py][0] = (xh[py][0] * gm + bt > 0.f) ? g0 : g0 * a.slope;
      gy[py][1] = (xh[py][1] * gm + bt > 0.f) ? g1 : g1 * a.slope;
    }
  };
  // pass 1: sum g, sum g * xhat
  float s[2] = {0.f, 0.f};
  float gy[2][2], xh[2][2];
  Walk3 wk(threadIdx.x, NT, H, W);
  for (int e = threadIdx.x; e < items; e += NT, wk.next()) {
    quad_g(wk.b, wk.i, wk.j, gy, xh);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s[0] += gy[k >> 1][k & 1];
      s[1] += gy[k >> 1][k & 1] * xh[k >> 1][k & 1];
    }
  }
  block_sum_n<2>(s, red);
  const float n = 4.f * (float)items, m1 = s[0] / n, m2 = s[1] / n, k0 = gm * rs;
  // combined weights of the data gradient: dx(i, j) = sum_{r, t = -1..2} cw[r+1][t+1] du(2i + r, 2j + t), where
  // cw = sum of w[ky][kx] over KY(r) x KX(t), KY(-1) = {2}, KY(0) = {1, 2}, KY(1) = {0, 1}, KY(2) = {0}
  float cw[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const bool iny = (r == 0 && ky == 2) || (r == 1 && ky >= 1) || (r == 2 && ky <= 1) || (r == 3 && ky == 0);
          const bool inx = (t == 0 && kx == 2) || (t == 1 && kx >= 1) || (t == 2 && kx <= 1) || (t == 3 && kx == 0);
          if (iny && inx) v += w[ky * 3 + kx];
        }
      cw[r][t] = v;
    }
  // weight-gradient sums in the basis of the combined weights (16 FMAs per quad), expanded to the 9 taps at the end
  float A[2][2][2][2];
#pragma unroll
  for (int k = 0; k < 16; ++k) A[k >> 3][(k >> 2) & 1][(k >> 1) & 1][k & 1] = 0.f;
  T* dx = (T*)a.y + (long)c * HW;
  for (int b0 = 0; b0 < a.B; b0 += G) {
    const int nb = a.B - b0 < G ? a.B - b0 : G;
    // (a) conv-output gradients of images b0 .. b0 + nb into LDS
    wk = Walk3(threadIdx.x, NT, H, W);
    for (int e = threadIdx.x; e < nb * HW; e += NT, wk.next()) {
      const int bl = wk.b, i = wk.i, j = wk.j;
      quad_g(b0 + bl, i, j, gy, xh);
      float* dp = du + (long)bl * DP + (2 * i + 1) * DW_ + 2 * j + 1;
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const float d = k0 * (gy[py][px] - m1 - xh[py][px] * m2);
          dp[py * DW_ + px] = d;
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) A[py][px][a2][c2] += d * S[py + a2][px + c2];
        }
    }
    __syncthreads();
    // (b) data gradient of the group's source pixels: the 4x4 window of padded rows 2i .. 2i + 3
    wk = Walk3(threadIdx.x, NT, H, W);
    for (int e = threadIdx.x; e < nb * HW; e += NT, wk.next()) {
      const int bl = wk.b, i = wk.i, j = wk.j, p = i * W + j;
      const float* dp = du + (long)bl * DP + (2 * i) * DW_ + 2 * j;
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) t += cw[r][k] * dp[r * DW_ + k];
      stf(dx + (long)(b0 + bl) * a.syb + p, t);
    }
    __syncthreads();
  }
  float fin[11];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float v = 0.f;
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) v += A[py][px][eucb_a2(py, ky)][eucb_a2(px, kx)];
      fin[ky * 3 + kx] = v;
    }
  fin[9] = 0.f, fin[10] = 0.f;
  block_sum_n<11>(fin, red);
  if (threadIdx.x < 9) a.dw[c * 9 + threadIdx.x] += fin[threadIdx.x];
  if (threadIdx.x == 9) a.dgamma[c] += s[1];
  if (threadIdx.x == 10) a.dbeta[c] += s[0];
}

// ---- CFAM "mid" chain (cfam.py:365-374 around nlb.py:140-148) --------------------------------------------------------------------
//   p = BatchNorm_nl(p_raw)            (the Non-local block's output conv, nlb.py:141-142)
//   z = (1 - w) m + w p                (nlb.py:148; m = the block's input, w a scalar parameter)
//   x1 = x0 + ls1 * z                  (cfam.py:370: layer-scale residual; x0 = the CFAM block's input)
//   y2 = BatchNorm_2(x1)               (cfam.py:372 norm2, input of the Mlp; x1 also feeds the second residual)
// Channel-local: workgroup = channel over the batch, three passes each way (the tensors are L2-resident at the levels this is
// used for).  Forward 5 - 6 launches -> 1, backward 6 - 7 -> 1.  Element e of the channel = (image e / HW, pixel e % HW);
// V = elements per access (4 when HW % 4 == 0).
struct MidArgs {
  const void *p_raw, *m, *x0;  // [B, C, HW] contiguous
  void *x1, *y2;               // forward outputs; backward: x1 is read
  const void *g_y2, *g_x1;     // backward inputs (g_x1 may be null)
  void *d_p, *d_m, *d_x0;      // backward outputs
  const float *gp, *bp, *g2, *b2, *w, *ls;
  float epsp, eps2;
  float *meanp, *varp, *mean2, *var2;
  float *rmp, *rvp, *rm2, *rv2;
  float momp, mom2;
  long *nbtp, *nbt2;
  float *dgp, *dbp, *dg2, *db2, *dw, *dls;  // ADDED into (dw: one float atomic per channel)
  int B, C, HW;
};

template <typename T, int V>
__device__ __forceinline__ long mid_off(const Walk2& k, int c, int C, int HWv) {  // element group (b, p) of channel c -> elements
  return ((long)k.b * C + c) * HWv * V + (long)k.p * V;
}
__device__ __forceinline__ void bn_publish(float* mean, float* var, float* rm, float* rv, float mom, long* nbt, int c, float mu,
                                           float v, float n) {
  mean[c] = mu;
  var[c] = v;
  if (rm) {
    rm[c] = (1.f - mom) * rm[c] + mom * mu;
    rv[c] = (1.f - mom) * rv[c] + mom * v * (n / (n - 1.f));
  }
  if (nbt && c == 0) nbt[0] += 1;
}
template <typename T>
__device__ __forceinline__ float round_to(float v) {
  return sizeof(T) == 2 ? cenet_bf2f(cenet_f2bf(v)) : v;
}

template <typename T, int V, int NT>
__global__ __launch_bounds__(NT) void cfam_mid_fwd_kernel(MidArgs a) {
  __shared__ float red[16 * 2];
  const int c = blockIdx.x, C = a.C, HWv = a.HW / V, items = a.B * HWv;
  const T *pr = (const T*)a.p_raw, *mm = (const T*)a.m, *x0 = (const T*)a.x0;
  T *x1 = (T*)a.x1, *y2 = (T*)a.y2;
  const float n = (float)a.B * (float)a.HW, wv = a.w[0], ls = a.ls[c];
  // pass 1: statistics of p_raw (shifted by the channel's first value)
  const float K = ldf(pr + (long)c * a.HW);
  float s[2] = {0.f, 0.f};
  for (Walk2 wk(threadIdx.x, NT, HWv); wk.b < a.B; wk.next()) {
    float v[V];
    ldv<V>(v, pr + mid_off<T, V>(wk, c, C, HWv));
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float d = v[k] - K;
      s[0] += d;
      s[1] += d * d;
    }
  }
  block_sum_n<2>(s, red);
  float mloc = s[0] / n;
  float varp = s[1] / n - mloc * mloc;
  if (varp < 0.f) varp = 0.f;
  const float mup = K + mloc;
  if (threadIdx.x == 0) bn_publish(a.meanp, a.varp, a.rmp, a.rvp, a.momp, a.nbtp, c, mup, varp, n);
  const float ap = a.gp[c] * rsqrtf(varp + a.epsp), cp = a.bp[c] - mup * ap;
  // pass 2: x1 (stored) and its statistics (of the STORED values: the backward pass normalises what it reads)
  const long o0 = (long)c * a.HW;
  const float K2 = round_to<T>(ldf(x0 + o0) + ls * ((1.f - wv) * ldf(mm + o0) + wv * (ap * ldf(pr + o0) + cp)));
  s[0] = s[1] = 0.f;
  for (Walk2 wk(threadIdx.x, NT, HWv); wk.b < a.B; wk.next()) {
    const long o = mid_off<T, V>(wk, c, C, HWv);
    float vp[V], vm[V], vx[V];
    ldv<V>(vp, pr + o);
    ldv<V>(vm, mm + o);
    ldv<V>(vx, x0 + o);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      vx[k] = round_to<T>(vx[k] + ls * ((1.f - wv) * vm[k] + wv * (ap * vp[k] + cp)));
      const float d = vx[k] - K2;
      s[0] += d;
      s[1] += d * d;
    }
    stv<V>(x1 + o, vx);
  }
  block_sum_n<2>(s, red);  // (its barriers also order the x1 stores before the reads below: same workgroup, same L1)
  mloc = s[0] / n;
  float var2 = s[1] / n - mloc * mloc;
  if (var2 < 0.f) var2 = 0.f;
  const float mu2 = K2 + mloc;
  if (threadIdx.x == 0) bn_publish(a.mean2, a.var2, a.rm2, a.rv2, a.mom2, a.nbt2, c, mu2, var2, n);
  const float a2 = a.g2[c] * rsqrtf(var2 + a.eps2), c2 = a.b2[c] - mu2 * a2;
  // pass 3: y2
  for (Walk2 wk(threadIdx.x, NT, HWv); wk.b < a.B; wk.next()) {
    const long o = mid_off<T, V>(wk, c, C, HWv);
    float v[V];
    ldv<V>(v, x1 + o);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = a2 * v[k] + c2;
    stv<V>(y2 + o, v);
  }
}

template <typename T, int V, int NT>
__global__ __launch_bounds__(NT) void cfam_mid_bwd_kernel(MidArgs a) {
  __shared__ float red[16 * 4];
  const int c = blockIdx.x, C = a.C, HWv = a.HW / V, items = a.B * HWv;
  const T *pr = (const T*)a.p_raw, *mm = (const T*)a.m, *x1 = (const T*)a.x1, *gy = (const T*)a.g_y2, *gx = (const T*)a.g_x1;
  T *dp = (T*)a.d_p, *dm = (T*)a.d_m, *dx0 = (T*)a.d_x0;
  const float n = (float)a.B * (float)a.HW, wv = a.w[0], ls = a.ls[c];
  const float mu2 = a.mean2[c], rs2 = rsqrtf(a.var2[c] + a.eps2), k2 = a.g2[c] * rs2;
  const float mup = a.meanp[c], rsp = rsqrtf(a.varp[c] + a.epsp), ap = a.gp[c] * rsp, cp = a.bp[c] - mup * ap;
  // pass 1: BatchNorm_2 sums
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (Walk2 wk(threadIdx.x, NT, HWv); wk.b < a.B; wk.next()) {
    const long o = mid_off<T, V>(wk, c, C, HWv);
    float g[V], v[V];
    ldv<V>(g, gy + o);
    ldv<V>(v, x1 + o);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      s[0] += g[k];
      s[1] += g[k] * ((v[k] - mu2) * rs2);
    }
  }
  block_sum_n<4>(s, red);
  const float m1 = s[0] / n, m2 = s[1] / n;
  const float dg2 = s[1], db2 = s[0];
  // pass 2: d x1 (= d x0, stored) and the sums of everything upstream of it
  s[0] = s[1] = s[2] = s[3] = 0.f;  // dls, dw, sum d p, sum d p * xhat_p
  for (Walk2 wk(threadIdx.x, NT, HWv); wk.b < a.B; wk.next()) {
    const long o = mid_off<T, V>(wk, c, C, HWv);
    float g[V], v[V], t[V], vp[V], vm[V];
    ldv<V>(g, gy + o);
    ldv<V>(v, x1 + o);
    if (gx) ldv<V>(t, gx + o);
    ldv<V>(vp, pr + o);
    ldv<V>(vm, mm + o);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float d = k2 * (g[k] - m1 - (v[k] - mu2) * rs2 * m2);
      if (gx) d += t[k];
      d = round_to<T>(d);  // (the value pass 3 reads back)
      g[k] = d;
      const float pn = ap * vp[k] + cp, z = (1.f - wv) * vm[k] + wv * pn, dz = d * ls, dpn = dz * wv;
      s[0] += d * z;
      s[1] += dz * (pn - vm[k]);
      s[2] += dpn;
      s[3] += dpn * ((vp[k] - mup) * rsp);
    }
    stv<V>(dx0 + o, g);
  }
  block_sum_n<4>(s, red);
  const float t1 = s[2] / n, t2 = s[3] / n;
  if (threadIdx.x == 0) {
    a.dg2[c] += dg2;
    a.db2[c] += db2;
    a.dls[c] += s[0];
    a.dgp[c] += s[3];
    a.dbp[c] += s[2];
    atomicAdd(a.dw, s[1]);
  }
  // pass 3: d p_raw, d m
  for (Walk2 wk(threadIdx.x, NT, HWv); wk.b < a.B; wk.next()) {
    const long o = mid_off<T, V>(wk, c, C, HWv);
    float d[V], vp[V], om[V];
    ldv<V>(d, dx0 + o);
    ldv<V>(vp, pr + o);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float dz = d[k] * ls, dpn = dz * wv;
      om[k] = dz * (1.f - wv);
      vp[k] = ap * (dpn - t1 - (vp[k] - mup) * rsp * t2);
    }
    stv<V>(dm + o, om);
    stv<V>(dp + o, vp);
  }
}

template <typename T>
static int cfam_mid_launch(const MidArgs& a, bool bwd, hipStream_t stream) {
  const long per = (long)a.B * a.HW;
  if (per < 2 || per > CENET_CHANLOC_MAX) return CENET_EUNSUPPORTED;
  const bool v4 = (a.HW & 3) == 0;
#define CENET_MID(Vv, NTv)                                                                                   \
  {                                                                                                          \
    if (bwd) CENET_LAUNCH((cfam_mid_bwd_kernel<T, Vv, NTv>), dim3(a.C), dim3(NTv), stream, a);               \
    else CENET_LAUNCH((cfam_mid_fwd_kernel<T, Vv, NTv>), dim3(a.C), dim3(NTv), stream, a);                   \
  }
  if (v4) {
    if (per / 4 <= 1024) CENET_MID(4, 256)
    else CENET_MID(4, 1024)
  } else {
    if (per <= 2048) CENET_MID(1, 256)
    else CENET_MID(1, 1024)
  }
#undef CENET_MID
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

template <typename T>
static int cfam_mid_fwd_impl(const T* p_raw, const T* m, const T* x0, T* x1, T* y2, const float* gp, const float* bp, float epsp,
                             float* meanp, float* varp, float* rmp, float* rvp, float momp, long* nbtp, const float* w,
                             const float* ls, const float* g2, const float* b2, float eps2, float* mean2, float* var2, float* rm2,
                             float* rv2, float mom2, long* nbt2, int B, int C, int HW, hipStream_t stream) {
  if (!p_raw || !m || !x0 || !x1 || !y2 || !gp || !bp || !meanp || !varp || !w || !ls || !g2 || !b2 || !mean2 || !var2 ||
      B <= 0 || C <= 0 || HW <= 0)
    return CENET_EINVAL;
  if ((((uintptr_t)p_raw | (uintptr_t)m | (uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)y2) & 15) != 0) return CENET_EUNSUPPORTED;
  MidArgs a;
  memset(&a, 0, sizeof(a));
  a.p_raw = p_raw; a.m = m; a.x0 = x0; a.x1 = x1; a.y2 = y2; a.gp = gp; a.bp = bp; a.epsp = epsp; a.meanp = meanp; a.varp = varp;
  a.rmp = rmp; a.rvp = rvp; a.momp = momp; a.nbtp = nbtp; a.w = w; a.ls = ls; a.g2 = g2; a.b2 = b2; a.eps2 = eps2;
  a.mean2 = mean2; a.var2 = var2; a.rm2 = rm2; a.rv2 = rv2; a.mom2 = mom2; a.nbt2 = nbt2; a.B = B; a.C = C; a.HW = HW;
  return cfam_mid_launch<T>(a, false, stream);
}
template <typename T>
static int cfam_mid_bwd_acc_impl(const T* g_y2, const T* g_x1, const T* p_raw, const T* m, const T* x1, T* d_p, T* d_m, T* d_x0,
                                 const float* gp, const float* bp, float epsp, const float* meanp, const float* varp,
                                 const float* w, const float* ls, const float* g2, float eps2, const float* mean2,
                                 const float* var2, float* dgp_acc, float* dbp_acc, float* dw_acc, float* dls_acc, float* dg2_acc,
                                 float* db2_acc, int B, int C, int HW, hipStream_t stream) {
  if (!g_y2 || !p_raw || !m || !x1 || !d_p || !d_m || !d_x0 || !gp || !bp || !meanp || !varp || !w || !ls || !g2 || !mean2 ||
      !var2 || !dgp_acc || !dbp_acc || !dw_acc || !dls_acc || !dg2_acc || !db2_acc || B <= 0 || C <= 0 || HW <= 0)
    return CENET_EINVAL;
  if ((((uintptr_t)g_y2 | (uintptr_t)g_x1 | (uintptr_t)p_raw | (uintptr_t)m | (uintptr_t)x1 | (uintptr_t)d_p | (uintptr_t)d_m |
        (uintptr_t)d_x0) & 15) != 0)
    return CENET_EUNSUPPORTED;
  MidArgs a;
  memset(&a, 0, sizeof(a));
  a.g_y2 = g_y2; a.g_x1 = g_x1; a.p_raw = p_raw; a.m = m; a.x1 = (void*)x1; a.d_p = d_p; a.d_m = d_m; a.d_x0 = d_x0; a.gp = gp;
  a.bp = bp; a.epsp = epsp; a.meanp = (float*)meanp; a.varp = (float*)varp; a.w = w; a.ls = ls; a.g2 = g2; a.eps2 = eps2;
  a.mean2 = (float*)mean2; a.var2 = (float*)var2; a.dgp = dgp_acc; a.dbp = dbp_acc; a.dw = dw_acc; a.dls = dls_acc;
  a.dg2 = dg2_acc; a.db2 = db2_acc; a.B = B; a.C = C; a.HW = HW;
  return cfam_mid_launch<T>(a, true, stream);
}

// ---- dilated depthwise branches of MultiOrderDWConv with their BatchNorm (cfam.py:227-241 over blocks.py:169-177) --------------
//   v[:, j*g + i] = ReLU(BatchNorm_train(DW3x3_{dil_j}(x[:, j*g + i])))   for the NB (<= 3) branches of g channels each,
//   rest = x[:, NB*g :]                                                   (the pooled branch's slice, copied)
// Forward: the conv output goes to v (rounded to the tensor type, as the launch chain stores it), its statistics are taken from
// those stored values, and after a barrier the workgroup normalises its own channel in place.  Backward recomputes the conv
// output from x (no saved copy), writes the conv-output gradient to an fp32 scratch plane set it alone reads back, and adds
// the OTHER gradients of x (g_add: the gate conv's data gradient; g_rest: the pooled branch's) while writing dx: the launch chain
// ran a depthwise kernel, a BatchNorm (2 launches), and backwards BatchNorm (2), depthwise data and weight gradients and a
// slice copy.  The channel's planes are staged in LDS when they fit (flat pointer either way).
struct DwBnArgs {
  const void* x;     // [B, Ctot, H, W] (batch stride sxb); the branch channels start at channel 0
  long sxb;
  void* y;           // forward: v [B, NB*g, H, W] (batch stride syb);  backward: dx [B, Ctot, H, W] (batch stride syb)
  long syb;
  void* rest;        // forward: [B, p, H, W] written;  backward: its gradient, read (batch stride srb)
  long srb;
  const void* g;     // backward: gradient of v (batch stride sgb)
  long sgb;
  const void* gadd;  // backward: other gradient of x [B, Ctot, H, W] (batch stride sab) or null
  long sab;
  float* du;         // backward scratch: [NB*g][B][HW] fp32
  const float* w[3];
  float* dw[3];
  int dil[3];
  int G, NB, P;      // channels per branch, branches, rest channels
  const float *gamma, *beta;
  float eps;
  float *mean, *var, *rmean, *rvar;
  float momentum;
  long* nbt;         // NB counters (one per branch) or null
  float *dgamma, *dbeta;
  int B, H, W;
};

template <typename T>
__device__ __forceinline__ float dw_tap9(const T* pl, int H, int W, int i, int j, int dil, const float (&w)[9], float (&tap)[9]) {
  float t = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = i + (ky - 1) * dil;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xx = j + (kx - 1) * dil;
      const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? ldf(pl + yy * W + xx) : 0.f;
      tap[ky * 3 + kx] = v;
      t += w[ky * 3 + kx] * v;
    }
  }
  return t;
}

// stage the channel's planes [B][HW] into LDS if they fit; returns the base and the image stride to read them with
template <typename T>
__device__ __forceinline__ const T* chan_stage(const T* x, long sxb, int B, int HW, unsigned char* smem, int sm_bytes, int NT,
                                               long& stride) {
  if ((long)B * HW * (long)sizeof(T) > sm_bytes) {
    stride = sxb;
    return x;
  }
  T* xs = (T*)smem;
  int e = threadIdx.x;
  for (Walk2 k(threadIdx.x, NT, HW); k.b < B; k.next(), e += NT) xs[e] = x[(long)k.b * sxb + k.p];
  __syncthreads();
  stride = HW;
  return xs;
}

template <typename T, int NT, int SM>
__global__ __launch_bounds__(NT) void dwbn_fwd_kernel(DwBnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  __shared__ float red[16 * 2];
  const int c = blockIdx.x, H = a.H, W = a.W, HW = H * W, items = a.B * HW;
  if (c >= a.NB * a.G) {  // the pooled branch's slice: a copy
    const int cr = c - a.NB * a.G;
    const T* x = (const T*)a.x + (long)c * HW;
    T* r = (T*)a.rest + (long)cr * HW;
    for (Walk2 k(threadIdx.x, NT, HW); k.b < a.B; k.next()) r[(long)k.b * a.srb + k.p] = x[(long)k.b * a.sxb + k.p];
    return;
  }
  const int j = c / a.G, cl = c - j * a.G, dil = a.dil[j];
  float w[9], tap[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = a.w[j][cl * 9 + k];
  long xst;
  const T* xs = chan_stage((const T*)a.x + (long)c * HW, a.sxb, a.B, HW, smem, SM, NT, xst);
  T* y = (T*)a.y + (long)c * HW;
  const float K = round_to<T>(dw_tap9(xs, H, W, H / 2, W / 2, dil, w, tap));
  float s[2] = {0.f, 0.f};
  for (Walk3 k(threadIdx.x, NT, H, W); k.b < a.B; k.next()) {
    const float u = round_to<T>(dw_tap9(xs + k.b * xst, H, W, k.i, k.j, dil, w, tap));
    stf(y + (long)k.b * a.syb + k.i * W + k.j, u);
    const float d = u - K;
    s[0] += d;
    s[1] += d * d;
  }
  block_sum_n<2>(s, red);  // (also orders the stores above before the in-place pass below: same workgroup)
  const float n = (float)items, m = s[0] / n;
  float var = s[1] / n - m * m;
  if (var < 0.f) var = 0.f;
  const float mu = K + m;
  if (threadIdx.x == 0) {
    a.mean[c] = mu;
    a.var[c] = var;
    if (a.rmean) {
      a.rmean[c] = (1.f - a.momentum) * a.rmean[c] + a.momentum * mu;
      a.rvar[c] = (1.f - a.momentum) * a.rvar[c] + a.momentum * var * (n / (n - 1.f));
    }
    if (a.nbt && cl == 0) a.nbt[j] += 1;
  }
  const float sc = a.gamma[c] * rsqrtf(var + a.eps), sh = a.beta[c] - mu * sc;
  for (Walk2 k(threadIdx.x, NT, HW); k.b < a.B; k.next()) {
    T* q = y + (long)k.b * a.syb + k.p;
    const float v = ldf(q) * sc + sh;
    stf(q, v > 0.f ? v : 0.f);
  }
}

template <typename T, int NT, int SM>
__global__ __launch_bounds__(NT) void dwbn_bwd_kernel(DwBnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  __shared__ float red[16 * 9];
  const int c = blockIdx.x, H = a.H, W = a.W, HW = H * W, items = a.B * HW;
  T* dx = (T*)a.y + (long)c * HW;
  const T* ga = a.gadd ? (const T*)a.gadd + (long)c * HW : nullptr;
  if (c >= a.NB * a.G) {  // pooled-branch channels: dx = g_rest + g_add
    const T* r = (const T*)a.rest + (long)(c - a.NB * a.G) * HW;
    for (Walk2 k(threadIdx.x, NT, HW); k.b < a.B; k.next()) {
      float v = ldf(r + (long)k.b * a.srb + k.p);
      if (ga) v += ldf(ga + (long)k.b * a.sab + k.p);
      stf(dx + (long)k.b * a.syb + k.p, v);
    }
    return;
  }
  const int j = c / a.G, cl = c - j * a.G, dil = a.dil[j];
  float w[9], tap[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = a.w[j][cl * 9 + k];
  long xst;
  const T* xs = chan_stage((const T*)a.x + (long)c * HW, a.sxb, a.B, HW, smem, SM, NT, xst);
  const T* g = (const T*)a.g + (long)c * HW;
  const float mu = a.mean[c], rs = rsqrtf(a.var[c] + a.eps), gm = a.gamma[c], bt = a.beta[c];
  // pass 1: sum g, sum g * xhat (through the ReLU mask); the conv output is recomputed and rounded as the forward stored it
  float s[2] = {0.f, 0.f};
  for (Walk3 k(threadIdx.x, NT, H, W); k.b < a.B; k.next()) {
    const float xh = (round_to<T>(dw_tap9(xs + k.b * xst, H, W, k.i, k.j, dil, w, tap)) - mu) * rs;
    const float gy = (xh * gm + bt > 0.f) ? ldf(g + (long)k.b * a.sgb + k.i * W + k.j) : 0.f;
    s[0] += gy;
    s[1] += gy * xh;
  }
  block_sum_n<2>(s, red);
  const float n = (float)items, m1 = s[0] / n, m2 = s[1] / n, k0 = gm * rs;
  // pass 2: conv-output gradient -> scratch; weight gradient
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  float* du = a.du + (long)c * items;
  for (Walk3 k(threadIdx.x, NT, H, W); k.b < a.B; k.next()) {
    const int p = k.i * W + k.j;
    const float xh = (round_to<T>(dw_tap9(xs + k.b * xst, H, W, k.i, k.j, dil, w, tap)) - mu) * rs;
    const float gy = (xh * gm + bt > 0.f) ? ldf(g + (long)k.b * a.sgb + p) : 0.f;
    const float d = k0 * (gy - m1 - xh * m2);
    du[(long)k.b * HW + p] = d;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] += d * tap[k];
  }
  block_sum_n<9>(acc, red);  // (its barriers order the scratch stores before the reads below)
  if (threadIdx.x < 9) a.dw[j][cl * 9 + threadIdx.x] += acc[threadIdx.x];
  if (threadIdx.x == 9) a.dgamma[c] += s[1];
  if (threadIdx.x == 10) a.dbeta[c] += s[0];
  // pass 3: data gradient (correlation with the flipped taps) + the other gradient of x
  for (Walk3 k(threadIdx.x, NT, H, W); k.b < a.B; k.next()) {
    const int b = k.b, i = k.i, jj = k.j, p = i * W + jj;
    const float* dp = du + (long)b * HW;
    float t = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = i - (ky - 1) * dil;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = jj - (kx - 1) * dil;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) t += w[ky * 3 + kx] * dp[yy * W + xx];
      }
    }
    if (ga) t += ldf(ga + (long)b * a.sab + p);
    stf(dx + (long)b * a.syb + p, t);
  }
}

constexpr int DWBN_SM = 64 * 1024;

template <typename T>
static int dwbn_launch(const DwBnArgs& a, bool bwd, hipStream_t stream) {
  const long per = (long)a.B * a.H * a.W;
  if (per < 2 || per > CENET_CHANLOC_MAX || a.NB < 1 || a.NB > 3 || a.G < 1 || a.P < 0) return CENET_EUNSUPPORTED;
  const int grid = a.NB * a.G + a.P;
  if (per <= 4096) {
    if (bwd) CENET_LAUNCH((dwbn_bwd_kernel<T, 256, DWBN_SM / 4>), dim3(grid), dim3(256), stream, a);
    else CENET_LAUNCH((dwbn_fwd_kernel<T, 256, DWBN_SM / 4>), dim3(grid), dim3(256), stream, a);
  } else {
    if (bwd) CENET_LAUNCH((dwbn_bwd_kernel<T, 1024, DWBN_SM>), dim3(grid), dim3(1024), stream, a);
    else CENET_LAUNCH((dwbn_fwd_kernel<T, 1024, DWBN_SM>), dim3(grid), dim3(1024), stream, a);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

template <typename T>
static int dwbn_fwd_impl(const T* x, long sxb, const float* const* w, const int* dil, int nb, int g, int p, T* v, long svb, T* rest,
                         long srb, const float* gamma, const float* beta, float eps, float* mean, float* var, float* rmean,
                         float* rvar, float momentum, long* nbt, int B, int H, int W, hipStream_t stream) {
  if (!x || !w || !dil || !v || !gamma || !beta || !mean || !var || (p > 0 && !rest) || B <= 0 || H <= 0 || W <= 0 || nb < 1 ||
      nb > 3)
    return CENET_EINVAL;
  DwBnArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.sxb = sxb; a.y = v; a.syb = svb; a.rest = rest; a.srb = srb;
  for (int j = 0; j < nb; ++j) {
    if (!w[j] || dil[j] < 1) return CENET_EINVAL;
    a.w[j] = w[j];
    a.dil[j] = dil[j];
  }
  a.G = g; a.NB = nb; a.P = p; a.gamma = gamma; a.beta = beta; a.eps = eps; a.mean = mean; a.var = var; a.rmean = rmean;
  a.rvar = rvar; a.momentum = momentum; a.nbt = nbt; a.B = B; a.H = H; a.W = W;
  return dwbn_launch<T>(a, false, stream);
}

template <typename T>
static int dwbn_bwd_acc_impl(const T* g_v, long sgb, const T* g_rest, long srb, const T* g_add, long sab, const T* x, long sxb,
                             const float* const* w, const int* dil, int nb, int g, int p, const float* gamma, const float* beta,
                             float eps, const float* mean, const float* var, T* dx, long sdb, float* du_ws, float* const* dw_acc,
                             float* dgamma_acc, float* dbeta_acc, int B, int H, int W, hipStream_t stream) {
  if (!g_v || !x || !w || !dil || !gamma || !beta || !mean || !var || !dx || !du_ws || !dw_acc || !dgamma_acc || !dbeta_acc ||
      (p > 0 && !g_rest) || B <= 0 || H <= 0 || W <= 0 || nb < 1 || nb > 3)
    return CENET_EINVAL;
  DwBnArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.sxb = sxb; a.y = dx; a.syb = sdb; a.rest = (void*)g_rest; a.srb = srb; a.g = g_v; a.sgb = sgb; a.gadd = g_add;
  a.sab = sab; a.du = du_ws;
  for (int j = 0; j < nb; ++j) {
    if (!w[j] || !dw_acc[j] || dil[j] < 1) return CENET_EINVAL;
    a.w[j] = w[j];
    a.dw[j] = dw_acc[j];
    a.dil[j] = dil[j];
  }
  a.G = g; a.NB = nb; a.P = p; a.gamma = gamma; a.beta = beta; a.eps = eps; a.mean = (float*)mean; a.var = (float*)var;
  a.dgamma = dgamma_acc; a.dbeta = dbeta_acc; a.B = B; a.H = H; a.W = W;
  return dwbn_launch<T>(a, true, stream);
}

constexpr int EUCB_SM_SMALL = 48 * 1024, EUCB_SM_LARGE = 152 * 1024;

// images per group of the backward's LDS gradient planes (0: does not fit)
template <typename T>
static inline int eucb_bwd_group(int B, int H, int W, int sm) {
  const long xs = (((long)B * (H + 2) * (W + 2) * sizeof(T)) + 15) & ~15L;
  const long plane = 4L * (2 * H + 2) * (2 * W + 2);
  long G = (sm - xs) / plane;
  if (G > B) G = B;
  return G < 1 ? 0 : (int)G;
}

template <typename T>
static int eucb_fwd_impl(const T* x, long sxb, const float* w, const float* gamma, const float* beta, float eps, float slope,
                         T* y, long syb, float* mean, float* var, float* rmean, float* rvar, float momentum, long* nbt, int B,
                         int C, int H, int W, hipStream_t stream) {
  if (!x || !w || !gamma || !beta || !y || !mean || !var || B <= 0 || C <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  const long need = (long)B * (H + 2) * (W + 2) * sizeof(T);
  if (need > EUCB_SM_LARGE || (syb & 1) || (((uintptr_t)y) & 7)) return CENET_EUNSUPPORTED;
  EucbArgs a;
  a.x = x; a.sxb = sxb; a.y = y; a.syb = syb; a.g = nullptr; a.sgb = 0; a.w = w; a.gamma = gamma; a.beta = beta; a.eps = eps;
  a.slope = slope; a.mean = mean; a.var = var; a.rmean = rmean; a.rvar = rvar; a.momentum = momentum; a.nbt = nbt;
  a.dw = a.dgamma = a.dbeta = nullptr; a.B = B; a.H = H; a.W = W;
  if (need <= EUCB_SM_SMALL) CENET_LAUNCH((eucb_fwd_kernel<T, 512, EUCB_SM_SMALL>), dim3(C), dim3(512), stream, a);
  else CENET_LAUNCH((eucb_fwd_kernel<T, 1024, EUCB_SM_LARGE>), dim3(C), dim3(1024), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

template <typename T>
static int eucb_bwd_acc_impl(const T* g, long sgb, const T* x, long sxb, const float* w, const float* gamma, const float* beta,
                             float eps, float slope, const float* mean, const float* var, T* dx, long sdb, float* dw_acc,
                             float* dgamma_acc, float* dbeta_acc, int B, int C, int H, int W, hipStream_t stream) {
  if (!g || !x || !w || !gamma || !beta || !mean || !var || !dx || !dw_acc || !dgamma_acc || !dbeta_acc || B <= 0 || C <= 0 ||
      H <= 0 || W <= 0)
    return CENET_EINVAL;
  if ((sgb & 1) || (((uintptr_t)g) & 7)) return CENET_EUNSUPPORTED;
  EucbArgs a;
  a.x = x; a.sxb = sxb; a.y = dx; a.syb = sdb; a.g = g; a.sgb = sgb; a.w = w; a.gamma = gamma; a.beta = beta; a.eps = eps;
  a.slope = slope; a.mean = (float*)mean; a.var = (float*)var; a.rmean = a.rvar = nullptr; a.momentum = 0.f; a.nbt = nullptr;
  a.dw = dw_acc; a.dgamma = dgamma_acc; a.dbeta = dbeta_acc; a.B = B; a.H = H; a.W = W;
  int G = eucb_bwd_group<T>(B, H, W, EUCB_SM_SMALL);
  if (G >= 4 || G == B) {
    CENET_LAUNCH((eucb_bwd_kernel<T, 512, EUCB_SM_SMALL>), dim3(C), dim3(512), stream, a, G);
  } else {
    G = eucb_bwd_group<T>(B, H, W, EUCB_SM_LARGE);
    if (G < 1) return CENET_EUNSUPPORTED;
    CENET_LAUNCH((eucb_bwd_kernel<T, 1024, EUCB_SM_LARGE>), dim3(C), dim3(1024), stream, a, G);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

}  // namespace

/* does the fused EUCB front (forward AND backward) take this shape?  esize = 2 (bf16) / 4 (fp32) */
extern "C" int cenet_eucb_supported(int B, int H, int W, int esize) {
  if (B <= 0 || H <= 0 || W <= 0 || (esize != 2 && esize != 4)) return 0;
  const long need = (long)B * (H + 2) * (W + 2) * esize;
  if (need > EUCB_SM_LARGE) return 0;
  return (esize == 2 ? eucb_bwd_group<bf16_t>(B, H, W, EUCB_SM_LARGE) : eucb_bwd_group<float>(B, H, W, EUCB_SM_LARGE)) >= 1;
}

CENET_TWIN(eucb_fwd, (const T* x, long sxb, const float* w, const float* gamma, const float* beta, float eps, float slope, T* y,
                      long syb, float* mean, float* var, float* running_mean, float* running_var, float momentum,
                      long* num_batches_tracked, int B, int C, int H, int W, hipStream_t stream),
           (x, sxb, w, gamma, beta, eps, slope, y, syb, mean, var, running_mean, running_var, momentum, num_batches_tracked, B, C,
            H, W, stream))
CENET_TWIN(eucb_bwd_acc, (const T* g, long sgb, const T* x, long sxb, const float* w, const float* gamma, const float* beta,
                          float eps, float slope, const float* mean, const float* var, T* dx, long sdb, float* dw_acc,
                          float* dgamma_acc, float* dbeta_acc, int B, int C, int H, int W, hipStream_t stream),
           (g, sgb, x, sxb, w, gamma, beta, eps, slope, mean, var, dx, sdb, dw_acc, dgamma_acc, dbeta_acc, B, C, H, W, stream))

/* a channel over the batch (B * HW elements) is small enough for the channel-local chains */
extern "C" int cenet_chanloc_supported(int B, int HW) { return (long)B * HW >= 2 && (long)B * HW <= CENET_CHANLOC_MAX; }

CENET_TWIN(cfam_mid_fwd, (const T* p_raw, const T* m, const T* x0, T* x1, T* y2, const float* gamma_p, const float* beta_p,
                          float eps_p, float* mean_p, float* var_p, float* rmean_p, float* rvar_p, float mom_p, long* nbt_p,
                          const float* w, const float* ls, const float* gamma_2, const float* beta_2, float eps_2, float* mean_2,
                          float* var_2, float* rmean_2, float* rvar_2, float mom_2, long* nbt_2, int B, int C, int HW,
                          hipStream_t stream),
           (p_raw, m, x0, x1, y2, gamma_p, beta_p, eps_p, mean_p, var_p, rmean_p, rvar_p, mom_p, nbt_p, w, ls, gamma_2, beta_2,
            eps_2, mean_2, var_2, rmean_2, rvar_2, mom_2, nbt_2, B, C, HW, stream))
CENET_TWIN(cfam_mid_bwd_acc, (const T* g_y2, const T* g_x1, const T* p_raw, const T* m, const T* x1, T* d_p_raw, T* d_m, T* d_x0,
                              const float* gamma_p, const float* beta_p, float eps_p, const float* mean_p, const float* var_p,
                              const float* w, const float* ls, const float* gamma_2, float eps_2, const float* mean_2,
                              const float* var_2, float* dgamma_p_acc, float* dbeta_p_acc, float* dw_acc, float* dls_acc,
                              float* dgamma_2_acc, float* dbeta_2_acc, int B, int C, int HW, hipStream_t stream),
           (g_y2, g_x1, p_raw, m, x1, d_p_raw, d_m, d_x0, gamma_p, beta_p, eps_p, mean_p, var_p, w, ls, gamma_2, eps_2, mean_2,
            var_2, dgamma_p_acc, dbeta_p_acc, dw_acc, dls_acc, dgamma_2_acc, dbeta_2_acc, B, C, HW, stream))

CENET_TWIN(dwbn_fwd, (const T* x, long sxb, const float* const* w, const int* dil, int nb, int g, int p, T* v, long svb, T* rest,
                      long srb, const float* gamma, const float* beta, float eps, float* mean, float* var, float* running_mean,
                      float* running_var, float momentum, long* num_batches_tracked, int B, int H, int W, hipStream_t stream),
           (x, sxb, w, dil, nb, g, p, v, svb, rest, srb, gamma, beta, eps, mean, var, running_mean, running_var, momentum,
            num_batches_tracked, B, H, W, stream))
CENET_TWIN(dwbn_bwd_acc, (const T* g_v, long sgb, const T* g_rest, long srb, const T* g_add, long sab, const T* x, long sxb,
                          const float* const* w, const int* dil, int nb, int g, int p, const float* gamma, const float* beta,
                          float eps, const float* mean, const float* var, T* dx, long sdb, float* du_ws, float* const* dw_acc,
                          float* dgamma_acc, float* dbeta_acc, int B, int H, int W, hipStream_t stream),
           (g_v, sgb, g_rest, srb, g_add, sab, x, sxb, w, dil, nb, g, p, gamma, beta, eps, mean, var, dx, sdb, du_ws, dw_acc,
            dgamma_acc, dbeta_acc, B, H, W, stream))
