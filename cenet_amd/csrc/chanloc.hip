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
// RO[py][ky] = index into S's rows (0: i-1, 1: i, 2: i+1) of tap ky for output parity py; columns alike.
__device__ __forceinline__ int eucb_ro(int par, int k) { return par == 0 ? (k == 0 ? 0 : 1) : (k == 2 ? 2 : 1); }

template <typename T>
__device__ __forceinline__ void eucb_load_nb(const T* xs, int b, int i, int j, int H, int W, float (&S)[3][3]) {
  const T* p = xs + (long)b * H * W;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int yy = i + a - 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int xx = j + c - 1;
      S[a][c] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? ldf(p + yy * W + xx) : 0.f;
    }
  }
}
// the four conv outputs of the quad: u[py][px]
__device__ __forceinline__ void eucb_quad(const float (&S)[3][3], const float (&w)[9], float (&u)[2][2]) {
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      float t = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) t += w[ky * 3 + kx] * S[eucb_ro(py, ky)][eucb_ro(px, kx)];
      u[py][px] = t;
    }
}

// grid = C, NT threads, SM bytes of LDS holding the channel's source planes [B][H][W] as T
template <typename T, int NT, int SM>
__global__ __launch_bounds__(NT) void eucb_fwd_kernel(EucbArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  __shared__ float red[16 * 2];
  T* xs = (T*)smem;
  const int c = blockIdx.x, H = a.H, W = a.W, HW = H * W, items = a.B * HW;
  const T* x = (const T*)a.x + (long)c * HW;
  for (int e = threadIdx.x; e < items; e += NT) {
    const int b = e / HW, p = e - b * HW;
    xs[e] = x[(long)b * a.sxb + p];
  }
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = a.w[c * 9 + k];
  __syncthreads();
  // pass 1: batch statistics of the conv output (shifted by the value at the plane centre of image 0: fp32 sums of squares)
  float S[3][3], u[2][2];
  eucb_load_nb(xs, 0, H / 2, W / 2, H, W, S);
  eucb_quad(S, w, u);
  const float K = u[0][0];
  float s[2] = {0.f, 0.f};
  for (int e = threadIdx.x; e < items; e += NT) {
    const int b = e / HW, p = e - b * HW, i = p / W, j = p - i * W;
    eucb_load_nb(xs, b, i, j, H, W, S);
    eucb_quad(S, w, u);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float d = u[q >> 1][q & 1] - K;
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
  for (int e = threadIdx.x; e < items; e += NT) {
    const int b = e / HW, p = e - b * HW, i = p / W, j = p - i * W;
    eucb_load_nb(xs, b, i, j, H, W, S);
    eucb_quad(S, w, u);
    T* yp = y + (long)b * a.syb + (long)(2 * i) * OW + 2 * j;
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

// backward.  LDS: the source planes [B][H][W] as T, then G planes [2H][2W] of fp32 conv-output gradients (one group of G images
// at a time: the depthwise data gradient needs the neighbours of an output pixel's gradient)
template <typename T, int NT, int SM>
__global__ __launch_bounds__(NT) void eucb_bwd_kernel(EucbArgs a, int G) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  __shared__ float red[16 * 11];
  const int c = blockIdx.x, H = a.H, W = a.W, HW = H * W, items = a.B * HW, OW = 2 * W, OH = 2 * H;
  T* xs = (T*)smem;
  float* du = (float*)(smem + (((long)items * sizeof(T) + 15) & ~15L));
  const T* x = (const T*)a.x + (long)c * HW;
  for (int e = threadIdx.x; e < items; e += NT) {
    const int b = e / HW, p = e - b * HW;
    xs[e] = x[(long)b * a.sxb + p];
  }
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = a.w[c * 9 + k];
  const float mu = a.mean[c], rs = rsqrtf(a.var[c] + a.eps), gm = a.gamma[c], bt = a.beta[c];
  __syncthreads();
  const T* g = (const T*)a.g + (long)c * 4 * HW;
  float S[3][3], u[2][2];
  // gradient of the quad's four conv outputs through LeakyReLU, and their normalised values
  auto quad_g = [&](int b, int i, int j, float (&gy)[2][2], float (&xh)[2][2]) __attribute__((always_inline)) {
    eucb_load_nb(xs, b, i, j, H, W, S);
    eucb_quad(S, w, u);
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
  for (int e = threadIdx.x; e < items; e += NT) {
    const int b = e / HW, p = e - b * HW, i = p / W, j = p - i * W;
    quad_g(b, i, j, gy, xh);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s[0] += gy[q >> 1][q & 1];
      s[1] += gy[q >> 1][q & 1] * xh[q >> 1][q & 1];
    }
  }
  block_sum_n<2>(s, red);
  const float n = 4.f * (float)items, m1 = s[0] / n, m2 = s[1] / n, k0 = gm * rs;
  // combined weights of the data gradient: dx(i, j) = sum_{r, q = -1..2} cw[r+1][q+1] du(2i + r, 2j + q), where
  // cw = sum of w[ky][kx] over KY(r) x KX(q), KY(-1) = {2}, KY(0) = {1, 2}, KY(1) = {0, 1}, KY(2) = {0}
  float cw[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float t = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const bool iny = (r == 0 && ky == 2) || (r == 1 && ky >= 1) || (r == 2 && ky <= 1) || (r == 3 && ky == 0);
          const bool inx = (q == 0 && kx == 2) || (q == 1 && kx >= 1) || (q == 2 && kx <= 1) || (q == 3 && kx == 0);
          if (iny && inx) t += w[ky * 3 + kx];
        }
      cw[r][q] = t;
    }
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  T* dx = (T*)a.y + (long)c * HW;
  const int OHW = 4 * HW;
  for (int b0 = 0; b0 < a.B; b0 += G) {
    const int nb = a.B - b0 < G ? a.B - b0 : G;
    // (a) conv-output gradients of images b0 .. b0 + nb into LDS; weight-gradient sums in registers
    for (int e = threadIdx.x; e < nb * HW; e += NT) {
      const int bl = e / HW, p = e - bl * HW, i = p / W, j = p - i * W;
      quad_g(b0 + bl, i, j, gy, xh);
      float d[2][2];
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          d[py][px] = k0 * (gy[py][px] - m1 - xh[py][px] * m2);
          du[(long)bl * OHW + (2 * i + py) * OW + 2 * j + px] = d[py][px];
        }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float t = 0.f;
#pragma unroll
          for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) t += d[py][px] * S[eucb_ro(py, ky)][eucb_ro(px, kx)];
          acc[ky * 3 + kx] += t;
        }
    }
    __syncthreads();
    // (b) data gradient of the group's source pixels
    for (int e = threadIdx.x; e < nb * HW; e += NT) {
      const int bl = e / HW, p = e - bl * HW, i = p / W, j = p - i * W;
      const float* dp = du + (long)bl * OHW;
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yy = 2 * i + r - 1;
        if (yy < 0 || yy >= OH) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int xx = 2 * j + q - 1;
          if (xx >= 0 && xx < OW) t += cw[r][q] * dp[yy * OW + xx];
        }
      }
      stf(dx + (long)(b0 + bl) * a.syb + p, t);
    }
    __syncthreads();
  }
  float fin[11];
#pragma unroll
  for (int k = 0; k < 9; ++k) fin[k] = acc[k];
  fin[9] = 0.f, fin[10] = 0.f;
  block_sum_n<11>(fin, red);
  if (threadIdx.x < 9) a.dw[c * 9 + threadIdx.x] += fin[threadIdx.x];
  if (threadIdx.x == 9) a.dgamma[c] += s[1];
  if (threadIdx.x == 10) a.dbeta[c] += s[0];
}

constexpr int EUCB_SM_SMALL = 48 * 1024, EUCB_SM_LARGE = 152 * 1024;

// images per group of the backward's LDS gradient planes (0: does not fit)
template <typename T>
static inline int eucb_bwd_group(int B, int H, int W, int sm) {
  const long xs = (((long)B * H * W * sizeof(T)) + 15) & ~15L;
  const long plane = 16L * H * W;
  long G = (sm - xs) / plane;
  if (G > B) G = B;
  return G < 1 ? 0 : (int)G;
}

template <typename T>
static int eucb_fwd_impl(const T* x, long sxb, const float* w, const float* gamma, const float* beta, float eps, float slope,
                         T* y, long syb, float* mean, float* var, float* rmean, float* rvar, float momentum, long* nbt, int B,
                         int C, int H, int W, hipStream_t stream) {
  if (!x || !w || !gamma || !beta || !y || !mean || !var || B <= 0 || C <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  const long need = (long)B * H * W * sizeof(T);
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
  const long need = (long)B * H * W * esize;
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
