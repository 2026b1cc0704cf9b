// loss_optim.hip — Dice + cross-entropy loss (forward/backward) and the fused SGD-momentum update.
//   Criterion / DiceLoss / CrossEntropyLoss   utils/core.py:44-80,161-188
//   SGD(momentum 0.9, weight decay)           utils/core.py:19-21 (torch.optim.SGD semantics)
// The loss never leaves the device: per-class Dice sums (over the WHOLE batch, as the reference does) and the CE sum
// are reduced with block partials + float atomics into a small accumulator that the backward kernel re-reads.
#include "common.h"
#include "../../include/cenet_hip.h"

#define LOSS_MAXK 16
// float atomics onto one cache line serialise (~12 ns each), so thousands of workgroups must not share a line.
// Round 5: one value PER 64-BYTE LINE — replica s of value v sits at acc[(v * ns + s) * 16], ns = loss_ns(K) replicas.  With the
// 13 - 65 values of a replica in one line (the layout until then), the ~10 000 adds of a forward launch queued on 16 lines: 26 of the
// kernel's 32 us.  The finalize kernel folds the replicas into the compact [0 .. 5K+1) layout the backward kernel reads.
#define LOSS_ACC_FLOATS 16384
__host__ __device__ static inline int loss_ns(int K) {
  const int ns = LOSS_ACC_FLOATS / ((4 * K + 1) * 16);
  return ns > 16 ? 16 : (ns < 1 ? 1 : ns);
}

// acc layout: [0..K) intersect, [K..2K) z_sum (p^2), [2K..3K) y_sum (t), [3K] ce_sum, [3K+1..4K+1) boundary pixel counts
// (BoundaryDoULoss, core.py:105-109), and after finalize [4K+1..5K+1) the per-class alpha of core.py:112-119
// KM: compile-time bound of the class loops (4 / 9 / 16: the predicated 16-way loops cost 4x the work at K = 4)
#define DICE_IT 8
template <typename T, int KM>
__global__ __launch_bounds__(256) void dice_ce_fwd_kernel(const T* __restrict__ logits, const float* __restrict__ labels,
                                                         float* __restrict__ acc, int K, int HW, long npix, int W,
                                                         int boundary) {
  __shared__ float red[16];
  float part[3 * KM + 1], bnd[KM];
#pragma unroll
  for (int i = 0; i < 3 * KM + 1; ++i) part[i] = 0.f;
#pragma unroll
  for (int i = 0; i < KM; ++i) bnd[i] = 0.f;
  // grid (ceil(HW / (256 IT)), B): a thread owns IT pixels of image blockIdx.y, all their loads issued together (the flat grid-stride
  // form put a 64-bit division and a dependent round trip of loads in front of every pixel: 32 us for 19 MB)
  constexpr int IT = KM > 9 ? 4 : DICE_IT;  // (16 classes x 8 pixels would not stay in registers)
  const int H = HW / W, b = blockIdx.y;
  const T* lb0 = logits + (long)b * K * HW;
  const float* lab = labels + (long)b * HW;
#pragma unroll 1
  for (int k0 = 0; k0 < IT; k0 += 2) {  // two pixels at a time (the fully unrolled eight ran slower than the loop it replaced)
  float vv[2][KM], tl[2];
  int pp[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = (blockIdx.x * IT + k0 + k) * 256 + threadIdx.x;
    pp[k] = p < HW ? p : -1;
    const int q = p < HW ? p : 0;
#pragma unroll
    for (int c = 0; c < KM; ++c)
      if (c < K) vv[k][c] = ldf(lb0 + (long)c * HW + q);
    tl[k] = lab[q];
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (pp[k] < 0) continue;
    const int p = pp[k];
    float v[KM];
    float mx = -3.4e38f;
#pragma unroll
    for (int c = 0; c < KM; ++c)
      if (c < K) {
        v[c] = vv[k][c];
        mx = fmaxf(mx, v[c]);
      }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < KM; ++c)
      if (c < K) {
        v[c] = fast_exp(v[c] - mx);  // (libm's expf / logf / the IEEE division: ~70 instructions per class and pixel)
        s += v[c];
      }
    const float inv = fast_rcp(s);
    const int t = (int)tl[k];
    bool edge = false;
    if (boundary) {  // a foreground pixel of class t is a boundary pixel unless its four neighbours (zero padded) share t
      const int py = p / W, px = p - py * W;
      const float* lb = lab;
      edge = !(py > 0 && (int)lb[p - W] == t && py + 1 < H && (int)lb[p + W] == t && px > 0 && (int)lb[p - 1] == t &&
               px + 1 < W && (int)lb[p + 1] == t);
    }
#pragma unroll
    for (int c = 0; c < KM; ++c)
      if (c < K) {
        if (c == t && edge) bnd[c] += 1.f;
        const float pc = v[c] * inv;
        part[KM + c] += pc * pc;
        if (c == t) {
          part[c] += pc;
          part[2 * KM + c] += 1.f;
          part[3 * KM] -= fast_log(fmaxf(pc, 1e-37f));
        }
      }
  }
  }
  const int ns = loss_ns(K), rep = (blockIdx.y * gridDim.x + blockIdx.x) % ns;
  auto put = [&](int v, float x) { atomicAdd(&acc[(v * ns + rep) * 16], x); };
#pragma unroll  // static indices: part[] must stay in registers
  for (int c = 0; c < KM; ++c)
    if (c < K) {
      float a0 = block_sum(part[c], red), a1 = block_sum(part[KM + c], red), a2 = block_sum(part[2 * KM + c], red);
      const float a3 = boundary ? block_sum(bnd[c], red) : 0.f;
      if (threadIdx.x == 0) {
        put(c, a0);
        put(K + c, a1);
        put(2 * K + c, a2);
        if (boundary) put(3 * K + 1 + c, a3);
      }
    }
  float ce = block_sum(part[3 * KM], red);
  if (threadIdx.x == 0) put(3 * K, ce);
}

__global__ void dice_ce_finalize_kernel(float* __restrict__ acc, float* __restrict__ loss, int K, float npix,
                                        float w_dice, float w_ce, float w_bd) {
  float t = 0.f;  // fold the replicas into the compact layout (what the backward kernel reads): all reads, then all writes
  const int ns = loss_ns(K);
  if (threadIdx.x < 4 * K + 1)
    for (int sidx = 0; sidx < ns; ++sidx) t += acc[(threadIdx.x * ns + sidx) * 16];
  __syncthreads();
  if (threadIdx.x < 4 * K + 1) acc[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float d = 0.f, bd = 0.f;
    for (int c = 0; c < K; ++c) {
      const float I = acc[c], z = acc[K + c], y = acc[2 * K + c];
      d += 1.f - (2.f * I + 1e-5f) / (z + y + 1e-5f);
      // BoundaryDoU (core.py:110-121): alpha from the boundary / area ratio of the class mask, truncated at 0.8
      float alpha = 2.f * (1.f - (acc[3 * K + 1 + c] + 1e-5f) / (y + 1e-5f)) - 1.f;
      if (alpha > 0.8f) alpha = 0.8f;
      acc[4 * K + 1 + c] = alpha;
      bd += (z + y - 2.f * I + 1e-5f) / (z + y - (1.f + alpha) * I + 1e-5f);
    }
    loss[0] = w_dice * d / K + w_ce * acc[3 * K] / npix + w_bd * bd / K;
  }
}

template <typename T, int KM>
__global__ __launch_bounds__(256) void dice_ce_bwd_kernel(const T* __restrict__ logits, const float* __restrict__ labels,
                                                         const float* __restrict__ acc, const float* __restrict__ gout,
                                                         T* __restrict__ dlogits, int K, int HW, long npix, float w_dice,
                                                         float w_ce, float w_bd) {
  const float go = gout[0];
  float A[KM], Bc[KM];  // dL/dp_c = A_c * t_c + Bc_c * p_c
#pragma unroll
  for (int c = 0; c < KM; ++c)
    if (c < K) {
      const float I = acc[c], zy = acc[K + c] + acc[2 * K + c];
      const float D = zy + 1e-5f;
      const float num = 2.f * I + 1e-5f;
      A[c] = -w_dice / K * 2.f / D;
      Bc[c] = w_dice / K * num * 2.f / (D * D);
      if (w_bd != 0.f) {  // L_c = N / Db, N = z+y-2I+eps, Db = z+y-(1+alpha)I+eps ; dz/dp = 2p, dI/dp = t
        const float alpha = acc[4 * K + 1 + c];
        const float N = zy - 2.f * I + 1e-5f, Db = zy - (1.f + alpha) * I + 1e-5f;
        A[c] += w_bd / K * (N * (1.f + alpha) - 2.f * Db) / (Db * Db);
        Bc[c] += w_bd / K * 2.f * (Db - N) / (Db * Db);
      }
    }
  const float cew = w_ce / (float)npix;
  constexpr int IT = 4;  // grid (ceil(HW / (256 IT)), B), as the forward kernel
  const int b = blockIdx.y;
  const T* lb0 = logits + (long)b * K * HW;
  T* db0 = dlogits + (long)b * K * HW;
  const float* lab = labels + (long)b * HW;
  float vv[IT][KM], tl[IT];
  int pp[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int p = (blockIdx.x * IT + k) * 256 + threadIdx.x;
    pp[k] = p < HW ? p : -1;
    const int q = p < HW ? p : 0;
#pragma unroll
    for (int c = 0; c < KM; ++c)
      if (c < K) vv[k][c] = ldf(lb0 + (long)c * HW + q);
    tl[k] = lab[q];
  }
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    if (pp[k] < 0) continue;
    float v[KM];
    float mx = -3.4e38f;
#pragma unroll
    for (int c = 0; c < KM; ++c)
      if (c < K) {
        v[c] = vv[k][c];
        mx = fmaxf(mx, v[c]);
      }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < KM; ++c)
      if (c < K) {
        v[c] = fast_exp(v[c] - mx);  // (libm's expf / logf / the IEEE division: ~70 instructions per class and pixel)
        s += v[c];
      }
    const float inv = fast_rcp(s);
    const int t = (int)tl[k];
    float gp[KM];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < KM; ++c)
      if (c < K) {
        v[c] *= inv;
        gp[c] = Bc[c] * v[c] + (c == t ? A[c] : 0.f);
        dot += gp[c] * v[c];
      }
#pragma unroll
    for (int c = 0; c < KM; ++c)
      if (c < K) stf(db0 + (long)c * HW + pp[k], go * (v[c] * (gp[c] - dot) + cew * (v[c] - (c == t ? 1.f : 0.f))));
  }
}

// p -= lr * buf, buf = momentum*buf + (g*gscale + wd*p)   (first step: buf = g*gscale + wd*p)
// hyper (device): [lr, momentum, weight_decay, gscale, first_step_flag]
// shadow (may be NULL): bf16 copy of the updated parameters, the operand the throughput-mode GEMMs read
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                 const float* __restrict__ hyper, long n, bf16_t* __restrict__ shadow) {
  const float lr = hyper[0], mom = hyper[1], wd = hyper[2], gs = hyper[3];
  const bool first = hyper[4] != 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pv = p[i];
    const float d = g[i] * gs + wd * pv;
    const float bv = first ? d : mom * buf[i] + d;
    buf[i] = bv;
    const float pn = pv - lr * bv;
    p[i] = pn;
    if (shadow) shadow[i] = (bf16_t)cenet_f2bf(pn);
  }
}
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) stf(y + i, ldf(x + i));
}
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast4_kernel(const TI* __restrict__ x, TO* __restrict__ y, long nq) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long)gridDim.x * 256) st4(y + 4 * i, ld4(x + 4 * i));
}

__global__ __launch_bounds__(256) void zero_kernel(float* __restrict__ p, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0.f;
}
int cenet_zero_async(float* p, long n, hipStream_t stream) {
  if (n <= 0) return CENET_OK;
  long blocks = (n + 1023) / 1024;
  if (blocks > 8192) blocks = 8192;
  CENET_LAUNCH(zero_kernel, dim3((unsigned)blocks), dim3(256), stream, p, n);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

// loss = w_dice * Dice + w_ce * CE + w_bd * BoundaryDoU, all from one pass over the logits (labels [B,H,W] as floats)
template <typename T>
static int seg_loss_fwd_impl(const T* logits, const float* labels, float* acc, float* loss, int B, int K, int H, int W,
                             float w_dice, float w_ce, float w_bd, hipStream_t stream) {
  if (B <= 0 || K <= 0 || K > LOSS_MAXK || H <= 0 || W <= 0) return CENET_EINVAL;
  if (cenet_zero_async(acc, (long)LOSS_ACC_FLOATS, stream) != CENET_OK) return CENET_EINVAL;
  const int HW = H * W;
  const long npix = (long)B * HW;
  if (B > 65535) return CENET_EUNSUPPORTED;
  const dim3 blocks(cdiv(HW, 256 * (K > 9 ? 4 : DICE_IT)), B);
  const int bnd = (int)(w_bd != 0.f);
  if (K <= 4) CENET_LAUNCH((dice_ce_fwd_kernel<T, 4>), blocks, dim3(256), stream, logits, labels, acc, K, HW, npix, W, bnd);
  else if (K <= 9) CENET_LAUNCH((dice_ce_fwd_kernel<T, 9>), blocks, dim3(256), stream, logits, labels, acc, K, HW, npix, W, bnd);
  else CENET_LAUNCH((dice_ce_fwd_kernel<T, LOSS_MAXK>), blocks, dim3(256), stream, logits, labels, acc, K, HW, npix, W, bnd);
  CENET_LAUNCH(dice_ce_finalize_kernel, dim3(1), dim3(128), stream, acc, loss, K, (float)npix, w_dice, w_ce, w_bd);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(seg_loss_fwd, (const T* logits, const float* labels, float* acc, float* loss, int B, int K, int H, int W, float w_dice,
                          float w_ce, float w_bd, hipStream_t stream),
           (logits, labels, acc, loss, B, K, H, W, w_dice, w_ce, w_bd, stream))

template <typename T>
static int seg_loss_bwd_impl(const T* logits, const float* labels, const float* acc, const float* gout, T* dlogits, int B, int K,
                             int H, int W, float w_dice, float w_ce, float w_bd, hipStream_t stream) {
  if (B <= 0 || K <= 0 || K > LOSS_MAXK || H <= 0 || W <= 0) return CENET_EINVAL;
  const int HW = H * W;
  const long npix = (long)B * HW;
  if (B > 65535) return CENET_EUNSUPPORTED;
  const dim3 blocks(cdiv(HW, 256 * 4), B);
  if (K <= 4) CENET_LAUNCH((dice_ce_bwd_kernel<T, 4>), blocks, dim3(256), stream, logits, labels, acc, gout, dlogits, K, HW, npix, w_dice, w_ce, w_bd);
  else if (K <= 9) CENET_LAUNCH((dice_ce_bwd_kernel<T, 9>), blocks, dim3(256), stream, logits, labels, acc, gout, dlogits, K, HW, npix, w_dice, w_ce, w_bd);
  else CENET_LAUNCH((dice_ce_bwd_kernel<T, LOSS_MAXK>), blocks, dim3(256), stream, logits, labels, acc, gout, dlogits, K, HW, npix, w_dice, w_ce, w_bd);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(seg_loss_bwd, (const T* logits, const float* labels, const float* acc, const float* gout, T* dlogits, int B, int K,
                          int H, int W, float w_dice, float w_ce, float w_bd, hipStream_t stream),
           (logits, labels, acc, gout, dlogits, B, K, H, W, w_dice, w_ce, w_bd, stream))

extern "C" int cenet_dice_ce_fwd_f32(const float* logits, const float* labels, float* acc, float* loss, int B, int K, int HW,
                                     float w_dice, float w_ce, hipStream_t stream) {
  return seg_loss_fwd_impl<float>(logits, labels, acc, loss, B, K, 1, HW, w_dice, w_ce, 0.f, stream);
}
extern "C" int cenet_dice_ce_bwd_f32(const float* logits, const float* labels, const float* acc, const float* gout,
                                     float* dlogits, int B, int K, int HW, float w_dice, float w_ce, hipStream_t stream) {
  return seg_loss_bwd_impl<float>(logits, labels, acc, gout, dlogits, B, K, 1, HW, w_dice, w_ce, 0.f, stream);
}
static int sgd_launch(float* p, const float* g, float* buf, const float* hyper5, long n, bf16_t* shadow, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  CENET_LAUNCH(sgd_kernel, dim3((unsigned)blocks), dim3(256), stream, p, g, buf, hyper5, n, shadow);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_sgd_step_f32(float* p, const float* g, float* buf, const float* hyper5, long n, hipStream_t stream) {
  return sgd_launch(p, g, buf, hyper5, n, nullptr, stream);
}
extern "C" int cenet_sgd_step_shadow_f32(float* p, const float* g, float* buf, const float* hyper5, long n,
                                         unsigned short* shadow_bf16, hipStream_t stream) {
  return sgd_launch(p, g, buf, hyper5, n, shadow_bf16, stream);
}
template <typename TI, typename TO>
static int cast_launch(const TI* x, TO* y, long n, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  if ((n & 3) == 0 && quad_aligned<TI>(x) && quad_aligned<TO>(y)) {
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    CENET_LAUNCH((cast4_kernel<TI, TO>), dim3((unsigned)blocks), dim3(256), stream, x, y, n / 4);
  } else {
    long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    CENET_LAUNCH((cast_kernel<TI, TO>), dim3((unsigned)blocks), dim3(256), stream, x, y, n);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_cast_f32_to_bf16(const float* x, unsigned short* y, long n, hipStream_t stream) {
  return cast_launch<float, bf16_t>(x, y, n, stream);
}
// y = bf16(x); x = 0: the fp32 accumulator a kernel added into atomically is rounded to the gradient's storage type AND left zero
// for its next user in one pass (the accumulator is a persistent workspace: no zero fill in front of every use)
__global__ __launch_bounds__(256) void cast_clear_kernel(float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    stf(y + i, x[i]);
    x[i] = 0.f;
  }
}
__global__ __launch_bounds__(256) void cast_clear4_kernel(float* __restrict__ x, bf16_t* __restrict__ y, long nq) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long)gridDim.x * 256) {
    st4(y + 4 * i, ld4(x + 4 * i));
    f4 z;
    z.v[0] = z.v[1] = z.v[2] = z.v[3] = 0.f;
    st4(x + 4 * i, z);
  }
}
// the same with a per-column bias (rows of N columns, N % 4 == 0): y = bf16(x + bias[col]); x = 0 — the tail of a split-K Linear
// whose partial sums were added atomically into the (zero-at-rest) fp32 accumulator x
__global__ __launch_bounds__(256) void cast_clear_bias4_kernel(float* __restrict__ x, bf16_t* __restrict__ y,
                                                              const float* __restrict__ bias, int N4, long nq) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long)gridDim.x * 256) {
    f4 v = ld4(x + 4 * i);
    const f4 b = ld4(bias + 4 * (int)(i % N4));
#pragma unroll
    for (int e = 0; e < 4; ++e) v.v[e] += b.v[e];
    st4(y + 4 * i, v);
    f4 z;
    z.v[0] = z.v[1] = z.v[2] = z.v[3] = 0.f;
    st4(x + 4 * i, z);
  }
}
extern "C" int cenet_cast_clear_bias_f32_to_bf16(float* x, unsigned short* y, const float* bias, int N, long n,
                                                 hipStream_t stream) {
  if (!x || !y || !bias || n <= 0 || N <= 0 || (n % N) != 0) return CENET_EINVAL;
  if ((N & 3) != 0 || !quad_aligned<float>(x) || !quad_aligned<bf16_t>((const bf16_t*)y) || !quad_aligned<float>(bias))
    return CENET_EUNSUPPORTED;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  CENET_LAUNCH(cast_clear_bias4_kernel, dim3((unsigned)blocks), dim3(256), stream, x, (bf16_t*)y, bias, N / 4, n / 4);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_cast_clear_f32_to_bf16(float* x, unsigned short* y, long n, hipStream_t stream) {
  if (!x || !y || n <= 0) return CENET_EINVAL;
  if ((n & 3) == 0 && quad_aligned<float>(x) && quad_aligned<bf16_t>((const bf16_t*)y)) {
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    CENET_LAUNCH(cast_clear4_kernel, dim3((unsigned)blocks), dim3(256), stream, x, (bf16_t*)y, n / 4);
  } else {
    long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    CENET_LAUNCH(cast_clear_kernel, dim3((unsigned)blocks), dim3(256), stream, x, (bf16_t*)y, n);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_cast_bf16_to_f32(const unsigned short* x, float* y, long n, hipStream_t stream) {
  return cast_launch<bf16_t, float>(x, y, n, stream);
}
extern "C" int cenet_zero_f32(float* p, long n, hipStream_t stream) {
  if (n <= 0) return CENET_EINVAL;
  return cenet_zero_async(p, n, stream);
}
// zero-fill of a byte range that need not be a whole number of aligned 32-bit words (a bf16 tensor with an odd element count,
// or a contiguous view that starts on an odd element): aligned words by the grid, the <= 3 head / tail bytes by workgroup 0
__global__ __launch_bounds__(256) void zero_bytes_kernel(unsigned char* __restrict__ p, long n) {
  long head = (long)((4 - ((uintptr_t)p & 3)) & 3);
  if (head > n) head = n;
  const long words = (n - head) >> 2;
  unsigned* w = (unsigned*)(p + head);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < words; i += (long)gridDim.x * 256) w[i] = 0u;
  if (blockIdx.x == 0) {
    const long tail0 = head + 4 * words;
    if ((long)threadIdx.x < head) p[threadIdx.x] = 0;
    if (tail0 + (long)threadIdx.x < n && threadIdx.x < 4) p[tail0 + threadIdx.x] = 0;
  }
}
extern "C" int cenet_zero_bytes(void* p, long nbytes, hipStream_t stream) {
  if (!p || nbytes <= 0) return CENET_EINVAL;
  long blocks = (nbytes / 4 + 1023) / 1024;
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  CENET_LAUNCH(zero_bytes_kernel, dim3((unsigned)blocks), dim3(256), stream, (unsigned char*)p, nbytes);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

// ---- evaluation: argmax mask + overlap counts (main_acdc.py:218-231, metrics_eval.py:24-34,46-49) ---------------------
// pred[b,p] = argmax_c logits[b,c,p] (first maximum, like torch.argmax; softmax is monotone so it is skipped);
// counts[c][0..2] = #(pred==c & gt==c), #(pred==c), #(gt==c) for c < K, and row K = the binary masks pred>0 / gt>0.
template <typename T>
__global__ __launch_bounds__(256) void argmax_counts_kernel(const T* __restrict__ logits, const float* __restrict__ labels,
                                                           float* __restrict__ pred, unsigned* __restrict__ counts, int K,
                                                           int HW, long npix) {
  __shared__ unsigned sc[(LOSS_MAXK + 1) * 3];
  for (int i = threadIdx.x; i < (K + 1) * 3; i += 256) sc[i] = 0u;
  __syncthreads();
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < npix; e += (long)gridDim.x * 256) {
    const long b = e / HW;
    const int p = (int)(e - b * HW);
    const T* lp = logits + b * (long)K * HW + p;
    float best = ldf(lp);
    int arg = 0;
    for (int c = 1; c < K; ++c) {
      const float v = ldf(lp + (long)c * HW);
      if (v > best) {
        best = v;
        arg = c;
      }
    }
    if (pred) pred[e] = (float)arg;
    if (labels) {
      const int t = (int)labels[e];
      atomicAdd(&sc[arg * 3 + 1], 1u);
      if (t >= 0 && t < K) atomicAdd(&sc[t * 3 + 2], 1u);
      if (t == arg) atomicAdd(&sc[arg * 3], 1u);
      if (arg > 0) atomicAdd(&sc[K * 3 + 1], 1u);
      if (t > 0) atomicAdd(&sc[K * 3 + 2], 1u);
      if (arg > 0 && t > 0) atomicAdd(&sc[K * 3], 1u);
    }
  }
  __syncthreads();
  if (labels)
    for (int i = threadIdx.x; i < (K + 1) * 3; i += 256)
      if (sc[i]) atomicAdd(&counts[i], sc[i]);
}

template <typename T>
static int argmax_counts_impl(const T* logits, const float* labels, float* pred, unsigned* counts, int B, int K, int HW,
                              hipStream_t stream) {
  if (!logits || B <= 0 || K <= 0 || K > LOSS_MAXK || HW <= 0 || (labels && !counts)) return CENET_EINVAL;
  const long npix = (long)B * HW;
  if (labels && cenet_zero_async((float*)counts, (K + 1) * 3L, stream) != CENET_OK) return CENET_EINVAL;
  long blocks = (npix + 2047) / 2048;
  if (blocks > 256) blocks = 256;
  CENET_LAUNCH((argmax_counts_kernel<T>), dim3((unsigned)blocks), dim3(256), stream, logits, labels, pred, counts, K, HW, npix);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(argmax_counts, (const T* logits, const float* labels, float* pred, unsigned* counts, int B, int K, int HW,
                           hipStream_t stream), (logits, labels, pred, counts, B, K, HW, stream))
