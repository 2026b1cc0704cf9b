// res_tail.hip — the tail of the output head's image branch on bf16 maps, fused (round 4):
//
//     w[c] * MaxPool2d(2,2)( LeakyReLU( BN2(x2) + BN3(x3) ) )                   out.py:60,70 over unet.py:201-214
//
// x2 = conv2's output, x3 = the 1x1 shortcut conv's output (both [B, C, H, W], raw, batch statistics already known).  Unfused this
// is bn_apply x 2 + add_act + maxpool (850 MB at 32 x 32 x 224 x 224: both normalised maps and their sum are written and read
// back) and, backward, maxpool_bwd + lrelu_bwd + (bn_bwd_partial + bn_bwd_apply) x 2 (1.5 GB).  Here:
//   forward : one pass reads x2, x3 and writes the pooled map (232 MB) — the pre-pool activation is never stored;
//   backward: the activation is RECOMPUTED from x2, x3 (which the BatchNorm backward needs anyway), twice:
//       pass A (grid C x S): per channel sums  sum d,  sum d xhat2,  sum d xhat3,  sum g maxpool(y)   (d = gradient of the sum
//               behind the LeakyReLU: non-zero at the first maximal element of each 2x2 window only)  -> partial rows, no atomics
//       pass B (grid planes x chunks): dx2, dx3 = BatchNorm backward of d through each norm; one workgroup per channel adds the
//               affine gradients and dw
// A thread owns 2 rows x 8 columns (two 16-byte loads per tensor) = four windows.
#include "common.h"
#include "../../include/cenet_hip.h"

struct ResTailArgs {
  const bf16_t *x2, *x3;        // [B, C, H, W]
  const float *mean2, *var2, *gamma2, *beta2;
  const float *mean3, *var3, *gamma3, *beta3;
  const float* w;               // [C] scale behind the pool
  float eps2, eps3, slope;
  int B, C, H, W;
  // forward
  bf16_t* out;                  // [B, C, H/2, W/2]
  // backward
  const bf16_t* g;              // [B, C, H/2, W/2]
  bf16_t *dx2, *dx3;
  float* part;                  // [C][S][4]
  int S;
  float *dgamma2, *dbeta2, *dgamma3, *dbeta3, *dw;  // += (may be null)
  // IMG form: x3 = w3[c] * img is not materialised.  img [B, 1, H, W]; its batch mean / variance (one channel); BN3's statistics
  // follow analytically (mean3 = w3 mean, var3 = w3^2 var) and its running statistics are updated by the forward kernel
  const bf16_t* img;
  const float *img_mean, *img_var, *w3;
  float *rmean3, *rvar3, *dw3;
  long* nbt3;
  float mom3;
};

struct ResTailCh {
  float a2, c2, a3, c3;   // y = lrelu(a2 x2 + c2 + a3 x3 + c3)
  float m2, r2, m3, r3;   // xhat_k = (x_k - m_k) r_k
};
template <bool IMG>
__device__ __forceinline__ ResTailCh res_tail_channel(const ResTailArgs& a, int c) {
  ResTailCh k;
  k.m2 = a.mean2[c];
  k.r2 = rsqrtf(a.var2[c] + a.eps2);
  k.a2 = a.gamma2[c] * k.r2;
  k.c2 = a.beta2[c] - k.a2 * k.m2;
  if (IMG) {
    // BN3(w3 x) = gamma3 w3 (x - mx) / sqrt(w3^2 vx + eps) + beta3: an affine map of the image.  The "x3" the kernels read is the
    // image itself; r3 = 1, so the third partial sum is T = sum d (x - mx), from which the parameter gradients follow (pass B)
    const float w3 = a.w3[c], mx = a.img_mean[0], vx = a.img_var[0];
    k.m3 = mx;
    k.r3 = 1.f;
    k.a3 = a.gamma3[c] * w3 * rsqrtf(w3 * w3 * vx + a.eps3);
    k.c3 = a.beta3[c] - k.a3 * mx;
  } else {
    k.m3 = a.mean3[c];
    k.r3 = rsqrtf(a.var3[c] + a.eps3);
    k.a3 = a.gamma3[c] * k.r3;
    k.c3 = a.beta3[c] - k.a3 * k.m3;
  }
  return k;
}
// the four windows of a 2 x 8 block: y of the window's four elements (t = 0, 1: upper row, 2, 3: lower row), maximum and the index of
// its FIRST occurrence in that order (PyTorch's tie rule, as maxpool2_bwd_kernel)
__device__ __forceinline__ void res_tail_windows(const ResTailCh& k, float slope, const float (&u2)[8], const float (&l2)[8],
                                                 const float (&u3)[8], const float (&l3)[8], float (&mv)[4], int (&am)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float y[4];
    y[0] = k.a2 * u2[2 * j] + k.c2 + (k.a3 * u3[2 * j] + k.c3);
    y[1] = k.a2 * u2[2 * j + 1] + k.c2 + (k.a3 * u3[2 * j + 1] + k.c3);
    y[2] = k.a2 * l2[2 * j] + k.c2 + (k.a3 * l3[2 * j] + k.c3);
    y[3] = k.a2 * l2[2 * j + 1] + k.c2 + (k.a3 * l3[2 * j + 1] + k.c3);
#pragma unroll
    for (int t = 0; t < 4; ++t) y[t] = y[t] > 0.f ? y[t] : slope * y[t];
    mv[j] = y[0];
    am[j] = 0;
#pragma unroll
    for (int t = 1; t < 4; ++t)
      if (y[t] > mv[j]) {
        mv[j] = y[t];
        am[j] = t;
      }
  }
}

template <bool IMG>
__global__ __launch_bounds__(256) void res_tail_fwd_kernel(ResTailArgs a) {
  const int bc = blockIdx.x, c = bc % a.C;
  const ResTailCh k = res_tail_channel<IMG>(a, c);
  const float wc = a.w[c];
  const int Wq = a.W / 8, Ho = a.H / 2, Wo = a.W / 2;
  const bf16_t* p2 = a.x2 + (long)bc * a.H * a.W;
  const bf16_t* p3 = IMG ? a.img + (long)(bc / a.C) * a.H * a.W : a.x3 + (long)bc * a.H * a.W;
  if (IMG && bc < a.C && blockIdx.y == 0 && threadIdx.x == 0 && a.rmean3) {
    // BatchNorm3's running statistics (torch: unbiased variance) from the image's, and its batch counter
    const float w3 = a.w3[c], n = (float)a.B * a.H * a.W;
    a.rmean3[c] = (1.f - a.mom3) * a.rmean3[c] + a.mom3 * (w3 * a.img_mean[0]);
    a.rvar3[c] = (1.f - a.mom3) * a.rvar3[c] + a.mom3 * (w3 * w3 * a.img_var[0] * (n / (n - 1.f)));
    if (c == 0 && a.nbt3) a.nbt3[0] += 1;
  }
  bf16_t* po = a.out + (long)bc * Ho * Wo;
  for (int q = blockIdx.y * 256 + threadIdx.x; q < Ho * Wq; q += gridDim.y * 256) {
    const int oy = q / Wq, xq = q - oy * Wq;
    const long base = (long)(2 * oy) * a.W + 8 * xq;
    float u2[8], l2[8], u3[8], l3[8], mv[4];
    int am[4];
    ldv<8>(u2, p2 + base);
    ldv<8>(l2, p2 + base + a.W);
    ldv<8>(u3, p3 + base);
    ldv<8>(l3, p3 + base + a.W);
    res_tail_windows(k, a.slope, u2, l2, u3, l3, mv, am);
#pragma unroll
    for (int j = 0; j < 4; ++j) mv[j] *= wc;
    stv<4>(po + (long)oy * Wo + 4 * xq, mv);
  }
}

// pass A: partial sums of channel c over slice s of its B * (H/2) * (W/8) blocks
template <bool IMG>
__global__ __launch_bounds__(256) void res_tail_bwd_partial_kernel(ResTailArgs a) {
  __shared__ float red[16];
  const int c = blockIdx.x, s = blockIdx.y;
  const ResTailCh k = res_tail_channel<IMG>(a, c);
  const float wc = a.w[c];
  const int Wq = a.W / 8, Ho = a.H / 2, Wo = a.W / 2;
  const long per = (long)Ho * Wq, total = per * a.B;
  float sd = 0.f, s2 = 0.f, s3 = 0.f, sw = 0.f;
  for (long q = (long)s * 256 + threadIdx.x; q < total; q += (long)a.S * 256) {
    const int b = (int)(q / per);
    const int r = (int)(q - (long)b * per);
    const int oy = r / Wq, xq = r - oy * Wq;
    const long plane = (long)b * a.C + c;
    const long base = plane * a.H * a.W + (long)(2 * oy) * a.W + 8 * xq;
    const long base3 = IMG ? (long)b * a.H * a.W + (long)(2 * oy) * a.W + 8 * xq : base;
    const bf16_t* x3p = IMG ? a.img : a.x3;
    float u2[8], l2[8], u3[8], l3[8], mv[4], gv[4];
    int am[4];
    ldv<8>(u2, a.x2 + base);
    ldv<8>(l2, a.x2 + base + a.W);
    ldv<8>(u3, x3p + base3);
    ldv<8>(l3, x3p + base3 + a.W);
    ldv<4>(gv, a.g + plane * Ho * Wo + (long)oy * Wo + 4 * xq);
    res_tail_windows(k, a.slope, u2, l2, u3, l3, mv, am);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sw += gv[j] * mv[j];
      const float d = gv[j] * wc * (mv[j] > 0.f ? 1.f : a.slope);
      // the window's maximal element (static indices: the loaded rows stay in registers)
      float v2 = u2[2 * j], v3 = u3[2 * j];
      if (am[j] == 1) v2 = u2[2 * j + 1], v3 = u3[2 * j + 1];
      if (am[j] == 2) v2 = l2[2 * j], v3 = l3[2 * j];
      if (am[j] == 3) v2 = l2[2 * j + 1], v3 = l3[2 * j + 1];
      sd += d;
      s2 += d * ((v2 - k.m2) * k.r2);
      s3 += d * ((v3 - k.m3) * k.r3);
    }
  }
  sd = block_sum(sd, red);
  s2 = block_sum(s2, red);
  s3 = block_sum(s3, red);
  sw = block_sum(sw, red);
  if (threadIdx.x == 0) {
    float* p = a.part + ((long)c * a.S + s) * 4;
    p[0] = sd, p[1] = s2, p[2] = s3, p[3] = sw;
  }
}

// pass B: dx2, dx3; workgroup (plane of image 0, chunk 0) of every channel adds the parameter gradients
template <bool IMG>
__global__ __launch_bounds__(256) void res_tail_bwd_apply_kernel(ResTailArgs a) {
  __shared__ float red[16];
  const int bc = blockIdx.x, b = bc / a.C, c = bc - b * a.C;
  const ResTailCh k = res_tail_channel<IMG>(a, c);
  const float wc = a.w[c];
  float p0 = 0.f, p1 = 0.f, p2s = 0.f, p3s = 0.f;
  for (int i = threadIdx.x; i < a.S; i += 256) {
    const float* p = a.part + ((long)c * a.S + i) * 4;
    p0 += p[0], p1 += p[1], p2s += p[2], p3s += p[3];
  }
  const float sd = block_sum(p0, red), s2 = block_sum(p1, red), s3 = block_sum(p2s, red), sw = block_sum(p3s, red);
  const float n = (float)a.B * a.H * a.W;
  const float m1 = sd / n, m22 = s2 / n, m23 = s3 / n;
  if (b == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    if (a.dgamma2) atomicAdd(&a.dgamma2[c], s2);
    if (a.dbeta2) atomicAdd(&a.dbeta2[c], sd);
    if (a.dbeta3) atomicAdd(&a.dbeta3[c], sd);
    if (a.dw) atomicAdd(&a.dw[c], sw);
    if (IMG) {
      // s3 = T = sum d (x - mx).  xhat3 = w3 (x - mx) / s,  s = sqrt(w3^2 vx + eps):  dgamma3 = T w3 / s;  the shortcut conv's weight
      // only reaches the output through eps (BatchNorm is invariant to the scale of its input):  dw3 = gamma3 eps / s^3 * T
      const float w3 = a.w3[c], rs = rsqrtf(w3 * w3 * a.img_var[0] + a.eps3);
      if (a.dgamma3) atomicAdd(&a.dgamma3[c], s3 * w3 * rs);
      if (a.dw3) atomicAdd(&a.dw3[c], a.gamma3[c] * a.eps3 * rs * rs * rs * s3);
    } else if (a.dgamma3) {
      atomicAdd(&a.dgamma3[c], s3);
    }
  }
  const int Wq = a.W / 8, Ho = a.H / 2, Wo = a.W / 2;
  const long pb = (long)bc * a.H * a.W;
  const bf16_t* gp = a.g + (long)bc * Ho * Wo;
  const bf16_t* x3p = IMG ? a.img + (long)b * a.H * a.W : a.x3 + pb;
  for (int q = blockIdx.y * 256 + threadIdx.x; q < Ho * Wq; q += gridDim.y * 256) {
    const int oy = q / Wq, xq = q - oy * Wq;
    const long off = (long)(2 * oy) * a.W + 8 * xq;
    const long base = pb + off;
    float u2[8], l2[8], u3[8], l3[8], mv[4], gv[4];
    int am[4];
    ldv<8>(u2, a.x2 + base);
    ldv<8>(l2, a.x2 + base + a.W);
    ldv<8>(u3, x3p + off);
    ldv<8>(l3, x3p + off + a.W);
    ldv<4>(gv, gp + (long)oy * Wo + 4 * xq);
    res_tail_windows(k, a.slope, u2, l2, u3, l3, mv, am);
    float du[8], dl[8];  // d at the 2 x 8 block (non-zero at the windows' maxima)
#pragma unroll
    for (int e = 0; e < 8; ++e) du[e] = dl[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = gv[j] * wc * (mv[j] > 0.f ? 1.f : a.slope);
      const int col = 2 * j + (am[j] & 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) {  // (static indices: the arrays stay in registers)
        if (e == col && !(am[j] & 2)) du[e] = d;
        if (e == col && (am[j] & 2)) dl[e] = d;
      }
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = k.a2 * (du[e] - m1 - (u2[e] - k.m2) * k.r2 * m22);
    stv<8>(a.dx2 + base, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = k.a2 * (dl[e] - m1 - (l2[e] - k.m2) * k.r2 * m22);
    stv<8>(a.dx2 + base + a.W, o);
    if (!IMG) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = k.a3 * (du[e] - m1 - (u3[e] - k.m3) * k.r3 * m23);
      stv<8>(a.dx3 + base, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = k.a3 * (dl[e] - m1 - (l3[e] - k.m3) * k.r3 * m23);
      stv<8>(a.dx3 + base + a.W, o);
    }
  }
}

static inline int res_tail_chunks(int blocks_per_plane) {
  int ch = (blocks_per_plane + 1023) / 1024;
  return ch < 1 ? 1 : (ch > 16 ? 16 : ch);
}
static int res_tail_check(const ResTailArgs& a) {
  if (!a.x2 || !a.mean2 || !a.var2 || !a.gamma2 || !a.beta2 || !a.gamma3 || !a.beta3 || !a.w) return CENET_EINVAL;
  if (a.img ? (!a.img_mean || !a.img_var || !a.w3) : (!a.x3 || !a.mean3 || !a.var3)) return CENET_EINVAL;
  if (a.B <= 0 || a.C <= 0 || a.H <= 0 || a.W <= 0) return CENET_EINVAL;
  if ((a.H & 1) || (a.W & 7) || ((((uintptr_t)a.x2 | (uintptr_t)a.x3 | (uintptr_t)a.img) & 15) != 0)) return CENET_EUNSUPPORTED;
  return CENET_OK;
}
extern "C" int cenet_res_tail_supported(int H, int W) { return H > 0 && W > 0 && (H & 1) == 0 && (W & 7) == 0; }

/* out[b, c] = w[c] * MaxPool2x2(LeakyReLU(BN(x2; mean2, var2, gamma2, beta2) + BN(x3; ...))), one launch */
extern "C" int cenet_res_tail_fwd_bf16(const bf16_t* x2, const bf16_t* x3, const float* mean2, const float* var2, const float* gamma2,
                                       const float* beta2, float eps2, const float* mean3, const float* var3, const float* gamma3,
                                       const float* beta3, float eps3, const float* w, float slope, bf16_t* out, int B, int C, int H,
                                       int W, hipStream_t stream) {
  ResTailArgs a = {};
  a.x2 = x2; a.x3 = x3; a.mean2 = mean2; a.var2 = var2; a.gamma2 = gamma2; a.beta2 = beta2; a.eps2 = eps2;
  a.mean3 = mean3; a.var3 = var3; a.gamma3 = gamma3; a.beta3 = beta3; a.eps3 = eps3; a.w = w; a.slope = slope;
  a.out = out; a.B = B; a.C = C; a.H = H; a.W = W;
  const int rc = res_tail_check(a);
  if (rc != CENET_OK) return rc;
  if (!out || (((uintptr_t)out) & 7) != 0) return CENET_EINVAL;
  CENET_LAUNCH(res_tail_fwd_kernel<false>, dim3(B * C, res_tail_chunks((H / 2) * (W / 8))), dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

/* The same with the shortcut branch NOT materialised (one-channel network input, unet.py conv3 = 1x1 conv 1 -> C): x3 = w3[c] * img, so
 * BN3(x3) is an affine map of the image whose coefficients follow from the image's batch mean / variance (img_mean, img_var: one
 * float each, from cenet_bn_stats_* on the [B, 1, H, W] image).  The kernel also updates BatchNorm3's running statistics and batch
 * counter (rmean3 / rvar3 / nbt3, momentum mom3; NULL: not wanted). */
extern "C" int cenet_res_tail_img_fwd_bf16(const bf16_t* x2, const bf16_t* img, const float* mean2, const float* var2,
                                           const float* gamma2, const float* beta2, float eps2, const float* img_mean,
                                           const float* img_var, const float* w3, const float* gamma3, const float* beta3, float eps3,
                                           float* rmean3, float* rvar3, long* nbt3, float mom3, const float* w, float slope,
                                           bf16_t* out, int B, int C, int H, int W, hipStream_t stream) {
  ResTailArgs a = {};
  a.x2 = x2; a.img = img; a.mean2 = mean2; a.var2 = var2; a.gamma2 = gamma2; a.beta2 = beta2; a.eps2 = eps2;
  a.img_mean = img_mean; a.img_var = img_var; a.w3 = w3; a.gamma3 = gamma3; a.beta3 = beta3; a.eps3 = eps3;
  a.rmean3 = rmean3; a.rvar3 = rvar3; a.nbt3 = nbt3; a.mom3 = mom3; a.w = w; a.slope = slope;
  a.out = out; a.B = B; a.C = C; a.H = H; a.W = W;
  const int rc = res_tail_check(a);
  if (rc != CENET_OK) return rc;
  if (!out || (((uintptr_t)out) & 7) != 0 || (rmean3 != nullptr) != (rvar3 != nullptr)) return CENET_EINVAL;
  CENET_LAUNCH(res_tail_fwd_kernel<true>, dim3(B * C, res_tail_chunks((H / 2) * (W / 8))), dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

/* floats of workspace the backward needs */
extern "C" long cenet_res_tail_bwd_ws_floats(int C) { return (long)C * 32 * 4; }

/* backward of cenet_res_tail_fwd_bf16 from x2, x3 and the batch statistics (training-mode BatchNorm): dx2, dx3 written;
 * dgamma / dbeta of both norms and dw ADDED into (any may be NULL).  Two launches. */
extern "C" int cenet_res_tail_bwd_bf16(const bf16_t* g, const bf16_t* x2, const bf16_t* x3, const float* mean2, const float* var2,
                                       const float* gamma2, const float* beta2, float eps2, const float* mean3, const float* var3,
                                       const float* gamma3, const float* beta3, float eps3, const float* w, float slope, bf16_t* dx2,
                                       bf16_t* dx3, float* dgamma2_acc, float* dbeta2_acc, float* dgamma3_acc, float* dbeta3_acc,
                                       float* dw_acc, float* ws, int B, int C, int H, int W, hipStream_t stream) {
  ResTailArgs a = {};
  a.x2 = x2; a.x3 = x3; a.mean2 = mean2; a.var2 = var2; a.gamma2 = gamma2; a.beta2 = beta2; a.eps2 = eps2;
  a.mean3 = mean3; a.var3 = var3; a.gamma3 = gamma3; a.beta3 = beta3; a.eps3 = eps3; a.w = w; a.slope = slope;
  a.g = g; a.dx2 = dx2; a.dx3 = dx3; a.part = ws; a.S = 32;
  a.dgamma2 = dgamma2_acc; a.dbeta2 = dbeta2_acc; a.dgamma3 = dgamma3_acc; a.dbeta3 = dbeta3_acc; a.dw = dw_acc;
  a.B = B; a.C = C; a.H = H; a.W = W;
  const int rc = res_tail_check(a);
  if (rc != CENET_OK) return rc;
  if (!g || !dx2 || !dx3 || !ws) return CENET_EINVAL;
  if (((((uintptr_t)dx2 | (uintptr_t)dx3) & 15) != 0) || (((uintptr_t)g) & 7) != 0) return CENET_EUNSUPPORTED;
  CENET_LAUNCH(res_tail_bwd_partial_kernel<false>, dim3(C, a.S), dim3(256), stream, a);
  CENET_LAUNCH(res_tail_bwd_apply_kernel<false>, dim3(B * C, res_tail_chunks((H / 2) * (W / 8))), dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

/* backward of cenet_res_tail_img_fwd_bf16: dx2 written; dgamma2 / dbeta2 / dgamma3 / dbeta3 / dw3 (the shortcut conv's weight
 * gradient: through eps only, BatchNorm being invariant to the scale of its input) / dw ADDED into (any may be NULL) */
extern "C" int cenet_res_tail_img_bwd_bf16(const bf16_t* g, const bf16_t* x2, const bf16_t* img, const float* mean2, const float* var2,
                                           const float* gamma2, const float* beta2, float eps2, const float* img_mean,
                                           const float* img_var, const float* w3, const float* gamma3, const float* beta3, float eps3,
                                           const float* w, float slope, bf16_t* dx2, float* dgamma2_acc, float* dbeta2_acc,
                                           float* dgamma3_acc, float* dbeta3_acc, float* dw3_acc, float* dw_acc, float* ws, int B,
                                           int C, int H, int W, hipStream_t stream) {
  ResTailArgs a = {};
  a.x2 = x2; a.img = img; a.mean2 = mean2; a.var2 = var2; a.gamma2 = gamma2; a.beta2 = beta2; a.eps2 = eps2;
  a.img_mean = img_mean; a.img_var = img_var; a.w3 = w3; a.gamma3 = gamma3; a.beta3 = beta3; a.eps3 = eps3; a.w = w; a.slope = slope;
  a.g = g; a.dx2 = dx2; a.part = ws; a.S = 32;
  a.dgamma2 = dgamma2_acc; a.dbeta2 = dbeta2_acc; a.dgamma3 = dgamma3_acc; a.dbeta3 = dbeta3_acc; a.dw3 = dw3_acc; a.dw = dw_acc;
  a.B = B; a.C = C; a.H = H; a.W = W;
  const int rc = res_tail_check(a);
  if (rc != CENET_OK) return rc;
  if (!g || !dx2 || !ws) return CENET_EINVAL;
  if (((((uintptr_t)dx2) & 15) != 0) || (((uintptr_t)g) & 7) != 0) return CENET_EUNSUPPORTED;
  CENET_LAUNCH(res_tail_bwd_partial_kernel<true>, dim3(C, a.S), dim3(256), stream, a);
  CENET_LAUNCH(res_tail_bwd_apply_kernel<true>, dim3(B * C, res_tail_chunks((H / 2) * (W / 8))), dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
