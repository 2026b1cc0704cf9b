// conv_c1.hip — stride-1 same-padded k x k convolution of a ONE-channel bf16 image into Cout <= 32 channels, forward and
// weight gradient (throughput mode).  These are the two convolutions of the output head's residual block that read the
// network input (out.py:41-44, unet.py:156-197: conv1 5x5 1->32 and the 1x1 shortcut conv3 1->32 at 224x224).  As
// implicit GEMMs their contraction is 25 (or 1) long and 32 wide — 105 / 158 us forward and 376 / 144 us weight gradient
// at B = 32 for ~100 MB of traffic each; here they are what they are: VALU stencils bound by the write (forward) or the
// read (weight gradient) of the 32-channel tensor.  The input image needs no gradient.
//
//   forward : workgroup = 8 rows x 128 columns of one image; the (8+k-1) x (128+k-1) halo tile sits in LDS as floats; a
//             thread takes 4 consecutive pixels, keeps their k x (k+3) window in registers and walks the output channels
//             (weights through scalar loads: the channel index is wave-uniform); 8-byte stores, 512 B per wave and plane.
//   wgrad   : same tile walk over TR rows per workgroup; thread = (4 output channels, 4 consecutive pixels): 4 * k*k
//             accumulators; lanes of a 32-lane half share their channel group -> xor-shuffle fold, LDS across waves, one set of
//             float atomics per workgroup.
#include "common.h"
#include <cstdlib>
#include "../../include/cenet_hip.h"

struct ConvC1Args {
  const bf16_t* x;   // [B, 1, H, W]
  const bf16_t* dy;  // [B, Cout, H, W] (wgrad)
  const float* w;    // [Cout, 1, k, k]
  bf16_t* y;         // [B, Cout, H, W] (forward)
  float* dw;         // [Cout, 1, k, k] accumulator (wgrad)
  int B, Cout, H, W, rows_per_wg;
};

template <int KS>
__global__ __launch_bounds__(256) void conv_c1_fwd_kernel(ConvC1Args a) {
  constexpr int PAD = KS / 2, TR = 8, TC = 128, LW = TC + 2 * PAD, LH = TR + 2 * PAD, WIN = KS + 3;
  __shared__ float tile[LH * LW];
  const cenet_bid bid = cenet_xcd_block();
  const int x0 = bid.x * TC, y0 = bid.y * TR, b = bid.z;
  const bf16_t* xb = a.x + (long)b * a.H * a.W;
  for (int i = threadIdx.x; i < LH * LW; i += 256) {
    const int r = i / LW, c = i - r * LW;
    const int iy = y0 - PAD + r, ix = x0 - PAD + c;
    tile[i] = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? cenet_bf2f(xb[(long)iy * a.W + ix]) : 0.f;
  }
  __syncthreads();
  const int row = threadIdx.x >> 5, q = threadIdx.x & 31;  // 8 rows x 32 pixel quads
  const int oy = y0 + row, ox = x0 + 4 * q;
  if (oy >= a.H || ox >= a.W) return;
  float win[KS][WIN];
#pragma unroll
  for (int r = 0; r < KS; ++r)
#pragma unroll
    for (int c = 0; c < WIN; ++c) win[r][c] = tile[(row + r) * LW + 4 * q + c];
  const long HW = (long)a.H * a.W;
  bf16_t* yb = a.y + (long)b * a.Cout * HW + (long)oy * a.W + ox;
  const bool full = ox + 3 < a.W && ((a.W & 3) == 0);
  for (int co = 0; co < a.Cout; ++co) {
    const float* wc = a.w + co * KS * KS;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < KS; ++r)
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const float wv = wc[r * KS + c];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += wv * win[r][c + e];
      }
    if (full) {
      st4v(yb + co * HW, acc);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ox + e < a.W) stf(yb + co * HW + e, acc[e]);
    }
  }
}

template <int KS>
__global__ __launch_bounds__(256) void conv_c1_wgrad_kernel(ConvC1Args a) {
  constexpr int PAD = KS / 2, TC = 128, LW = TC + 2 * PAD, WIN = KS + 3, KK = KS * KS;
  constexpr int TRM = 8;  // rows staged at a time
  __shared__ float tile[(TRM + 2 * PAD) * LW];
  __shared__ float red[8][4 * KK];
  const cenet_bid bid = cenet_xcd_block();
  const int x0 = bid.x * TC, b = bid.z;
  const int ybeg = bid.y * a.rows_per_wg, yend = ybeg + a.rows_per_wg < a.H ? ybeg + a.rows_per_wg : a.H;
  const bf16_t* xb = a.x + (long)b * a.H * a.W;
  const long HW = (long)a.H * a.W;
  const int cg = threadIdx.x >> 5, q = threadIdx.x & 31;  // 8 channel groups (4 channels each) x 32 pixel quads
  const int ox = x0 + 4 * q;
  const bool cok = 4 * cg < a.Cout, full = ox + 3 < a.W && ((a.W & 3) == 0);
  float acc[4][KK];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[c][t] = 0.f;
  for (int y0 = ybeg; y0 < yend; y0 += TRM) {
    __syncthreads();
    for (int i = threadIdx.x; i < (TRM + 2 * PAD) * LW; i += 256) {
      const int r = i / LW, c = i - r * LW;
      const int iy = y0 - PAD + r, ix = x0 - PAD + c;
      tile[i] = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? cenet_bf2f(xb[(long)iy * a.W + ix]) : 0.f;
    }
    __syncthreads();
    if (!cok || ox >= a.W) continue;
    for (int row = 0; row < TRM && y0 + row < yend; ++row) {
      float g[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16_t* gp = a.dy + ((long)b * a.Cout + 4 * cg + c) * HW + (long)(y0 + row) * a.W + ox;
        if (4 * cg + c < a.Cout) {
          if (full) {
            ld4v(g[c], gp);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[c][e] = ox + e < a.W ? ldf(gp + e) : 0.f;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) g[c][e] = 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < KS; ++r) {
        float wr[WIN];
#pragma unroll
        for (int c = 0; c < WIN; ++c) wr[c] = tile[(row + r) * LW + 4 * q + c];
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[c][r * KS + kx] += g[c][e] * wr[kx + e];
      }
    }
  }
  // fold the 32 pixel-quad lanes of each channel group, then one atomic per (channel, tap) and workgroup
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      float v = acc[c][t];
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
      if ((lane & 31) == 0) red[cg][c * KK + t] = v;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 4 * KK; i += 256) {
    const int g8 = i / (4 * KK), rem = i - g8 * 4 * KK, c = rem / KK, t = rem - c * KK;
    const int co = 4 * g8 + c;
    if (co < a.Cout) atomicAdd(&a.dw[co * KK + t], red[g8][rem]);
  }
}

extern "C" int cenet_conv_c1_supported(int Cin, int Cout, int k, int stride, int pad) {
  return Cin == 1 && Cout >= 1 && Cout <= 32 && (k == 1 || k == 3 || k == 5) && stride == 1 && pad == k / 2;
}

extern "C" int cenet_conv_c1_fwd_bf16(const bf16_t* x, const float* w, bf16_t* y, int B, int Cout, int H, int W, int k,
                                      hipStream_t stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  if (!cenet_conv_c1_supported(1, Cout, k, 1, k / 2) || B > 65535) return CENET_EUNSUPPORTED;
  ConvC1Args a;
  a.x = x; a.dy = nullptr; a.w = w; a.y = y; a.dw = nullptr; a.B = B; a.Cout = Cout; a.H = H; a.W = W; a.rows_per_wg = 8;
  const dim3 grid(cdiv(W, 128), cdiv(H, 8), B);
  if (k == 1) CENET_LAUNCH((conv_c1_fwd_kernel<1>), grid, dim3(256), stream, a);
  else if (k == 3) CENET_LAUNCH((conv_c1_fwd_kernel<3>), grid, dim3(256), stream, a);
  else CENET_LAUNCH((conv_c1_fwd_kernel<5>), grid, dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_conv_c1_wgrad_bf16(const bf16_t* x, const bf16_t* dy, float* dw_acc, int B, int Cout, int H, int W, int k,
                                        hipStream_t stream) {
  if (!x || !dy || !dw_acc || B <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  if (!cenet_conv_c1_supported(1, Cout, k, 1, k / 2) || B > 65535) return CENET_EUNSUPPORTED;
  ConvC1Args a;
  a.x = x; a.dy = dy; a.w = nullptr; a.y = nullptr; a.dw = dw_acc; a.B = B; a.Cout = Cout; a.H = H; a.W = W;
  // ~1024 workgroups: enough rows per workgroup to amortise the fold (hundreds of shuffles) and keep the atomics few
  static const char* e = getenv("CENET_C1_WGS");  // measurement aid
  const long target = e ? atol(e) : (k >= 3 ? 512 : 1024);  // measured at 224x224, B = 32: 5x5 105 us at 512 (128 at 1024)
  int rows = 8;
  while (rows < H && (long)cdiv(W, 128) * cdiv(H, rows) * B > target) rows += 8;
  a.rows_per_wg = rows;
  const dim3 grid(cdiv(W, 128), cdiv(H, rows), B);
  if (k == 1) CENET_LAUNCH((conv_c1_wgrad_kernel<1>), grid, dim3(256), stream, a);
  else if (k == 3) CENET_LAUNCH((conv_c1_wgrad_kernel<3>), grid, dim3(256), stream, a);
  else CENET_LAUNCH((conv_c1_wgrad_kernel<5>), grid, dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Square 1x1 convolutions over a handful of channels (bf16, no bias): the pointwise convs of the three dilated SepConvBN
// branches (cfam.py:208-212; g = 5C/16 = 20 / 40 channels at the 56x56 / 28x28 decoder levels, as `groups` independent G x G
// products on the channel groups of one tensor) and the conv of the pooled branch (cfam.py:213-219; C/16 = 4 ... 32 channels on
// 7x7 maps).  As GEMMs these are M = K = G <= 40 problems: 21-51 us per launch on 32x256 tiles that are 60-90 % padding, for
// 6-12 MB of traffic.  Here a thread owns two neighbouring pixels of one (image, group): it keeps their G inputs in registers,
// walks the G outputs with the weight row read from LDS (every lane the same address: a broadcast), and stores 4 bytes per
// output plane.  TRANS: the data gradient (the same products with the transposed weights).  HBM-bound: 2 * G * HW * 2 bytes
// per (image, group).
//   x, y [B, groups * G, HW] (group j = channels j*G .. j*G+G-1), W [groups, G, G] bf16 (the arena's shadow of the fp32 weights).
// ---------------------------------------------------------------------------------------------------------------------------
template <int G, bool TRANS>
__global__ __launch_bounds__(256) void pw_small_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                      bf16_t* __restrict__ y, int groups, int HW) {
  __shared__ float w[G * G];  // w[o][i]
  const int bj = blockIdx.y, j = bj % groups;
  for (int t = threadIdx.x; t < G * G; t += 256) {
    const int o = t / G, i = t - o * G;
    w[t] = cenet_bf2f(W[(long)j * G * G + (TRANS ? i * G + o : o * G + i)]);
  }
  __syncthreads();
  const int p = 2 * (blockIdx.x * 256 + threadIdx.x);
  if (p >= HW) return;
  const bf16_t* xb = x + (long)bj * G * HW + p;
  bf16_t* yb = y + (long)bj * G * HW + p;
  const bool pair = p + 1 < HW && (HW & 1) == 0;  // (odd planes: 2-byte accesses, the last pixel alone)
  float x0[G], x1[G];
#pragma unroll
  for (int i = 0; i < G; ++i) {
    if (pair) {
      unsigned u;
      memcpy(&u, xb + (long)i * HW, 4);
      x0[i] = cenet_bf2f(u & 0xFFFFu);
      x1[i] = cenet_bf2f(u >> 16);
    } else {
      x0[i] = cenet_bf2f(xb[(long)i * HW]);
      x1[i] = p + 1 < HW ? cenet_bf2f(xb[(long)i * HW + 1]) : 0.f;
    }
  }
#pragma unroll 2
  for (int o = 0; o < G; ++o) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const float wv = w[o * G + i];
      a0 += wv * x0[i];
      a1 += wv * x1[i];
    }
    if (pair) {
      const unsigned u = cenet_pack_bf2(a0, a1);
      memcpy(yb + (long)o * HW, &u, 4);
    } else {
      stf(yb + (long)o * HW, a0);
      if (p + 1 < HW) stf(yb + (long)o * HW + 1, a1);
    }
  }
}

extern "C" int cenet_pw_small_supported(int G) { return G == 4 || G == 8 || G == 20 || G == 32 || G == 40; }

extern "C" int cenet_pw_small_bf16(const bf16_t* x, const bf16_t* W, bf16_t* y, int B, int groups, int G, long HW, int transpose,
                                   hipStream_t stream) {
  if (!x || !W || !y || B <= 0 || groups <= 0 || HW <= 0 || HW > 0x7FFFFFFF) return CENET_EINVAL;
  if (!cenet_pw_small_supported(G) || (long)B * groups > 65535) return CENET_EUNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)y) & 3) != 0) return CENET_EINVAL;
  const dim3 grid(cdiv((int)((HW + 1) / 2), 256), B * groups);
#define PW_GO(G_)                                                                                              \
  do {                                                                                                         \
    if (transpose) CENET_LAUNCH((pw_small_kernel<G_, true>), grid, dim3(256), stream, x, W, y, groups, (int)HW); \
    else CENET_LAUNCH((pw_small_kernel<G_, false>), grid, dim3(256), stream, x, W, y, groups, (int)HW);         \
  } while (0)
  if (G == 4) PW_GO(4);
  else if (G == 8) PW_GO(8);
  else if (G == 20) PW_GO(20);
  else if (G == 32) PW_GO(32);
  else PW_GO(40);
#undef PW_GO
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
