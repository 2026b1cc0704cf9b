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

// ---------------------------------------------------------------------------------------------------------------------------
// 5x5, 1 -> 32 channels on the matrix cores (round 5).  The VALU stencils above spend 25 multiply-adds per output value: 1.3 G
// of them per pass at 224x224, B = 32 — 66 us forward, 122 us weight gradient for 103 MB of traffic.  As MFMA products with the
// 25 taps (padded to 32) as one of the dimensions:
//   forward  out[px][co] = sum_t X[px + d_t] W[co][t]:   A = 16 pixels x 32 taps (a lane gathers the 8 taps of ITS tap group from
//            the fp32 LDS halo tile: per-lane constant offsets), B = the weights (two 16-channel fragments, loaded once);
//            a lane receives 4 consecutive pixels of one channel -> 8-byte stores.
//   wgrad    dW[co][t] = sum_px dY[co][px] X[px + d_t]:   A = dY rows straight from HBM (8 consecutive pixels = one 16-byte load),
//            B = 32 pixels x 16 taps (a lane reads 8 consecutive tile floats at ITS tap's offset); four accumulator tiles per wave
//            over all its pixels, one LDS fold and one set of float atomics per workgroup.
// ~30 instructions per 16 pixels x 32 channels instead of ~200.  The tile holds the bf16 inputs widened to fp32, so re-packing
// two of them is a byte permute (exact).
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned c1_pack(float lo, float hi) {  // (both are bf16 values held as fp32: low halves are zero)
  return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xFFFF0000u);
}
#define C1M_TR 8
#define C1M_TC 128
#define C1M_LW (C1M_TC + 4)
// stages rows y0 - 2 .. y0 + NR + 1, columns x0 - 2 .. x0 + 129 of image xb into tile[(NR + 4) * C1M_LW] (zero outside the image)
template <int NR>
__device__ __forceinline__ void c1m_stage(const bf16_t* xb, int H, int W, int y0, int x0, float* tile) {
  constexpr int LH = NR + 4;
  const int c = threadIdx.x & 127, r0 = threadIdx.x >> 7;
  float v[LH / 2];
#pragma unroll
  for (int k = 0; k < LH / 2; ++k) {
    const int r = r0 + 2 * k, iy = y0 - 2 + r, ix = x0 - 2 + c;
    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
    v[k] = cenet_bf2f(xb[ok ? (long)iy * W + ix : 0]);
    if (!ok) v[k] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < LH / 2; ++k) tile[(r0 + 2 * k) * C1M_LW + c] = v[k];
  if (threadIdx.x < LH * 4) {  // the last four columns
    const int r = threadIdx.x >> 2, cc = 128 + (threadIdx.x & 3), iy = y0 - 2 + r, ix = x0 - 2 + cc;
    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
    const float t = cenet_bf2f(xb[ok ? (long)iy * W + ix : 0]);
    tile[r * C1M_LW + cc] = ok ? t : 0.f;
  }
}

// grid (ceil(W / 128), ceil(H / 8), B); W % 4 == 0
__global__ __launch_bounds__(256) void conv_c1_fwd_mfma5_kernel(ConvC1Args a) {
  __shared__ float tile[(C1M_TR + 4) * C1M_LW];
  const cenet_bid bid = cenet_xcd_block();
  const int x0 = bid.x * C1M_TC, y0 = bid.y * C1M_TR, b = bid.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  c1m_stage<C1M_TR>(a.x + (long)b * a.H * a.W, a.H, a.W, y0, x0, tile);
  // this lane's tap group: t = 8 fq + j -> tile offset (ky, kx); taps 25 .. 31 read offset 0 against a zero weight
  int toff[8];
  bf16x8 wf[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int t = 8 * fq + j, ky = t / 5, kx = t - 5 * ky;
    toff[j] = t < 25 ? ky * C1M_LW + kx : 0;
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    unsigned u[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int t0 = 8 * fq + 2 * h, co = 16 * ct + fr;
      const float w0 = (t0 < 25 && co < a.Cout) ? a.w[co * 25 + t0] : 0.f, w1 = (t0 + 1 < 25 && co < a.Cout) ? a.w[co * 25 + t0 + 1] : 0.f;
      u[h] = cenet_pack_bf2(w0, w1);
    }
    memcpy(&wf[ct], u, 16);
  }
  __syncthreads();
  const long HW = (long)a.H * a.W;
  bf16_t* yb = a.y + (long)b * a.Cout * HW;
#pragma unroll 1
  for (int rr = 0; rr < 2; ++rr) {
    const int row = 2 * wave + rr, oy = y0 + row;
    if (oy >= a.H) break;  // (wave-uniform)
#pragma unroll 2
    for (int g = 0; g < C1M_TC / 16; ++g) {
      const int xs = 16 * g;
      if (x0 + xs >= a.W) break;  // (wave-uniform; W % 16 == 0 for the fast path)
      const float* tp = tile + row * C1M_LW + xs + fr;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tp[toff[j]];
      unsigned u[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) u[h] = c1_pack(v[2 * h], v[2 * h + 1]);
      bf16x8 af;
      memcpy(&af, u, 16);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[ct], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const int co = 16 * ct + fr;
        if (co < a.Cout) {
          const float o[4] = {d[0], d[1], d[2], d[3]};
          st4v(yb + (long)co * HW + (long)oy * a.W + x0 + xs + 4 * fq, o);
        }
      }
    }
  }
}

// grid (ceil(W / 128), ceil(H / rows_per_wg), B); W % 32 == 0
__global__ __launch_bounds__(256) void conv_c1_wgrad_mfma5_kernel(ConvC1Args a) {
  __shared__ float tile[(C1M_TR + 4) * C1M_LW];
  __shared__ float red[4][4][256];
  const cenet_bid bid = cenet_xcd_block();
  const int x0 = bid.x * C1M_TC, b = bid.z;
  const int ybeg = bid.y * a.rows_per_wg, yend = ybeg + a.rows_per_wg < a.H ? ybeg + a.rows_per_wg : a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const long HW = (long)a.H * a.W;
  const bf16_t* xb = a.x + (long)b * HW;
  const int xs = 32 * wave;  // this wave's 32-pixel segment of the tile's columns
  const bool wok = x0 + xs < a.W;
  // B operand: tap t = 16 tt + fr -> tile offset of (ky, kx); taps >= 25 alias tap 0 (their columns are never written back)
  int toff[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int t = 16 * tt + fr, ky = t / 5, kx = t - 5 * ky;
    toff[tt] = t < 25 ? ky * C1M_LW + kx : 0;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) acc[ct][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // A operand: dY[co = 16 ct + fr][row][x0 + xs + 8 fq .. + 7]
  const bf16_t* gp[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int co = 16 * ct + fr < a.Cout ? 16 * ct + fr : 0;
    gp[ct] = a.dy + ((long)b * a.Cout + co) * HW + x0 + xs + 8 * fq;
  }
  for (int y0 = ybeg; y0 < yend; y0 += C1M_TR) {
    __syncthreads();
    c1m_stage<C1M_TR>(xb, a.H, a.W, y0, x0, tile);
    __syncthreads();
    if (!wok) continue;  // (wave-uniform)
    const int nr = yend - y0 < C1M_TR ? yend - y0 : C1M_TR;
    bf16x8 af[2], nx[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) memcpy(&af[ct], gp[ct] + (long)y0 * a.W, 16);
    for (int row = 0; row < nr; ++row) {
      const int rn = row + 1 < nr ? row + 1 : row;  // next row's dY in flight under this row's products
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) memcpy(&nx[ct], gp[ct] + (long)(y0 + rn) * a.W, 16);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const float* tp = tile + row * C1M_LW + xs + 8 * fq + toff[tt];
        unsigned u[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) u[h] = c1_pack(tp[2 * h], tp[2 * h + 1]);
        bf16x8 bfv;
        memcpy(&bfv, u, 16);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ct][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct], bfv, acc[ct][tt], 0, 0, 0);
      }
      af[0] = nx[0];
      af[1] = nx[1];
    }
  }
  // fold the four waves, then one atomic per (channel, tap): acc[ct][tt][r] = dW[co = 16 ct + 4 fq + r][t = 16 tt + fr]
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][ct * 2 + tt][lane * 4 + r] = acc[ct][tt][r];
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * 256; i += 256) {
    const int q = i >> 8, e = i & 255, ct = q >> 1, tt = q & 1, ln = e >> 2, r = e & 3;
    const int co = 16 * ct + 4 * (ln >> 4) + r, t = 16 * tt + (ln & 15);
    if (co < a.Cout && t < 25) atomicAdd(&a.dw[co * 25 + t], red[0][q][e] + red[1][q][e] + red[2][q][e] + red[3][q][e]);
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
  static const bool no_mfma = getenv("CENET_C1_NO_MFMA") != nullptr;  // measurement aid
  if (k == 5 && (W & 15) == 0 && !no_mfma && ((uintptr_t)y & 7) == 0) {
    CENET_LAUNCH(conv_c1_fwd_mfma5_kernel, grid, dim3(256), stream, a);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
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
  static const bool no_mfma = getenv("CENET_C1_NO_MFMA") != nullptr;  // measurement aid
  if (k == 5 && (W & 31) == 0 && !no_mfma && ((uintptr_t)dy & 15) == 0) {
    CENET_LAUNCH(conv_c1_wgrad_mfma5_kernel, grid, dim3(256), stream, a);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (k == 1) CENET_LAUNCH((conv_c1_wgrad_kernel<1>), grid, dim3(256), stream, a);
  else if (k == 3) CENET_LAUNCH((conv_c1_wgrad_kernel<3>), grid, dim3(256), stream, a);
  else CENET_LAUNCH((conv_c1_wgrad_kernel<5>), grid, dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Pixel-pair access for the thread-per-pixel 1x1 kernels below: a thread owns pixels p, p + 1 (p even) of a plane of HW bf16
// elements.  EVEN (HW even: every plane starts 4-byte aligned and p + 1 < HW always): one 4-byte access, no condition — the
// kernels branch ONCE on the parity of HW (wave-uniform), never per load (per-load conditions serialise the loads).
// ---------------------------------------------------------------------------------------------------------------------------
template <bool EVEN>
__device__ __forceinline__ void pw_ld2(const bf16_t* q, bool has1, float& a, float& b) {
  if (EVEN) {
    unsigned u;
    memcpy(&u, q, 4);
    a = cenet_bf2f(u & 0xFFFFu);
    b = cenet_bf2f(u >> 16);
  } else {
    a = cenet_bf2f(q[0]);
    b = has1 ? cenet_bf2f(q[1]) : 0.f;
  }
}
template <bool EVEN>
__device__ __forceinline__ void pw_st2(bf16_t* q, bool has1, float a, float b) {
  if (EVEN) {
    const unsigned u = cenet_pack_bf2(a, b);
    memcpy(q, &u, 4);
  } else {
    stf(q, a);
    if (has1) stf(q + 1, b);
  }
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
template <int G, bool EVEN>
__device__ __forceinline__ void pw_small_body(const bf16_t* xb, bf16_t* yb, const float* w, int HW, bool has1) {
  float x0[G], x1[G];
#pragma unroll
  for (int i = 0; i < G; ++i) pw_ld2<EVEN>(xb + (long)i * HW, has1, x0[i], x1[i]);
#pragma unroll 2
  for (int o = 0; o < G; ++o) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const float wv = w[o * G + i];
      a0 += wv * x0[i];
      a1 += wv * x1[i];
    }
    pw_st2<EVEN>(yb + (long)o * HW, has1, a0, a1);
  }
}
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
  if ((HW & 1) == 0) pw_small_body<G, true>(xb, yb, w, HW, true);
  else pw_small_body<G, false>(xb, yb, w, HW, p + 1 < HW);
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

// ---------------------------------------------------------------------------------------------------------------------------
// 1x1 convolution from 64 channels to a FEW (<= 16) output channels, bf16, with bias: the segmentation head's last layer
// (unet.py:200-217 UnetOutBlock; out.py:49: 64 -> num_classes at 112x112).  As a GEMM it is an M = 4 problem on 32-row tiles
// (80 us forward, 62 us weight gradient, 39 us data gradient at B = 32 for 51 MB of traffic).  Here:
//   forward : thread = two neighbouring pixels, their 64 inputs in registers, weights [CO][64] broadcast from LDS; reads the
//             64-channel tensor once (HBM-bound), writes CO planes.
//   dgrad   : thread = two pixels, their CO output gradients in registers (16 slots, zero weights beyond CO), walks the 64
//             input channels; writes the 64-channel tensor once.
//   wgrad   : workgroup = 1024 pixels of one image; wave w owns input channels 16w .. 16w+15, lane = pixel pair: CO x 16
//             accumulators per thread, xor-shuffle fold over the 64 lanes, one atomic per (o, i) and workgroup; wave 0 also
//             sums dY into the bias gradient.
// ---------------------------------------------------------------------------------------------------------------------------
#define PWF_CI 64
#define PWF_COMAX 16
template <bool EVEN>
__device__ __forceinline__ void pwf_fwd_body(const bf16_t* xb, bf16_t* yb, const float* w, const float* bsh, int CO, int HW,
                                             bool has1) {
  float x0[PWF_CI], x1[PWF_CI];
#pragma unroll
  for (int i = 0; i < PWF_CI; ++i) pw_ld2<EVEN>(xb + (long)i * HW, has1, x0[i], x1[i]);
  for (int o = 0; o < CO; ++o) {
    float a0 = bsh[o], a1 = bsh[o];
#pragma unroll
    for (int i = 0; i < PWF_CI; ++i) {
      const float wv = w[o * PWF_CI + i];
      a0 += wv * x0[i];
      a1 += wv * x1[i];
    }
    pw_st2<EVEN>(yb + (long)o * HW, has1, a0, a1);
  }
}
__global__ __launch_bounds__(256) void pw_fewout_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                           const float* __restrict__ bias, bf16_t* __restrict__ y, int CO, int HW) {
  __shared__ float w[PWF_COMAX * PWF_CI];
  __shared__ float bsh[PWF_COMAX];
  for (int t = threadIdx.x; t < CO * PWF_CI; t += 256) w[t] = cenet_bf2f(W[t]);
  if (threadIdx.x < CO) bsh[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const int b = blockIdx.y, p = 2 * (blockIdx.x * 256 + threadIdx.x);
  if (p >= HW) return;
  const bf16_t* xb = x + (long)b * PWF_CI * HW + p;
  bf16_t* yb = y + (long)b * CO * HW + p;
  if ((HW & 1) == 0) pwf_fwd_body<true>(xb, yb, w, bsh, CO, HW, true);
  else pwf_fwd_body<false>(xb, yb, w, bsh, CO, HW, p + 1 < HW);
}

// COT: compile-time bound of the output-channel loop (2, 4, 9, or 16 for anything else <= 16; rows >= CO hold zero weights)
template <int COT, bool EVEN>
__device__ __forceinline__ void pwf_dgrad_body(const bf16_t* gb, bf16_t* db, const float* w, int CO, int HW, bool has1) {
  float g0[COT], g1[COT];
#pragma unroll
  for (int o = 0; o < COT; ++o) {
    g0[o] = g1[o] = 0.f;
    if (o < CO) pw_ld2<EVEN>(gb + (long)o * HW, has1, g0[o], g1[o]);  // (CO is wave-uniform)
  }
#pragma unroll 4
  for (int i = 0; i < PWF_CI; ++i) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int o = 0; o < COT; ++o) {
      const float wv = w[o * PWF_CI + i];
      a0 += wv * g0[o];
      a1 += wv * g1[o];
    }
    pw_st2<EVEN>(db + (long)i * HW, has1, a0, a1);
  }
}
template <int COT>
__global__ __launch_bounds__(256) void pw_fewout_dgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ W,
                                                             bf16_t* __restrict__ dx, int CO, int HW) {
  __shared__ float w[COT * PWF_CI];  // rows >= CO zero
  for (int t = threadIdx.x; t < COT * PWF_CI; t += 256) w[t] = t < CO * PWF_CI ? cenet_bf2f(W[t]) : 0.f;
  __syncthreads();
  const int b = blockIdx.y, p = 2 * (blockIdx.x * 256 + threadIdx.x);
  if (p >= HW) return;
  const bf16_t* gb = dy + (long)b * CO * HW + p;
  bf16_t* db = dx + (long)b * PWF_CI * HW + p;
  if ((HW & 1) == 0) pwf_dgrad_body<COT, true>(gb, db, w, CO, HW, true);
  else pwf_dgrad_body<COT, false>(gb, db, w, CO, HW, p + 1 < HW);
}

template <int CO, bool EVEN>
__device__ __forceinline__ void pwf_wgrad_body(const bf16_t* xb, const bf16_t* gb, int HW, int p0, int p1, int lane,
                                               float (&acc)[CO][16], float (&accb)[CO]) {
  for (int p = p0 + 2 * lane; p < p1; p += 128) {
    const bool has1 = p + 1 < p1;
    float g0[CO], g1[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      pw_ld2<EVEN>(gb + (long)o * HW + p, has1, g0[o], g1[o]);
      accb[o] += g0[o] + g1[o];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float x0, x1;
      pw_ld2<EVEN>(xb + (long)i * HW + p, has1, x0, x1);
#pragma unroll
      for (int o = 0; o < CO; ++o) acc[o][i] += g0[o] * x0 + g1[o] * x1;
    }
  }
}
template <int CO>
__global__ __launch_bounds__(256) void pw_fewout_wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                             float* __restrict__ dW, float* __restrict__ dbias, int HW) {
  __shared__ float red[4][CO * 16 + CO];
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p0 = blockIdx.x * 1024, p1 = p0 + 1024 < HW ? p0 + 1024 : HW;
  const bf16_t* xb = x + (long)b * PWF_CI * HW + (long)(16 * wave) * HW;
  const bf16_t* gb = dy + (long)b * CO * HW;
  float acc[CO][16], accb[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    accb[o] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[o][i] = 0.f;
  }
  if ((HW & 1) == 0) pwf_wgrad_body<CO, true>(xb, gb, HW, p0, p1, lane, acc, accb);
  else pwf_wgrad_body<CO, false>(xb, gb, HW, p0, p1, lane, acc, accb);
  // fold the 64 lanes of each wave (xor shuffles), meet in LDS, one atomic per (o, i) and workgroup
  float vals[CO * 16 + CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) {
#pragma unroll
    for (int i = 0; i < 16; ++i) vals[o * 16 + i] = acc[o][i];
    vals[CO * 16 + o] = accb[o];
  }
#pragma unroll
  for (int k = 0; k < CO * 16 + CO; ++k) {
    float v = vals[k];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 4 * CO * 16; k += 256) {
    const int wv = k / (CO * 16), r = k - wv * (CO * 16), o = r / 16, i = r - o * 16;
    atomicAdd(&dW[o * PWF_CI + 16 * wv + i], red[wv][r]);
  }
  if (dbias && threadIdx.x < CO) atomicAdd(&dbias[threadIdx.x], red[0][CO * 16 + threadIdx.x]);
}

extern "C" int cenet_pw_fewout_supported(int Cin, int Cout) { return Cin == PWF_CI && Cout >= 1 && Cout <= PWF_COMAX; }
extern "C" int cenet_pw_fewout_wgrad_supported(int Cin, int Cout) { return Cin == PWF_CI && (Cout == 2 || Cout == 4 || Cout == 9); }

extern "C" int cenet_pw_fewout_fwd_bf16(const bf16_t* x, const bf16_t* W, const float* bias, bf16_t* y, int B, int Cin, int Cout,
                                        long HW, hipStream_t stream) {
  if (!x || !W || !y || B <= 0 || HW <= 0 || HW > 0x7FFFFFFF || B > 65535) return CENET_EINVAL;
  if (!cenet_pw_fewout_supported(Cin, Cout)) return CENET_EUNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)y) & 3) != 0) return CENET_EINVAL;
  CENET_LAUNCH(pw_fewout_fwd_kernel, dim3(cdiv((int)((HW + 1) / 2), 256), B), dim3(256), stream, x, W, bias, y, Cout, (int)HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_pw_fewout_dgrad_bf16(const bf16_t* dy, const bf16_t* W, bf16_t* dx, int B, int Cin, int Cout, long HW,
                                          hipStream_t stream) {
  if (!dy || !W || !dx || B <= 0 || HW <= 0 || HW > 0x7FFFFFFF || B > 65535) return CENET_EINVAL;
  if (!cenet_pw_fewout_supported(Cin, Cout)) return CENET_EUNSUPPORTED;
  if ((((uintptr_t)dy | (uintptr_t)dx) & 3) != 0) return CENET_EINVAL;
  const dim3 grid(cdiv((int)((HW + 1) / 2), 256), B);
  if (Cout <= 2) CENET_LAUNCH((pw_fewout_dgrad_kernel<2>), grid, dim3(256), stream, dy, W, dx, Cout, (int)HW);
  else if (Cout <= 4) CENET_LAUNCH((pw_fewout_dgrad_kernel<4>), grid, dim3(256), stream, dy, W, dx, Cout, (int)HW);
  else if (Cout <= 9) CENET_LAUNCH((pw_fewout_dgrad_kernel<9>), grid, dim3(256), stream, dy, W, dx, Cout, (int)HW);
  else CENET_LAUNCH((pw_fewout_dgrad_kernel<16>), grid, dim3(256), stream, dy, W, dx, Cout, (int)HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_pw_fewout_wgrad_bf16(const bf16_t* x, const bf16_t* dy, float* dW_acc, float* dbias_acc, int B, int Cin,
                                          int Cout, long HW, hipStream_t stream) {
  if (!x || !dy || !dW_acc || B <= 0 || HW <= 0 || HW > 0x7FFFFFFF || B > 65535) return CENET_EINVAL;
  if (!cenet_pw_fewout_wgrad_supported(Cin, Cout)) return CENET_EUNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)dy) & 3) != 0) return CENET_EINVAL;
  const dim3 grid(cdiv((int)HW, 1024), B);
  if (Cout == 2) CENET_LAUNCH((pw_fewout_wgrad_kernel<2>), grid, dim3(256), stream, x, dy, dW_acc, dbias_acc, (int)HW);
  else if (Cout == 4) CENET_LAUNCH((pw_fewout_wgrad_kernel<4>), grid, dim3(256), stream, x, dy, dW_acc, dbias_acc, (int)HW);
  else CENET_LAUNCH((pw_fewout_wgrad_kernel<9>), grid, dim3(256), stream, x, dy, dW_acc, dbias_acc, (int)HW);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
