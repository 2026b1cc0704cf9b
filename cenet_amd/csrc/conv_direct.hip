// conv_direct.hip — direct ("LDS halo") convolution for the stride-1 same-padded k x k convs of the output head in the
// bf16-operand mode: forward and data-gradient of out.rb.0.conv2 (5x5, 32->32 @224^2), out.out.0.conv{1,2} (3x3, 64->64
// @112^2) and out.up.up.1 (3x3, 64->32 @112^2)  [reference out.py:41-49,59; unet.py:156-197; blocks.py:211].
//
// Why: as an implicit GEMM the 25-tap gather fetched 1.16 GB per launch for 411 MB of algorithmic traffic
// (profiles/r01_pmc_roofline_kernel.csv).  Here every input element is read from HBM/L2 ONCE per output tile:
//   * a persistent 256-thread workgroup keeps the WHOLE weight tensor in LDS as W[tap][ci/8][co][8] (bf16) and walks a
//     list of 8x32-pixel output tiles;
//   * the input halo tile of all input channels is staged as X[ci/8][y][x][8] (bf16): a lane gathers the 8 channels of one
//     pixel with 8 lane-coalesced dword loads and writes ONE 16-byte LDS slot, so staging and fragment reads are both
//     conflict-free and an MFMA operand fragment (8 consecutive input channels of one pixel / one output channel) is one
//     ds_read_b128;
//   * the next tile's halo is prefetched into registers while the current tile runs its KK*(CIN/32) k-steps of
//     v_mfma_f32_16x16x32_bf16 (pixels on the MFMA row so a lane owns 4 consecutive x -> 16-byte stores);
//   * the data-gradient is the same kernel reading the weights transposed and flipped.
// Throughput mode only: activations (x, y, dY) are bf16 in HBM and go to LDS without conversion; weights are read from their
// fp32 master copy (a few KB per launch) and rounded once per workgroup; fp32 accumulate.  fp32 tensors (parity mode) take
// the exact implicit-GEMM path of gemm_core.h instead.
#include "common.h"
#include "../../include/cenet_hip.h"

#define TH 8
#define TW 32
typedef bf16_t bf;

__device__ __forceinline__ unsigned cd_f2bf(float f) { return cenet_f2bf(f); }

struct ConvDirectArgs {
  const bf16_t* x;  // [B, CIN, H, W]
  const float* w;   // fwd: [COUT, CIN, KS, KS] ; dgrad: [CIN(kernel in = conv out), COUT(kernel out = conv in), KS, KS]
  bf16_t* y;        // [B, COUT, H, W]
  int B, H, W, dgrad, tiles_x, tiles_y, ntiles;
};

// NW waves per workgroup: 4, or 8 for the 64 -> 64 instance — its LDS image (117 KB) allows one workgroup per CU, and at four waves
// (365 registers: 64 accumulators, 88 for the prefetched halo) that was ONE wave per SIMD with nothing to cover an LDS read; eight
// waves own one tile row each (32 accumulators, half the halo slots: under 256 registers, two waves per SIMD)
template <int CIN, int COUT, int KS, int NW>
__global__ __launch_bounds__(64 * NW, (CIN == 64 && COUT == 64) ? 1 : 2) void conv_direct_bf16_kernel(ConvDirectArgs a) {
  constexpr int NT = 64 * NW, RPW = TH / NW, NI = 2 * RPW;  // tile rows per wave, (row, x-segment) pairs per wave
  constexpr int CQ = CIN / 8, HH = TH + KS - 1, HW_ = TW + KS - 1, KK = KS * KS, PADK = KS / 2;
  constexpr int UNITS = CQ * HH * HW_;           // 16-byte halo slots
  constexpr int UPT = (UNITS + NT - 1) / NT;     // slots per thread
  constexpr int MT = COUT / 16;                  // output-channel tiles
  __shared__ __attribute__((aligned(16))) bf Ws[KK * CQ * COUT * 8];
  __shared__ __attribute__((aligned(16))) bf Xs[UNITS * 8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, fq = lane >> 4;

  // ---- weights -> LDS once per workgroup: Ws[((tap*CQ + ci/8)*COUT + co)*8 + ci%8]
  for (int e = tid; e < KK * CIN * COUT; e += NT) {
    const int t = e % KK, rest = e / KK;
    int co, ci;
    float v;
    if (!a.dgrad) {  // w[co][ci][t]
      ci = rest % CIN;
      co = rest / CIN;
      v = a.w[e];
    } else {         // original tensor w[i = kernel-in][o = kernel-out][t], used flipped: Wd[o][i][t] = w[i][o][KK-1-t]
      co = rest % COUT;
      ci = rest / COUT;
      v = a.w[(long)(ci * COUT + co) * KK + (KK - 1 - t)];
    }
    Ws[((t * CQ + (ci >> 3)) * COUT + co) * 8 + (ci & 7)] = (bf)cd_f2bf(v);
  }

  const int HWp = a.H * a.W;
  unsigned pre[UPT][8];  // raw bf16 bit patterns of the 8 channels of one halo pixel
  // a thread's halo slots are the same in every tile: their (channel octet, row, column) split and element offset are
  // computed once; a tile only adds its origin
  int uhy[UPT], uhx[UPT];
  long uoff[UPT];
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    const int s = tid + NT * u;
    const int cq = s / (HH * HW_), rem = s - cq * (HH * HW_);
    uhy[u] = (s < UNITS) ? rem / HW_ - PADK : -(1 << 20);  // out-of-range slots fail every row test
    uhx[u] = rem % HW_ - PADK;
    uoff[u] = (long)(cq * 8) * HWp + (long)uhy[u] * a.W + uhx[u];
  }
  auto load_halo = [&](int tile) {
    const int b = tile / (a.tiles_x * a.tiles_y), tt = tile - b * (a.tiles_x * a.tiles_y);
    const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
    const bf16_t* xt = a.x + (long)b * CIN * HWp + (long)(ty * TH) * a.W + tx * TW;
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int iy = ty * TH + uhy[u], ix = tx * TW + uhx[u];
      const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const bf16_t* p = xt + uoff[u];
#pragma unroll
      for (int j = 0; j < 8; ++j) pre[u][j] = ok ? (unsigned)p[(long)j * HWp] : 0u;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int s = tid + NT * u;
      if (s < UNITS) {
        unsigned pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = pre[u][2 * j] | (pre[u][2 * j + 1] << 16);
        memcpy(&Xs[s * 8], pk, 16);
      }
    }
  };

  // XCD-aware tile walk: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its own L2.  XCD x walks
  // the contiguous eighth [x*per, (x+1)*per) of the tile list (whole images), its workgroups taking consecutive tiles, so
  // the tiles that share halo rows / columns and 128-byte lines run on the same L2 at about the same time.  (Dealing tiles
  // by blockIdx instead put every neighbour on another XCD: 866 MB fetched + written for 411 MB algorithmic; now 417 MB.)
  const int G = gridDim.x < 8 ? (int)gridDim.x : 8;
  const int xcd = blockIdx.x % G, nloc = ((int)gridDim.x + G - 1 - xcd) / G;  // workgroups on this XCD
  const int per = (a.ntiles + G - 1) / G;
  const int t_end = (xcd + 1) * per < a.ntiles ? (xcd + 1) * per : a.ntiles;
  int tile = xcd * per + (int)blockIdx.x / G;
  if (tile < t_end) load_halo(tile);
  for (; tile < t_end; tile += nloc) {
    __syncthreads();  // previous tile's fragments consumed (and, first time, weights visible after the next barrier)
    store_halo();
    __syncthreads();
    const int nxt = tile + nloc;
    if (nxt < t_end) load_halo(nxt);  // next halo flies under this tile's MFMAs

    f32x4 acc[MT][NI];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // wave w owns tile rows RPW w .. RPW w + RPW - 1 ; ni -> (row = RPW w + ni/2, xseg = 16*(ni&1))
    // fully unrolled: every LDS address below is one per-lane base plus a compile-time offset (the rolled loop spent ~35
    // VALU instructions per tap on divisions and address arithmetic — the kernel was VALU-bound, not MFMA- or HBM-bound)
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      constexpr int KS_ = KS;
      const int ky = t / KS_, kx = t - ky * KS_;
#pragma unroll
      for (int kb = 0; kb < CIN / 32; ++kb) {
        bf16x8 wf[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) memcpy(&wf[mi], &Ws[((t * CQ + kb * 4 + fq) * COUT + mi * 16 + fr) * 8], 16);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int row = RPW * wave + (ni >> 1), xs = 16 * (ni & 1);
          bf16x8 xf;
          memcpy(&xf, &Xs[(((kb * 4 + fq) * HH + row + ky) * HW_ + xs + fr + kx) * 8], 16);
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wf[mi], acc[mi][ni], 0, 0, 0);
        }
      }
    }
    // epilogue: acc[mi][ni][r] = out[co = 16 mi + fr][y = ty*TH + row][x = tx*TW + xs + 4 fq + r]
    const int b = tile / (a.tiles_x * a.tiles_y), tt = tile - b * (a.tiles_x * a.tiles_y);
    const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
    bf16_t* yb = a.y + (long)b * COUT * HWp;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int oy = ty * TH + RPW * wave + (ni >> 1), ox = tx * TW + 16 * (ni & 1) + 4 * fq;
      if (oy < a.H && ox < a.W) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          bf16_t* dst = yb + (long)(mi * 16 + fr) * HWp + (long)oy * a.W + ox;
          if (ox + 3 < a.W && (a.W & 3) == 0) {
            float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
            st4v(dst, v);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (ox + r < a.W) stf(dst + r, acc[mi][ni][r]);
          }
        }
      }
    }
  }
}

extern "C" int cenet_conv_direct_supported(int Cin, int Cout, int k, int stride, int pad) {
  if (stride != 1 || pad != k / 2) return 0;
  if (k == 5 && Cin == 32 && Cout == 32) return 1;
  if (k == 3 && Cin == 64 && (Cout == 64 || Cout == 32)) return 1;
  if (k == 3 && Cin == 32 && Cout == 64) return 1;  // data-gradient of the 64->32 conv
  return 0;
}

// dgrad = 0: y[B,Cout,H,W] = conv(x[B,Cin,H,W], w[Cout,Cin,k,k]);  dgrad = 1: y = dX[B,Cout,H,W] from x = dY[B,Cin,H,W]
// and the ORIGINAL forward weight w[Cin,Cout,k,k] (Cin/Cout here name the kernel's input/output channel counts).
extern "C" int cenet_conv_direct_bf16(const bf16_t* x, const float* w, bf16_t* y, int B, int Cin, int Cout, int H, int W, int k,
                                      int dgrad, hipStream_t stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  if (((uintptr_t)y & 7) != 0) return CENET_EINVAL;  // output rows are stored as 8-byte quads
  if (!cenet_conv_direct_supported(Cin, Cout, k, 1, k / 2)) return CENET_EUNSUPPORTED;
  ConvDirectArgs a;
  a.x = x; a.w = w; a.y = y; a.B = B; a.H = H; a.W = W; a.dgrad = dgrad;
  a.tiles_x = cdiv(W, TW);
  a.tiles_y = cdiv(H, TH);
  a.ntiles = B * a.tiles_x * a.tiles_y;
  // persistent workgroups: two per CU where two copies of the LDS image (weights + halo) fit in 160 KB — the staging of one
  // then overlaps the MFMAs of the other — else one per CU.  (16-byte halo loads were tried and changed nothing.)
  const bool two = !(k == 3 && Cin == 64 && Cout == 64);
  const int slots = two ? 512 : 256;
  int grid = a.ntiles < slots ? a.ntiles : slots;
  if (k == 5 && Cin == 32 && Cout == 32) {
    CENET_LAUNCH((conv_direct_bf16_kernel<32, 32, 5, 4>), dim3(grid), dim3(256), stream, a);
  } else if (k == 3 && Cin == 64 && Cout == 64) {
    CENET_LAUNCH((conv_direct_bf16_kernel<64, 64, 3, 8>), dim3(grid), dim3(512), stream, a);
  } else if (k == 3 && Cin == 64 && Cout == 32) {
    CENET_LAUNCH((conv_direct_bf16_kernel<64, 32, 3, 8>), dim3(grid), dim3(512), stream, a);  // (same time as four waves; no scratch)
  } else {
    CENET_LAUNCH((conv_direct_bf16_kernel<32, 64, 3, 4>), dim3(grid), dim3(256), stream, a);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

// =====================================================================================================================
// Weight gradient of the same convolutions, direct form:  dW[co][ci][ky][kx] += sum_{b,y,x} dY[b][co][y][x] * X[b][ci][y+ky-p][x+kx-p]
//
// As an implicit GEMM (K = pixels) the im2col operand was gathered element by element from HBM/L2 (1.96 ms for the 5x5
// 32->32 conv at 224^2 against ~0.1 ms of traffic).  Here a workgroup stages one TH x 32 pixel tile: dY[co][y][x] and the
// X halo [ci][y][x] as bf16 in LDS, pixels contiguous.  For one image row of the tile and one tap, the product over the
// row's 32 pixels is ONE v_mfma_f32_16x16x32_bf16 per (16 output channels x 16 input channels):
//     A = dY[co = fr][8 px at 8fq]            one aligned ds_read_b128
//     B = X [ci = fr][8 px at 8fq + kx - p]   a window shifted by whole bf16 elements: six dwords are read once per (row, ky)
//                                             and the KS shifted fragments are cut out of them with funnel shifts
// Each wave owns a fixed set of (co-tile, ci-tile) pairs and keeps all KS*KS accumulator tiles of them in registers across
// the persistent tile loop; two workgroups share a CU so that one stages while the other multiplies.  Partial sums leave as
// coalesced 16-byte stores into a per-workgroup slab and a second kernel folds the slabs into dW.
// =====================================================================================================================
struct ConvWgradArgs {
  const bf16_t* x;   // [B, CIN, H, W]
  const bf16_t* dy;  // [B, COUT, H, W]
  float* ws;         // [gridDim.x][COUT*CIN*KS*KS] partial sums, register order
  int B, H, W, tiles_x, tiles_y, ntiles;
  int vec;           // 16-byte staging: W % 8 == 0 and 16-byte aligned tensors (else pixel pairs, 4 bytes at a time)
};

__device__ __forceinline__ unsigned funnel16(unsigned lo, unsigned hi) { return (lo >> 16) | (hi << 16); }

// NW waves per workgroup: 4 (two workgroups per CU) or 8 (one: the 64 x 64 instance — 16 pairs x 9 taps = 144 accumulator registers
// per lane at 4 waves, 356 bytes of scratch per lane under the 256-register cap; at 8 waves a lane holds 72)
// KSPLIT = 2 (the 5 x 5 32 -> 32 instance: only four (co, ci) pairs, 25 taps each = 100 accumulator registers, 136 bytes of scratch
// at four waves): two waves share a pair and take taps 0 .. 12 / 13 .. 24 — 52 registers each, two 8-wave workgroups per CU
template <int V>
struct cd_ic {
  static constexpr int value = V;
};
template <int CIN, int COUT, int KS, int TH_, int NW, int KSPLIT = 1>
__global__ __launch_bounds__(64 * NW, (NW == 4 || KSPLIT == 2) ? 2 : 1) void conv_wgrad_direct_kernel(ConvWgradArgs a) {
  constexpr int NT = 64 * NW;
  constexpr int P = KS / 2, HH = TH_ + KS - 1, KK = KS * KS;
  constexpr int XROW = 24;                                   // dwords per halo row: 8 px pad + 32 px + 8 px pad
  constexpr int XPL = HH * XROW + (12 - (HH * XROW) % 8) % 8;  // ci plane stride in dwords, == 4 (mod 8): conflict-free b128
  constexpr int GPL = TH_ * 16 + (12 - (TH_ * 16) % 8) % 8;    // co plane stride of dY
  static_assert(XPL % 8 == 4 && GPL % 8 == 4, "plane strides must be 4 mod 8 dwords");
  constexpr int NCI = CIN / 16, NCO = COUT / 16, PAIRS = NCI * NCO, NPW = PAIRS * KSPLIT / NW;
  constexpr int TPP = (KS * KS + KSPLIT - 1) / KSPLIT;  // taps per wave
  static_assert((PAIRS * KSPLIT) % NW == 0 && NPW >= 1 && (NPW <= NCI) && (NCI % NPW == 0) && (KSPLIT == 1 || NPW == 1),
                "a wave's pairs share one output-channel tile");
  __shared__ __attribute__((aligned(16))) unsigned Xs[CIN * XPL];
  __shared__ __attribute__((aligned(16))) unsigned Gs[COUT * GPL];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  const int pw = wave / KSPLIT, part = wave % KSPLIT;                 // pair-set index; which share of the taps
  const int cog = (pw * NPW) / NCI, cig0 = (pw * NPW) % NCI;  // this wave: co tile cog, ci tiles cig0 .. cig0+NPW-1
  const int HWp = a.H * a.W;
  const bool w_even = (a.W & 1) == 0;
  f32x4 acc[NPW][TPP];  // acc[j][t - part * TPP]
#pragma unroll
  for (int j = 0; j < NPW; ++j)
#pragma unroll
    for (int t = 0; t < TPP; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // XCD-aware tile walk, as in the forward kernel: each XCD owns a contiguous eighth of the tile list
  const int G = gridDim.x < 8 ? (int)gridDim.x : 8;
  const int xcd = blockIdx.x % G, nloc = ((int)gridDim.x + G - 1 - xcd) / G;
  const int per = (a.ntiles + G - 1) / G;
  const int t_end = (xcd + 1) * per < a.ntiles ? (xcd + 1) * per : a.ntiles;
  // ---- staging, 16-byte form (a.vec): a halo row of X is the 48 pixels x0 - 8 .. x0 + 39 = six aligned 16-byte chunks that map
  // one to one onto the 24 dwords of its LDS row (8 px pad | 32 px | 8 px pad), a dY row is four chunks.  A thread's chunks are the
  // same for every tile (chunk number = tid + 256 q): their LDS offsets and image-relative offsets are computed once.  The chunks of
  // tile t + 1 are fetched into registers BEFORE tile t is multiplied and written to LDS after it — the kernel was parked on its
  // loads and barriers 62 - 66 % of the time (SQ_WAIT_ANY) with 4-byte loads issued and waited for between the tiles.
  // A thread's chunks are the same for every tile (chunk number = tid + NT q); their coordinates are recomputed from that number
  // at every use (divisions by constants) rather than held in registers: the arrays cost 39 registers of the 256.
  constexpr int NXC = CIN * HH * 6, NGC = COUT * TH_ * 4;
  constexpr int QX = (NXC + NT - 1) / NT, QG = (NGC + NT - 1) / NT;
  uint4 xv[QX], gv[QG];
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int b = tile / (a.tiles_x * a.tiles_y), tt = tile - b * (a.tiles_x * a.tiles_y);
    const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
    const int y0 = ty * TH_, x0 = tx * 32;
    const bf16_t* xb = a.x + (long)b * CIN * HWp + (long)(y0 - P) * a.W + x0;
    const bf16_t* gb = a.dy + (long)b * COUT * HWp + (long)y0 * a.W + x0;
#pragma unroll
    for (int q = 0; q < QX; ++q) {
      const int u = tid + NT * q;
      const int c = u % 6, r2 = u / 6, hy = r2 % HH, ci = r2 / HH;
      const int iy = y0 - P + hy, ix = x0 - 8 + 8 * c;
      xv[q] = uint4{0u, 0u, 0u, 0u};
      if (u < NXC && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) memcpy(&xv[q], xb + (ci * HWp + hy * a.W + 8 * c - 8), 16);
    }
#pragma unroll
    for (int q = 0; q < QG; ++q) {
      const int u = tid + NT * q;
      const int c = u & 3, r2 = u >> 2, yy = r2 % TH_, co = r2 / TH_;
      const int iy = y0 + yy, ix = x0 + 8 * c;
      gv[q] = uint4{0u, 0u, 0u, 0u};
      if (u < NGC && iy < a.H && ix < a.W) memcpy(&gv[q], gb + (co * HWp + yy * a.W + 8 * c), 16);
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < QX; ++q) {
      const int u = tid + NT * q;
      const int c = u % 6, r2 = u / 6, hy = r2 % HH, ci = r2 / HH;
      if (u < NXC) memcpy(&Xs[ci * XPL + hy * XROW + 4 * c], &xv[q], 16);
    }
#pragma unroll
    for (int q = 0; q < QG; ++q) {
      const int u = tid + NT * q;
      const int c = u & 3, r2 = u >> 2, yy = r2 % TH_, co = r2 / TH_;
      if (u < NGC) memcpy(&Gs[co * GPL + yy * 16 + 4 * c], &gv[q], 16);
    }
  };
  const int tile0 = xcd * per + (int)blockIdx.x / G;
  if (a.vec && tile0 < t_end) {
    fetch(tile0);
    stage();
  }
  for (int tile = tile0; tile < t_end; tile += nloc) {
    if (a.vec) {
      if (tile + nloc < t_end) fetch(tile + nloc);  // in flight while this tile is multiplied
    } else {
    const int b = tile / (a.tiles_x * a.tiles_y), tt = tile - b * (a.tiles_x * a.tiles_y);
    const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
    const int y0 = ty * TH_, x0 = tx * 32;
    const bf16_t* xb = a.x + (long)b * CIN * HWp;
    const bf16_t* gb = a.dy + (long)b * COUT * HWp;
    __syncthreads();  // the previous tile's fragments are consumed
    // ---- X halo: dwords 3..20 of every (ci, halo row) = pixels x0-2 .. x0+33 ; 8 pairs in flight per thread
    constexpr int NXU = CIN * HH * 18;
    for (int u0 = tid; u0 < NXU; u0 += NT * 8) {
      unsigned v[8];  // packed pixel pairs, raw bf16
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int u = u0 + NT * q;
        v[q] = 0u;
        if (u < NXU) {
          const int d = u % 18, r2 = u / 18, hy = r2 % HH, ci = r2 / HH;
          const int iy = y0 - P + hy, ix = x0 - 2 + 2 * d;
          if (iy >= 0 && iy < a.H) {
            const bf16_t* p = xb + (long)ci * HWp + (long)iy * a.W + ix;
            if (w_even && ix >= 0 && ix + 1 < a.W) {
              memcpy(&v[q], p, 4);
            } else {
              if (ix >= 0 && ix < a.W) v[q] = p[0];
              if (ix + 1 >= 0 && ix + 1 < a.W) v[q] |= (unsigned)p[1] << 16;
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int u = u0 + NT * q;
        if (u < NXU) {
          const int d = u % 18, r2 = u / 18, hy = r2 % HH, ci = r2 / HH;
          Xs[ci * XPL + hy * XROW + 3 + d] = v[q];
        }
      }
    }
    // ---- dY tile
    constexpr int NGU = COUT * TH_ * 16;
    for (int u0 = tid; u0 < NGU; u0 += NT * 8) {
      unsigned v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int u = u0 + NT * q;
        v[q] = 0u;
        if (u < NGU) {
          const int d = u & 15, r2 = u >> 4, yy = r2 % TH_, co = r2 / TH_;
          const int iy = y0 + yy, ix = x0 + 2 * d;
          if (iy < a.H) {
            const bf16_t* p = gb + (long)co * HWp + (long)iy * a.W + ix;
            if (w_even && ix + 1 < a.W) {
              memcpy(&v[q], p, 4);
            } else {
              if (ix < a.W) v[q] = p[0];
              if (ix + 1 < a.W) v[q] |= (unsigned)p[1] << 16;
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int u = u0 + NT * q;
        if (u < NGU) {
          const int d = u & 15, r2 = u >> 4, yy = r2 % TH_, co = r2 / TH_;
          Gs[co * GPL + yy * 16 + d] = v[q];
        }
      }
    }
    }
    __syncthreads();
    // ---- one k-step (32 pixels) per tile row
    auto rows = [&](auto pc) __attribute__((always_inline)) {
      constexpr int PART = decltype(pc)::value, T0 = PART * TPP, T1 = (T0 + TPP < KK) ? T0 + TPP : KK;
      for (int row = 0; row < TH_; ++row) {
        bf16x8 af;
        memcpy(&af, &Gs[(16 * cog + fr) * GPL + row * 16 + 4 * fq], 16);
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
#pragma unroll
          for (int ky = 0; ky < KS; ++ky) {
            if (ky * KS + KS <= T0 || ky * KS >= T1) continue;  // (compile-time: no tap of this kernel row is this wave's)
            const unsigned* xr = &Xs[(16 * (cig0 + j) + fr) * XPL + (row + ky) * XROW + 4 + 4 * fq];
            unsigned w[6];
            w[0] = xr[-1];
            memcpy(&w[1], xr, 16);
            w[5] = xr[4];
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
              constexpr int KS_ = KS;
              const int t = ky * KS_ + kx;
              if (t < T0 || t >= T1) continue;  // (compile-time)
              const int s = kx - P + 2;  // first window element of this tap, counted from w[0]'s low half
              unsigned f[4];
#pragma unroll
              for (int d = 0; d < 4; ++d) f[d] = (s & 1) ? funnel16(w[(s - 1) / 2 + d], w[(s + 1) / 2 + d]) : w[s / 2 + d];
              bf16x8 bfv;
              memcpy(&bfv, f, 16);
              acc[j][t - T0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfv, acc[j][t - T0], 0, 0, 0);
            }
          }
        }
      }
    };
    if (KSPLIT == 1 || part == 0) rows(cd_ic<0>());  // (wave-uniform)
    else rows(cd_ic<KSPLIT - 1>());
    if (a.vec && tile + nloc < t_end) {
      __syncthreads();  // this tile's fragments are consumed
      stage();
    }
  }
  // ---- partial sums, register order: ws[block][((wave*NPW + j)*KK + t)*256 + lane*4 + r]
  float* slab = a.ws + (long)blockIdx.x * (COUT * CIN * KK);
#pragma unroll
  for (int j = 0; j < NPW; ++j)
#pragma unroll
    for (int tt = 0; tt < TPP; ++tt) {
      const int t = part * TPP + tt;
      if (t < KK) {
        float v[4] = {acc[j][tt][0], acc[j][tt][1], acc[j][tt][2], acc[j][tt][3]};
        memcpy(slab + ((pw * NPW + j) * KK + t) * 256 + lane * 4, v, 16);
      }
    }
}

// dW[co][ci][t] += sum over a slice of the slabs; grid (PSIZE/256, slices)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslabs,
                                                               int psize, int CIN, int KK) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= psize) return;
  const int per = (nslabs + gridDim.y - 1) / gridDim.y;
  const int s0 = blockIdx.y * per, s1 = (s0 + per < nslabs) ? s0 + per : nslabs;
  // (eight independent loads in flight per thread: the plain loop was one dependent add per ~1 us round trip — SQ_WAIT_ANY 0.96)
  float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int sidx = s0;
  for (; sidx + 8 <= s1; sidx += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc8[k] += ws[(long)(sidx + k) * psize + e];
  }
  for (; sidx < s1; ++sidx) acc8[0] += ws[(long)sidx * psize + e];
  const float sum = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
  const int within = e & 255, tile = e >> 8;
  const int lane = within >> 2, r = within & 3, fr = lane & 15, fq = lane >> 4;
  const int pr = tile / KK, t = tile - pr * KK;
  const int nci = CIN / 16, cog = pr / nci, cig = pr - cog * nci;
  const int co = 16 * cog + 4 * fq + r, ci = 16 * cig + fr;
  atomicAdd(&dw[((long)co * CIN + ci) * KK + t], sum);
}

#define CENET_WGRAD_SLABS 512
extern "C" int cenet_conv_wgrad_direct_supported(int Cin, int Cout, int k, int stride, int pad) {
  if (stride != 1 || pad != k / 2) return 0;
  return (k == 5 && Cin == 32 && Cout == 32) || (k == 3 && Cin == 64 && (Cout == 64 || Cout == 32));
}
extern "C" long cenet_conv_wgrad_direct_ws_floats(int Cin, int Cout, int k) {
  return cenet_conv_wgrad_direct_supported(Cin, Cout, k, 1, k / 2) ? (long)CENET_WGRAD_SLABS * Cin * Cout * k * k : 0;
}
extern "C" int cenet_conv_wgrad_direct_bf16(const bf16_t* x, const bf16_t* dy, float* dw_acc, float* ws, int B, int Cin, int Cout,
                                            int H, int W, int k, hipStream_t stream) {
  if (!x || !dy || !dw_acc || !ws || B <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  if ((((uintptr_t)x | (uintptr_t)dy) & 3) != 0) return CENET_EINVAL;  // pixel pairs are fetched as 4-byte words
  if (!cenet_conv_wgrad_direct_supported(Cin, Cout, k, 1, k / 2)) return CENET_EUNSUPPORTED;
  ConvWgradArgs a;
  a.x = x; a.dy = dy; a.ws = ws; a.B = B; a.H = H; a.W = W;
  {
    static const bool no_vec = getenv("CENET_WGRAD_NO_VEC") != nullptr;  // measurement aid
    a.vec = !no_vec && (W & 7) == 0 && ((((uintptr_t)x | (uintptr_t)dy) & 15) == 0) && H < 256 &&
            (long)(Cin > Cout ? Cin : Cout) * H * W < (1L << 30);  // (row index in 8 bits, image-relative offsets in an int)
  }
  const int th = (k == 5) ? 8 : 4;
  a.tiles_x = cdiv(W, 32);
  a.tiles_y = cdiv(H, th);
  a.ntiles = B * a.tiles_x * a.tiles_y;
  int grid = a.ntiles < CENET_WGRAD_SLABS ? a.ntiles : CENET_WGRAD_SLABS;
  if (const char* e = getenv("CENET_WGRAD_GRID")) {  // test aid: few workgroups, so that each one walks several tiles
    const int gset = atoi(e);
    if (gset >= 1 && gset < grid) grid = gset;
  }
  const bool wide = k == 3 && Cout == 64;  // eight waves, one workgroup per CU
  if (wide && grid > CENET_WGRAD_SLABS / 2) grid = CENET_WGRAD_SLABS / 2;
  if (k == 5) CENET_LAUNCH((conv_wgrad_direct_kernel<32, 32, 5, 8, 8, 2>), dim3(grid), dim3(512), stream, a);
  else if (wide) CENET_LAUNCH((conv_wgrad_direct_kernel<64, 64, 3, 4, 8>), dim3(grid), dim3(512), stream, a);
  else CENET_LAUNCH((conv_wgrad_direct_kernel<64, 32, 3, 4, 4>), dim3(grid), dim3(256), stream, a);
  const int psize = Cin * Cout * k * k;
  CENET_LAUNCH(conv_wgrad_reduce_kernel, dim3(psize / 256, 8), dim3(256), stream, (const float*)ws, dw_acc, grid, psize, Cin,
               k * k);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
