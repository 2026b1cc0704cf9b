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
// fp32 in HBM, fp32 accumulate; only the MFMA operands are bf16 (same contract as the bf16 GEMM core).
#include "common.h"
#include "../../include/cenet_hip.h"

#define TH 8
#define TW 32
typedef unsigned short bf;

__device__ __forceinline__ unsigned cd_f2bf(float f) { return cenet_f2bf(f); }

struct ConvDirectArgs {
  const float* x;   // [B, CIN, H, W]
  const float* w;   // fwd: [COUT, CIN, KS, KS] ; dgrad: [CIN(kernel in = conv out), COUT(kernel out = conv in), KS, KS]
  float* y;         // [B, COUT, H, W]
  int B, H, W, dgrad, tiles_x, tiles_y, ntiles;
};

template <int CIN, int COUT, int KS>
__global__ __launch_bounds__(256) void conv_direct_bf16_kernel(ConvDirectArgs a) {
  constexpr int CQ = CIN / 8, HH = TH + KS - 1, HW_ = TW + KS - 1, KK = KS * KS, PADK = KS / 2;
  constexpr int UNITS = CQ * HH * HW_;           // 16-byte halo slots
  constexpr int UPT = (UNITS + 255) / 256;       // slots per thread
  constexpr int MT = COUT / 16;                  // output-channel tiles
  __shared__ __attribute__((aligned(16))) bf Ws[KK * CQ * COUT * 8];
  __shared__ __attribute__((aligned(16))) bf Xs[UNITS * 8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, fq = lane >> 4;

  // ---- weights -> LDS once per workgroup: Ws[((tap*CQ + ci/8)*COUT + co)*8 + ci%8]
  for (int e = tid; e < KK * CIN * COUT; e += 256) {
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
  float pre[UPT][8];
  auto load_halo = [&](int tile) {
    const int b = tile / (a.tiles_x * a.tiles_y), tt = tile - b * (a.tiles_x * a.tiles_y);
    const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
    const float* xb = a.x + (long)b * CIN * HWp;
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int s = tid + 256 * u;
      const int cq = s / (HH * HW_), rem = s - cq * (HH * HW_);
      const int hy = rem / HW_, hx = rem - hy * HW_;
      const int iy = ty * TH + hy - PADK, ix = tx * TW + hx - PADK;
      const bool ok = (s < UNITS) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const float* p = xb + (long)(cq * 8) * HWp + (long)iy * a.W + ix;
#pragma unroll
      for (int j = 0; j < 8; ++j) pre[u][j] = ok ? p[(long)j * HWp] : 0.f;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int s = tid + 256 * u;
      if (s < UNITS) {
        unsigned pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = cenet_pack_bf2(pre[u][2 * j], pre[u][2 * j + 1]);
        memcpy(&Xs[s * 8], pk, 16);
      }
    }
  };

  int tile = blockIdx.x;
  if (tile < a.ntiles) load_halo(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();  // previous tile's fragments consumed (and, first time, weights visible after the next barrier)
    store_halo();
    __syncthreads();
    const int nxt = tile + gridDim.x;
    if (nxt < a.ntiles) load_halo(nxt);  // next halo flies under this tile's MFMAs

    f32x4 acc[MT][4];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // wave w owns tile rows 2w, 2w+1 ; ni -> (row = 2w + ni/2, xseg = 16*(ni&1))
    for (int t = 0; t < KK; ++t) {
      const int ky = t / KS, kx = t - ky * KS;
#pragma unroll
      for (int kb = 0; kb < CIN / 32; ++kb) {
        bf16x8 wf[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) memcpy(&wf[mi], &Ws[((t * CQ + kb * 4 + fq) * COUT + mi * 16 + fr) * 8], 16);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int row = 2 * wave + (ni >> 1), xs = 16 * (ni & 1);
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
    float* yb = a.y + (long)b * COUT * HWp;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int oy = ty * TH + 2 * wave + (ni >> 1), ox = tx * TW + 16 * (ni & 1) + 4 * fq;
      if (oy < a.H && ox < a.W) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          float* dst = yb + (long)(mi * 16 + fr) * HWp + (long)oy * a.W + ox;
          if (ox + 3 < a.W && (a.W & 3) == 0) {
            float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
            memcpy(dst, v, 16);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (ox + r < a.W) dst[r] = acc[mi][ni][r];
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
extern "C" int cenet_conv_direct_bf16(const float* x, const float* w, float* y, int B, int Cin, int Cout, int H, int W, int k,
                                      int dgrad, hipStream_t stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  if (!cenet_conv_direct_supported(Cin, Cout, k, 1, k / 2)) return CENET_EUNSUPPORTED;
  ConvDirectArgs a;
  a.x = x; a.w = w; a.y = y; a.B = B; a.H = H; a.W = W; a.dgrad = dgrad;
  a.tiles_x = cdiv(W, TW);
  a.tiles_y = cdiv(H, TH);
  a.ntiles = B * a.tiles_x * a.tiles_y;
  int grid = a.ntiles < 256 ? a.ntiles : 256;  // one persistent workgroup per CU (LDS-limited)
  if (k == 5 && Cin == 32 && Cout == 32) {
    CENET_LAUNCH((conv_direct_bf16_kernel<32, 32, 5>), dim3(grid), dim3(256), stream, a);
  } else if (k == 3 && Cin == 64 && Cout == 64) {
    CENET_LAUNCH((conv_direct_bf16_kernel<64, 64, 3>), dim3(grid), dim3(256), stream, a);
  } else if (k == 3 && Cin == 64 && Cout == 32) {
    CENET_LAUNCH((conv_direct_bf16_kernel<64, 32, 3>), dim3(grid), dim3(256), stream, a);
  } else {
    CENET_LAUNCH((conv_direct_bf16_kernel<32, 64, 3>), dim3(grid), dim3(256), stream, a);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
