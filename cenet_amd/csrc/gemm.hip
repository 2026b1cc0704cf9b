// gemm.hip — batched strided GEMM / implicit-GEMM convolution core on the gfx950 matrix cores.
//
// One kernel serves every dense contraction of the CENet hot path:
//   * nn.Linear fwd/bwd in token layout                      (pvtv2.py:41,45,90,98,106; multihead_diffattn.py:79-81,126)
//   * 1x1 convs in NCHW fwd/bwd                              (cfam.py:149,158,299,302; nlb.py:106-115,142; blocks.py:178,320; dseb.py:164)
//   * dense k x k convs as implicit GEMM (fwd, dgrad, wgrad)  (pvtv2.py:164,67; unet.py:156-197; blocks.py:211)
//   * materialised attention products for large head dims
//
// C[b] (+)= epilogue( alpha * sum_kb A[b,kb] (MxK) * B[b,kb] (KxN) )
// A is always a plain strided matrix; B is plain or an on-the-fly im2col / transposed-gather view of an image tensor.
// Tiles: 64x64x32 per 256-thread workgroup (4 waves, each 32x32 = 2x2 v_mfma_f32_16x16x4_f32 tiles); operands are
// staged k-contiguous in LDS with a row pitch of 34 dwords, which makes the ds_read_b32 fragment reads of a 32-lane
// group hit 32 distinct banks.  fp32 in / fp32 accumulate == exact f32 FMA chain (parity mode).
#include "common.h"
#include "../../include/cenet_hip.h"

#define BM 64
#define BN 64
#define BK 32
#define LDP 34

struct GemmArgs {
  cenet_mat_t A, B;
  cenet_epi_t E;
  int M, N, K, nkb, splits, nb_inner;
};

// KDIM: 0 = the row index r is the k index (B operand), 1 = the column index c is the k index (A operand)
template <bool IM2COL, int KDIM>
__device__ __forceinline__ float mat_fetch(const cenet_mat_t& d, const float* base, int r, int c) {
  if (!IM2COL) {
    if (d.kinner > 0) {
      if (KDIM == 1) {
        int ko = c / d.kinner, ki = c - ko * d.kinner;
        return base[(long)r * d.sr + (long)ko * d.sk_outer + (long)ki * d.sc];
      } else {
        int ko = r / d.kinner, ki = r - ko * d.kinner;
        return base[(long)ko * d.sk_outer + (long)ki * d.sr + (long)c * d.sc];
      }
    }
    return base[(long)r * d.sr + (long)c * d.sc];
  } else {
    int e = d.patch_is_row ? r : c;
    int p = d.patch_is_row ? c : r;
    int kk = d.KH * d.KW;
    int ci = e / kk;
    int rem = e - ci * kk;
    int ky = rem / d.KW;
    int kx = rem - ky * d.KW;
    int py = p / d.Pw;
    int px = p - py * d.Pw;
    int iy, ix;
    if (!d.transposed) {
      iy = py * d.stride - d.pad + ky * d.dil;
      ix = px * d.stride - d.pad + kx * d.dil;
    } else {
      int ty = py + d.pad - ky * d.dil;
      int tx = px + d.pad - kx * d.dil;
      if (ty < 0 || tx < 0) return 0.f;
      iy = ty / d.stride;
      ix = tx / d.stride;
      if (iy * d.stride != ty || ix * d.stride != tx) return 0.f;
    }
    if (iy < 0 || iy >= d.Hs || ix < 0 || ix >= d.Ws) return 0.f;
    return base[(long)ci * d.sci + (long)iy * d.sy + (long)ix * d.sx];
  }
}

template <bool B_IM2COL>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ float As[BM * LDP];
  __shared__ float Bs[BN * LDP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int batch = blockIdx.z / g.splits, split = blockIdx.z - batch * g.splits;
  const int bo = batch / g.nb_inner, bi = batch - bo * g.nb_inner;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ktiles = (g.K + BK - 1) / BK;
  const int total = g.nkb * ktiles;
  const int chunk = (total + g.splits - 1) / g.splits;
  const int it0 = split * chunk;
  const int it1 = (it0 + chunk < total) ? it0 + chunk : total;

  for (int it = it0; it < it1; ++it) {
    const int kb = it / ktiles;
    const int k0 = (it - kb * ktiles) * BK;
    const float* baseA = g.A.ptr + (long)bo * g.A.sb + (long)bi * g.A.sb2 + (long)kb * g.A.skb;
    const float* baseB = g.B.ptr + (long)bo * g.B.sb + (long)bi * g.B.sb2 + (long)kb * g.B.skb;
    // ---- stage A (BM x BK), k-contiguous in LDS ----
    if (g.A.kfast) {
      const int kk = tid & 31, r0 = tid >> 5;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int row = r0 + 8 * i;
        float v = 0.f;
        if (m0 + row < g.M && k0 + kk < g.K) v = mat_fetch<false, 1>(g.A, baseA, m0 + row, k0 + kk);
        As[row * LDP + kk] = v;
      }
    } else {
      const int row = tid & 63, q0 = tid >> 6;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int kk = q0 + 4 * i;
        float v = 0.f;
        if (m0 + row < g.M && k0 + kk < g.K) v = mat_fetch<false, 1>(g.A, baseA, m0 + row, k0 + kk);
        As[row * LDP + kk] = v;
      }
    }
    // ---- stage B (BK x BN) as Bs[n][k] ----
    if (g.B.kfast) {
      const int kk = tid & 31, c0 = tid >> 5;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int col = c0 + 8 * i;
        float v = 0.f;
        if (n0 + col < g.N && k0 + kk < g.K) v = mat_fetch<B_IM2COL, 0>(g.B, baseB, k0 + kk, n0 + col);
        Bs[col * LDP + kk] = v;
      }
    } else {
      const int col = tid & 63, q0 = tid >> 6;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int kk = q0 + 4 * i;
        float v = 0.f;
        if (n0 + col < g.N && k0 + kk < g.K) v = mat_fetch<B_IM2COL, 0>(g.B, baseB, k0 + kk, n0 + col);
        Bs[col * LDP + kk] = v;
      }
    }
    __syncthreads();
    // ---- 8 k-steps of 4: 2x2 MFMA tiles per wave ----
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      float a0 = As[(wm * 32 + fr) * LDP + ks * 4 + fk];
      float a1 = As[(wm * 32 + 16 + fr) * LDP + ks * 4 + fk];
      float b0 = Bs[(wn * 32 + fr) * LDP + ks * 4 + fk];
      float b1 = Bs[(wn * 32 + 16 + fr) * LDP + ks * 4 + fk];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue ----
  const cenet_epi_t& E = g.E;
  float* Cb = E.C + (long)bo * E.scb + (long)bi * E.scb2;
  const float* Rb = E.R ? E.R + (long)bo * E.srb + (long)bi * E.srb2 : nullptr;
  const float bs = E.bscale ? E.bscale[batch] : 1.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        int col = n0 + wn * 32 + j * 16 + (lane & 15);
        if (row < g.M && col < g.N) {
          float v = acc[i][j][r] * E.alpha;
          if (E.atomic) {
            atomicAdd(&Cb[(long)row * E.scr + (long)col * E.scc], v);
          } else {
            if (E.bias) v += E.bias[E.bias_on_row ? row : col];
            v = act_fwd(E.act, v, E.slope);
            v *= bs;
            if (Rb) v += Rb[(long)row * E.srr + (long)col * E.src];
            Cb[(long)row * E.scr + (long)col * E.scc] = v;
          }
        }
      }
}

extern "C" int cenet_gemm_f32(const cenet_mat_t* A, const cenet_mat_t* B, const cenet_epi_t* E, int M, int N, int K,
                              int nbatch, int nb_inner, int nkb, int splits, hipStream_t stream) {
  if (!A || !B || !E || !A->ptr || !B->ptr || !E->C) return CENET_EINVAL;
  if (M <= 0 || N <= 0 || K <= 0 || nbatch <= 0 || nb_inner <= 0 || nkb <= 0 || splits <= 0) return CENET_EINVAL;
  if (A->mode != 0) return CENET_EUNSUPPORTED;
  if (splits > 1 && !E->atomic) return CENET_EINVAL;
  if (E->atomic && (E->bias || E->R || E->act != ACT_NONE || E->bscale)) return CENET_EINVAL;
  GemmArgs g;
  g.A = *A;
  g.B = *B;
  g.E = *E;
  g.M = M; g.N = N; g.K = K; g.nkb = nkb; g.splits = splits; g.nb_inner = nb_inner;
  dim3 grid(cdiv(N, BN), cdiv(M, BM), nbatch * splits);
  if (grid.y > 65535 || grid.z > 65535) return CENET_EUNSUPPORTED;
  if (B->mode == 0) {
    CENET_LAUNCH((gemm_f32_kernel<false>), grid, dim3(256), stream, g);
  } else {
    CENET_LAUNCH((gemm_f32_kernel<true>), grid, dim3(256), stream, g);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
