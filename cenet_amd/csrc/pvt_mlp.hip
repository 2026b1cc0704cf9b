// pvt_mlp.hip — the MLP half of a PVTv2 block (reference pvtv2.py:40-47, 145-149, 364-370) as ONE kernel per pass on bf16
// token tensors:
//     y = x + s_b * ( fc2( GELU( DW3x3( fc1( LN(x) ) ) + bd ) ) + b2 )          (s_b: per-sample DropPath scale or 1)
// The hidden tensor ([tokens, 8C]: 102 MB per block at 56x56 / batch 32) and the LayerNorm output never reach HBM in the
// forward pass; the backward pass recomputes them tile by tile and writes only what the grouped weight-gradient launch reads.
//
// Workgroup (512 threads) = one (TH x TW)-token tile of one image.  The (TH+2) x (TW+2) halo tile of x is normalised into a
// k-fast bf16 LDS image (the layout of gemm_ring.h: 128-byte rows, 16-byte chunks XOR-swizzled); the hidden dimension is then
// walked in slabs of 64 channels:
//   P1 (MFMA)  h^T[64 ch, halo tokens] = W1_slab . xn^T + b1     -> bf16 LDS plane, rows of out-of-image tokens forced to 0
//                                                                    (the depthwise conv zero-pads h, not x)
//   P2 (VALU)  a = GELU(DW3x3(h) + bd) on the interior tokens     -> k-fast bf16 LDS image (thread = channel pair x strip)
//   P3 (MFMA)  acc^T[C, interior tokens] += W2[:, slab] . a^T     (fp32 registers across the slabs)
// and the epilogue adds bias, DropPath scale and the residual.  Weight slabs arrive by LDS-DMA one phase ahead of their use.
// Every rounding point of the unfused chain (LN output, h, GELU output: bf16) is kept, so the fused result equals the
// chain of launches it replaces up to the order of fp32 additions.
#include "gemm_ring.h"

struct PvtMlpArgs {
  const bf16_t* x;
  const float* ln_g;
  const float* ln_b;
  const bf16_t* w1;   // [HD, C]
  const float* b1;    // [HD]
  const float* wd;    // [HD, 9]
  const float* bd;    // [HD]
  const bf16_t* w2;   // [C, HD]
  const float* b2;    // [C]
  const float* bscale;  // [B] or null
  bf16_t* y;
  // backward only
  const bf16_t* dy;   // [B, H*W, C] gradient of y
  bf16_t* dx;         // [B, H*W, C] gradient of x (residual path included)
  bf16_t* xn_out;     // [B, H*W, C]  LN(x)            (operand of the fc1 weight gradient)
  bf16_t* dys_out;    // [B, H*W, C]  s_b * dy          (operand of the fc2 weight gradient; null when bscale is null)
  bf16_t* a_out;      // [B, H*W, HD] GELU output       (operand of the fc2 weight gradient)
  bf16_t* dh_out;     // [B, H*W, HD] gradient of fc1's output (operand of the fc1 weight gradient)
  float* dwd;         // [HD, 9] +=
  float* dbd;         // [HD] +=
  float* dln_g;       // [C] +=
  float* dln_b;       // [C] +=
  int H, W, HD, tiles_x, tiles_per_img;
  float eps;
  int dbg;  // measurement aid (CENET_PVT_DBG): bit mask of phases to skip; 0 in production
};

#ifdef CENET_HOSTSIM_BUILD
#define PVT_WAVE_ID(tid) ((tid) >> 6)
#define PVT_SCHED_FENCE()
#else
#define PVT_WAVE_ID(tid) __builtin_amdgcn_readfirstlane((tid) >> 6)
#define PVT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// 1 / x by v_rcp_f32 (1 ulp): a full-precision division is ten instructions, and the P2 phase is bound by instruction issue
__device__ __forceinline__ float pvt_rcp(float x) {
#ifdef CENET_HOSTSIM_BUILD
  return 1.f / x;
#else
  return __builtin_amdgcn_rcpf(x);
#endif
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, as in dwconv.hip's tiled kernels)
__device__ __forceinline__ float pvt_gelu(float u) {
  const float ax = fabsf(u) * 0.70710678118654752f;
  const float t = pvt_rcp(1.f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.f - poly * fast_exp(-ax * ax);
  return 0.5f * u * (1.f + copysignf(r, u));
}
// GELU(u) and GELU'(u) = Phi(u) + u phi(u) from one exponential
__device__ __forceinline__ void pvt_gelu2(float u, float& act, float& grad) {
  const float ax = fabsf(u) * 0.70710678118654752f;
  const float t = pvt_rcp(1.f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = fast_exp(-0.5f * u * u);
  const float cdf = 0.5f * (1.f + copysignf(1.f - poly * e, u));
  act = u * cdf;
  grad = cdf + u * 0.3989422804014327f * e;
}

template <int LPT>
__device__ __forceinline__ float pvt_subrow_sum(float v) {
#pragma unroll
  for (int o = LPT / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// 64 rows x 64 k of a k-contiguous bf16 matrix (row pitch ld elements) -> k-fast LDS image, one LDS-DMA per wave (8 waves)
__device__ __forceinline__ void pvt_load_kf64(const bf16_t* src, long ld, unsigned char* img, int wave, int lane) {
  const int S = wave * 64 + lane, row = S >> 3, c = (S & 7) ^ kf_key(row);
  ring_glds16(src + (long)row * ld + 8 * c, img + wave * 1024, lane);
}

// LayerNorm of the halo tile into the k-fast image(s) XN[C / 64][PTP][64]; tokens outside the image (or beyond PT) become 0.
// mean / rstd of every halo token go to `stat` (2 floats per token) when it is non-null.
template <int C, int PW, int PT, int PTP>
__device__ __forceinline__ void pvt_ln_tile(const bf16_t* ximg, const float* ln_g, const float* ln_b, float eps, int H, int W,
                                            int y0, int x0, int halo, unsigned char* XN, float* stat, int tid) {
  constexpr int LPT = C / 8, TPP = 512 / LPT;
  const int j = tid % LPT, tsub = tid / LPT;
  float gm[8], bt[8];
  {
    const f4 g0 = ld4(ln_g + 8 * j), g1 = ld4(ln_g + 8 * j + 4), b0 = ld4(ln_b + 8 * j), b1 = ld4(ln_b + 8 * j + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) gm[e] = g0.v[e], gm[4 + e] = g1.v[e], bt[e] = b0.v[e], bt[4 + e] = b1.v[e];
  }
  const float invC = 1.f / C;
#pragma unroll
  for (int p0 = 0; p0 < PTP; p0 += TPP) {  // (unrolled: the loads of all passes are issued before the first reduction)
    const int p = p0 + tsub;
    const int ty = p / PW, tx = p - ty * PW;
    const int iy = y0 - halo + ty, ix = x0 - halo + tx;
    const bool ok = p < PT && iy >= 0 && iy < H && ix >= 0 && ix < W;
    float v[8];
    if (ok) ldv<8>(v, ximg + ((long)iy * W + ix) * C + 8 * j);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    const float mu = pvt_subrow_sum<LPT>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = v[e] - mu;
      q += d * d;
    }
    const float rs = rsqrtf(pvt_subrow_sum<LPT>(q) * invC + eps);
    unsigned o[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float a0 = ok ? (v[2 * h] - mu) * rs * gm[2 * h] + bt[2 * h] : 0.f;
      const float a1 = ok ? (v[2 * h + 1] - mu) * rs * gm[2 * h + 1] + bt[2 * h + 1] : 0.f;
      o[h] = cenet_pack_bf2(a0, a1);
    }
    if (p < PTP) {
      *(uint4*)(XN + (j >> 3) * (PTP * 128) + p * 128 + ((j & 7) ^ kf_key(p)) * 16) = uint4{o[0], o[1], o[2], o[3]};
      if (stat && j == 0) stat[2 * p] = mu, stat[2 * p + 1] = rs;
    }
  }
}

// ============================================================================================================================
// forward
// ============================================================================================================================
template <int C, int TH, int TW>
struct PvtFwdGeo {
  static constexpr int PW = TW + 2, PH = TH + 2, PT = PH * PW, PTP = (PT + 15) / 16 * 16, MTH = PTP / 16;
  static constexpr int NT = TH * TW, MTI = (NT + 15) / 16, SL = TW / 2;
  static constexpr int KB = C / 64, HS = 136;
  static constexpr int XN_B = KB * PTP * 128, A_B = MTI * 16 * 128, W1_B = KB * 8192, W2_B = C * 128, H_B = PTP * HS;
  static constexpr int O_XN = 0, O_A = XN_B, O_W1 = O_A + A_B, O_W2 = O_W1 + W1_B, O_H = O_W2 + W2_B, LDS = O_H + H_B;
};

template <int C, int TH, int TW>
__global__ __launch_bounds__(512, 2) void pvt_mlp_fwd_kernel(PvtMlpArgs a) {
  typedef PvtFwdGeo<C, TH, TW> G;
  constexpr int PW = G::PW, PT = G::PT, PTP = G::PTP, MTH = G::MTH, NT = G::NT, MTI = G::MTI, SL = G::SL, KB = G::KB, HS = G::HS;
  static_assert(C == 64 || C == 128, "channels");
  static_assert(MTI <= 8 && 2 * TH <= 16 && TW % 2 == 0, "tile shape");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS];
  unsigned char* const XN = lds + G::O_XN;
  unsigned char* const AI = lds + G::O_A;
  unsigned char* const W1I = lds + G::O_W1;
  unsigned char* const W2I = lds + G::O_W2;
  unsigned char* const HI = lds + G::O_H;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = PVT_WAVE_ID(tid);
  const cenet_bid bid = cenet_xcd_block();
  const int b = bid.x / a.tiles_per_img, t = bid.x - b * a.tiles_per_img;
  const int y0 = (t / a.tiles_x) * TH, x0 = (t % a.tiles_x) * TW;
  const long img = (long)b * a.H * a.W;  // first token of the image
  const bf16_t* ximg = a.x + img * C;
  const int nslab = a.HD / 64;

  // weight slabs of slab 0 (in flight under the LayerNorm phase)
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) pvt_load_kf64(a.w1 + kb * 64, C, W1I + kb * 8192, wave, lane);
#pragma unroll
  for (int r = 0; r < C / 64; ++r) pvt_load_kf64(a.w2 + (long)(r * 64) * a.HD, a.HD, W2I + r * 8192, wave, lane);

  pvt_ln_tile<C, PW, PT, PTP>(ximg, a.ln_g, a.ln_b, a.eps, a.H, a.W, y0, x0, 1, XN, nullptr, tid);

  f32x4 acc[C / 16];
#pragma unroll
  for (int i = 0; i < C / 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // P2 geometry of this thread: channel pair cp of the slab, strip of SL outputs in row oy
  const int cp = tid & 31, strip = tid >> 5;
  const int oy = strip >> 1, ox0 = (strip & 1) * SL;
  const bool p2_on = strip < 2 * TH;
  // P1 geometry: channel tile ct of the slab, token tiles (wave >> 2), +2, ...
  const int ct = wave & 3;

  ring_wait_vm<0>();
  __syncthreads();

  for (int s = 0; s < nslab; ++s) {
    // depthwise weights / biases of this thread's channel pair: 18 + 2 floats, issued before P1, used in P2
    float wdv[18], bdv[2], b1v[4];
    {
      const float* wp = a.wd + (long)(s * 64 + 2 * cp) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const float2 v = *(const float2*)(wp + 2 * i);  // (typed accesses: byte-wise memcpy made hipcc reassemble every value from bytes)
        wdv[2 * i] = v.x, wdv[2 * i + 1] = v.y;
      }
      bdv[0] = a.bd[s * 64 + 2 * cp], bdv[1] = a.bd[s * 64 + 2 * cp + 1];
      const f4 bb = ld4(a.b1 + s * 64 + ct * 16 + (lane >> 4) * 4);
      b1v[0] = bb.v[0], b1v[1] = bb.v[1], b1v[2] = bb.v[2], b1v[3] = bb.v[3];
    }
    // ---- P1: h^T = W1_slab . xn^T + b1 ------------------------------------------------------------------------------------
    {
      bf16x8 wf[KB * 2];
#pragma unroll
      for (int k = 0; k < KB * 2; ++k) wf[k] = ring_frag_kf(W1I + (k >> 1) * 8192, ct * 16, k & 1, lane);
      for (int tt = wave >> 2; tt < MTH; tt += 2) {
        f32x4 h = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KB * 2; ++k) {
          const bf16x8 xf = ring_frag_kf(XN + (k >> 1) * (PTP * 128), tt * 16, k & 1, lane);
          h = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[k], xf, h, 0, 0, 0);
        }
        const int p = tt * 16 + (lane & 15);
        const int ty = p / PW, tx = p - ty * PW;
        const int iy = y0 - 1 + ty, ix = x0 - 1 + tx;
        const bool ok = p < PT && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        unsigned o[2];
        o[0] = ok ? cenet_pack_bf2(h[0] + b1v[0], h[1] + b1v[1]) : 0u;
        o[1] = ok ? cenet_pack_bf2(h[2] + b1v[2], h[3] + b1v[3]) : 0u;
        *(uint2*)(HI + p * HS + (ct * 16 + (lane >> 4) * 4) * 2) = uint2{o[0], o[1]};
      }
    }
    __syncthreads();  // h complete; the W1 image is free
    if (s + 1 < nslab) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) pvt_load_kf64(a.w1 + (long)(s + 1) * 64 * C + kb * 64, C, W1I + kb * 8192, wave, lane);
    }
    // ---- P2: a = GELU(DW3x3(h) + bd) on the interior tokens ---------------------------------------------------------------
    if (p2_on) {
      float u[SL][2];
#pragma unroll
      for (int i = 0; i < SL; ++i) u[i][0] = bdv[0], u[i][1] = bdv[1];
      const unsigned char* hp = HI + (oy * PW + ox0) * HS + cp * 4;
#pragma unroll
      for (int kxx = 0; kxx < SL + 2; ++kxx) {
        float h0[3], h1[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const unsigned v = *(const unsigned*)(hp + (ky * PW + kxx) * HS);
          h0[ky] = __uint_as_float(v << 16);
          h1[ky] = __uint_as_float(v & 0xFFFF0000u);
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int i = kxx - kx;
          if (i >= 0 && i < SL) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              u[i][0] += wdv[ky * 3 + kx] * h0[ky];
              u[i][1] += wdv[9 + ky * 3 + kx] * h1[ky];
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < SL; ++i) {
        const int row = oy * TW + ox0 + i;
        const unsigned v = cenet_pack_bf2(pvt_gelu(u[i][0]), pvt_gelu(u[i][1]));
        *(unsigned*)(AI + row * 128 + (((cp >> 2) ^ kf_key(row)) * 16) + (cp & 3) * 4) = v;
      }
    }
    ring_wait_vm<0>();  // this wave's part of the next W1 slab (and of this slab's W2, issued one slab ago) has landed
    __syncthreads();    // a complete; h free
    // ---- P3: acc^T += W2[:, slab] . a^T -----------------------------------------------------------------------------------
    if (wave < MTI) {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        const bf16x8 af = ring_frag_kf(AI, wave * 16, kc, lane);
#pragma unroll
        for (int nt = 0; nt < C / 16; ++nt) {
          const bf16x8 wf = ring_frag_kf(W2I + (nt >> 2) * 8192, (nt & 3) * 16, kc, lane);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[nt], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // a and the W2 image are free
    if (s + 1 < nslab) {
#pragma unroll
      for (int r = 0; r < C / 64; ++r)
        pvt_load_kf64(a.w2 + (long)(r * 64) * a.HD + (s + 1) * 64, a.HD, W2I + r * 8192, wave, lane);
    }
  }
  // ---- epilogue: y = x + s_b (acc + b2) -------------------------------------------------------------------------------------
  if (wave < MTI) {
    const int ti = wave * 16 + (lane & 15);
    const int ry = ti / TW, rx = ti - ry * TW;
    const int gy = y0 + ry, gx = x0 + rx;
    if (ti < NT && gy < a.H && gx < a.W) {
      const float sc = a.bscale ? a.bscale[b] : 1.f;
      const long tok = img + (long)gy * a.W + gx;
#pragma unroll
      for (int nt = 0; nt < C / 16; ++nt) {
        const int n0 = nt * 16 + (lane >> 4) * 4;
        const f4 xr = ld4(a.x + tok * C + n0);
        const f4 bb = ld4(a.b2 + n0);
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = xr.v[e] + sc * (acc[nt][e] + bb.v[e]);
        st4(a.y + tok * C + n0, o);
      }
    }
  }
}

// ============================================================================================================================
// backward
// ============================================================================================================================
// One persistent workgroup per CU walks (TH x TW)-token tiles.  Per tile: LayerNorm of the 2-halo tile of x and the scaled
// 1-halo tile of dy are staged through LDS ONCE into MFMA B fragments that stay in registers for all slabs (a wave owns token
// tiles w and w + 8), then per slab of 64 hidden channels:
//   Pa (MFMA)  h^T  = W1_s . xn^T + b1         on the 2-halo tile          -> H2 (bf16 plane, 0 outside the image)
//   Pb (MFMA)  da^T = W2[:, s]^T . dys^T        on the 1-halo tile          -> DU (bf16 plane)
//   Pc (VALU)  u = DW(h) + bd, (a, g') = GELU / GELU'(u), du = da g' on the 1-halo tile (thread = channel pair x column, sliding
//              down); interior tokens: a -> HBM, depthwise weight / bias gradient accumulated in registers
//   Pd (VALU)  dh = DW^T(du) on the interior tokens -> HBM and the k-fast image DH
//   Pe (MFMA)  dxn^T += W1_s^T . dh^T           (fp32 registers across the slabs; runs merged with Pa / Pb of the next slab)
// and per tile the LayerNorm backward + residual produce dx.  Parameter-gradient partials of a workgroup (depthwise weights /
// biases, LayerNorm affine) live in a workgroup-private row of a workspace (plain read-modify-writes, no float atomics in
// HBM); pvt_mlp_fold_kernel adds the rows into the gradient arena.  a, dh, xn and s dy are written once in bf16 for the
// grouped weight-gradient launch of fc1 / fc2 (gemm_group.hip).
template <int C, int TH, int TW>
struct PvtBwdGeo {
  static constexpr int PW1 = TW + 2, PH1 = TH + 2, PT1 = PH1 * PW1, MT1 = PT1 / 16;
  static constexpr int PW2 = TW + 4, PH2 = TH + 4, PT2 = PH2 * PW2, PTP2 = (PT2 + 15) / 16 * 16, MT2 = PTP2 / 16;
  static constexpr int NT = TH * TW, MTI = (NT + 15) / 16, SL = TW / 2;
  static constexpr int KB = C / 64, KS = C / 32, HS = 136;
  static constexpr int H_B = PTP2 * HS, DU_B = PT1 * HS, DH_B = MTI * 16 * 128;
  static constexpr int STG_B = KB * PTP2 * 128;
  static constexpr int S_RAW = (H_B + DU_B + DH_B) > STG_B ? (H_B + DU_B + DH_B) : STG_B;
  static constexpr int S_B = (S_RAW + 1023) / 1024 * 1024;
  static constexpr int W1K_B = KB * 8192, W1R_B = 64 * C * 2, W2R_B = KB * 8192;
  static constexpr int O_S = 0, O_H = 0, O_DU = H_B, O_DH = H_B + DU_B;
  static constexpr int O_W1K = S_B, O_W1R = O_W1K + W1K_B, O_W2R = O_W1R + W1R_B, O_STAT = O_W2R + W2R_B;
  static constexpr int STAT_B = PTP2 * 8, O_SA = O_STAT + STAT_B, SA_B = 2 * 640 * 4, O_LN = O_SA + SA_B, LN_B = 2 * C * 4;
  static constexpr int O_VEC = O_LN + LN_B;  // b1 | bd for all slabs: 2 * HD floats (dynamic part: HD <= PVT_MAX_HD)
};
#define PVT_MAX_HD 1024

// 64 k-rows x BX columns of a row-major bf16 matrix (pitch ld) -> row-fast LDS image [64][BX] (gemm_ring.h layout)
template <int BX>
__device__ __forceinline__ void pvt_load_rf(const bf16_t* src, long ld, unsigned char* img, int wave, int lane) {
  constexpr int CH = BX / 8;
#pragma unroll
  for (int j = 0; j < BX / 64; ++j) {
    const int S = (j * 8 + wave) * 64 + lane;
    const int k = S / CH, c = (S % CH) ^ rf_key<BX>(k);
    ring_glds16(src + (long)k * ld + 8 * c, img + (j * 8 + wave) * 1024, lane);
  }
}

template <int C, int TH, int TW>
__global__ __launch_bounds__(512, 2) void pvt_mlp_bwd_kernel(PvtMlpArgs a, float* __restrict__ ws, int ws_stride, int total_tiles) {
  typedef PvtBwdGeo<C, TH, TW> G;
  constexpr int PW1 = G::PW1, PH1 = G::PH1, PT1 = G::PT1, MT1 = G::MT1, PW2 = G::PW2, PT2 = G::PT2, PTP2 = G::PTP2, MT2 = G::MT2;
  constexpr int NT = G::NT, MTI = G::MTI, SL = G::SL, KB = G::KB, KS = G::KS, HS = G::HS;
  static_assert(C == 64 || C == 128, "channels");
  static_assert(PW1 == 16 && MT2 <= 16 && MT1 <= 16 && MTI <= 8 && 2 * TH <= 16, "tile shape");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[G::O_VEC + 2 * PVT_MAX_HD * 4];
  unsigned char* const STG = lds + G::O_S;
  unsigned char* const H2 = lds + G::O_H;
  unsigned char* const DU = lds + G::O_DU;
  unsigned char* const DH = lds + G::O_DH;
  unsigned char* const W1K = lds + G::O_W1K;
  unsigned char* const W1R = lds + G::O_W1R;
  unsigned char* const W2R = lds + G::O_W2R;
  float* const STAT = (float*)(lds + G::O_STAT);
  float* const SA = (float*)(lds + G::O_SA);
  float* const LNACC = (float*)(lds + G::O_LN);
  float* const B1S = (float*)(lds + G::O_VEC);
  float* const BDS = B1S + a.HD;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = PVT_WAVE_ID(tid);
  const int nslab = a.HD / 64;
  float* const wsrow = ws + (long)blockIdx.x * ws_stride;

  for (int i = tid; i < 2 * 640; i += 512) SA[i] = 0.f;
  for (int i = tid; i < 2 * C; i += 512) LNACC[i] = 0.f;
  for (int i = tid; i < a.HD; i += 512) B1S[i] = a.b1[i], BDS[i] = a.bd[i];

  // thread geometry of the VALU phases
  const int cp = tid & 31;
  const int col = tid >> 5;                                 // Pc: column of the 1-halo tile
  const int oy = (tid >> 5) >> 1, ox0 = ((tid >> 5) & 1) * SL;  // Pd: strip of SL interior tokens in row oy
  const bool pd_on = (tid >> 5) < 2 * TH;

  bool first = true;
  for (int vt = blockIdx.x; vt < total_tiles; vt += gridDim.x) {
    // XCD-aware tile order: the virtual tile list is dealt round-robin over the XCDs (gridDim.x % 8 == 0), so give each XCD a
    // contiguous range of real tiles (neighbouring tiles share their halos in one L2)
    int tile = vt;
    if ((gridDim.x & 7) == 0 && total_tiles >= 64) {
      const int per = total_tiles >> 3, rem = total_tiles & 7, xcd = vt & 7, idx = vt >> 3;
      tile = xcd * per + (xcd < rem ? xcd : rem) + idx;
    }
    const int b = tile / a.tiles_per_img, t = tile - b * a.tiles_per_img;
    const int y0 = (t / a.tiles_x) * TH, x0 = (t % a.tiles_x) * TW;
    const long img = (long)b * a.H * a.W;
    const bf16_t* ximg = a.x + img * C;
    const float sc = a.bscale ? a.bscale[b] : 1.f;

    __syncthreads();  // the previous tile is done with every LDS region
    // weight slabs of slab 0 (in flight under the staging phases)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) pvt_load_kf64(a.w1 + kb * 64, C, W1K + kb * 8192, wave, lane);
    pvt_load_rf<C>(a.w1, C, W1R, wave, lane);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) pvt_load_rf<64>(a.w2 + (long)(kb * 64) * a.HD, a.HD, W2R + kb * 8192, wave, lane);

    // ---- staging 1: LayerNorm of the 2-halo tile -> k-fast image -> B fragments in registers; xn of the interior -> HBM ------
    pvt_ln_tile<C, PW2, PT2, PTP2>(ximg, a.ln_g, a.ln_b, a.eps, a.H, a.W, y0, x0, 2, STG, STAT, tid);
    __syncthreads();
    bf16x8 xf[2][KS];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int tt = wave + 8 * i;
#pragma unroll
      for (int k = 0; k < KS; ++k)
        xf[i][k] = ring_frag_kf(STG + (k >> 1) * (PTP2 * 128), (tt < MT2 ? tt : 0) * 16, k & 1, lane);
    }
    for (int i = tid; i < NT * (C / 8); i += 512) {
      const int ti = i / (C / 8), j = i - ti * (C / 8);
      const int ry = ti / TW, rx = ti - ry * TW;
      const int p2 = (ry + 2) * PW2 + rx + 2;
      if (y0 + ry < a.H && x0 + rx < a.W) {
        const uint4 v = *(const uint4*)(STG + (j >> 3) * (PTP2 * 128) + p2 * 128 + ((j & 7) ^ kf_key(p2)) * 16);
        *(uint4*)(a.xn_out + (img + (long)(y0 + ry) * a.W + x0 + rx) * C + 8 * j) = v;
      }
    }
    __syncthreads();
    // ---- staging 2: s_b dy on the 1-halo tile -> k-fast image -> B fragments; interior -> HBM (operand of the fc2 weight gradient)
    {
      constexpr int LPT = C / 8, TPP = 512 / LPT;
      const int j = tid % LPT, tsub = tid / LPT;
#pragma unroll
      for (int p0 = 0; p0 < PT1; p0 += TPP) {
        const int p = p0 + tsub;
        const int ty = p / PW1, tx = p - ty * PW1;
        const int iy = y0 - 1 + ty, ix = x0 - 1 + tx;
        const bool ok = p < PT1 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (ok) {
          const long off = (img + (long)iy * a.W + ix) * C + 8 * j;
          v = *(const uint4*)(a.dy + off);
          if (a.bscale) {
            float f[8];
            ldv<8>(f, (const bf16_t*)&v);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= sc;
            v = uint4{cenet_pack_bf2(f[0], f[1]), cenet_pack_bf2(f[2], f[3]), cenet_pack_bf2(f[4], f[5]), cenet_pack_bf2(f[6], f[7])};
            if (ty >= 1 && ty <= TH && tx >= 1 && tx <= TW) *(uint4*)(a.dys_out + off) = v;
          }
        }
        if (p < PT1) *(uint4*)(STG + (j >> 3) * (PT1 * 128) + p * 128 + ((j & 7) ^ kf_key(p)) * 16) = v;
      }
    }
    __syncthreads();
    bf16x8 df[2][KS];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int tt = wave + 8 * i;
#pragma unroll
      for (int k = 0; k < KS; ++k)
        df[i][k] = ring_frag_kf(STG + (k >> 1) * (PT1 * 128), (tt < MT1 ? tt : 0) * 16, k & 1, lane);
    }
    // per-lane validity of the tokens this lane stores in Pa / Pb
    bool ok2[2], ok1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p2 = (wave + 8 * i) * 16 + (lane & 15);
      const int ty2 = p2 / PW2, tx2 = p2 - ty2 * PW2;
      const int iy2 = y0 - 2 + ty2, ix2 = x0 - 2 + tx2;
      ok2[i] = p2 < PT2 && iy2 >= 0 && iy2 < a.H && ix2 >= 0 && ix2 < a.W;
      const int p1 = (wave + 8 * i) * 16 + (lane & 15);
      const int ty1 = p1 / PW1, tx1 = p1 - ty1 * PW1;
      const int iy1 = y0 - 1 + ty1, ix1 = x0 - 1 + tx1;
      ok1[i] = p1 < PT1 && iy1 >= 0 && iy1 < a.H && ix1 >= 0 && ix1 < a.W;
    }
    f32x4 acc[C / 16];
#pragma unroll
    for (int i = 0; i < C / 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    ring_wait_vm<0>();
    __syncthreads();  // fragments are in registers: the staging region becomes H2 | DU | DH; slab-0 weights have landed

    // Pe of slab s (dxn^T += W1_s^T . dh^T)
    auto pe = [&]() __attribute__((always_inline)) {
      if (wave < MTI) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          const bf16x8 hf = ring_frag_kf(DH, wave * 16, kc, lane);
#pragma unroll
          for (int nt = 0; nt < C / 16; ++nt) {
            const bf16x8 wf = ring_frag_rf<C>(W1R, nt * 16, kc, lane);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hf, acc[nt], 0, 0, 0);
          }
        }
      }
    };

    float wdv[18];
    {
      const float* wp = a.wd + (long)(2 * cp) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const float2 v = *(const float2*)(wp + 2 * i);
        wdv[2 * i] = v.x, wdv[2 * i + 1] = v.y;
      }
    }
    for (int s = 0; s < nslab; ++s) {
      // depthwise weights of the next slab (registers, one slab ahead)
      float wdn[18];
      {
        const float* wp = a.wd + (long)((s + 1 < nslab ? s + 1 : s) * 64 + 2 * cp) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const float2 v = *(const float2*)(wp + 2 * i);
          wdn[2 * i] = v.x, wdn[2 * i + 1] = v.y;
        }
      }
      // ---- MFMA phase: Pe(s - 1), Pa(s), Pb(s) --------------------------------------------------------------------------------
      if (s > 0) pe();
      if (!(a.dbg & 16))
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        f32x4 h[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        f32x4 d[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const bf16x8 w1f = ring_frag_kf(W1K + (k >> 1) * 8192, ct * 16, k & 1, lane);
          const bf16x8 w2f = ring_frag_rf<64>(W2R + (k >> 1) * 8192, ct * 16, k & 1, lane);
          h[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f, xf[0][k], h[0], 0, 0, 0);
          d[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f, df[0][k], d[0], 0, 0, 0);
          if (wave + 8 < MT2) h[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f, xf[1][k], h[1], 0, 0, 0);
          if (wave + 8 < MT1) d[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f, df[1][k], d[1], 0, 0, 0);
        }
        const int ch0 = ct * 16 + (lane >> 4) * 4;
        const f4 bb = ld4(B1S + s * 64 + ch0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int p = (wave + 8 * i) * 16 + (lane & 15);
          if (wave + 8 * i < MT2) {
            const unsigned o0 = ok2[i] ? cenet_pack_bf2(h[i][0] + bb.v[0], h[i][1] + bb.v[1]) : 0u;
            const unsigned o1 = ok2[i] ? cenet_pack_bf2(h[i][2] + bb.v[2], h[i][3] + bb.v[3]) : 0u;
            *(uint2*)(H2 + p * HS + ch0 * 2) = uint2{o0, o1};
          }
          if (wave + 8 * i < MT1) {
            const unsigned o0 = ok1[i] ? cenet_pack_bf2(d[i][0], d[i][1]) : 0u;
            const unsigned o1 = ok1[i] ? cenet_pack_bf2(d[i][2], d[i][3]) : 0u;
            *(uint2*)(DU + p * HS + ch0 * 2) = uint2{o0, o1};
          }
        }
      }
      __syncthreads();  // B1: h and da complete; W1K / W2R / W1R are free
      if (s + 1 < nslab) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) pvt_load_kf64(a.w1 + (long)(s + 1) * 64 * C + kb * 64, C, W1K + kb * 8192, wave, lane);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
          pvt_load_rf<64>(a.w2 + (long)(kb * 64) * a.HD + (s + 1) * 64, a.HD, W2R + kb * 8192, wave, lane);
      }
      if (s > 0) pvt_load_rf<C>(a.w1 + (long)s * 64 * C, C, W1R, wave, lane);
      // ---- Pc: column `col` of the 1-halo tile, sliding down --------------------------------------------------------------
      if (!(a.dbg & 4)) {
        const float bd0 = BDS[s * 64 + 2 * cp], bd1 = BDS[s * 64 + 2 * cp + 1];
        float wacc[2][10];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int k = 0; k < 10; ++k) wacc[e][k] = 0.f;
        float win[3][3][2];
        const unsigned char* hp = H2 + col * HS + cp * 4;  // 2-halo column col (= 1-halo column col - 1 + 1) .. col + 2
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const unsigned v = *(const unsigned*)(hp + (r * PW2 + kx) * HS);
            win[r][kx][0] = __uint_as_float(v << 16);
            win[r][kx][1] = __uint_as_float(v & 0xFFFF0000u);
          }
        const int gx = x0 - 1 + col;
        const bool col_in = gx >= 0 && gx < a.W;
        const bool col_int = col >= 1 && col <= TW && gx < a.W;
#pragma unroll
        for (int r = 0; r < PH1; ++r) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const unsigned v = *(const unsigned*)(hp + ((r + 2) * PW2 + kx) * HS);
            win[2][kx][0] = __uint_as_float(v << 16);
            win[2][kx][1] = __uint_as_float(v & 0xFFFF0000u);
          }
          float u0 = bd0, u1 = bd1;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              u0 += wdv[ky * 3 + kx] * win[ky][kx][0];
              u1 += wdv[9 + ky * 3 + kx] * win[ky][kx][1];
            }
          float a0, g0, a1, g1;
          pvt_gelu2(u0, a0, g0);
          pvt_gelu2(u1, a1, g1);
          const int gy = y0 - 1 + r;
          const bool in = col_in && gy >= 0 && gy < a.H;
          unsigned* dup = (unsigned*)(DU + (r * PW1 + col) * HS + cp * 4);
          const unsigned dv = *dup;
          const float du0 = in ? __uint_as_float(dv << 16) * g0 : 0.f;
          const float du1 = in ? __uint_as_float(dv & 0xFFFF0000u) * g1 : 0.f;
          *dup = cenet_pack_bf2(du0, du1);
          if (r >= 1 && r <= TH && col_int && gy < a.H) {  // interior token
            if (!(a.dbg & 2)) *(unsigned*)(a.a_out + (img + (long)gy * a.W + gx) * a.HD + s * 64 + 2 * cp) = cenet_pack_bf2(a0, a1);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) {
                wacc[0][ky * 3 + kx] += du0 * win[ky][kx][0];
                wacc[1][ky * 3 + kx] += du1 * win[ky][kx][1];
              }
            wacc[0][9] += du0;
            wacc[1][9] += du1;
          }
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            win[0][kx][0] = win[1][kx][0], win[0][kx][1] = win[1][kx][1];
            win[1][kx][0] = win[2][kx][0], win[1][kx][1] = win[2][kx][1];
          }
        }
        // the sixteen columns meet in LDS (ds_add_f32; the two columns of a wave hit the same address and are serialised there,
        // cheaper than 20 cross-lane exchanges each followed by a divergent branch)
        float* sa = SA + (s & 1) * 640 + 2 * cp * 10;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int k = 0; k < 10; ++k) atomicAdd(&sa[e * 10 + k], wacc[e][k]);
      }
      if (!(a.dbg & 32)) ring_wait_vm<0>();
      __syncthreads();  // B2: du and the slab's depthwise-gradient sums are complete
      // slab sums -> this workgroup's workspace row (plain read-modify-write: the row is private)
      if (!(a.dbg & 1)) {
        float* sa = SA + (s & 1) * 640;
        for (int i = tid; i < 640; i += 512) {
          const int ch = i / 10, k = i - ch * 10;
          const long wi = k < 9 ? (long)(s * 64 + ch) * 9 + k : (long)a.HD * 9 + s * 64 + ch;
          const float v = sa[i];
          wsrow[wi] = first ? v : wsrow[wi] + v;
          sa[i] = 0.f;
        }
      }
      // ---- Pd: dh = DW^T(du) on the interior tokens --------------------------------------------------------------------------
      if (pd_on && !(a.dbg & 8)) {
        float dhv[SL][2];
#pragma unroll
        for (int i = 0; i < SL; ++i) dhv[i][0] = 0.f, dhv[i][1] = 0.f;
        const unsigned char* dp = DU + (oy * PW1 + ox0) * HS + cp * 4;
#pragma unroll
        for (int kxx = 0; kxx < SL + 2; ++kxx) {
          float d0[3], d1[3];
#pragma unroll
          for (int jy = 0; jy < 3; ++jy) {
            const unsigned v = *(const unsigned*)(dp + (jy * PW1 + kxx) * HS);
            d0[jy] = __uint_as_float(v << 16);
            d1[jy] = __uint_as_float(v & 0xFFFF0000u);
          }
#pragma unroll
          for (int jx = 0; jx < 3; ++jx) {
            const int i = kxx - jx;
            if (i >= 0 && i < SL) {
#pragma unroll
              for (int jy = 0; jy < 3; ++jy) {
                dhv[i][0] += wdv[(2 - jy) * 3 + (2 - jx)] * d0[jy];
                dhv[i][1] += wdv[9 + (2 - jy) * 3 + (2 - jx)] * d1[jy];
              }
            }
          }
        }
        const int gy = y0 + oy;
#pragma unroll
        for (int i = 0; i < SL; ++i) {
          const int row = oy * TW + ox0 + i;
          const unsigned v = cenet_pack_bf2(dhv[i][0], dhv[i][1]);
          *(unsigned*)(DH + row * 128 + (((cp >> 2) ^ kf_key(row)) * 16) + (cp & 3) * 4) = v;
          const int gx = x0 + ox0 + i;
          if (gy < a.H && gx < a.W && !(a.dbg & 2)) *(unsigned*)(a.dh_out + (img + (long)gy * a.W + gx) * a.HD + s * 64 + 2 * cp) = v;
        }
      }
#pragma unroll
      for (int i = 0; i < 18; ++i) wdv[i] = wdn[i];
      ring_wait_vm<0>();
      __syncthreads();  // B3: dh complete; H2 / DU free; the next slab's W1K / W2R and this slab's W1R have landed
    }
    pe();
    // ---- LayerNorm backward + residual: dx = dy + rstd (g - mean(g) - xhat mean(g xhat)), g = dxn gamma -------------------------
    if (wave < MTI) {
      const int ti = wave * 16 + (lane & 15);
      const int ry = ti / TW, rx = ti - ry * TW;
      const int gy = y0 + ry, gx = x0 + rx;
      const bool on = ti < NT && gy < a.H && gx < a.W;
      const int p2 = (ry + 2) * PW2 + rx + 2;
      const float mu = on ? STAT[2 * p2] : 0.f, rs = on ? STAT[2 * p2 + 1] : 0.f;
      const long tok = img + (long)(on ? gy : y0) * a.W + (on ? gx : x0);
      float xh[C / 16][4], gv[C / 16][4];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int nt = 0; nt < C / 16; ++nt) {
        const int n0 = nt * 16 + (lane >> 4) * 4;
        const f4 xr = ld4(a.x + tok * C + n0);
        const f4 gm = ld4(a.ln_g + n0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dxn = on ? acc[nt][e] : 0.f;
          xh[nt][e] = (xr.v[e] - mu) * rs;
          gv[nt][e] = dxn * gm.v[e];
          s1 += gv[nt][e];
          s2 += gv[nt][e] * xh[nt][e];
          // LayerNorm affine gradients: column sums over the tile's tokens (the 16 lanes of a group hold 16 tokens)
          float dg = dxn * xh[nt][e], db = dxn;
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) dg += __shfl_xor(dg, o), db += __shfl_xor(db, o);
          if ((lane & 15) == 0) {
            atomicAdd(&LNACC[n0 + e], dg);
            atomicAdd(&LNACC[C + n0 + e], db);
          }
        }
      }
      s1 += __shfl_xor(s1, 16), s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
      const float m1 = s1 * (1.f / C), m2 = s2 * (1.f / C);
      if (on) {
#pragma unroll
        for (int nt = 0; nt < C / 16; ++nt) {
          const int n0 = nt * 16 + (lane >> 4) * 4;
          const f4 dyr = ld4(a.dy + tok * C + n0);
          f4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o.v[e] = dyr.v[e] + rs * (gv[nt][e] - m1 - xh[nt][e] * m2);
          st4(a.dx + tok * C + n0, o);
        }
      }
    }
    first = false;
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += 512) wsrow[(long)a.HD * 10 + i] = LNACC[i];
}

// out += sum over the workgroup rows of the workspace: [HD * 9] depthwise weights | [HD] depthwise bias | [C] LN gamma | [C] LN beta
__global__ __launch_bounds__(256) void pvt_mlp_fold_kernel(const float* __restrict__ ws, int ws_stride, int nrows, int HD, int C,
                                                          float* __restrict__ dwd, float* __restrict__ dbd,
                                                          float* __restrict__ dln_g, float* __restrict__ dln_b) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = HD * 10 + 2 * C;
  if (i >= n) return;
  float s = 0.f;
  for (int r = 0; r < nrows; ++r) s += ws[(long)r * ws_stride + i];
  if (i < HD * 9) dwd[i] += s;
  else if (i < HD * 10) dbd[i - HD * 9] += s;
  else if (i < HD * 10 + C) dln_g[i - HD * 10] += s;
  else dln_b[i - HD * 10 - C] += s;
}

static bool pvt_mlp_geo(int H, int W, int& TH, int& TW) {
  if (W % 14 != 0) return false;
  TW = 14;
  if (H % 8 == 0) TH = 8;
  else if (H % 7 == 0) TH = 7;
  else return false;
  return true;
}

/* 1 when (C, HD, H, W) has a fused instance */
extern "C" int cenet_pvt_mlp_supported(int C, int HD, int H, int W) {
  int TH, TW;
  return (C == 64 || C == 128) && HD % 64 == 0 && HD >= 64 && pvt_mlp_geo(H, W, TH, TW) ? 1 : 0;
}

extern "C" int cenet_pvt_mlp_fwd_bf16(const bf16_t* x, const float* ln_g, const float* ln_b, float eps, const bf16_t* w1,
                                      const float* b1, const float* wd, const float* bd, const bf16_t* w2, const float* b2,
                                      const float* bscale, bf16_t* y, int B, int H, int W, int C, int HD, hipStream_t stream) {
  if (!x || !ln_g || !ln_b || !w1 || !b1 || !wd || !bd || !w2 || !b2 || !y || B <= 0) return CENET_EINVAL;
  int TH, TW;
  if (!cenet_pvt_mlp_supported(C, HD, H, W) || !pvt_mlp_geo(H, W, TH, TW)) return CENET_EUNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)ln_g | (uintptr_t)ln_b | (uintptr_t)b1 |
        (uintptr_t)b2) & 15) != 0 || (((uintptr_t)wd | (uintptr_t)bd) & 7) != 0)
    return CENET_EUNSUPPORTED;
  PvtMlpArgs a = {};
  a.x = x; a.ln_g = ln_g; a.ln_b = ln_b; a.w1 = w1; a.b1 = b1; a.wd = wd; a.bd = bd; a.w2 = w2; a.b2 = b2; a.bscale = bscale;
  a.y = y; a.H = H; a.W = W; a.HD = HD; a.eps = eps;
  a.tiles_x = W / TW;
  a.tiles_per_img = a.tiles_x * (H / TH);
  const dim3 grid(B * a.tiles_per_img);
#define PVT_FWD_GO(C_, TH_)                                                                  \
  if (C == C_ && TH == TH_) {                                                                 \
    CENET_LAUNCH((pvt_mlp_fwd_kernel<C_, TH_, 14>), grid, dim3(512), stream, a);              \
    CENET_CHECK_LAUNCH();                                                                     \
    return CENET_OK;                                                                          \
  }
  PVT_FWD_GO(64, 8)
  PVT_FWD_GO(64, 7)
  PVT_FWD_GO(128, 8)
  PVT_FWD_GO(128, 7)
#undef PVT_FWD_GO
  return CENET_EUNSUPPORTED;
}

/* floats of workspace cenet_pvt_mlp_bwd_bf16 needs */
extern "C" long cenet_pvt_mlp_bwd_ws_floats(int C, int HD) { return 256L * (HD * 10 + 2 * C); }

extern "C" int cenet_pvt_mlp_bwd_bf16(const bf16_t* x, const bf16_t* dy, const float* ln_g, const float* ln_b, float eps,
                                      const bf16_t* w1, const float* b1, const float* wd, const float* bd, const bf16_t* w2,
                                      const float* bscale, bf16_t* dx, bf16_t* xn_out, bf16_t* dys_out, bf16_t* a_out,
                                      bf16_t* dh_out, float* dwd_acc, float* dbd_acc, float* dln_g_acc, float* dln_b_acc,
                                      float* ws, int B, int H, int W, int C, int HD, hipStream_t stream) {
  if (!x || !dy || !ln_g || !ln_b || !w1 || !b1 || !wd || !bd || !w2 || !dx || !xn_out || !a_out || !dh_out || !dwd_acc ||
      !dbd_acc || !dln_g_acc || !dln_b_acc || !ws || B <= 0 || (bscale && !dys_out))
    return CENET_EINVAL;
  int TH, TW;
  if (!cenet_pvt_mlp_supported(C, HD, H, W) || !pvt_mlp_geo(H, W, TH, TW) || HD > PVT_MAX_HD) return CENET_EUNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)xn_out | (uintptr_t)dys_out | (uintptr_t)a_out |
        (uintptr_t)dh_out | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)ln_g | (uintptr_t)ln_b | (uintptr_t)b1) & 15) != 0 ||
      (((uintptr_t)wd | (uintptr_t)bd) & 7) != 0)
    return CENET_EUNSUPPORTED;
  PvtMlpArgs a = {};
  a.x = x; a.dy = dy; a.ln_g = ln_g; a.ln_b = ln_b; a.w1 = w1; a.b1 = b1; a.wd = wd; a.bd = bd; a.w2 = w2; a.bscale = bscale;
  a.dx = dx; a.xn_out = xn_out; a.dys_out = dys_out; a.a_out = a_out; a.dh_out = dh_out;
  a.H = H; a.W = W; a.HD = HD; a.eps = eps;
  {
    static const char* e = getenv("CENET_PVT_DBG");
    a.dbg = e ? atoi(e) : 0;
  }
  a.tiles_x = W / TW;
  a.tiles_per_img = a.tiles_x * (H / TH);
  const int total = B * a.tiles_per_img;
  int nwg = total < 256 ? total : 256;
  if (nwg >= 8) nwg &= ~7;
  const int stride = HD * 10 + 2 * C;
#define PVT_BWD_GO(C_, TH_)                                                                             \
  if (C == C_ && TH == TH_) {                                                                            \
    CENET_LAUNCH((pvt_mlp_bwd_kernel<C_, TH_, 14>), dim3(nwg), dim3(512), stream, a, ws, stride, total); \
    CENET_CHECK_LAUNCH();                                                                                \
  } else
  PVT_BWD_GO(64, 8)
  PVT_BWD_GO(64, 7)
  PVT_BWD_GO(128, 8)
  PVT_BWD_GO(128, 7)
  return CENET_EUNSUPPORTED;
#undef PVT_BWD_GO
  CENET_LAUNCH(pvt_mlp_fold_kernel, dim3(cdiv(stride, 256)), dim3(256), stream, (const float*)ws, stride, nwg, HD, C, dwd_acc,
               dbd_acc, dln_g_acc, dln_b_acc);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
