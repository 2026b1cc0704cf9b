// pvt_mlp.hip — the MLP half of a PVTv2 block (reference pvtv2.py:40-47, 145-149, 364-370) as ONE kernel per pass on bf16
// token tensors:
//     y = x + s_b * ( fc2( GELU( DW3x3( fc1( LN(x) ) ) + bd ) ) + b2 )          (s_b: per-sample DropPath scale or 1)
// Inference: the hidden tensor ([tokens, 8C]: 102 MB per block at 56x56 / batch 32) and the LayerNorm output never reach HBM.
// Training: the kernel also stores what the backward chain reads (LN output + statistics, fc1 output, GELU output), each ONCE
// and from registers / LDS it already holds; the four reads and two launch boundaries of the unfused chain are gone either way.
// (A fused backward was built and measured too: one persistent workgroup per CU recomputing the hidden tile with its halo is
// bound by vector-instruction issue in the depthwise / GELU' phases and lost to the chain of launches, 576 vs 430 us at
// 56x56)
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
  // saved for the backward chain (all null for inference)
  bf16_t* xn_out;     // [B, H*W, C]  LN(x)
  float* mean_out;    // [B * H*W]
  float* rstd_out;    // [B * H*W]
  bf16_t* h_out;      // [B, H*W, HD] fc1 output
  bf16_t* a_out;      // [B, H*W, HD] s_b * GELU output (the DropPath scale rides on the operand of the fc2 weight gradient)
  int H, W, HD, tiles_x, tiles_per_img;
  float eps;
};

#ifdef CENET_HOSTSIM_BUILD
#define PVT_WAVE_ID(tid) ((tid) >> 6)
#else
#define PVT_WAVE_ID(tid) __builtin_amdgcn_readfirstlane((tid) >> 6)
#endif

// 1 / x by v_rcp_f32 (1 ulp): a full-precision division is ten instructions, and the P2 phase is bound by instruction issue
__device__ __forceinline__ float pvt_rcp(float x) {
#ifdef CENET_HOSTSIM_BUILD
  return 1.f / x;
#else
  return __builtin_amdgcn_rcpf(x);
#endif
}
// GELU by common.h's gelu_as (Abramowitz & Stegun 7.1.26 erf, |error| <= 1.5e-7, in the fewest vector instructions)
__device__ __forceinline__ float pvt_gelu(float u) { return gelu_as(u); }
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

// 64 k-rows x BX columns of a row-major bf16 matrix (pitch ld) -> row-fast LDS image [64][BX] (gemm_ring.h layout; 8 waves)
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

// LayerNorm of the halo tile into the k-fast image(s) XN[C / 64][PTP][64]; tokens outside the image (or beyond PT) become 0.
// Interior tokens (halo <= ty < halo + TH, halo <= tx < halo + TW) also go to xn_out / mean_out / rstd_out when those are set
// (pointers to the IMAGE's first token).
template <int C, int PW, int PT, int PTP, int TH, int TW>
__device__ __forceinline__ void pvt_ln_tile(const bf16_t* ximg, const float* ln_g, const float* ln_b, float eps, int H, int W,
                                            int y0, int x0, int halo, unsigned char* XN, bf16_t* xn_out, float* mean_out,
                                            float* rstd_out, int tid) {
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
    if (p < PTP) *(uint4*)(XN + (j >> 3) * (PTP * 128) + p * 128 + ((j & 7) ^ kf_key(p)) * 16) = uint4{o[0], o[1], o[2], o[3]};
    if (xn_out && ok && ty >= halo && ty < halo + TH && tx >= halo && tx < halo + TW) {
      const long tk = (long)iy * W + ix;
      *(uint4*)(xn_out + tk * C + 8 * j) = uint4{o[0], o[1], o[2], o[3]};
      if (j == 0) mean_out[tk] = mu, rstd_out[tk] = rs;
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

  pvt_ln_tile<C, PW, PT, PTP, TH, TW>(ximg, a.ln_g, a.ln_b, a.eps, a.H, a.W, y0, x0, 1, XN, a.xn_out ? a.xn_out + img * C : nullptr,
                                      a.mean_out + img, a.rstd_out + img, tid);

  f32x4 acc[C / 16];
#pragma unroll
  for (int i = 0; i < C / 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float sc_b = a.bscale ? a.bscale[b] : 1.f;
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
    unsigned hkeep[SL], akeep[SL];  // this thread's h / a values (packed channel pair), stored to HBM after the barrier
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
          if (ky == 1 && kxx >= 1 && kxx <= SL) hkeep[kxx - 1] = v;
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
        const float a0 = pvt_gelu(u[i][0]), a1 = pvt_gelu(u[i][1]);
        const unsigned v = cenet_pack_bf2(a0, a1);
        *(unsigned*)(AI + row * 128 + (((cp >> 2) ^ kf_key(row)) * 16) + (cp & 3) * 4) = v;
        // saved for the fc2 weight gradient dW2 = (s_b g)^T a = g^T (s_b a): the scale rides on a, so no scaled copy of g is needed
        akeep[i] = a.bscale ? cenet_pack_bf2(a0 * sc_b, a1 * sc_b) : v;
      }
    }
    ring_wait_vm<0>();  // this wave's part of the next W1 slab (and of this slab's W2, issued one slab ago) has landed
    __syncthreads();    // a complete; h free
    if (a.h_out && p2_on) {  // (after the wait above: the stores' acknowledgements are not waited for before the next slab's)
      const long o = (img + (long)(y0 + oy) * a.W + x0 + ox0) * a.HD + s * 64 + 2 * cp;  // (exact tiles: inside the image)
#pragma unroll
      for (int i = 0; i < SL; ++i) {
        *(unsigned*)(a.h_out + o + (long)i * a.HD) = hkeep[i];
        *(unsigned*)(a.a_out + o + (long)i * a.HD) = akeep[i];
      }
    }
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
      const float sc = sc_b;
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
// backward: two kernels instead of scale_batch + fc2 dgrad GEMM + depthwise/GELU backward + depthwise dgrad + fc1 dgrad GEMM +
// LayerNorm backward.  (One kernel for everything was built and measured first — see the header.  What these two keep from it:
// no halo RECOMPUTATION — the halo tiles of h and gu are read from memory — and two workgroups per CU.)
//   K1  gu = s_b (g . W2[:, slab]) * GELU'(DW3x3(h) + bd)      workgroup = (slab of 64 hidden channels, run of tiles, image):
//       the depthwise weight / bias gradients of the slab stay in registers over the workgroup's tiles
//   K2  dh = DW3x3^T(gu) -> HBM (operand of the fc1 weight gradient);  dxn += dh . W1_slab over the slabs (registers);
//       dx = g + LayerNormBackward(dxn)                        workgroup = tile, slabs walked inside
// ============================================================================================================================
__device__ __forceinline__ float pvt_gelu_grad(float u) { return gelu_as_grad(u); }

struct PvtBwdArgs {
  const bf16_t* g;       // [B, N, C]  gradient of the block output y
  const float* bscale;   // [B] or null
  const bf16_t* w1;      // [HD, C]
  const bf16_t* w2;      // [C, HD]
  const float* wd;       // [HD, 9]
  const float* bd;       // [HD]
  const bf16_t* h;       // [B, N, HD] fc1 output (saved by the forward kernel)
  bf16_t* gu;            // [B, N, HD] gradient of the depthwise conv's output
  bf16_t* dh;            // [B, N, HD] gradient of fc1's output
  float* dwd;            // [HD, 9] +=
  float* dbd;            // [HD] +=
  const bf16_t* x;       // [B, N, C]  block input (LayerNorm input)
  const float* ln_g;
  const float* mean;     // [B * N]
  const float* rstd;
  bf16_t* dx;            // [B, N, C]
  const float* up_scale; // [B] or null: DropPath scale of the branch that produced x ...
  bf16_t* dxs;           // ... and where K2 leaves up_scale_b * dx for that branch's backward (both null: not wanted)
  float* ws;             // K2: [tiles][3 C] partials: LayerNorm gamma / beta gradients, fc2 bias gradient
  int H, W, HD, tiles_x, tiles_per_img, tpw;
};

// (rows x 64 channels) of a token-major bf16 tensor (pitch ld elements) -> LDS plane with 128-byte rows by LDS-DMA; `tok(p)`
// gives the source token of plane row p or -1 for a zero row.  NI = ceil(rows * 8 / 512) instructions per wave.
template <int ROWS, typename F>
__device__ __forceinline__ void pvt_dma_plane(const bf16_t* base, long ld, unsigned char* plane, int wave, int lane, F tok) {
  constexpr int NI = (ROWS * 8 + 511) / 512;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int S = (j * 8 + wave) * 64 + lane, p = S >> 3, c = S & 7;
    if ((j * 8 + wave) * 64 < ROWS * 8) {  // (wave-uniform)
      const long t = p < ROWS ? tok(p) : -1;
      ring_glds16(t >= 0 ? (const void*)(base + t * ld + 8 * c) : (const void*)ring_zero16, plane + (j * 8 + wave) * 1024, lane);
    }
  }
}

template <int C, int TH, int TW>
struct PvtB1Geo {
  static constexpr int PW = TW + 2, PH = TH + 2, PT = PH * PW, NT = TH * TW, MTI = (NT + 15) / 16, SL = TW / 2;
  static constexpr int KB = C / 64, KS = C / 32, HS = 136;
  static constexpr int HP_B = (PT * 128 + 1023) / 1024 * 1024, GS_B = KB * MTI * 16 * 128, W2_B = KB * 8192, DA_B = MTI * 16 * HS;
  static constexpr int O_HP = 0, O_GS = 2 * HP_B, O_W2 = O_GS + GS_B, O_DA = O_W2 + W2_B, LDS = O_DA + DA_B;
};

template <int C, int TH, int TW>
__global__ __launch_bounds__(512, 4) void pvt_mlp_bwd1_kernel(PvtBwdArgs a) {
  typedef PvtB1Geo<C, TH, TW> G;
  constexpr int PW = G::PW, PT = G::PT, NT = G::NT, MTI = G::MTI, SL = G::SL, KB = G::KB, KS = G::KS, HS = G::HS;
  static_assert(G::LDS >= 8 * 32 * 20 * 4, "the end-of-kernel reduction reuses the buffers");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS];
  unsigned char* const GS = lds + G::O_GS;
  unsigned char* const W2R = lds + G::O_W2;
  unsigned char* const DA = lds + G::O_DA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = PVT_WAVE_ID(tid);
  const cenet_bid bid = cenet_xcd_block();
  const int s = bid.x, b = bid.z;
  const long img = (long)b * a.H * a.W;
  const float sc = a.bscale ? a.bscale[b] : 1.f;
  const int cp = tid & 31, strip = tid >> 5;
  const int oy = strip >> 1, ox0 = (strip & 1) * SL;
  const bool on = strip < 2 * TH;
  const int ct = wave & 3;

  // once per workgroup: the W2 slab (row-fast: k = output channel n, columns = the slab's hidden channels), depthwise weights
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) pvt_load_rf<64>(a.w2 + (long)(kb * 64) * a.HD + s * 64, a.HD, W2R + kb * 8192, wave, lane);
  float wdv[18], wacc[2][10];
  {
    const float* wp = a.wd + (long)(s * 64 + 2 * cp) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float2 v = *(const float2*)(wp + 2 * i);
      wdv[2 * i] = v.x, wdv[2 * i + 1] = v.y;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int k = 0; k < 10; ++k) wacc[e][k] = 0.f;
  }
  const float bd0 = a.bd[s * 64 + 2 * cp], bd1 = a.bd[s * 64 + 2 * cp + 1];

  const int t_begin = bid.y * a.tpw;
  const int t_end = t_begin + a.tpw < a.tiles_per_img ? t_begin + a.tpw : a.tiles_per_img;
  auto issue_h = [&](int tile, int buf) __attribute__((always_inline)) {
    const int y0 = (tile / a.tiles_x) * TH, x0 = (tile % a.tiles_x) * TW;
    pvt_dma_plane<PT>(a.h + img * a.HD + s * 64, a.HD, lds + G::O_HP + buf * G::HP_B, wave, lane, [&](int p) -> long {
      const int ty = p / PW, tx = p - ty * PW;
      const int iy = y0 - 1 + ty, ix = x0 - 1 + tx;
      return (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? (long)iy * a.W + ix : -1;
    });
  };
  // g of a tile -> k-fast image (B operand of the dgrad product) by LDS-DMA: row ti of the image = interior token ti
  auto issue_g = [&](int tile) __attribute__((always_inline)) {
    const int y0 = (tile / a.tiles_x) * TH, x0 = (tile % a.tiles_x) * TW;
    constexpr int NI = (MTI * 16 * 8 + 511) / 512;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if ((j * 8 + wave) * 64 < MTI * 16 * 8) {  // (wave-uniform)
          const int S = (j * 8 + wave) * 64 + lane, ti = S >> 3, c = (S & 7) ^ kf_key(ti);
          const int ry = ti / TW, rx = ti - ry * TW;
          const bool ok = ti < NT && y0 + ry < a.H && x0 + rx < a.W;
          const void* src = ok ? (const void*)(a.g + (img + (long)(y0 + ry) * a.W + x0 + rx) * C + kb * 64 + 8 * c)
                               : (const void*)ring_zero16;
          ring_glds16(src, GS + kb * (MTI * 16 * 128) + (j * 8 + wave) * 1024, lane);
        }
      }
  };
  issue_h(t_begin, 0);
  issue_g(t_begin);
  int buf = 0;
  for (int tile = t_begin; tile < t_end; ++tile, buf ^= 1) {
    const int y0 = (tile / a.tiles_x) * TH, x0 = (tile % a.tiles_x) * TW;
    ring_wait_vm<0>();
    __syncthreads();  // this tile's h plane and g image (and, first time, the W2 slab) are in LDS
    if (tile + 1 < t_end) issue_h(tile + 1, buf ^ 1);
    // ---- da^T[slab channels, tokens] = W2[:, slab]^T . (s g)^T --------------------------------------------------------------
    for (int tt = wave >> 2; tt < MTI; tt += 2) {
      f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const bf16x8 wf = ring_frag_rf<64>(W2R + (k >> 1) * 8192, ct * 16, k & 1, lane);
        const bf16x8 gf = ring_frag_kf(GS + (k >> 1) * (MTI * 16 * 128), tt * 16, k & 1, lane);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, gf, d, 0, 0, 0);
      }
      const int p = tt * 16 + (lane & 15);
      *(uint2*)(DA + p * HS + (ct * 16 + (lane >> 4) * 4) * 2) =
          uint2{cenet_pack_bf2(sc * d[0], sc * d[1]), cenet_pack_bf2(sc * d[2], sc * d[3])};
    }
    __syncthreads();
    if (tile + 1 < t_end) issue_g(tile + 1);  // (the image is free; lands under the depthwise phase)
    // ---- gu = da * GELU'(DW3x3(h) + bd); depthwise weight / bias gradient -------------------------------------------------------
    if (on) {
      const unsigned char* hp = lds + G::O_HP + buf * G::HP_B + (oy * PW + ox0) * 128 + cp * 4;
      float win[3][3][2];
#pragma unroll
      for (int kxx = 0; kxx < 2; ++kxx)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const unsigned v = *(const unsigned*)(hp + (ky * PW + kxx) * 128);
          win[ky][kxx][0] = __uint_as_float(v << 16);
          win[ky][kxx][1] = __uint_as_float(v & 0xFFFF0000u);
        }
      bf16_t* const gup = a.gu + (img + (long)(y0 + oy) * a.W + x0 + ox0) * a.HD + s * 64 + 2 * cp;
#pragma unroll
      for (int i = 0; i < SL; ++i) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const unsigned v = *(const unsigned*)(hp + (ky * PW + i + 2) * 128);
          win[ky][2][0] = __uint_as_float(v << 16);
          win[ky][2][1] = __uint_as_float(v & 0xFFFF0000u);
        }
        float u0 = bd0, u1 = bd1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            u0 += wdv[ky * 3 + kx] * win[ky][kx][0];
            u1 += wdv[9 + ky * 3 + kx] * win[ky][kx][1];
          }
        const int row = oy * TW + ox0 + i;
        const unsigned dv = *(const unsigned*)(DA + row * HS + cp * 4);
        // (tiles are exact — pvt_mlp_geo: H % TH == 0, W % TW == 0 — so interior tokens are always inside the image)
        const float g0 = __uint_as_float(dv << 16) * pvt_gelu_grad(u0);
        const float g1 = __uint_as_float(dv & 0xFFFF0000u) * pvt_gelu_grad(u1);
        *(unsigned*)(gup + (long)i * a.HD) = cenet_pack_bf2(g0, g1);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            wacc[0][ky * 3 + kx] += g0 * win[ky][kx][0];
            wacc[1][ky * 3 + kx] += g1 * win[ky][kx][1];
          }
        wacc[0][9] += g0;
        wacc[1][9] += g1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          win[ky][0][0] = win[ky][1][0], win[ky][0][1] = win[ky][1][1];
          win[ky][1][0] = win[ky][2][0], win[ky][1][1] = win[ky][2][1];
        }
      }
    }
    __syncthreads();  // the g image, the da plane and this h plane are free
  }
  // ---- the slab's depthwise gradients: two strips per wave fold by a shuffle, the eight waves meet in LDS, one atomic each ------
  float* red = (float*)lds;
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int k = 0; k < 10; ++k) wacc[e][k] += __shfl_xor(wacc[e][k], 32);
  if (lane < 32) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int k = 0; k < 10; ++k) red[(wave * 32 + cp) * 20 + e * 10 + k] = wacc[e][k];
  }
  __syncthreads();
  for (int i = tid; i < 640; i += 512) {
    const int ch = i / 10, k = i - ch * 10;  // channel 0..63 of the slab, tap 0..8 or 9 = bias
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[(w * 32 + (ch >> 1)) * 20 + (ch & 1) * 10 + k];
    if (k < 9) atomicAdd(&a.dwd[(long)(s * 64 + ch) * 9 + k], v);
    else atomicAdd(&a.dbd[s * 64 + ch], v);
  }
}

template <int C, int TH, int TW>
struct PvtB2Geo {
  static constexpr int PW = TW + 2, PH = TH + 2, PT = PH * PW, NT = TH * TW, MTI = (NT + 15) / 16, SL = TW / 2;
  static constexpr int GU_B = (PT * 128 + 1023) / 1024 * 1024, W1_B = 64 * C * 2, DH_B = MTI * 16 * 128;
  static constexpr int O_GU = 0, O_W1 = 2 * GU_B, O_DH = O_W1 + 2 * W1_B, O_LN = O_DH + DH_B, LDS = O_LN + 3 * C * 4;
};

template <int C, int TH, int TW>
__global__ __launch_bounds__(512, C == 64 ? 4 : 2) void pvt_mlp_bwd2_kernel(PvtBwdArgs a) {
  typedef PvtB2Geo<C, TH, TW> G;
  constexpr int PW = G::PW, PT = G::PT, NT = G::NT, MTI = G::MTI, SL = G::SL;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS];
  unsigned char* const DH = lds + G::O_DH;
  float* const LNS = (float*)(lds + G::O_LN);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = PVT_WAVE_ID(tid);
  const cenet_bid bid = cenet_xcd_block();
  const int b = bid.x / a.tiles_per_img, t = bid.x - b * a.tiles_per_img;
  const int y0 = (t / a.tiles_x) * TH, x0 = (t % a.tiles_x) * TW;
  const long img = (long)b * a.H * a.W;
  const int nslab = a.HD / 64;
  const int cp = tid & 31, strip = tid >> 5;
  const int oy = strip >> 1, ox0 = (strip & 1) * SL;
  const bool on = strip < 2 * TH;
  for (int i = tid; i < 3 * C; i += 512) LNS[i] = 0.f;
  const float sc = a.bscale ? a.bscale[b] : 1.f;

  auto issue = [&](int s, int buf) __attribute__((always_inline)) {
    pvt_dma_plane<PT>(a.gu + img * a.HD + s * 64, a.HD, lds + G::O_GU + buf * G::GU_B, wave, lane, [&](int p) -> long {
      const int ty = p / PW, tx = p - ty * PW;
      const int iy = y0 - 1 + ty, ix = x0 - 1 + tx;
      return (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? (long)iy * a.W + ix : -1;
    });
    pvt_load_rf<C>(a.w1 + (long)s * 64 * C, C, lds + G::O_W1 + buf * G::W1_B, wave, lane);
  };
  f32x4 acc[C / 16];
#pragma unroll
  for (int i = 0; i < C / 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  issue(0, 0);
  float wdv[18];
  {
    const float* wp = a.wd + (long)(2 * cp) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float2 v = *(const float2*)(wp + 2 * i);
      wdv[2 * i] = v.x, wdv[2 * i + 1] = v.y;
    }
  }
  for (int s = 0, buf = 0; s < nslab; ++s, buf ^= 1) {
    ring_wait_vm<0>();
    __syncthreads();  // slab s (gu plane, W1 image) has landed; everybody is done with the other buffer and with DH
    if (s + 1 < nslab) issue(s + 1, buf ^ 1);
    float wdn[18];  // depthwise weights of the next slab (one slab ahead)
    {
      const float* wp = a.wd + (long)((s + 1 < nslab ? s + 1 : s) * 64 + 2 * cp) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const float2 v = *(const float2*)(wp + 2 * i);
        wdn[2 * i] = v.x, wdn[2 * i + 1] = v.y;
      }
    }
    // ---- dh = DW3x3^T(gu): the mirrored taps ---------------------------------------------------------------------------------
    if (on) {
      float dhv[SL][2];
#pragma unroll
      for (int i = 0; i < SL; ++i) dhv[i][0] = 0.f, dhv[i][1] = 0.f;
      const unsigned char* gp = lds + G::O_GU + buf * G::GU_B + (oy * PW + ox0) * 128 + cp * 4;
#pragma unroll
      for (int kxx = 0; kxx < SL + 2; ++kxx) {
        float d0[3], d1[3];
#pragma unroll
        for (int jy = 0; jy < 3; ++jy) {
          const unsigned v = *(const unsigned*)(gp + (jy * PW + kxx) * 128);
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
      bf16_t* const dhp = a.dh + (img + (long)(y0 + oy) * a.W + x0 + ox0) * a.HD + s * 64 + 2 * cp;
#pragma unroll
      for (int i = 0; i < SL; ++i) {
        const int row = oy * TW + ox0 + i;
        const unsigned v = cenet_pack_bf2(dhv[i][0], dhv[i][1]);
        *(unsigned*)(DH + row * 128 + (((cp >> 2) ^ kf_key(row)) * 16) + (cp & 3) * 4) = v;
        *(unsigned*)(dhp + (long)i * a.HD) = v;  // (exact tiles: always inside the image)
      }
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) wdv[i] = wdn[i];
    __syncthreads();  // dh image complete
    // ---- dxn^T[C, tokens] += W1_slab^T . dh^T ------------------------------------------------------------------------------------
    if (wave < MTI) {
      const unsigned char* W1R = lds + G::O_W1 + buf * G::W1_B;
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
  }
  // ---- LayerNorm backward + the residual connection: dx = g + rstd (q - mean(q) - xhat mean(q xhat)), q = dxn gamma -----------
  if (wave < MTI) {
    const int ti = wave * 16 + (lane & 15);
    const int ry = ti / TW, rx = ti - ry * TW;
    const int gy = y0 + ry, gx = x0 + rx;
    const bool in = ti < NT && gy < a.H && gx < a.W;
    const long tok = img + (long)(in ? gy : y0) * a.W + (in ? gx : x0);
    const float mu = a.mean[tok], rs = a.rstd[tok];
    float xh[C / 16][4], qv[C / 16][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int nt = 0; nt < C / 16; ++nt) {
      const int n0 = nt * 16 + (lane >> 4) * 4;
      const f4 xr = ld4(a.x + tok * C + n0);
      const f4 gm = ld4(a.ln_g + n0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dxn = in ? acc[nt][e] : 0.f;
        xh[nt][e] = (xr.v[e] - mu) * rs;
        qv[nt][e] = dxn * gm.v[e];
        s1 += qv[nt][e];
        s2 += qv[nt][e] * xh[nt][e];
        // affine gradients: sums over the 16 tokens of this lane group, then over the workgroup in LDS
        float dg = dxn * xh[nt][e], db = dxn;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) dg += __shfl_xor(dg, o), db += __shfl_xor(db, o);
        if ((lane & 15) == 0) {
          atomicAdd(&LNS[n0 + e], dg);
          atomicAdd(&LNS[C + n0 + e], db);
        }
      }
    }
    s1 += __shfl_xor(s1, 16), s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
    const float m1 = s1 * (1.f / C), m2 = s2 * (1.f / C);
#pragma unroll
    for (int nt = 0; nt < C / 16; ++nt) {
      const int n0 = nt * 16 + (lane >> 4) * 4;
      const f4 gr = ld4(a.g + tok * C + n0);
      if (in) {
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = gr.v[e] + rs * (qv[nt][e] - m1 - xh[nt][e] * m2);
        st4(a.dx + tok * C + n0, o);
        if (a.dxs) {  // (what scale_batch(dx, up_scale) would write: the rounded dx times the scale)
          const float us = a.up_scale[b];
#pragma unroll
          for (int e = 0; e < 4; ++e) o.v[e] = us * cenet_bf2f(cenet_f2bf(o.v[e]));
          st4(a.dxs + tok * C + n0, o);
        }
      }
      // fc2 bias gradient: column sums of s_b g (the grouped weight-gradient launch gets the UNscaled g, whose row sums
      // would miss the DropPath scale)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = in ? sc * gr.v[e] : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((lane & 15) == 0) atomicAdd(&LNS[2 * C + n0 + e], v);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 3 * C; i += 512) a.ws[(long)bid.x * 3 * C + i] = LNS[i];
}

// dln_g[c] += sum_rows ws[row][c], dln_b[c] += sum_rows ws[row][C + c], db2[c] += sum_rows ws[row][2 C + c]
__global__ __launch_bounds__(256) void pvt_mlp_lnfold_kernel(const float* __restrict__ ws, int rows, int C, float* __restrict__ dg,
                                                            float* __restrict__ db, float* __restrict__ db2) {
  __shared__ float red[256];
  const int c = blockIdx.x, tid = threadIdx.x;  // one workgroup per output
  float s = 0.f;
  for (int r = tid; r < rows; r += 256) s += ws[(long)r * 3 * C + c];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    if (c < C) dg[c] += red[0];
    else if (c < 2 * C) db[c - C] += red[0];
    else if (db2) db2[c - 2 * C] += red[0];
  }
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
                                      const float* bscale, bf16_t* y, bf16_t* xn_out, float* mean_out, float* rstd_out,
                                      bf16_t* h_out, bf16_t* a_out, int B, int H, int W, int C, int HD, hipStream_t stream) {
  if (!x || !ln_g || !ln_b || !w1 || !b1 || !wd || !bd || !w2 || !b2 || !y || B <= 0) return CENET_EINVAL;
  const int nsave = (xn_out != nullptr) + (mean_out != nullptr) + (rstd_out != nullptr) + (h_out != nullptr) + (a_out != nullptr);
  if (nsave != 0 && nsave != 5) return CENET_EINVAL;  // all of the saved tensors or none
  if ((((uintptr_t)xn_out | (uintptr_t)h_out | (uintptr_t)a_out) & 15) != 0) return CENET_EUNSUPPORTED;
  int TH, TW;
  if (!cenet_pvt_mlp_supported(C, HD, H, W) || !pvt_mlp_geo(H, W, TH, TW)) return CENET_EUNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)ln_g | (uintptr_t)ln_b | (uintptr_t)b1 |
        (uintptr_t)b2) & 15) != 0 || (((uintptr_t)wd | (uintptr_t)bd) & 7) != 0)
    return CENET_EUNSUPPORTED;
  PvtMlpArgs a = {};
  a.x = x; a.ln_g = ln_g; a.ln_b = ln_b; a.w1 = w1; a.b1 = b1; a.wd = wd; a.bd = bd; a.w2 = w2; a.b2 = b2; a.bscale = bscale;
  a.y = y; a.xn_out = xn_out; a.mean_out = mean_out; a.rstd_out = rstd_out; a.h_out = h_out; a.a_out = a_out;
  a.H = H; a.W = W; a.HD = HD; a.eps = eps;
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
extern "C" long cenet_pvt_mlp_bwd_ws_floats(int B, int H, int W, int C) {
  int TH, TW;
  if (!pvt_mlp_geo(H, W, TH, TW)) return 0;
  return (long)B * (W / TW) * (H / TH) * 3 * C;
}

/* backward of cenet_pvt_mlp_fwd_bf16 from its saved tensors, two launches (+ a fold of the LayerNorm affine gradients):
 * gu, dh: [B, H*W, HD] outputs (dh: the operand of the fc1 weight gradient; the fc2 weight gradient is g^T a with the saved,
 * already scaled a); dx = g + dLayerNorm (and, with up_scale [B] + dxs, dxs = up_scale_b * dx: x was itself residual +
 * up_scale_b * branch, and that branch's backward starts from the scaled gradient); dwd / dbd / dln_g / dln_b and db2 (fc2 bias: column sums of s_b g; may be NULL) are
 * ADDED into. */
extern "C" int cenet_pvt_mlp_bwd_bf16(const bf16_t* g, const float* bscale, const bf16_t* w1, const bf16_t* w2, const float* wd,
                                      const float* bd, const bf16_t* h, const bf16_t* x, const float* ln_g, const float* mean,
                                      const float* rstd, bf16_t* gu, bf16_t* dh, bf16_t* dx, const float* up_scale, bf16_t* dxs,
                                      float* dwd_acc, float* dbd_acc, float* dln_g_acc, float* dln_b_acc, float* db2_acc,
                                      float* ws, int B, int H, int W, int C, int HD, hipStream_t stream) {
  if (!g || !w1 || !w2 || !wd || !bd || !h || !x || !ln_g || !mean || !rstd || !gu || !dh || !dx || !dwd_acc || !dbd_acc ||
      !dln_g_acc || !dln_b_acc || !ws || B <= 0 || (up_scale != nullptr) != (dxs != nullptr))
    return CENET_EINVAL;
  int TH, TW;
  if (!cenet_pvt_mlp_supported(C, HD, H, W) || !pvt_mlp_geo(H, W, TH, TW)) return CENET_EUNSUPPORTED;
  if ((((uintptr_t)g | (uintptr_t)h | (uintptr_t)x | (uintptr_t)gu | (uintptr_t)dh | (uintptr_t)dx | (uintptr_t)dxs |
        (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)ln_g) & 15) != 0 || (((uintptr_t)wd | (uintptr_t)bd) & 7) != 0)
    return CENET_EUNSUPPORTED;
  PvtBwdArgs a = {};
  a.g = g; a.bscale = bscale; a.w1 = w1; a.w2 = w2; a.wd = wd; a.bd = bd; a.h = h; a.gu = gu; a.dh = dh;
  a.dwd = dwd_acc; a.dbd = dbd_acc; a.x = x; a.ln_g = ln_g; a.mean = mean; a.rstd = rstd; a.dx = dx; a.ws = ws;
  a.up_scale = up_scale; a.dxs = dxs;
  a.H = H; a.W = W; a.HD = HD;
  a.tiles_x = W / TW;
  a.tiles_per_img = a.tiles_x * (H / TH);
  {
    static const char* e = getenv("CENET_PVT_TPW");
    a.tpw = e ? atoi(e) : (a.tiles_per_img >= 16 ? 4 : 2);
    if (a.tpw < 1) a.tpw = 1;
    if (a.tpw > a.tiles_per_img) a.tpw = a.tiles_per_img;
  }
  // K1 and K2 meet only through gu [B, N, HD] (token-indexed) and are tiled INDEPENDENTLY: K2 (workgroup = tile, the LayerNorm
  // partials in ws are per K2 tile) keeps the forward's tile; K1 (workgroup = slab x run of tiles x image) takes 4-row tiles where
  // they divide the map — more, shorter runs: 94 -> 75 us at 28 x 28 (the forward kernel with 4-row tiles is SLOWER: 90 -> 116 us,
  // its depthwise phase then fills half the threads)
  static const char* k1e = getenv("CENET_PVT_K1_TH");  // measurement aid: 0 = K1 on the forward's tile
  const int k1th = (k1e ? atoi(k1e) : 4) == 4 && TH == 7 && H % 4 == 0 ? 4 : TH;  // (at 56 x 56 the 8-row tile wins: 97 vs 126 us)
  PvtBwdArgs a1 = a;
  a1.tiles_per_img = a.tiles_x * (H / k1th);
  {  // runs of tiles per (slab, image): as few as still give ~1 024 workgroups (two rounds at two per CU) — a run keeps the W2 slab,
     // the depthwise weights and the weight-gradient accumulators of its slab; 4 / 2 tiles per run measured 0.04 ms slower than 7 / 7
    int runs = 1024 / ((HD / 64) * B);
    if (runs < 1) runs = 1;
    a1.tpw = cdiv(a1.tiles_per_img, runs);
  }
  {
    static const char* e = getenv("CENET_PVT_TPW");
    if (e) a1.tpw = atoi(e);
    if (a1.tpw < 1) a1.tpw = 1;
    if (a1.tpw > a1.tiles_per_img) a1.tpw = a1.tiles_per_img;
  }
  const dim3 g1(HD / 64, cdiv(a1.tiles_per_img, a1.tpw), B), g2(B * a.tiles_per_img);
  if (g1.y > 65535 || g1.z > 65535) return CENET_EUNSUPPORTED;
  static const char* only_e = getenv("CENET_PVT_ONLY");  // measurement aid: 1 / 2 = launch only that kernel
  const int only = only_e ? atoi(only_e) : 0;
#define PVT_BWD_GO(C_, TH_)                                                                              \
  if (C == C_ && TH == TH_) {                                                                             \
    if (only != 2) {                                                                                      \
      if (k1th == 4) CENET_LAUNCH((pvt_mlp_bwd1_kernel<C_, 4, 14>), g1, dim3(512), stream, a1);           \
      else CENET_LAUNCH((pvt_mlp_bwd1_kernel<C_, TH_, 14>), g1, dim3(512), stream, a1);                   \
    }                                                                                                     \
    if (only != 1) CENET_LAUNCH((pvt_mlp_bwd2_kernel<C_, TH_, 14>), g2, dim3(512), stream, a);            \
  } else
  PVT_BWD_GO(64, 8)
  PVT_BWD_GO(64, 7)
  PVT_BWD_GO(128, 8)
  PVT_BWD_GO(128, 7)
  return CENET_EUNSUPPORTED;
#undef PVT_BWD_GO
  CENET_LAUNCH(pvt_mlp_lnfold_kernel, dim3(3 * C), dim3(256), stream, (const float*)ws, (int)g2.x, C, dln_g_acc, dln_b_acc, db2_acc);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
