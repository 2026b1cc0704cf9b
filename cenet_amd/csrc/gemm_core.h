// gemm_core.h — batched strided GEMM / implicit-GEMM convolution core on the gfx950 matrix cores (v3).
//
// One kernel family serves every dense contraction of the CENet hot path:
//   * nn.Linear fwd/bwd in token layout                      (pvtv2.py:41,45,90,98,106; multihead_diffattn.py:79-81,126)
//   * 1x1 convs in NCHW fwd/bwd                              (cfam.py:149,158,299,302; nlb.py:106-115,142; blocks.py:178,320; dseb.py:164)
//   * dense k x k convs as implicit GEMM (fwd, dgrad, wgrad)  (pvtv2.py:164,67; unet.py:156-197; blocks.py:211)
//   * materialised attention products for large head dims
//
// C[b] (+)= epilogue( alpha * sum_kb A[b,kb] (MxK) * B[b,kb] (KxN) )
// A is a plain strided matrix; B is plain or an on-the-fly im2col / transposed-gather view of an image tensor; the
// epilogue can scatter through a col2im map (data-gradient of strided convolutions).
//
// Structure (per 256-thread workgroup = 4 waves in a 2x2 grid, tile BM x BN, K step 32):
//   * operands are read from HBM in whichever orientation is contiguous (kfast: 16-byte loads along k when alignment
//     allows, else dwords; mfast: lane-contiguous dwords), one K-tile AHEAD into registers (software double buffering:
//     the loads of tile t+1 fly under the MFMAs of tile t), then written k-contiguous into LDS;
//   * fragments are fetched with 16-byte ds_read_b128: for fp32, lane (r = lane&15, q = lane>>4) owns k' = 8q..8q+7 and
//     MFMA step s multiplies slot s of A with slot s of B (a permutation of the k order, which a sum does not care about);
//   * OpT = float : fp32 tensors in HBM, v_mfma_f32_16x16x4_f32, exact fp32 FMA chain (parity mode)
//     OpT = bf16  : bf16 tensors in HBM (throughput mode: activations, activation gradients and the bf16 shadow of the
//                   weights), copied to LDS without conversion (8-byte k-contiguous quads, or row pairs re-packed with one
//                   v_perm per dword when the operand is row-contiguous), v_mfma_f32_16x16x32_bf16, fp32 accumulate; C / R
//                   are bf16 except in atomic (split-K / scatter-add) epilogues, which always add into fp32;
//   * SWAP: when C is row-major the MFMA is issued as (B-fragment, A-fragment) so a lane's four accumulator registers are
//     four CONSECUTIVE COLUMNS of C and the epilogue moves 16 bytes per instruction (bias / residual / store);
//   * im2col addressing never divides per element: forward/dgrad threads walk (kx,ky,ci) with carries, the
//     weight-gradient view reads a per-workgroup LDS table of patch decompositions.
#pragma once
#include "common.h"
#include "../../include/cenet_hip.h"

struct GemmArgs {
  cenet_mat_t A, B;
  cenet_epi_t E;
  int M, N, K, nkb, splits, nb_inner;
  int avec, bvec, cvec;  // quad (16-byte fp32 / 8-byte bf16) global access is legal for A / B staging / the epilogue
  int cvec8;             // bf16 C: 16-byte stores along its contiguous axis are legal (the staged epilogue of the ring kernel)
  int apair, bpair;      // bf16, row-contiguous operand: adjacent rows may be fetched as 4-byte pairs
  // LDS-DMA ring kernel (gemm_ring.h) on operands whose pitch / extent / base is not a multiple of 8 elements: 16-byte
  // chunks then start at 2-byte aligned addresses (the hardware takes them) and the last chunk of a row runs into the
  // next row; a_end / b_end = one past the last element of the operand (a chunk that would cross it is fetched by hand)
  int ring_unal;
  const void *a_end, *b_end;
};

__device__ __forceinline__ unsigned f2bf_bits(float f) { return cenet_f2bf(f); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) { return cenet_pack_bf2(lo, hi); }

// LDS row pitch in elements for a K step of KT: KT + 4 floats (144 / 272-byte rows) or KT + 8 bf16 (80 / 144-byte rows)
template <typename OpT, int KT> struct OpTraits { static constexpr int PITCH = KT + (sizeof(OpT) == 4 ? 4 : 8); };
#define BK 32  // K step of the implicit-GEMM (im2col) instances; plain bf16 instances may use 64

struct KEntry {
  int off;     // patch side: ci*sci
  int dy, dx;  // patch side: ky*dil, kx*dil ; pixel side: py*stride-pad (or py+pad), px*stride-pad (or px+pad)
};

template <int KDIM>
__device__ __forceinline__ long plain_off(const cenet_mat_t& d, int r, int c) {
  if (d.kinner > 0) {
    if (KDIM == 1) {
      int ko = c / d.kinner, ki = c - ko * d.kinner;
      return (long)r * d.sr + (long)ko * d.sk_outer + (long)ki * d.sc;
    } else {
      int ko = r / d.kinner, ki = r - ko * d.kinner;
      return (long)ko * d.sk_outer + (long)ki * d.sr + (long)c * d.sc;
    }
  }
  return (long)r * d.sr + (long)c * d.sc;
}

__device__ __forceinline__ KEntry im2col_entry(const cenet_mat_t& d, int idx, bool is_patch) {
  KEntry e;
  if (is_patch) {
    int kk = d.KH * d.KW;
    int ci = idx / kk;
    int rem = idx - ci * kk;
    int ky = rem / d.KW, kx = rem - ky * d.KW;
    e.dy = ky * d.dil;
    e.dx = kx * d.dil;
    e.off = ci * (int)d.sci;
  } else {
    int py = idx / d.Pw, px = idx - py * d.Pw;
    if (!d.transposed) {
      e.dy = py * d.stride - d.pad;
      e.dx = px * d.stride - d.pad;
    } else {
      e.dy = py + d.pad;
      e.dx = px + d.pad;
    }
    e.off = 0;
  }
  return e;
}

template <typename GT>
__device__ __forceinline__ float im2col_load(const cenet_mat_t& d, const GT* base, const KEntry& pat, const KEntry& pix) {
  int iy, ix;
  if (!d.transposed) {
    iy = pix.dy + pat.dy;
    ix = pix.dx + pat.dx;
  } else {
    int ty = pix.dy - pat.dy, tx = pix.dx - pat.dx;
    if (ty < 0 || tx < 0) return 0.f;
    if (d.stride == 1) {
      iy = ty;
      ix = tx;
    } else {
      iy = ty / d.stride;
      ix = tx / d.stride;
      if (iy * d.stride != ty || ix * d.stride != tx) return 0.f;
    }
  }
  if (iy < 0 || iy >= d.Hs || ix < 0 || ix >= d.Ws) return 0.f;
  return ldf(base + (long)pat.off + (long)iy * d.sy + (long)ix * d.sx);
}

// write NV consecutive-k values of one LDS row (NV multiple of 4)
template <typename OpT, int NV>
__device__ __forceinline__ void lds_put(OpT* dst, const float* v) {
  if (sizeof(OpT) == 4) {
#pragma unroll
    for (int q = 0; q < NV / 4; ++q) memcpy((char*)dst + 16 * q, v + 4 * q, 16);
  } else {
#pragma unroll
    for (int q = 0; q < NV / 4; ++q) {
      unsigned pk[2] = {pack_bf2(v[4 * q], v[4 * q + 1]), pack_bf2(v[4 * q + 2], v[4 * q + 3])};
      memcpy((char*)dst + 8 * q, pk, 8);
    }
  }
}
template <typename OpT>
__device__ __forceinline__ void lds_put1(OpT* dst, float v) {
  if (sizeof(OpT) == 4) memcpy(dst, &v, 4);
  else *dst = (OpT)f2bf_bits(v);
}


// Staging geometry of one PLAIN operand, hoisted out of the K loop: a thread's registers walk either the operand's rows
// (k-contiguous forms) or its k index (row-contiguous form) with a constant element stride, so the loop only adds.
struct PlainStage {
  long toff;    // thread's element offset inside a K tile (relative to the tile origin at row 0, k0)
  long dj;      // element stride between consecutive registers (register quads for the 16-byte form)
  unsigned ok;  // bit j: register j's row lies inside the matrix (k-contiguous forms); bit 0: the thread's row does
                // (pair form: bit 1 = the second row of the pair does)
  int kk;       // k index inside the tile of register 0
  int pair;     // bf16, row-contiguous operand: the thread owns TWO adjacent rows (4-byte loads) and NX/2 consecutive k
};
// staging registers of one operand: fp32 values (fp32 operands; bf16 fallback paths, converted when they enter LDS) or raw
// 32-bit words (bf16 fast paths: packed k-quads / row pairs)
template <int N>
union StageRegs {
  float f[N];
  unsigned u[N];
};
// (lo halves, hi halves) of two packed words: w0 = (a0 | b0 << 16), w1 = (a1 | b1 << 16)  ->  (a0 | a1 << 16), (b0 | b1 << 16)
__device__ __forceinline__ unsigned pack_lo16(unsigned w0, unsigned w1) {
#ifdef CENET_HOSTSIM_BUILD
  return (w0 & 0xFFFFu) | (w1 << 16);
#else
  return __builtin_amdgcn_perm(w1, w0, 0x05040100u);
#endif
}
__device__ __forceinline__ unsigned pack_hi16(unsigned w0, unsigned w1) {
#ifdef CENET_HOSTSIM_BUILD
  return (w0 >> 16) | (w1 & 0xFFFF0000u);
#else
  return __builtin_amdgcn_perm(w1, w0, 0x07060302u);
#endif
}

// rs / ks: element strides along the operand's row (M or N) index and along k; x0 / X: tile origin and matrix extent
template <int BX, int NX, int KT>
__device__ __forceinline__ PlainStage plain_stage(long rs, long ks, int kfast, int vec, int x0, int X, int tid, int pair) {
  PlainStage s;
  s.ok = 0;
  s.pair = 0;
  if (!kfast && pair) {
    const int row = 2 * (tid % (BX / 2)), kq = tid / (BX / 2);
    s.pair = 1;
    s.kk = kq * (NX / 2);
    s.toff = (long)(x0 + row) * rs + (long)s.kk * ks;
    s.dj = ks;
    s.ok = (x0 + row < X ? 1u : 0u) | (x0 + row + 1 < X ? 2u : 0u);
    return s;
  }
  if (kfast) {
    // vector form (vec = 4 or 8 elements per access): KT/vec lanes per row; scalar form: KT lanes per row
    const int lpr = vec ? KT / vec : KT;
    const int rstep = 256 / lpr, r0 = tid / lpr;
    s.kk = vec ? (tid % lpr) * vec : (tid % lpr);
    s.toff = (long)(x0 + r0) * rs + (long)s.kk * ks;
    s.dj = (long)rstep * rs;
    const int nreg = vec ? NX / vec : NX;
    for (int j = 0; j < nreg; ++j)
      if (x0 + r0 + j * rstep < X) s.ok |= 1u << j;
  } else {
    const int row = tid % BX, kq = tid / BX;
    s.kk = kq * NX;  // NX == KT / (256 / BX) consecutive k per thread
    s.toff = (long)(x0 + row) * rs + (long)s.kk * ks;
    s.dj = ks;
    s.ok = (x0 + row < X) ? 1u : 0u;
  }
  return s;
}

// tile = operand base of this batch / K-batch advanced to k0; klim = K - k0; interior: whole BX x BK tile inside the matrix
// bf16 operands: k-contiguous quads as raw 8-byte words, row pairs as raw 4-byte words; the scalar forms convert to fp32
template <int NX>
__device__ __forceinline__ void plain_fetch(const PlainStage& s, const bf16_t* tile, int klim, int kfast, int vec,
                                            bool interior, StageRegs<NX>& rg) {
  const bf16_t* p = tile + s.toff;
  // the interior forms are branch-free runs of loads (a per-load runtime condition makes the compiler branch around every
  // load and wait for each one: cdna_hip_programming.md "Three .s-level traps" (c))
  if (s.pair) {
    if (interior) {
#pragma unroll
      for (int j = 0; j < NX / 2; ++j, p += s.dj) memcpy(&rg.u[j], p, 4);
      return;
    }
    const int lim = klim - s.kk;
    const bool both = (s.ok & 3u) == 3u, one = (s.ok & 1u) != 0;
#pragma unroll
    for (int j = 0; j < NX / 2; ++j, p += s.dj) {
      unsigned w = 0;
      if (j < lim) {
        if (both) memcpy(&w, p, 4);
        else if (one) w = *p;
      }
      rg.u[j] = w;
    }
    return;
  }
  if (kfast && vec == 8) {  // 16-byte accesses: 8 consecutive k
    if (interior) {
#pragma unroll
      for (int j = 0; j < NX / 8; ++j, p += s.dj) memcpy(&rg.u[4 * j], p, 16);
      return;
    }
    const bool kok = s.kk < klim;
#pragma unroll
    for (int j = 0; j < NX / 8; ++j, p += s.dj) {
      if (kok && ((s.ok >> j) & 1u)) memcpy(&rg.u[4 * j], p, 16);
      else rg.u[4 * j] = rg.u[4 * j + 1] = rg.u[4 * j + 2] = rg.u[4 * j + 3] = 0u;
    }
    return;
  }
  if (kfast && vec) {
    if (interior) {
#pragma unroll
      for (int j = 0; j < NX / 4; ++j, p += s.dj) memcpy(&rg.u[2 * j], p, 8);
      return;
    }
    const bool kok = s.kk < klim;
#pragma unroll
    for (int j = 0; j < NX / 4; ++j, p += s.dj) {
      if (kok && ((s.ok >> j) & 1u)) memcpy(&rg.u[2 * j], p, 8);
      else rg.u[2 * j] = rg.u[2 * j + 1] = 0u;
    }
    return;
  }
  if (interior) {
#pragma unroll
    for (int j = 0; j < NX; ++j, p += s.dj) rg.f[j] = cenet_bf2f(*p);
    return;
  }
  if (kfast) {
    const bool kok = s.kk < klim;
#pragma unroll
    for (int j = 0; j < NX; ++j, p += s.dj) rg.f[j] = (kok && ((s.ok >> j) & 1u)) ? cenet_bf2f(*p) : 0.f;
  } else {
    const int lim = (s.ok & 1u) ? klim - s.kk : 0;
#pragma unroll
    for (int j = 0; j < NX; ++j, p += s.dj) rg.f[j] = (j < lim) ? cenet_bf2f(*p) : 0.f;
  }
}
template <int NX>
__device__ __forceinline__ void plain_fetch(const PlainStage& s, const float* tile, int klim, int kfast, int vec,
                                            bool interior, StageRegs<NX>& rg) {
  float* r = rg.f;
  const float* p = tile + s.toff;
  if (interior) {
    if (kfast && vec) {
#pragma unroll
      for (int j = 0; j < NX / 4; ++j, p += s.dj) memcpy(&r[4 * j], p, 16);
    } else {
#pragma unroll
      for (int j = 0; j < NX; ++j, p += s.dj) r[j] = *p;
    }
    return;
  }
  if (kfast) {
    const bool kok = s.kk < klim;
    if (vec) {
#pragma unroll
      for (int j = 0; j < NX / 4; ++j, p += s.dj) {
        if (kok && ((s.ok >> j) & 1u)) memcpy(&r[4 * j], p, 16);
        else r[4 * j] = r[4 * j + 1] = r[4 * j + 2] = r[4 * j + 3] = 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NX; ++j, p += s.dj) r[j] = (kok && ((s.ok >> j) & 1u)) ? *p : 0.f;
    }
  } else {
    const int lim = (s.ok & 1u) ? klim - s.kk : 0;
#pragma unroll
    for (int j = 0; j < NX; ++j, p += s.dj) r[j] = (j < lim) ? *p : 0.f;
  }
}

// Epilogue shared by the GEMM kernels: acc[i][j] is the 16x16 fragment (i, j) of this wave's (BM/2)x(BN/2) quadrant
// (waves 2x2 over the tile).  cstrip: LDS scratch of 4*16*(BN/2+1) floats, used by the atomic split-K path (!SWAP).
// STAGED (ring kernel, bf16 C): interior tiles go through an LDS image of the whole tile, so that a wave's store instruction
// writes whole rows (16 bytes per lane, 128 / 256 contiguous bytes per row) instead of 16 rows x 32 bytes; cstrip then holds
// at least BO * (BI + 8) * 2 bytes.  Same values, same single rounding as the direct path.
template <typename GT, int BM, int BN, bool SWAP, bool STAGED = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[BM / 32][BN / 32], float* cstrip, int m0, int n0,
                                              int bo, int bi, int batch, int wave, int lane) {
  constexpr int MI = BM / 32, NJ = BN / 32;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  // ---- epilogue ----
  // !SWAP: acc[i][j][r] = C[row = .. + fq*4 + r][col = .. + fr]   (4 consecutive rows per lane)
  //  SWAP: acc[i][j][r] = C[row = .. + fr][col = .. + fq*4 + r]   (4 consecutive columns per lane)
  const cenet_epi_t& E = g.E;
  float* const Cf = (float*)E.C + (long)bo * E.scb + (long)bi * E.scb2;  // atomic epilogues: C is fp32 whatever OpT is
  GT* const Cb = (GT*)E.C + (long)bo * E.scb + (long)bi * E.scb2;
  if (!SWAP && E.atomic && !E.cmode && E.scc == 1) {
    // split-K accumulation into a row-major C: float atomics reach their chip-wide rate only as 256 contiguous bytes per
    // wave instruction, but an MFMA accumulator register spans 4 rows x 16 floats.  Each wave therefore transposes one
    // 16-row strip at a time through a private LDS strip and issues the atomics with lane = column.
    constexpr int WN = BN / 2;
    float* strip = cstrip + wave * (16 * (WN + 1));
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) strip[(fq * 4 + r) * (WN + 1) + j * 16 + fr] = acc[i][j][r] * E.alpha;
      __syncthreads();
      const int row0 = m0 + wm * (BM / 2) + i * 16, col0 = n0 + wn * WN;
      for (int idx = lane; idx < 16 * WN; idx += 64) {
        const int r = idx / WN, c = idx - r * WN;
        if (row0 + r < g.M && col0 + c < g.N) {
          // atomic == 2: the only contribution to this element (no split): a plain fp32 store, C needs no zero fill
          if (E.atomic == 2) Cf[(long)(row0 + r) * E.scr + col0 + c] = strip[r * (WN + 1) + c];
          else atomicAdd(&Cf[(long)(row0 + r) * E.scr + col0 + c], strip[r * (WN + 1) + c]);
        }
      }
    }
    return;
  }
  const GT* Rb = E.R ? (const GT*)E.R + (long)bo * E.srb + (long)bi * E.srb2 : nullptr;
  const int bsr = E.bscale ? E.bscale_rows : 0;  // > 0: per-sample scale looked up by row (flat batch)
  const float bs = (E.bscale && bsr == 0) ? E.bscale[batch] : 1.f;
  const int rwave = m0 + wm * (BM / 2) + (SWAP ? fr : fq * 4);
  const int cwave = n0 + wn * (BN / 2) + (SWAP ? fq * 4 : fr);
  if (g.cvec && !E.atomic && !E.cmode && E.act == ACT_NONE && m0 + BM <= g.M && n0 + BN <= g.N) {
    // interior tile of the common case (plain store, optional bias / per-batch scale / residual): 16 bytes per lane and
    // fragment, no bounds tests, one 64-bit offset per lane and constant strides per fragment (the general version below
    // spends ~70 VALU instructions per fragment; on the K <= 128 GEMMs of stages 1-2 that made the epilogue the longest part
    // of the kernel)
    const float scale = E.alpha * bs;
    GT* crow = Cb + (long)rwave * E.scr + (long)cwave * E.scc;
    const GT* rrow = Rb ? Rb + (long)rwave * E.srr + (long)cwave * E.src : nullptr;
    const long ci = 16 * E.scr, cj = 16 * E.scc, ri = 16 * E.srr, rj = 16 * E.src;
    const bool bias_vec = E.bias && (SWAP != (bool)E.bias_on_row) && (((uintptr_t)E.bias & 15) == 0);
    constexpr int BI = SWAP ? BN : BM, BO = SWAP ? BM : BN, TP = BI + 8;  // staged image: [BO][BI] + 16-byte row padding
    const bool staged = STAGED && sizeof(GT) == 2 && g.cvec8;
    bf16_t* const tile = (bf16_t*)cstrip;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int rbase = rwave + i * 16, cbase = cwave + j * 16;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
        if (E.bias) {
          if (bias_vec) {  // the bias index runs along the lane's four elements
            float bb[4];
            memcpy(bb, E.bias + (SWAP ? cbase : rbase), 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] * E.alpha + bb[r];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              v[r] = v[r] * E.alpha + E.bias[E.bias_on_row ? (SWAP ? rbase : rbase + r) : (SWAP ? cbase + r : cbase)];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= bs;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= scale;
        }
        if (bsr > 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= E.bscale[(SWAP ? rbase : rbase + r) / bsr];
        }
        if (rrow) {
          float rr[4];
          ld4v(rr, rrow + i * ri + j * rj);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rr[r];
        }
        if (staged) {
          const int in0 = (SWAP ? wn * (BN / 2) + j * 16 : wm * (BM / 2) + i * 16) + fq * 4;
          const int out0 = (SWAP ? wm * (BM / 2) + i * 16 : wn * (BN / 2) + j * 16) + fr;
          st4v(tile + out0 * TP + in0, v);
        } else {
          st4v(crow + i * ci + j * cj, v);
        }
      }
    if (staged) {
      __syncthreads();
      const long so = SWAP ? E.scr : E.scc;
      bf16_t* cb = (bf16_t*)Cb + (long)(SWAP ? m0 : n0) * so + (SWAP ? n0 : m0);
      constexpr int CPR = BI / 8;
      for (int c = wave * 64 + lane; c < BO * CPR; c += 256) {
        const int out = c / CPR, ch = c - out * CPR;
        unsigned q[4];
        memcpy(q, tile + out * TP + 8 * ch, 16);
        memcpy(cb + (long)out * so + 8 * ch, q, 16);
      }
    }
    return;
  }
  if (g.cvec && !E.atomic && !E.cmode && E.act == ACT_NONE) {
    // common case (plain store, optional bias / per-batch scale / residual), 16 bytes per lane and fragment.  This loop
    // nest MUST stay small enough to unroll fully: if it does not, acc[][] is indexed dynamically, lives in scratch
    // memory and is spilled there on every K step.
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int rbase = rwave + i * 16, cbase = cwave + j * 16;
        const bool full = SWAP ? (rbase < g.M && cbase + 3 < g.N) : (rbase + 3 < g.M && cbase < g.N);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * E.alpha;
        if (full) {
          if (E.bias) {
            if (SWAP != (bool)E.bias_on_row && ((uintptr_t)E.bias & 15) == 0) {  // bias index runs along the lane's 4 elements
              float bb[4];
              memcpy(bb, E.bias + (SWAP ? cbase : rbase), 16);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bb[r];
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += E.bias[E.bias_on_row ? (SWAP ? rbase : rbase + r) : (SWAP ? cbase + r : cbase)];
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= bsr > 0 ? E.bscale[(SWAP ? rbase : rbase + r) / bsr] : bs;
          if (Rb) {
            float rr[4];
            ld4v(rr, Rb + (long)rbase * E.srr + (long)cbase * E.src);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rr[r];
          }
          st4v(Cb + (long)rbase * E.scr + (long)cbase * E.scc, v);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = SWAP ? rbase : rbase + r, col = SWAP ? cbase + r : cbase;
            if (row < g.M && col < g.N) {
              float t = v[r];
              if (E.bias) t += E.bias[E.bias_on_row ? row : col];
              t *= bsr > 0 ? E.bscale[row / bsr] : bs;
              if (Rb) t += ldf(Rb + (long)row * E.srr + (long)col * E.src);
              stf(Cb + (long)row * E.scr + (long)col * E.scc, t);
            }
          }
        }
      }
    return;
  }
  // everything else (activations, col2im scatter, strided atomics, unaligned C): a RUNTIME loop over the fragments; the
  // fragment is picked with a chain of selects on static indices so that acc[][] stays in registers
  for (int f = 0; f < MI * NJ; ++f) {
    f32x4 t4 = acc[0][0];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (f == i * NJ + j) t4 = acc[i][j];
    const int rbase = rwave + (f / NJ) * 16, cbase = cwave + (f % NJ) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = SWAP ? rbase : rbase + r, col = SWAP ? cbase + r : cbase;
      if (row < g.M && col < g.N) {
        float v = t4[r] * E.alpha;
        if (E.cmode) {  // col2im scatter: row = (ci,ky,kx), col = (py,px)
          const int kkw = E.cKH * E.cKW;
          const int ci = row / kkw, rem = row - ci * kkw;
          const int ky = rem / E.cKW, kx = rem - ky * E.cKW;
          const int py = col / E.cPw, px = col - py * E.cPw;
          const int iy = py * E.cstride - E.cpad + ky, ix = px * E.cstride - E.cpad + kx;
          if (iy >= 0 && iy < E.cHs && ix >= 0 && ix < E.cWs) {
            const long o = (long)ci * E.csci + (long)iy * E.csy + (long)ix * E.csx;
            if (E.atomic) atomicAdd(&Cf[o], v);
            else stf(&Cb[o], v);
          }
        } else if (E.atomic == 2) {
          Cf[(long)row * E.scr + (long)col * E.scc] = v;
        } else if (E.atomic) {
          atomicAdd(&Cf[(long)row * E.scr + (long)col * E.scc], v);
        } else {
          if (E.bias) v += E.bias[E.bias_on_row ? row : col];
          v = act_fwd(E.act, v, E.slope);
          v *= bsr > 0 ? E.bscale[row / bsr] : bs;
          if (Rb) v += ldf(Rb + (long)row * E.srr + (long)col * E.src);
          stf(&Cb[(long)row * E.scr + (long)col * E.scc], v);
        }
      }
    }
  }
}

template <typename OpT, int BM, int BN, bool B_IM2COL, bool SWAP, int KT>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128 ? 2 : 1)) void gemm_kernel(GemmArgs g) {  // 128x128: keep two workgroups per CU
  static_assert(KT == 32 || (KT == 64 && !B_IM2COL), "K step: 32, or 64 for plain operands");
  constexpr int P = OpTraits<OpT, KT>::PITCH;
  constexpr bool BF = (sizeof(OpT) == 2);
  constexpr int MI = BM / 32, NJ = BN / 32;              // 16x16 tiles per wave in each direction
  constexpr int NA = BM * KT / 256, NB = BN * KT / 256;  // prefetch registers per thread
  typedef OpT GT;                                          // element type of A, B (and of C / R in non-atomic epilogues)
  __shared__ __attribute__((aligned(16))) OpT As[BM * P];
  __shared__ __attribute__((aligned(16))) OpT Bs[BN * P];
  __shared__ KEntry ntab[B_IM2COL ? BN : 1];
  __shared__ float cstrip[SWAP ? 1 : 4 * 16 * (BN / 2 + 1)];  // per-wave epilogue strips (atomic split-K path)

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (linear id % 8, MI355X_MICROARCH.md "Workgroup
  // dispatch"), so re-number them such that each XCD owns a contiguous range of tiles: neighbours along N re-read their
  // A rows, and the tiles of one split-K chunk their operands, from that XCD's own L2.  Speed only, never correctness.
  int bx, by, bz;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    const int T = gx * gy * (int)gridDim.z;
    int L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if (T >= 64) {
      const int per = T >> 3, rem = T & 7, xcd = L & 7, idx = L >> 3;
      L = xcd * per + (xcd < rem ? xcd : rem) + idx;
    }
    bx = L % gx;
    const int t = L / gx;
    by = t % gy;
    bz = t / gy;
  }
  const int batch = bz / g.splits, split = bz - batch * g.splits;
  const int bo = batch / g.nb_inner, bi = batch - bo * g.nb_inner;
  const int m0 = by * BM, n0 = bx * BN;

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ktiles = (g.K + KT - 1) / KT;
  const int total = g.nkb * ktiles;
  const int chunk = (total + g.splits - 1) / g.splits;
  const int it0 = split * chunk;
  const int it1 = (it0 + chunk < total) ? it0 + chunk : total;

  // thread -> tile-element maps
  //   kfast scalar : kk = tid % 32,  rows r = tid / 32 + 8 j
  //   kfast vec    : k4 = tid % 8,   rows r = tid / 8 + 32 j   (4 consecutive k per register quad)
  //   mfast        : row = tid % BMN, kq = tid / BMN ; k = kq * KPT + j  (KPT consecutive k per thread)
  constexpr int A_KPT = KT / (256 / BM), B_KPT = KT / (256 / BN);
  static_assert(BM <= 256 && BN <= 256 && NA % 4 == 0 && NB % 4 == 0 && NA <= 64 && NB <= 64, "unsupported tile");
  StageRegs<NA> rga;
  StageRegs<NB> rgb;
  float* const ra = rga.f;
  float* const rb = rgb.f;
  // plain operands without the (ko,ki) split of k use hoisted pointer walks; the split form keeps per-element addressing
  const bool a_fast = g.A.kinner == 0, b_fast = !B_IM2COL && g.B.kinner == 0;
  // 16-byte (8-element) staging needs at least 8 elements per thread and operand; else the quad form
  const int avec = (g.avec == 8 && NA % 8 != 0) ? 4 : g.avec, bvec = (g.bvec == 8 && NB % 8 != 0) ? 4 : g.bvec;
  const PlainStage sa = plain_stage<BM, NA, KT>(g.A.sr, g.A.sc, g.A.kfast, avec, m0, g.M, tid, BF && a_fast && g.apair);
  const PlainStage sb = plain_stage<BN, NB, KT>(g.B.sc, g.B.sr, g.B.kfast, bvec, n0, g.N, tid, BF && b_fast && g.bpair);
  const bool a_in = m0 + BM <= g.M, b_in = n0 + BN <= g.N;

  KEntry nent;
  nent.off = nent.dy = nent.dx = 0;
  bool n_ok = true;
  if (B_IM2COL) {
    int ncol = g.B.kfast ? 0 : n0 + (tid % BN);
    n_ok = ncol < g.N;
    if (!g.B.kfast) nent = im2col_entry(g.B, n_ok ? ncol : 0, !g.B.patch_is_row);
  }

  auto fetch = [&](int it) __attribute__((always_inline)) {
    const int kb = it / ktiles;
    const int k0 = (it - kb * ktiles) * KT;
    const GT* baseA = (const GT*)g.A.ptr + (long)bo * g.A.sb + (long)bi * g.A.sb2 + (long)kb * g.A.skb;
    const GT* baseB = (const GT*)g.B.ptr + (long)bo * g.B.sb + (long)bi * g.B.sb2 + (long)kb * g.B.skb;
    const int klim = g.K - k0;
    if (a_fast) {
      plain_fetch<NA>(sa, baseA + (long)k0 * g.A.sc, klim, g.A.kfast, avec, a_in && klim >= KT, rga);
    } else if (g.A.kfast) {
      const int kk = tid % KT, r0 = tid / KT;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int row = r0 + j * (256 / KT);
        ra[j] = (m0 + row < g.M && k0 + kk < g.K) ? ldf(baseA + plain_off<1>(g.A, m0 + row, k0 + kk)) : 0.f;
      }
    } else {
      const int row = tid % BM, kq = tid / BM;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int kk = kq * A_KPT + j;
        ra[j] = (m0 + row < g.M && k0 + kk < g.K) ? ldf(baseA + plain_off<1>(g.A, m0 + row, k0 + kk)) : 0.f;
      }
    }
    if (!B_IM2COL) {
      if (b_fast) {
        plain_fetch<NB>(sb, baseB + (long)k0 * g.B.sr, klim, g.B.kfast, bvec, b_in && klim >= KT, rgb);
      } else if (g.B.kfast) {
        const int kk = tid % KT, c0 = tid / KT;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          int col = c0 + j * (256 / KT);
          rb[j] = (n0 + col < g.N && k0 + kk < g.K) ? ldf(baseB + plain_off<0>(g.B, k0 + kk, n0 + col)) : 0.f;
        }
      } else {
        const int col = tid % BN, kq = tid / BN;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          int kk = kq * B_KPT + j;
          rb[j] = (n0 + col < g.N && k0 + kk < g.K) ? ldf(baseB + plain_off<0>(g.B, k0 + kk, n0 + col)) : 0.f;
        }
      }
    } else {
      const cenet_mat_t& d = g.B;
      if (d.kfast) {
        // weight-gradient view: this thread's k is ONE pixel of the tile, its NB columns are patch elements whose
        // (ci,ky,kx) decomposition sits in the per-workgroup LDS table ntab (n0 is fixed for the workgroup)
        const int kk = tid & 31, c0 = tid >> 5;
        const bool kok = k0 + kk < g.K;
        const KEntry pe = im2col_entry(d, kok ? k0 + kk : 0, false);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          int col = c0 + j * 8;
          float v = 0.f;
          if (kok && n0 + col < g.N) v = im2col_load(d, baseB, ntab[col], pe);
          rb[j] = v;
        }
      } else {
        // forward / data-gradient view: this thread owns ONE pixel (nent) and B_KPT consecutive patch indices; walk
        // (kx,ky,ci) with carries instead of decomposing every index
        const int kq = tid / BN;
        int e = k0 + kq * B_KPT;
        const int kkw = d.KH * d.KW;
        int ci = e / kkw;
        int rem = e - ci * kkw;
        int ky = rem / d.KW, kx = rem - ky * d.KW;
        const bool fast = !(d.transposed && d.stride != 1);
        if (fast) {
          const int sg = d.transposed ? -1 : 1;
          const int dsx = sg * d.dil * (int)d.sx, dsy = sg * d.dil * (int)d.sy, dd = sg * d.dil;
          int iy = nent.dy + sg * ky * d.dil, ix = nent.dx + sg * kx * d.dil;
          int off = ci * (int)d.sci + iy * (int)d.sy + ix * (int)d.sx;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const bool ok = n_ok && (e + j < g.K) && iy >= 0 && iy < d.Hs && ix >= 0 && ix < d.Ws;
            rb[j] = ok ? ldf(baseB + off) : 0.f;
            ++kx; ix += dd; off += dsx;
            if (kx == d.KW) {
              kx = 0; ix -= d.KW * dd; off -= d.KW * dsx;
              ++ky; iy += dd; off += dsy;
              if (ky == d.KH) {
                ky = 0; iy -= d.KH * dd; off += (int)d.sci - d.KH * dsy;
              }
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            float v = 0.f;
            if (n_ok && e + j < g.K) v = im2col_load(d, baseB, im2col_entry(d, e + j, true), nent);
            rb[j] = v;
          }
        }
      }
    }
  };

  auto store_lds = [&]() __attribute__((always_inline)) {
    constexpr int LV = KT / 4, RV = 256 / LV;  // 16-byte form: lanes per row, rows per pass
    constexpr int RS = 256 / KT;               // scalar form: rows per pass
    if (BF && sa.pair) {
      // row pairs -> two k-contiguous runs of NA / 2 elements (one v_perm per dword)
      const int row = 2 * (tid % (BM / 2)), kq = tid / (BM / 2);
      unsigned lo[NA / 4 > 0 ? NA / 4 : 1], hi[NA / 4 > 0 ? NA / 4 : 1];
#pragma unroll
      for (int q = 0; q < NA / 4; ++q) {
        lo[q] = pack_lo16(rga.u[2 * q], rga.u[2 * q + 1]);
        hi[q] = pack_hi16(rga.u[2 * q], rga.u[2 * q + 1]);
      }
      memcpy(&As[row * P + kq * (NA / 2)], lo, NA);
      memcpy(&As[(row + 1) * P + kq * (NA / 2)], hi, NA);
    } else if (g.A.kfast) {
      if (BF && avec == 8) {
        constexpr int L8 = KT / 8, R8 = 256 / L8;
        const int k8 = (tid % L8) * 8, r0 = tid / L8;
#pragma unroll
        for (int j = 0; j < NA / 8; ++j) memcpy(&As[(r0 + j * R8) * P + k8], &rga.u[4 * j], 16);
      } else if (avec) {
        const int k4 = (tid % LV) * 4, r0 = tid / LV;
        if (BF) {
#pragma unroll
          for (int j = 0; j < NA / 4; ++j) memcpy(&As[(r0 + j * RV) * P + k4], &rga.u[2 * j], 8);
        } else {
#pragma unroll
          for (int j = 0; j < NA / 4; ++j) lds_put<OpT, 4>(&As[(r0 + j * RV) * P + k4], &ra[4 * j]);
        }
      } else {
        const int kk = tid % KT, r0 = tid / KT;
#pragma unroll
        for (int j = 0; j < NA; ++j) lds_put1<OpT>(&As[(r0 + j * RS) * P + kk], ra[j]);
      }
    } else {
      const int row = tid % BM, kq = tid / BM;
      lds_put<OpT, NA>(&As[row * P + kq * A_KPT], ra);
    }
    if (BF && !B_IM2COL && sb.pair) {
      const int col = 2 * (tid % (BN / 2)), kq = tid / (BN / 2);
      unsigned lo[NB / 4 > 0 ? NB / 4 : 1], hi[NB / 4 > 0 ? NB / 4 : 1];
#pragma unroll
      for (int q = 0; q < NB / 4; ++q) {
        lo[q] = pack_lo16(rgb.u[2 * q], rgb.u[2 * q + 1]);
        hi[q] = pack_hi16(rgb.u[2 * q], rgb.u[2 * q + 1]);
      }
      memcpy(&Bs[col * P + kq * (NB / 2)], lo, NB);
      memcpy(&Bs[(col + 1) * P + kq * (NB / 2)], hi, NB);
    } else if (g.B.kfast) {
      if (BF && !B_IM2COL && bvec == 8) {
        constexpr int L8 = KT / 8, R8 = 256 / L8;
        const int k8 = (tid % L8) * 8, c0 = tid / L8;
#pragma unroll
        for (int j = 0; j < NB / 8; ++j) memcpy(&Bs[(c0 + j * R8) * P + k8], &rgb.u[4 * j], 16);
      } else if (!B_IM2COL && bvec) {
        const int k4 = (tid % LV) * 4, c0 = tid / LV;
        if (BF) {
#pragma unroll
          for (int j = 0; j < NB / 4; ++j) memcpy(&Bs[(c0 + j * RV) * P + k4], &rgb.u[2 * j], 8);
        } else {
#pragma unroll
          for (int j = 0; j < NB / 4; ++j) lds_put<OpT, 4>(&Bs[(c0 + j * RV) * P + k4], &rb[4 * j]);
        }
      } else {
        const int kk = tid % KT, c0 = tid / KT;
#pragma unroll
        for (int j = 0; j < NB; ++j) lds_put1<OpT>(&Bs[(c0 + j * RS) * P + kk], rb[j]);
      }
    } else {
      const int col = tid % BN, kq = tid / BN;
      lds_put<OpT, NB>(&Bs[col * P + kq * B_KPT], rb);
    }
  };

  if (B_IM2COL && g.B.kfast) {
    for (int c = tid; c < BN; c += 256) ntab[c] = im2col_entry(g.B, n0 + c < g.N ? n0 + c : 0, true);
    __syncthreads();
  }
  if (it0 < it1) fetch(it0);
  const int fr = lane & 15, fq = lane >> 4;
  for (int it = it0; it < it1; ++it) {
    store_lds();
    __syncthreads();
    if (it + 1 < it1) fetch(it + 1);  // next tile's HBM loads fly under this tile's MFMAs
#pragma unroll
    for (int kc = 0; kc < KT / 32; ++kc) {
      if (!BF) {
        // lane owns k' = 8*fq .. 8*fq+7 of this 32-chunk ; step s pairs slot s of A with slot s of B
        float a[MI][8];
#pragma unroll
        for (int i = 0; i < MI; ++i) memcpy(a[i], &As[(wm * (BM / 2) + i * 16 + fr) * P + kc * 32 + fq * 8], 32);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float b[8];
          memcpy(b, &Bs[(wn * (BN / 2) + j * 16 + fr) * P + kc * 32 + fq * 8], 32);
#pragma unroll
          for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < MI; ++i)
              acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], a[i][s], acc[i][j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[s], acc[i][j], 0, 0, 0);
        }
      } else {
        bf16x8 a[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) memcpy(&a[i], &As[(wm * (BM / 2) + i * 16 + fr) * P + kc * 32 + fq * 8], 16);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          bf16x8 b;
          memcpy(&b, &Bs[(wn * (BN / 2) + j * 16 + fr) * P + kc * 32 + fq * 8], 16);
#pragma unroll
          for (int i = 0; i < MI; ++i)
            acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a[i], acc[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  gemm_epilogue<GT, BM, BN, SWAP>(g, acc, cstrip, m0, n0, bo, bi, batch, wave, lane);
}

template <typename OpT, bool IM, bool SWAP, int KT>
static int launch_tile(const GemmArgs& g, int bm, int bn, int nbatch, hipStream_t stream) {
  dim3 grid(cdiv(g.N, bn), cdiv(g.M, bm), nbatch * g.splits);
  if (grid.y > 65535 || grid.z > 65535) return CENET_EUNSUPPORTED;
#define CENET_TILE(BMv, BNv)                                                                   \
  if (bm == BMv && bn == BNv) {                                                                \
    CENET_LAUNCH((gemm_kernel<OpT, BMv, BNv, IM, SWAP, KT>), grid, dim3(256), stream, g);      \
    return CENET_OK;                                                                           \
  }
  if (KT == 32) {
    CENET_TILE(128, 128)
    CENET_TILE(128, 64)
    CENET_TILE(64, 128)
    CENET_TILE(64, 64)
    CENET_TILE(32, 256)
    CENET_TILE(32, 64)
  }
  return CENET_EUNSUPPORTED;
}
// K step 64 (plain bf16 operands; with bf16 tensors the prefetch registers hold packed pairs, so the 128-wide tiles keep
// two workgroups per CU at K step 64 too)
template <typename OpT, bool SWAP>
static int launch_tile_k64(const GemmArgs& g, int bm, int bn, int nbatch, hipStream_t stream) {
  dim3 grid(cdiv(g.N, bn), cdiv(g.M, bm), nbatch * g.splits);
  if (grid.y > 65535 || grid.z > 65535) return CENET_EUNSUPPORTED;
#define CENET_TILE64(BMv, BNv)                                                                 \
  if (bm == BMv && bn == BNv) {                                                                \
    CENET_LAUNCH((gemm_kernel<OpT, BMv, BNv, false, SWAP, 64>), grid, dim3(256), stream, g);   \
    return CENET_OK;                                                                           \
  }
  CENET_TILE64(64, 64)
  CENET_TILE64(32, 64)
#undef CENET_TILE64
  return CENET_EUNSUPPORTED;
}
#undef CENET_TILE

// one translation unit per (operand type, B view): gemm_inst_*.hip
#define CENET_GEMM_INSTANCE(NAME, T, IM)                                                                     \
  int NAME(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream) {                   \
    return swap ? launch_tile<T, IM, true, 32>(g, bm, bn, nbatch, stream) : launch_tile<T, IM, false, 32>(g, bm, bn, nbatch, stream); \
  }
#define CENET_GEMM_INSTANCE_K64(NAME, T)                                                                     \
  int NAME(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream) {                   \
    return swap ? launch_tile_k64<T, true>(g, bm, bn, nbatch, stream) : launch_tile_k64<T, false>(g, bm, bn, nbatch, stream); \
  }
int cenet_gemm_launch_f32_plain(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream);
int cenet_gemm_launch_f32_im2col(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream);
int cenet_gemm_launch_bf16_plain(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream);
int cenet_gemm_launch_bf16_im2col(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream);
int cenet_gemm_launch_bf16_plain_k64(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream);
