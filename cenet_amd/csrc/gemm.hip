// gemm.hip — batched strided GEMM / implicit-GEMM convolution core on the gfx950 matrix cores (v2).
//
// One kernel family serves every dense contraction of the CENet hot path:
//   * nn.Linear fwd/bwd in token layout                      (pvtv2.py:41,45,90,98,106; multihead_diffattn.py:79-81,126)
//   * 1x1 convs in NCHW fwd/bwd                              (cfam.py:149,158,299,302; nlb.py:106-115,142; blocks.py:178,320; dseb.py:164)
//   * dense k x k convs as implicit GEMM (fwd, dgrad, wgrad)  (pvtv2.py:164,67; unet.py:156-197; blocks.py:211)
//   * materialised attention products for large head dims
//
// C[b] (+)= epilogue( alpha * sum_kb A[b,kb] (MxK) * B[b,kb] (KxN) )
// A is a plain strided matrix; B is plain or an on-the-fly im2col / transposed-gather view of an image tensor.
//
// Structure (per 256-thread workgroup = 4 waves in a 2x2 grid, tile BM x BN, K step BK):
//   * operands are read from HBM with lane-contiguous dword loads in whichever orientation is contiguous (kfast /
//     mfast), one K-tile AHEAD into registers (software double buffering: the loads of tile t+1 fly under the MFMAs
//     of tile t), then written k-contiguous into LDS with a 144-byte row pitch;
//   * fragments are fetched with 16-byte ds_read_b128: lane (r = lane&15, q = lane>>4) owns k' = 8q..8q+7 of each
//     32-deep slab, and MFMA step s multiplies slot s of A with slot s of B (a permutation of the k order, which a
//     sum does not care about) — 4x fewer LDS instructions than dword reads;
//   * OpT = float : v_mfma_f32_16x16x4_f32, exact fp32 FMA chain (parity mode, BK = 32)
//     OpT = bf16  : operands rounded to bf16 when they enter LDS, v_mfma_f32_16x16x32_bf16, fp32 accumulate
//                   (throughput mode of BASELINE configs[1], BK = 64);
//   * im2col addressing never divides per element: the k-side decomposition (ci,ky,kx | py,px) of each K-tile is
//     written to a small LDS table by BK threads, the n-side decomposition is computed once per thread.
#include "common.h"
#include "../../include/cenet_hip.h"

struct GemmArgs {
  cenet_mat_t A, B;
  cenet_epi_t E;
  int M, N, K, nkb, splits, nb_inner;
};

__device__ __forceinline__ unsigned f2bf_bits(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) { return f2bf_bits(lo) | (f2bf_bits(hi) << 16); }

template <typename OpT> struct OpTraits;
template <> struct OpTraits<float> {
  static constexpr int BK = 32;     // k elements per tile
  static constexpr int PITCH = 36;  // LDS row pitch in elements (144 B)
};
template <> struct OpTraits<unsigned short> {
  static constexpr int BK = 64;
  static constexpr int PITCH = 72;  // 144 B
};

struct KEntry {  // per-k im2col table entry
  int off;       // ci*sci (+ ky*dil*sy + kx*dil*sx for the non-transposed patch side)  or  iy0*sy + ix0*sx
  int dy, dx;    // patch side: ky*dil, kx*dil ; pixel side: iy0, ix0
};

// plain element address (supports the split k index used by the conv data-gradient weight view)
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

// decompose a patch index e=(ci,ky,kx) or a pixel index p=(py,px) into a table entry
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

// combine a patch entry and a pixel entry into a value
__device__ __forceinline__ float im2col_load(const cenet_mat_t& d, const float* base, const KEntry& pat, const KEntry& pix) {
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
  return base[(long)pat.off + (long)iy * d.sy + (long)ix * d.sx];
}

template <typename OpT, int BM, int BN, bool B_IM2COL>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int BK = OpTraits<OpT>::BK, P = OpTraits<OpT>::PITCH;
  constexpr bool BF = (sizeof(OpT) == 2);
  constexpr int MI = BM / 32, NJ = BN / 32;  // 16x16 tiles per wave in each direction
  constexpr int NA = BM * BK / 256, NB = BN * BK / 256;  // prefetch registers per thread
  __shared__ __attribute__((aligned(16))) OpT As[BM * P];
  __shared__ __attribute__((aligned(16))) OpT Bs[BN * P];
  __shared__ KEntry ntab[B_IM2COL ? BN : 1];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int batch = blockIdx.z / g.splits, split = blockIdx.z - batch * g.splits;
  const int bo = batch / g.nb_inner, bi = batch - bo * g.nb_inner;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ktiles = (g.K + BK - 1) / BK;
  const int total = g.nkb * ktiles;
  const int chunk = (total + g.splits - 1) / g.splits;
  const int it0 = split * chunk;
  const int it1 = (it0 + chunk < total) ? it0 + chunk : total;

  // ---- thread -> tile-element maps (kfast: lanes along k ; mfast: lanes along the row index) ----
  // kfast: kk = tid % BK, rows r = tid / BK + j * (256 / BK)
  // mfast: row = tid % BMN, kq = tid / BMN ; k = kq * (BK / (256/BMN)) + j   (consecutive k per thread)
  constexpr int A_RPP = 256 / BK;           // rows per pass (kfast)
  constexpr int A_KG = 256 / BM > 0 ? 256 / BM : 1, A_KPT = BK / (256 / BM > 0 ? 256 / BM : 1);
  constexpr int B_KG = 256 / BN > 0 ? 256 / BN : 1, B_KPT = BK / (256 / BN > 0 ? 256 / BN : 1);
  static_assert(BM <= 256 && BN <= 256, "tile too large for the mfast map");
  // for BN == 256 (BM == 32): one thread per column, all BK k's per thread
  float ra[NA], rb[NB];

  // n-side im2col decomposition (fixed per thread): patch_is_row -> n is a pixel ; else n is a patch element
  KEntry nent;
  nent.off = nent.dy = nent.dx = 0;
  bool n_ok = true;
  if (B_IM2COL) {
    int ncol = g.B.kfast ? 0 : n0 + (tid % BN);
    n_ok = ncol < g.N;
    if (!g.B.kfast) nent = im2col_entry(g.B, n_ok ? ncol : 0, !g.B.patch_is_row);
  }

  auto fetch = [&](int it) {
    const int kb = it / ktiles;
    const int k0 = (it - kb * ktiles) * BK;
    const float* baseA = g.A.ptr + (long)bo * g.A.sb + (long)bi * g.A.sb2 + (long)kb * g.A.skb;
    const float* baseB = g.B.ptr + (long)bo * g.B.sb + (long)bi * g.B.sb2 + (long)kb * g.B.skb;
    if (g.A.kfast) {
      const int kk = tid % BK, r0 = tid / BK;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int row = r0 + j * A_RPP;
        ra[j] = (m0 + row < g.M && k0 + kk < g.K) ? baseA[plain_off<1>(g.A, m0 + row, k0 + kk)] : 0.f;
      }
    } else {
      const int row = tid % BM, kq = tid / BM;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int kk = kq * A_KPT + j;
        ra[j] = (m0 + row < g.M && k0 + kk < g.K) ? baseA[plain_off<1>(g.A, m0 + row, k0 + kk)] : 0.f;
      }
    }
    if (!B_IM2COL) {
      if (g.B.kfast) {
        const int kk = tid % BK, c0 = tid / BK;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          int col = c0 + j * A_RPP;
          rb[j] = (n0 + col < g.N && k0 + kk < g.K) ? baseB[plain_off<0>(g.B, k0 + kk, n0 + col)] : 0.f;
        }
      } else {
        const int col = tid % BN, kq = tid / BN;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          int kk = kq * B_KPT + j;
          rb[j] = (n0 + col < g.N && k0 + kk < g.K) ? baseB[plain_off<0>(g.B, k0 + kk, n0 + col)] : 0.f;
        }
      }
    } else {
      const cenet_mat_t& d = g.B;
      if (d.kfast) {
        // weight-gradient view: this thread's k is ONE pixel of the tile, its NB columns are patch elements whose
        // (ci,ky,kx) decomposition sits in the per-block LDS table ntab (n0 is fixed for the workgroup)
        const int kk = tid % BK, c0 = tid / BK;
        const bool kok = k0 + kk < g.K;
        const KEntry pe = im2col_entry(d, kok ? k0 + kk : 0, false);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          int col = c0 + j * A_RPP;
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
            rb[j] = ok ? baseB[off] : 0.f;
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

  auto store_lds = [&]() {
    if (g.A.kfast) {
      const int kk = tid % BK, r0 = tid / BK;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int row = r0 + j * A_RPP;
        if (BF) As[row * P + kk] = (OpT)f2bf_bits(ra[j]);
        else memcpy(&As[row * P + kk], &ra[j], 4);
      }
    } else {
      const int row = tid % BM, kq = tid / BM;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int kk = kq * A_KPT + j;
        if (BF) As[row * P + kk] = (OpT)f2bf_bits(ra[j]);
        else memcpy(&As[row * P + kk], &ra[j], 4);
      }
    }
    if (g.B.kfast) {
      const int kk = tid % BK, c0 = tid / BK;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        int col = c0 + j * A_RPP;
        if (BF) Bs[col * P + kk] = (OpT)f2bf_bits(rb[j]);
        else memcpy(&Bs[col * P + kk], &rb[j], 4);
      }
    } else {
      const int col = tid % BN, kq = tid / BN;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        int kk = kq * B_KPT + j;
        if (BF) Bs[col * P + kk] = (OpT)f2bf_bits(rb[j]);
        else memcpy(&Bs[col * P + kk], &rb[j], 4);
      }
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
    if (!BF) {
      // lane owns k' = 8*fq .. 8*fq+7 ; step s pairs slot s of A with slot s of B
      float a[MI][8];
#pragma unroll
      for (int i = 0; i < MI; ++i) memcpy(a[i], &As[(wm * (BM / 2) + i * 16 + fr) * P + fq * 8], 32);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float b[8];
        memcpy(b, &Bs[(wn * (BN / 2) + j * 16 + fr) * P + fq * 8], 32);
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[s], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 a[MI], b[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) memcpy(&a[i], &As[(wm * (BM / 2) + i * 16 + fr) * P + ks * 32 + fq * 8], 16);
#pragma unroll
        for (int j = 0; j < NJ; ++j) memcpy(&b[j], &Bs[(wn * (BN / 2) + j * 16 + fr) * P + ks * 32 + fq * 8], 16);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue ----
  const cenet_epi_t& E = g.E;
  float* Cb = E.C + (long)bo * E.scb + (long)bi * E.scb2;
  const float* Rb = E.R ? E.R + (long)bo * E.srb + (long)bi * E.srb2 : nullptr;
  const float bs = E.bscale ? E.bscale[batch] : 1.f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = m0 + wm * (BM / 2) + i * 16 + fq * 4 + r;
        int col = n0 + wn * (BN / 2) + j * 16 + fr;
        if (row < g.M && col < g.N) {
          float v = acc[i][j][r] * E.alpha;
          if (E.cmode) {  // col2im scatter: row = (ci,ky,kx), col = (py,px)
            const int kkw = E.cKH * E.cKW;
            const int ci = row / kkw, rem = row - ci * kkw;
            const int ky = rem / E.cKW, kx = rem - ky * E.cKW;
            const int py = col / E.cPw, px = col - py * E.cPw;
            const int iy = py * E.cstride - E.cpad + ky, ix = px * E.cstride - E.cpad + kx;
            if (iy >= 0 && iy < E.cHs && ix >= 0 && ix < E.cWs) {
              float* dst = &Cb[(long)ci * E.csci + (long)iy * E.csy + (long)ix * E.csx];
              if (E.atomic) atomicAdd(dst, v);
              else *dst = v;
            }
          } else if (E.atomic) {
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

template <typename OpT, bool IM>
static int launch_tile(const GemmArgs& g, int bm, int bn, int nbatch, hipStream_t stream) {
  dim3 grid(cdiv(g.N, bn), cdiv(g.M, bm), nbatch * g.splits);
  if (grid.y > 65535 || grid.z > 65535) return CENET_EUNSUPPORTED;
#define CENET_TILE(BMv, BNv)                                                              \
  if (bm == BMv && bn == BNv) {                                                           \
    CENET_LAUNCH((gemm_kernel<OpT, BMv, BNv, IM>), grid, dim3(256), stream, g);           \
    return CENET_OK;                                                                      \
  }
  CENET_TILE(128, 128)
  CENET_TILE(128, 64)
  CENET_TILE(64, 128)
  CENET_TILE(64, 64)
  CENET_TILE(32, 256)
  CENET_TILE(32, 64)
#undef CENET_TILE
  return CENET_EUNSUPPORTED;
}

static void pick_tile(int M, int N, int nbatch, int splits, int* bm, int* bn) {
  int m = M >= 96 ? 128 : (M >= 48 ? 64 : 32);
  int n = N >= 96 ? 128 : 64;
  if (m == 32) n = N >= 192 ? 256 : 64;
  // keep the chip busy: fall back to 64x64 when the big tile leaves most CUs idle
  long blocks = (long)cdiv(M, m) * cdiv(N, n) * nbatch * splits;
  if (blocks < 256 && m > 32) {
    int m2 = m > 64 ? 64 : m, n2 = n > 64 ? 64 : n;
    if (M >= 48) {
      m = m2;
      n = n2;
    }
  }
  *bm = m;
  *bn = n;
}

static int g_compute_bf16 = 0;
extern "C" int cenet_set_compute_bf16(int on) {
  int old = g_compute_bf16;
  g_compute_bf16 = on ? 1 : 0;
  return old;
}
extern "C" int cenet_get_compute_bf16() { return g_compute_bf16; }

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
  int bm, bn;
  pick_tile(M, N, nbatch, splits, &bm, &bn);
  if (B->mode == 1 && B->kfast && bm == 32) bn = 64;  // weight-gradient view: keep the per-thread gather list short
  int rc;
  if (g_compute_bf16) {
    rc = (B->mode == 0) ? launch_tile<unsigned short, false>(g, bm, bn, nbatch, stream)
                        : launch_tile<unsigned short, true>(g, bm, bn, nbatch, stream);
  } else {
    rc = (B->mode == 0) ? launch_tile<float, false>(g, bm, bn, nbatch, stream)
                        : launch_tile<float, true>(g, bm, bn, nbatch, stream);
  }
  if (rc != CENET_OK) return rc;
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
