// attn_bf16.hip — bf16-operand tiled attention (throughput mode), register-resident probabilities.
//
// Same contract as attn.hip (generic strides, fp32 online softmax / LSE / delta / accumulators) on bf16 tensors: q, k, v, o,
// dO, dq (and dk / dv unless `dkv_f32`) are bf16 in HBM.  Reference sites: pvtv2.py:101-105, nlb.py:117-138, multihead_diffattn.py:96-116.
//
// Layout idea (v_mfma_f32_16x16x32_bf16: lane (fr = lane&15, fq = lane>>4) supplies A[row fr][k 8fq..8fq+7] and
// B[k 8fq..8fq+7][col fr], and receives D[row 4fq+r][col fr], r = 0..3):
//   * every score tile is computed TRANSPOSED with respect to the operand that stays fixed for the whole K loop, so the
//     fixed side (queries in fwd / dQ, keys in dK/dV) is the MFMA column: one lane = one query (key), its softmax
//     state (running max, sum, lse, delta) is a per-lane scalar, and row reductions are 16 in-lane values + 2 shuffles;
//   * the probabilities / dS a lane holds (4 rows of two adjacent 16-row tiles = 8 values) are exactly one B fragment
//     of the next product if that product's k index is enumerated in the same permuted order:
//         k slot (fq, s)  <->  row 32c + 16 (s >> 2) + 4 fq + (s & 3)
//     so P / dS never go through LDS; the other operand is read with two 8-byte LDS loads at rows 32c+4fq and
//     32c+16+4fq of a [feature][row] (transposed) copy of the streamed tile;
//   * the fixed side's own fragments (Q^T, dO^T or K^T, V^T) are loaded once from HBM into registers;
//   * the streamed tiles are register-prefetched one tile ahead and double-buffered in LDS (one barrier per tile);
//   * exponentials are v_exp_f32 (base 2): log2(e) is folded into the query scaling.
#include "common.h"
#include "../../include/cenet_hip.h"

#define TK 64
#define NEG_BIG (-1.0e30f)
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define PT 72  // pitch (elements) of [feature][64 rows] transposed tiles: 144-byte rows
typedef unsigned short bf;

struct AttnArgsB {
  const bf *q, *k, *v;
  bf* o;
  float* lse;
  const bf* dout;
  bf* dq;
  void *dk, *dv;  // bf16, or fp32 accumulators when dkv_f32 (atomic adds: shared V heads, query-range slices)
  float* delta;
  int dkv_f32;
  long qsb, qsh, qsi, qsd, ksb, ksh, ksi, ksd, vsb, vsh, vsi, vsd, osb, osh, osi, osd;
  int B, H, Nq, Nk, D, Dv, v_head_div;
  int q_al, k_al, v_al, o_al;  // every stride a multiple of 4 floats and the base 16-byte aligned
  int dv_atomic, qsplit, tiles_per_split;
  float scale;
};

// (fast_exp2: common.h)
__device__ __forceinline__ bf16x8 pack8(const float* v) {
  unsigned u[4] = {cenet_pack_bf2(v[0], v[1]), cenet_pack_bf2(v[2], v[3]), cenet_pack_bf2(v[4], v[5]),
                   cenet_pack_bf2(v[6], v[7])};
  bf16x8 o;
  memcpy(&o, u, 16);
  return o;
}
// B fragment from two adjacent accumulator tiles (rows 4fq+r of tile 2c and of tile 2c+1)
__device__ __forceinline__ bf16x8 pack_tiles(const f32x4& lo, const f32x4& hi) {
  unsigned u[4] = {cenet_pack_bf2(lo[0], lo[1]), cenet_pack_bf2(lo[2], lo[3]), cenet_pack_bf2(hi[0], hi[1]),
                   cenet_pack_bf2(hi[2], hi[3])};
  bf16x8 o;
  memcpy(&o, u, 16);
  return o;
}
// A fragment in the permuted k order from a transposed tile T[feature][row]: rows 32c+4fq..+3 and 32c+16+4fq..+3
__device__ __forceinline__ bf16x8 frag_perm(const bf* T, int feature, int c, int fq) {
  bf16x8 o;
  const bf* p = T + feature * PT + 32 * c + 4 * fq;
  memcpy(&o, p, 8);
  memcpy((char*)&o + 8, p + 16, 8);
  return o;
}
__device__ __forceinline__ bf16x8 frag_rm(const bf* T, int pitch, int row, int c, int fq) {
  bf16x8 o;
  memcpy(&o, T + row * pitch + 32 * c + 8 * fq, 16);
  return o;
}
// fixed-side fragment straight from HBM: X[row][32c + 8fq .. +7] * mul, zero outside [0,nrows) x [0,cols)
__device__ __forceinline__ bf16x8 frag_global(const bf* x, long s_row, long s_col, int row, int nrows, int c, int fq,
                                              int cols, float mul, int al) {
  float v[8];
  const int c0 = 32 * c + 8 * fq;
  if (row < nrows && s_col == 1 && al && c0 + 7 < cols) {
    ld4v(v, x + (long)row * s_row + c0);
    ld4v(v + 4, x + (long)row * s_row + c0 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= mul;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[e] = (row < nrows && c0 + e < cols) ? ldf(x + (long)row * s_row + (long)(c0 + e) * s_col) * mul : 0.f;
  }
  return pack8(v);
}

// Register prefetch of a streamed [64 rows x COLS] bf16 tile (as fp32: it may be scaled) and its way into LDS, row-major
// dst_rm[row][col] and / or transposed dst_tr[col][row].  A thread owns groups of 4 elements along the axis that is
// contiguous in the LDS copy it must write packed (8-byte stores); the other copy, if wanted, gets 2-byte stores.
template <int COLS>
struct StageT {
  static constexpr int NG = 64 * COLS / 1024;
  float r[NG][4];
  int own_cols, rowadj, vec;
  __device__ __forceinline__ void setup(long s_row, long s_col, bool need_rm, bool need_tr, int aligned) {
    if (s_col == 1) {
      own_cols = need_rm;
      rowadj = 0;
      vec = own_cols && aligned;
    } else if (s_row == 1) {
      own_cols = !need_tr;
      rowadj = 1;
      vec = !own_cols && aligned;
    } else {
      own_cols = need_rm;
      rowadj = 0;
      vec = 0;
    }
  }
  __device__ __forceinline__ void coords(int g, int& row, int& col) const {
    if (own_cols) {
      if (!rowadj) {
        row = g / (COLS / 4);
        col = (g - row * (COLS / 4)) * 4;
      } else {
        row = g & 63;
        col = (g >> 6) * 4;
      }
    } else {
      if (!rowadj) {
        col = g % COLS;
        row = (g / COLS) * 4;
      } else {
        row = (g & 15) * 4;
        col = g >> 4;
      }
    }
  }
  __device__ __forceinline__ void load(const bf* src, long s_row, long s_col, int row0, int nrows, int cols) {
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      int row, col;
      coords(threadIdx.x + 256 * j, row, col);
      const long dr = own_cols ? 0 : s_row, dc = own_cols ? s_col : 0;
      const int er = own_cols ? 0 : 1, ec = own_cols ? 1 : 0;
      const bf* p = src + (long)(row0 + row) * s_row + (long)col * s_col;
      if (vec && row0 + row + 3 * er < nrows && col + 3 * ec < cols) {
        ld4v(r[j], p);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          r[j][e] = (row0 + row + e * er < nrows && col + e * ec < cols) ? ldf(p + e * (dr + dc)) : 0.f;
      }
    }
  }
  __device__ __forceinline__ void store(bf* dst_rm, int p_rm, bf* dst_tr, float mul) const {
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      int row, col;
      coords(threadIdx.x + 256 * j, row, col);
      const unsigned pk[2] = {cenet_pack_bf2(r[j][0] * mul, r[j][1] * mul), cenet_pack_bf2(r[j][2] * mul, r[j][3] * mul)};
      // the scattered copy converts each element on its own: one conversion whose low half feeds ds_write_b16 directly is
      // cheaper than cutting the packed pairs apart again (shift / mask / permute per element)
      if (own_cols) {
        if (dst_rm) memcpy(dst_rm + row * p_rm + col, pk, 8);
        if (dst_tr) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst_tr[(col + e) * PT + row] = (bf)cenet_f2bf(r[j][e] * mul);
        }
      } else {
        if (dst_tr) memcpy(dst_tr + col * PT + row, pk, 8);
        if (dst_rm) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst_rm[(row + e) * p_rm + col] = (bf)cenet_f2bf(r[j][e] * mul);
        }
      }
    }
  }
};

__device__ __forceinline__ float quad_sum(float v) {  // over the 4 lanes that share fr (fq = 0..3)
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

// write 4 consecutive features (f0..f0+3) of one row; T = bf (plain stores) or float (fp32 accumulator, optionally atomic)
template <typename T>
__device__ __forceinline__ void put4(T* base, long s_row, long s_col, int row, int f0, int nf, const float* v, int al,
                                     bool atomic) {
  T* p = base + (long)row * s_row + (long)f0 * s_col;
  if (atomic) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (f0 + e < nf) atomicAdd((float*)p + e * s_col, v[e]);
    }
  } else if (s_col == 1 && al && f0 + 3 < nf) {
    st4v(p, v);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (f0 + e < nf) stf(p + e * s_col, v[e]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward: one workgroup = 64*NQT queries of one (batch, head); wave w owns NQT 16-query tiles
// ---------------------------------------------------------------------------------------------------------------
template <int DQ, int DV, int NQT>
__global__ __launch_bounds__(256, (DV <= 64 ? 2 : 1)) void flashc_fwd_kernel(AttnArgsB a) {
  constexpr int PQ = DQ + 8, NC = DQ / 32, NU = DV / 16;
  __shared__ __attribute__((aligned(16))) bf Ks[2][TK * PQ];  // [key][d]
  __shared__ __attribute__((aligned(16))) bf Vt[2][DV * PT];  // [dv][key]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, hv = h / a.v_head_div;
  const int q0 = blockIdx.x * (64 * NQT) + wave * (16 * NQT);
  const bf* qb = a.q + (long)b * a.qsb + (long)h * a.qsh;
  const bf* kb = a.k + (long)b * a.ksb + (long)h * a.ksh;
  const bf* vb = a.v + (long)b * a.vsb + (long)hv * a.vsh;

  bf16x8 qf[NQT][NC];
#pragma unroll
  for (int t = 0; t < NQT; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c)
      qf[t][c] = frag_global(qb, a.qsi, a.qsd, q0 + 16 * t + fr, a.Nq, c, fq, a.D, a.scale * LOG2E, a.q_al);
  f32x4 o[NQT][NU];
  float m[NQT], l[NQT];
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    m[t] = NEG_BIG;
    l[t] = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) o[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  StageT<DQ> sk;
  StageT<DV> sv;
  sk.setup(a.ksi, a.ksd, true, false, a.k_al);
  sv.setup(a.vsi, a.vsd, false, true, a.v_al);
  sk.load(kb, a.ksi, a.ksd, 0, a.Nk, a.D);
  sv.load(vb, a.vsi, a.vsd, 0, a.Nk, a.Dv);
  sk.store(Ks[0], PQ, nullptr, 1.f);
  sv.store(nullptr, 0, Vt[0], 1.f);
  __syncthreads();
  int cur = 0;
  for (int j0 = 0; j0 < a.Nk; j0 += TK) {
    const bool more = j0 + TK < a.Nk;
    if (more) {
      sk.load(kb, a.ksi, a.ksd, j0 + TK, a.Nk, a.D);
      sv.load(vb, a.vsi, a.vsd, j0 + TK, a.Nk, a.Dv);
    }
    const bf* K_ = Ks[cur];
    const bf* V_ = Vt[cur];
    f32x4 st[4][NQT];  // S^T: rows = keys 16kt + 4fq + r, column = query fr
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int t = 0; t < NQT; ++t) st[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const bf16x8 kf = frag_rm(K_, PQ, 16 * kt + fr, c, fq);
#pragma unroll
        for (int t = 0; t < NQT; ++t) st[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][c], st[kt][t], 0, 0, 0);
      }
    }
    if (j0 + TK > a.Nk) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j0 + 16 * kt + 4 * fq + r >= a.Nk) {
#pragma unroll
            for (int t = 0; t < NQT; ++t) st[kt][t][r] = NEG_BIG;
          }
    }
    bf16x8 pb[NQT][2];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
      float mx = NEG_BIG;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][t][r]);
      mx = quad_max(mx);
      const float mnew = fmaxf(m[t], mx);
      const float alpha = fast_exp2(m[t] - mnew);
      m[t] = mnew;
      float rs = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp2(st[kt][t][r] - mnew);
          st[kt][t][r] = p;
          rs += p;
        }
      l[t] = l[t] * alpha + rs;  // per-lane partial sum (this lane's keys); the 4 fq lanes are added once at the end
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t][u][r] *= alpha;
      pb[t][0] = pack_tiles(st[0][t], st[1][t]);
      pb[t][1] = pack_tiles(st[2][t], st[3][t]);
    }
    // O^T[dv][query] += V^T[dv][keys] . P^T[keys][query]
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8 vf = frag_perm(V_, 16 * u + fr, c, fq);
#pragma unroll
        for (int t = 0; t < NQT; ++t) o[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[t][c], o[t][u], 0, 0, 0);
      }
    if (more) {
      sk.store(Ks[cur ^ 1], PQ, nullptr, 1.f);
      sv.store(nullptr, 0, Vt[cur ^ 1], 1.f);
    }
    __syncthreads();
    cur ^= 1;
  }
  bf* ob = a.o + (long)b * a.osb + (long)h * a.osh;
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    const float lt = quad_sum(l[t]);
    const int i = q0 + 16 * t + fr;
    if (i < a.Nq) {
      const float inv = 1.f / lt;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = o[t][u][r] * inv;
        put4(ob, a.osi, a.osd, i, 16 * u + 4 * fq, a.Dv, v, a.o_al, false);
      }
      if (fq == 0) a.lse[((long)b * a.H + h) * a.Nq + i] = m[t] * LN2 + logf(lt);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, dQ (and delta): one workgroup = 64*NQT queries; streams key tiles
// ---------------------------------------------------------------------------------------------------------------
// MINB = 2 caps the registers for two workgroups per CU; the 64-dim instances then spill a few dwords, which pays on long
// key loops but costs a scratch set-up per launch on tiny problems — those take the MINB = 1 instance
template <int DQ, int DV, int NQT, int MINB>
__global__ __launch_bounds__(256, MINB) void flashc_bwd_dq_kernel(AttnArgsB a) {
  constexpr int PQ = DQ + 8, PV = DV + 8, NC = DQ / 32, NCV = DV / 32, ND = DQ / 16;
  __shared__ __attribute__((aligned(16))) bf Ks[2][TK * PQ];  // [key][d]
  __shared__ __attribute__((aligned(16))) bf Kt[2][DQ * PT];  // [d][key]
  __shared__ __attribute__((aligned(16))) bf Vs[2][TK * PV];  // [key][dv]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, hv = h / a.v_head_div;
  const int q0 = blockIdx.x * (64 * NQT) + wave * (16 * NQT);
  const bf* qb = a.q + (long)b * a.qsb + (long)h * a.qsh;
  const bf* kb = a.k + (long)b * a.ksb + (long)h * a.ksh;
  const bf* vb = a.v + (long)b * a.vsb + (long)hv * a.vsh;
  const bf* ob = a.o + (long)b * a.osb + (long)h * a.osh;
  const bf* gb = a.dout + (long)b * a.osb + (long)h * a.osh;

  bf16x8 qf[NQT][NC], gf[NQT][NCV];
  float lse2[NQT], dl[NQT];
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    const int i = q0 + 16 * t + fr;
#pragma unroll
    for (int c = 0; c < NC; ++c) qf[t][c] = frag_global(qb, a.qsi, a.qsd, i, a.Nq, c, fq, a.D, a.scale * LOG2E, a.q_al);
#pragma unroll
    for (int c = 0; c < NCV; ++c) gf[t][c] = frag_global(gb, a.osi, a.osd, i, a.Nq, c, fq, a.Dv, 1.f, a.o_al);
    // delta = rowsum(dO * O) in fp32 from HBM: this lane's feature slots, then the 4 fq lanes
    float sacc = 0.f;
    if (i < a.Nq) {
#pragma unroll
      for (int c = 0; c < NCV; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int d = 32 * c + 8 * fq + e;
          if (d < a.Dv) sacc += ldf(gb + (long)i * a.osi + (long)d * a.osd) * ldf(ob + (long)i * a.osi + (long)d * a.osd);
        }
    }
    sacc = quad_sum(sacc);
    dl[t] = sacc;
    lse2[t] = (i < a.Nq) ? a.lse[((long)b * a.H + h) * a.Nq + i] * LOG2E : -NEG_BIG;
    if (fq == 0 && i < a.Nq) a.delta[((long)b * a.H + h) * a.Nq + i] = sacc;
  }
  f32x4 dq[NQT][ND];
#pragma unroll
  for (int t = 0; t < NQT; ++t)
#pragma unroll
    for (int u = 0; u < ND; ++u) dq[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  StageT<DQ> sk;
  StageT<DV> sv;
  sk.setup(a.ksi, a.ksd, true, true, a.k_al);
  sv.setup(a.vsi, a.vsd, true, false, a.v_al);
  sk.load(kb, a.ksi, a.ksd, 0, a.Nk, a.D);
  sv.load(vb, a.vsi, a.vsd, 0, a.Nk, a.Dv);
  sk.store(Ks[0], PQ, Kt[0], 1.f);
  sv.store(Vs[0], PV, nullptr, 1.f);
  __syncthreads();
  int cur = 0;
  for (int j0 = 0; j0 < a.Nk; j0 += TK) {
    const bool more = j0 + TK < a.Nk;
    if (more) {
      sk.load(kb, a.ksi, a.ksd, j0 + TK, a.Nk, a.D);
      sv.load(vb, a.vsi, a.vsd, j0 + TK, a.Nk, a.Dv);
    }
    const bf* K_ = Ks[cur];
    const bf* Kt_ = Kt[cur];
    const bf* V_ = Vs[cur];
    // the 64 keys of the tile in two halves of 32 (= one permuted-k chunk each)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f32x4 st[2][NQT], dp[2][NQT];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kt = 2 * c + e;
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
          st[e][t] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[e][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
          const bf16x8 kf = frag_rm(K_, PQ, 16 * kt + fr, cc, fq);
#pragma unroll
          for (int t = 0; t < NQT; ++t) st[e][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][cc], st[e][t], 0, 0, 0);
        }
#pragma unroll
        for (int cc = 0; cc < NCV; ++cc) {
          const bf16x8 vf = frag_rm(V_, PV, 16 * kt + fr, cc, fq);
#pragma unroll
          for (int t = 0; t < NQT; ++t) dp[e][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gf[t][cc], dp[e][t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fast_exp2(st[e][t][r] - lse2[t]);
            st[e][t][r] = p * (dp[e][t][r] - dl[t]);
          }
      }
      // dQ^T[d][query] += K^T[d][keys] . dS^T[keys][query]   (keys beyond Nk carry zero K rows)
#pragma unroll
      for (int t = 0; t < NQT; ++t) {
        const bf16x8 dsb = pack_tiles(st[0][t], st[1][t]);
#pragma unroll
        for (int u = 0; u < ND; ++u)
          if (16 * u < a.D) dq[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_perm(Kt_, 16 * u + fr, c, fq), dsb, dq[t][u], 0, 0, 0);
      }
    }
    if (more) {
      sk.store(Ks[cur ^ 1], PQ, Kt[cur ^ 1], 1.f);
      sv.store(Vs[cur ^ 1], PV, nullptr, 1.f);
    }
    __syncthreads();
    cur ^= 1;
  }
  bf* dqb = a.dq + (long)b * a.qsb + (long)h * a.qsh;
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    const int i = q0 + 16 * t + fr;
    if (i < a.Nq) {
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = dq[t][u][r] * a.scale;
        put4(dqb, a.qsi, a.qsd, i, 16 * u + 4 * fq, a.D, v, a.q_al, false);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, dK / dV: one workgroup = 64*NKT keys; streams query tiles (optionally only a slice of them: blockIdx.z)
// ---------------------------------------------------------------------------------------------------------------
template <int DQ, int DV, int NKT, int MINB>
__global__ __launch_bounds__(256, MINB) void flashc_bwd_dkv_kernel(AttnArgsB a) {
  constexpr int PQ = DQ + 8, PV = DV + 8, NC = DQ / 32, NCV = DV / 32, ND = DQ / 16, NU = DV / 16;
  constexpr int NBUF = (DV <= 64) ? 2 : 1;  // the widest instance keeps one LDS stage (two barriers per tile)
  __shared__ __attribute__((aligned(16))) bf Qs[NBUF][64 * PQ];   // [query][d]   (scaled by scale*log2e)
  __shared__ __attribute__((aligned(16))) bf Qt[NBUF][DQ * PT];   // [d][query]
  __shared__ __attribute__((aligned(16))) bf Gs[NBUF][64 * PV];   // dO [query][dv]
  __shared__ __attribute__((aligned(16))) bf Gt[NBUF][DV * PT];   // dO^T [dv][query]
  __shared__ __attribute__((aligned(16))) float lse_s[NBUF][64], del_s[NBUF][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, hv = h / a.v_head_div;
  const int k0 = blockIdx.x * (64 * NKT) + wave * (16 * NKT);
  const bf* qb = a.q + (long)b * a.qsb + (long)h * a.qsh;
  const bf* kb = a.k + (long)b * a.ksb + (long)h * a.ksh;
  const bf* vb = a.v + (long)b * a.vsb + (long)hv * a.vsh;
  const bf* gb = a.dout + (long)b * a.osb + (long)h * a.osh;
  const long stat0 = ((long)b * a.H + h) * a.Nq;

  bf16x8 kf[NKT][NC], vf[NKT][NCV];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
#pragma unroll
    for (int c = 0; c < NC; ++c) kf[t][c] = frag_global(kb, a.ksi, a.ksd, k0 + 16 * t + fr, a.Nk, c, fq, a.D, 1.f, a.k_al);
#pragma unroll
    for (int c = 0; c < NCV; ++c) vf[t][c] = frag_global(vb, a.vsi, a.vsd, k0 + 16 * t + fr, a.Nk, c, fq, a.Dv, 1.f, a.v_al);
  }
  f32x4 dk[NKT][ND], dv[NKT][NU];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
#pragma unroll
    for (int u = 0; u < ND; ++u) dk[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u) dv[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int ibeg = blockIdx.z * a.tiles_per_split * 64;
  int iend = ibeg + a.tiles_per_split * 64;
  if (iend > a.Nq) iend = a.Nq;
  StageT<DQ> sq;
  StageT<DV> sg;
  sq.setup(a.qsi, a.qsd, true, true, a.q_al);
  sg.setup(a.osi, a.osd, true, true, a.o_al);
  float pl = 0.f, pd = 0.f;  // prefetched lse / delta of query ibeg + threadIdx.x (threads 0..63)
  auto load_tile = [&](int i0) __attribute__((always_inline)) {
    sq.load(qb, a.qsi, a.qsd, i0, iend, a.D);
    sg.load(gb, a.osi, a.osd, i0, iend, a.Dv);
    if (threadIdx.x < 64) {
      const int i = i0 + threadIdx.x;
      pl = (i < iend) ? a.lse[stat0 + i] * LOG2E : -NEG_BIG;
      pd = (i < iend) ? a.delta[stat0 + i] : 0.f;
    }
  };
  auto store_tile = [&](int buf) __attribute__((always_inline)) {
    sq.store(Qs[buf], PQ, Qt[buf], a.scale * LOG2E);
    sg.store(Gs[buf], PV, Gt[buf], 1.f);
    if (threadIdx.x < 64) {
      lse_s[buf][threadIdx.x] = pl;
      del_s[buf][threadIdx.x] = pd;
    }
  };
  if (ibeg < iend) load_tile(ibeg);
  int cur = 0;
  for (int i0 = ibeg; i0 < iend; i0 += 64) {
    const bool more = i0 + 64 < iend;
    if (NBUF == 1 || i0 == ibeg) {
      if (NBUF == 1) __syncthreads();
      store_tile(cur);
      __syncthreads();
    }
    if (more) load_tile(i0 + 64);
    const bf* Q_ = Qs[cur];
    const bf* Qt_ = Qt[cur];
    const bf* G_ = Gs[cur];
    const bf* Gt_ = Gt[cur];
    // the 64 queries of the tile in two halves of 32 (= one permuted-k chunk each), one 16-key tile at a time, to keep the
    // live score registers low (the Q / dO fragments are re-read from LDS per key tile)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        f32x4 s[2], dp[2];  // rows = queries 32c + 16e + 4fq + r, column = key fr
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int qt = 2 * c + e;
          s[e] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int cc = 0; cc < NC; ++cc)
            s[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm(Q_, PQ, 16 * qt + fr, cc, fq), kf[t][cc], s[e], 0, 0, 0);
#pragma unroll
          for (int cc = 0; cc < NCV; ++cc)
            dp[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm(G_, PV, 16 * qt + fr, cc, fq), vf[t][cc], dp[e], 0, 0, 0);
          float l4[4], d4[4];
          memcpy(l4, &lse_s[cur][16 * qt + 4 * fq], 16);
          memcpy(d4, &del_s[cur][16 * qt + 4 * fq], 16);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fast_exp2(s[e][r] - l4[r]);  // queries beyond the slice carry lse = +big -> p = 0
            s[e][r] = p;
            dp[e][r] = p * (dp[e][r] - d4[r]);
          }
        }
        const bf16x8 pb = pack_tiles(s[0], s[1]), db = pack_tiles(dp[0], dp[1]);
#pragma unroll
        for (int u = 0; u < NU; ++u)
          if (16 * u < a.Dv) dv[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_perm(Gt_, 16 * u + fr, c, fq), pb, dv[t][u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < ND; ++u)
          if (16 * u < a.D) dk[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_perm(Qt_, 16 * u + fr, c, fq), db, dk[t][u], 0, 0, 0);
      }
    }
    if (NBUF == 2) {
      if (more) store_tile(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  const long dko = (long)b * a.ksb + (long)h * a.ksh, dvo = (long)b * a.vsb + (long)hv * a.vsh;
  const bool split = a.qsplit > 1;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int j = k0 + 16 * t + fr;
    if (j < a.Nk) {
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = dk[t][u][r] * LN2;  // Q was staged scaled by scale*log2e
        if (a.dkv_f32) put4((float*)a.dk + dko, a.ksi, a.ksd, j, 16 * u + 4 * fq, a.D, v, a.k_al, split);
        else put4((bf*)a.dk + dko, a.ksi, a.ksd, j, 16 * u + 4 * fq, a.D, v, a.k_al, false);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = dv[t][u][r];
        if (a.dkv_f32) put4((float*)a.dv + dvo, a.vsi, a.vsd, j, 16 * u + 4 * fq, a.Dv, v, a.v_al, split || a.dv_atomic);
        else put4((bf*)a.dv + dvo, a.vsi, a.vsd, j, 16 * u + 4 * fq, a.Dv, v, a.v_al, false);
      }
    }
  }
}

// quads of bf16 (8 bytes)
static int al4(const void* p, long s0, long s1, long s2, long s3) {
  auto m4 = [](long v) { return v == 1 || (v & 3) == 0; };
  return (((uintptr_t)p & 7) == 0) && m4(s0) && m4(s1) && m4(s2) && m4(s3);
}

static void fill_b(AttnArgsB& a, const cenet_attn_t* p) {
  a.q = (const bf*)p->q; a.k = (const bf*)p->k; a.v = (const bf*)p->v; a.o = (bf*)p->o; a.lse = p->lse;
  a.dout = (const bf*)p->dout; a.dq = (bf*)p->dq; a.dk = p->dk; a.dv = p->dv; a.delta = p->delta;
  a.dkv_f32 = p->dkv_f32;
  a.qsb = p->qsb; a.qsh = p->qsh; a.qsi = p->qsi; a.qsd = p->qsd;
  a.ksb = p->ksb; a.ksh = p->ksh; a.ksi = p->ksi; a.ksd = p->ksd;
  a.vsb = p->vsb; a.vsh = p->vsh; a.vsi = p->vsi; a.vsd = p->vsd;
  a.osb = p->osb; a.osh = p->osh; a.osi = p->osi; a.osd = p->osd;
  a.B = p->B; a.H = p->H; a.Nq = p->Nq; a.Nk = p->Nk; a.D = p->D; a.Dv = p->Dv;
  a.v_head_div = p->v_head_div > 0 ? p->v_head_div : 1;
  // 16-byte accesses: the base of every tensor the problem touches and all its strides
  a.q_al = al4(p->q, p->qsb, p->qsh, p->qsi, p->qsd) && (!p->dq || ((uintptr_t)p->dq & 7) == 0);
  a.k_al = al4(p->k, p->ksb, p->ksh, p->ksi, p->ksd) && (!p->dk || ((uintptr_t)p->dk & 15) == 0);
  a.v_al = al4(p->v, p->vsb, p->vsh, p->vsi, p->vsd) && (!p->dv || ((uintptr_t)p->dv & 15) == 0);
  a.o_al = al4(p->o, p->osb, p->osh, p->osi, p->osd) && (!p->dout || ((uintptr_t)p->dout & 7) == 0);
  a.dv_atomic = (a.v_head_div > 1);
  a.qsplit = 1;
  a.tiles_per_split = cdiv(a.Nq, 64);
  a.scale = p->scale;
}

static int pick_b(int D, int Dv) {
  if (D <= 32 && Dv <= 32) return 0;
  if (D <= 32 && Dv <= 64) return 1;
  if (D <= 64 && Dv <= 64) return 2;
  if (D <= 64 && Dv <= 128) return 3;
  return -1;
}

// called by attn.hip's entry points when the bf16-operand mode is on
int cenet_flashb_fwd(const cenet_attn_t* p, hipStream_t stream) {
  AttnArgsB a;
  fill_b(a, p);
  const int cls = pick_b(a.D, a.Dv);
  if (cls < 0) return CENET_EUNSUPPORTED;
  const bool wide = a.Nq >= 1024 && cls != 3;  // two 16-query tiles per wave when there are enough queries to fill the chip
  dim3 grid(cdiv(a.Nq, wide ? 128 : 64), a.B * a.H);
#define CENET_FWD(DQv, DVv)                                                                      \
  if (wide) CENET_LAUNCH((flashc_fwd_kernel<DQv, DVv, 2>), grid, dim3(256), stream, a);          \
  else CENET_LAUNCH((flashc_fwd_kernel<DQv, DVv, 1>), grid, dim3(256), stream, a);
  switch (cls) {
    case 0: CENET_FWD(32, 32) break;
    case 1: CENET_FWD(32, 64) break;
    case 2: CENET_FWD(64, 64) break;
    default: CENET_LAUNCH((flashc_fwd_kernel<64, 128, 1>), grid, dim3(256), stream, a); break;
  }
#undef CENET_FWD
  return CENET_OK;
}

int cenet_flashb_bwd(const cenet_attn_t* p, hipStream_t stream) {
  AttnArgsB a;
  fill_b(a, p);
  const int cls = pick_b(a.D, a.Dv);
  if (cls < 0) return CENET_EUNSUPPORTED;
  const bool wide_q = a.Nq >= 1024 && cls != 3;
  const bool wide_k = a.Nk >= 1024 && cls <= 1;  // two 16-key tiles per wave (measured 4.2 vs 5.4 ms on the DSEB-56^2 problem)
  dim3 gq(cdiv(a.Nq, wide_q ? 128 : 64), a.B * a.H);
  // few key tiles under many queries (spatial-reduction attention: 49 keys): slice the query range over workgroups and
  // accumulate dK / dV atomically — only when the caller guarantees zero-filled dk / dv
  const int ktiles = cdiv(a.Nk, wide_k ? 128 : 64), qtiles = cdiv(a.Nq, 64);
  if (a.dv_atomic && !a.dkv_f32) return CENET_EINVAL;  // shared V heads add atomically: dk / dv must be fp32 accumulators
  if (p->dkv_zeroed && a.dkv_f32 && (long)ktiles * a.B * a.H < 512 && qtiles > 1) {
    int s = cdiv(1024, (long)ktiles * a.B * a.H);
    if (s > qtiles) s = qtiles;
    a.tiles_per_split = cdiv(qtiles, s);
    a.qsplit = cdiv(qtiles, a.tiles_per_split);
  }
  dim3 gk(ktiles, a.B * a.H, a.qsplit);
  const bool big = (long)a.Nq * a.Nk >= 1024L * 1024L;  // long loops: the register-capped 64-dim instances
#define CENET_DQ(DQv, DVv, NQTv, MB) CENET_LAUNCH((flashc_bwd_dq_kernel<DQv, DVv, NQTv, MB>), gq, dim3(256), stream, a)
#define CENET_DKV(DQv, DVv, NKTv, MB) CENET_LAUNCH((flashc_bwd_dkv_kernel<DQv, DVv, NKTv, MB>), gk, dim3(256), stream, a)
  switch (cls) {
    case 0:
      // the wide dQ instance runs three workgroups per CU at the price of a 20-byte spill: 1.59 -> 1.22 ms on DSEB-56^2
      if (wide_q) CENET_DQ(32, 32, 2, 3); else CENET_DQ(32, 32, 1, 2);
      // wide dK/dV: key tiles one at a time (fewer live score registers) lets it run three workgroups per CU with a 92-byte
      // spill: 2.56 -> 2.44 ms on DSEB-56^2 (the same order at two workgroups per CU is slower: 2.8 ms)
      if (wide_k) CENET_DKV(32, 32, 2, 3); else CENET_DKV(32, 32, 1, 2);
      break;
    case 1:
      if (wide_q) CENET_DQ(32, 64, 2, 2); else CENET_DQ(32, 64, 1, 2);
      if (wide_k) CENET_DKV(32, 64, 2, 1); else CENET_DKV(32, 64, 1, 2);  // two workgroups per CU (20-byte spill)
      break;
    case 2:
      if (wide_q && big) CENET_DQ(64, 64, 2, 2); else if (wide_q) CENET_DQ(64, 64, 2, 1); else CENET_DQ(64, 64, 1, 2);
      if (big) CENET_DKV(64, 64, 1, 2); else CENET_DKV(64, 64, 1, 1);
      break;
    default:
      CENET_DQ(64, 128, 1, 1);
      CENET_DKV(64, 128, 1, 1);
      break;
  }
#undef CENET_DQ
#undef CENET_DKV
  return CENET_OK;
}
