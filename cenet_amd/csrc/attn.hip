// attn.hip — tiled ("flash"-style) softmax attention forward/backward with generic strides, plus row softmax.
//
// One kernel family serves the three attention sites of CENet without ever materialising N x N maps:
//   * spatial-reduction attention  q@k^T*scale -> softmax -> @v                      pvtv2.py:101-105
//   * Non-local block              softmax(theta^T phi / sqrt(C)) g   (NCHW operands)  nlb.py:117-138
//   * differential attention       2H softmax heads sharing H value heads           multihead_diffattn.py:96-116
// Element (b,h,i,d) of Q lives at q[b*sb + h*sh + i*si + d*sd] (same for K, V, O), so token-layout heads and
// channel-major NCHW maps are both addressed in place.  Value head = h / v_head_div.
//
// Workgroup = 4 waves = one 64-row query tile (forward, dQ) or one 64-row key tile (dK/dV); tiles are staged in
// LDS k-contiguous; all products run on v_mfma_f32_16x16x4_f32 (exact fp32); the online softmax lives in the MFMA
// accumulator layout (row = (lane>>4)*4+r, col = lane&15) and P / dS cross LDS once to become an A operand.
// Backward recomputes P from the saved log-sum-exp.
#include "common.h"
#include "../../include/cenet_hip.h"

#define TQ 64
#define TK 64
#define NEG_BIG (-1.0e30f)

struct AttnArgs {
  const float *q, *k, *v;
  float* o;
  float* lse;  // [B,H,Nq]
  // backward
  const float *dout;
  float *dq, *dk, *dv, *delta;
  long qsb, qsh, qsi, qsd;
  long ksb, ksh, ksi, ksd;
  long vsb, vsh, vsi, vsd;
  long osb, osh, osi, osd;
  int B, H, Nq, Nk, D, Dv, v_head_div;
  int q_dfast, k_dfast, v_dfast, o_dfast;  // 1: d is the contiguous dim (token layout); 0: token index contiguous (NCHW)
  int dv_atomic;
  float scale;
  int finite_scores;  // forward: q pre-scaled, scores through nan_to_num (multihead_diffattn.py:95,106)
};

// torch.nan_to_num on one fp32 score: NaN -> 0, +-inf -> +-FLT_MAX
__device__ __forceinline__ float score_nan_to_num(float x) {
  if (x != x) return 0.f;
  return fminf(fmaxf(x, -3.402823466e+38f), 3.402823466e+38f);
}

// stage a [64 x cols] tile (zero padded to colsp columns) into LDS dst[r*pitch + c]
__device__ __forceinline__ void stage_tile(float* dst, int pitch, const float* src, long s_row, long s_col, int row0,
                                           int nrows, int cols, int colsp, int dfast) {
  const int total = 64 * colsp;
  if (dfast) {
    for (int idx = threadIdx.x; idx < total; idx += 256) {
      int r = idx / colsp, c = idx - r * colsp;
      float v = 0.f;
      if (row0 + r < nrows && c < cols) v = src[(long)(row0 + r) * s_row + (long)c * s_col];
      dst[r * pitch + c] = v;
    }
  } else {
    for (int idx = threadIdx.x; idx < total; idx += 256) {
      int c = idx >> 6, r = idx & 63;
      float v = 0.f;
      if (row0 + r < nrows && c < cols) v = src[(long)(row0 + r) * s_row + (long)c * s_col];
      dst[r * pitch + c] = v;
    }
  }
}


// Register prefetch of a [64 x COLS] tile (same element map as stage_tile): load() issues the HBM reads, store() writes
// them to LDS one phase later, so a tile's loads fly under the previous tile's MFMAs.
template <int COLS>
struct TilePrefetch {
  float r[64 * COLS / 256];
  __device__ __forceinline__ void load(const float* src, long s_row, long s_col, int row0, int nrows, int cols, int dfast) {
#pragma unroll
    for (int j = 0; j < 64 * COLS / 256; ++j) {
      const int idx = threadIdx.x + 256 * j;
      int rr, c;
      if (dfast) {
        rr = idx / COLS;
        c = idx - rr * COLS;
      } else {
        c = idx >> 6;
        rr = idx & 63;
      }
      r[j] = (row0 + rr < nrows && c < cols) ? src[(long)(row0 + rr) * s_row + (long)c * s_col] : 0.f;
    }
  }
  __device__ __forceinline__ void store(float* dst, int pitch, int dfast) const {
#pragma unroll
    for (int j = 0; j < 64 * COLS / 256; ++j) {
      const int idx = threadIdx.x + 256 * j;
      int rr, c;
      if (dfast) {
        rr = idx / COLS;
        c = idx - rr * COLS;
      } else {
        c = idx >> 6;
        rr = idx & 63;
      }
      dst[rr * pitch + c] = r[j];
    }
  }
};

// acc[t] (16x16 tiles over 64 columns) = A[arow0..+16][0..kdim) * B^T where both operands are k-contiguous in LDS:
// A frag: As[(arow0 + lane&15)*pa + k + (lane>>4)], B frag: Bs[(16t + lane&15)*pb + k + (lane>>4)]
__device__ __forceinline__ void mma_rowsxrows(f32x4 acc[4], const float* As, int pa, int arow0, const float* Bs, int pb,
                                              int kdim, int lane) {
  const int fr = lane & 15, fk = lane >> 4;
  for (int k = 0; k < kdim; k += 4) {
    float a = As[(arow0 + fr) * pa + k + fk];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float b = Bs[(16 * t + fr) * pb + k + fk];
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  }
}

// acc[t] (16 x 16*NT) += A[arow0..+16][0..64) * B[0..64)[0..16*NT) with A k-contiguous (pitch pa) and B row-major
// [k][n] (pitch pb): B frag = Bs[(k + lane>>4)*pb + 16t + lane&15]
template <int NT>
__device__ __forceinline__ void mma_rowsxcols(f32x4 acc[NT], const float* As, int pa, int arow0, const float* Bs, int pb,
                                              int lane) {
  const int fr = lane & 15, fk = lane >> 4;
#pragma unroll 4
  for (int k = 0; k < 64; k += 4) {
    float a = As[(arow0 + fr) * pa + k + fk];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float b = Bs[(k + fk) * pb + 16 * t + fr];
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  }
}

template <int DQ, int DV>
__global__ __launch_bounds__(256) void flash_fwd_kernel(AttnArgs a) {
  constexpr int PQ = DQ + 2, PV = DV + 16, PP = 66, NV = DV / 16;
  __shared__ float Qs[TQ * PQ];
  __shared__ float Ks[TK * PQ];
  __shared__ float Vs[TK * PV];
  __shared__ float Ps[TQ * PP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, hv = h / a.v_head_div;
  const int i0 = blockIdx.x * TQ;
  const int Dp = (a.D + 3) & ~3;  // k extent actually multiplied (LDS tiles are zero padded to DQ / DV)
  const float* qb = a.q + (long)b * a.qsb + (long)h * a.qsh;
  const float* kb = a.k + (long)b * a.ksb + (long)h * a.ksh;
  const float* vb = a.v + (long)b * a.vsb + (long)hv * a.vsh;
  stage_tile(Qs, PQ, qb, a.qsi, a.qsd, i0, a.Nq, a.D, DQ, a.q_dfast);
  // finite_scores: the reference scales q first (`q *= self.scaling`, multihead_diffattn.py:95), then multiplies — which products
  // overflow depends on it; each thread rescales exactly the elements it staged (same index map as stage_tile)
  const bool fin = a.finite_scores != 0;
  const float lo = fin ? -3.402823466e+38f : NEG_BIG;  // below every nan_to_num'd score
  if (fin)
    for (int idx = threadIdx.x; idx < 64 * DQ; idx += 256) {
      const int r = a.q_dfast ? idx / DQ : idx & 63, c = a.q_dfast ? idx - (idx / DQ) * DQ : idx >> 6;
      Qs[r * PQ + c] *= a.scale;
    }

  f32x4 o[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m[r] = lo;
    l[r] = 0.f;
  }
  const int fr = lane & 15, fq = lane >> 4;
  TilePrefetch<DQ> pk;
  TilePrefetch<DV> pv;
  pk.load(kb, a.ksi, a.ksd, 0, a.Nk, a.D, a.k_dfast);
  pv.load(vb, a.vsi, a.vsd, 0, a.Nk, a.Dv, a.v_dfast);

  for (int j0 = 0; j0 < a.Nk; j0 += TK) {
    __syncthreads();  // previous tile fully consumed (also orders the Q staging on the first pass)
    pk.store(Ks, PQ, a.k_dfast);
    pv.store(Vs, PV, a.v_dfast);
    __syncthreads();
    if (j0 + TK < a.Nk) {  // next tile's loads fly under this tile's MFMAs
      pk.load(kb, a.ksi, a.ksd, j0 + TK, a.Nk, a.D, a.k_dfast);
      pv.load(vb, a.vsi, a.vsd, j0 + TK, a.Nk, a.Dv, a.v_dfast);
    }
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_rowsxrows(s, Qs, PQ, wave * 16, Ks, PQ, Dp, lane);
    // online softmax on rows (fq*4 + r), columns 16t + fr
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = lo;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v = (j0 + 16 * t + fr < a.Nk) ? (fin ? score_nan_to_num(s[t][r]) : s[t][r] * a.scale) : lo;
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
#pragma unroll
      for (int o_ = 1; o_ < 16; o_ <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o_));
      float mnew = fmaxf(m[r], mx);
      alpha[r] = fast_exp(m[r] - mnew);
      float rs = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float p = (j0 + 16 * t + fr < a.Nk) ? fast_exp(s[t][r] - mnew) : 0.f;
        s[t][r] = p;
        rs += p;
      }
#pragma unroll
      for (int o_ = 1; o_ < 16; o_ <<= 1) rs += __shfl_xor(rs, o_);
      l[r] = l[r] * alpha[r] + rs;
      m[r] = mnew;
    }
#pragma unroll
    for (int t = 0; t < NV; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[t][r] *= alpha[r];
    // P -> LDS (wave-private 16 rows) in [row][key] order, then back as an A operand
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ps[(wave * 16 + fq * 4 + r) * PP + 16 * t + fr] = s[t][r];
    __syncthreads();
    mma_rowsxcols<NV>(o, Ps, PP, wave * 16, Vs, PV, lane);
  }
  // epilogue: O / l, LSE
  float* ob = a.o + (long)b * a.osb + (long)h * a.osh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + wave * 16 + fq * 4 + r;
    if (i < a.Nq) {
      const float inv = 1.f / l[r];
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int d = 16 * t + fr;
        if (d < a.Dv) ob[(long)i * a.osi + (long)d * a.osd] = o[t][r] * inv;
      }
      if (fr == 0) a.lse[((long)b * a.H + h) * a.Nq + i] = m[r] + logf(l[r]);
    }
  }
}

// dQ (and delta = rowsum(dO*O)) : workgroup = 64 query rows, sweeps key tiles
template <int DQ, int DV>
__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(AttnArgs a) {
  // pitch = width+2 for tiles only read k-contiguous, width+18 for tiles also read as row-major [k][n] B operands
  constexpr int PQ = DQ + 2, PK = DQ + 18, PV = DV + 2, PP = 66, NQ = DQ / 16;
  __shared__ float Qs[TQ * PQ];
  __shared__ float Ks[TK * PK];  // [j][d]: k-contiguous for S, row-major [k=j][n=d] for dQ
  __shared__ float Vs[TK * PV];
  __shared__ float dOs[TQ * PV];
  __shared__ float Ps[TQ * PP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, hv = h / a.v_head_div;
  const int i0 = blockIdx.x * TQ;
  const int Dp = (a.D + 3) & ~3, Dvp = (a.Dv + 3) & ~3;
  const float* qb = a.q + (long)b * a.qsb + (long)h * a.qsh;
  const float* kb = a.k + (long)b * a.ksb + (long)h * a.ksh;
  const float* vb = a.v + (long)b * a.vsb + (long)hv * a.vsh;
  const float* ob = a.o + (long)b * a.osb + (long)h * a.osh;
  const float* gb = a.dout + (long)b * a.osb + (long)h * a.osh;
  const int fr = lane & 15, fq = lane >> 4;
  stage_tile(Qs, PQ, qb, a.qsi, a.qsd, i0, a.Nq, a.D, DQ, a.q_dfast);
  stage_tile(dOs, PV, gb, a.osi, a.osd, i0, a.Nq, a.Dv, DV, a.o_dfast);
  stage_tile(Vs, PV, ob, a.osi, a.osd, i0, a.Nq, a.Dv, DV, a.o_dfast);  // O staged temporarily in Vs
  __syncthreads();
  // delta for rows fq*4+r of this wave: each of the 16 lanes (fr) sums a strided part of the row
  float dl[4], ls[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = wave * 16 + fq * 4 + r;
    float sacc = 0.f;
    for (int d = fr; d < Dvp; d += 16) sacc += dOs[row * PV + d] * Vs[row * PV + d];
#pragma unroll
    for (int o_ = 1; o_ < 16; o_ <<= 1) sacc += __shfl_xor(sacc, o_);
    dl[r] = sacc;
    const int i = i0 + row;
    ls[r] = (i < a.Nq) ? a.lse[((long)b * a.H + h) * a.Nq + i] : 0.f;
    if (fr == 0 && i < a.Nq) a.delta[((long)b * a.H + h) * a.Nq + i] = sacc;
  }
  f32x4 dq[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  TilePrefetch<DQ> pk;
  TilePrefetch<DV> pv;
  pk.load(kb, a.ksi, a.ksd, 0, a.Nk, a.D, a.k_dfast);
  pv.load(vb, a.vsi, a.vsd, 0, a.Nk, a.Dv, a.v_dfast);
  for (int j0 = 0; j0 < a.Nk; j0 += TK) {
    __syncthreads();
    pk.store(Ks, PK, a.k_dfast);
    pv.store(Vs, PV, a.v_dfast);
    __syncthreads();
    if (j0 + TK < a.Nk) {
      pk.load(kb, a.ksi, a.ksd, j0 + TK, a.Nk, a.D, a.k_dfast);
      pv.load(vb, a.vsi, a.vsd, j0 + TK, a.Nk, a.Dv, a.v_dfast);
    }
    f32x4 s[4], dp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    mma_rowsxrows(s, Qs, PQ, wave * 16, Ks, PK, Dp, lane);
    mma_rowsxrows(dp, dOs, PV, wave * 16, Vs, PV, Dvp, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = (j0 + 16 * t + fr < a.Nk) ? fast_exp(s[t][r] * a.scale - ls[r]) : 0.f;
        float ds = p * (dp[t][r] - dl[r]) * a.scale;
        Ps[(wave * 16 + fq * 4 + r) * PP + 16 * t + fr] = ds;
      }
    __syncthreads();
    mma_rowsxcols<NQ>(dq, Ps, PP, wave * 16, Ks, PK, lane);
  }
  float* dqb = a.dq + (long)b * a.qsb + (long)h * a.qsh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + wave * 16 + fq * 4 + r;
    if (i < a.Nq) {
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        const int d = 16 * t + fr;
        if (d < a.D) dqb[(long)i * a.qsi + (long)d * a.qsd] = dq[t][r];
      }
    }
  }
}

// dK, dV : workgroup = 64 key rows, sweeps query tiles; works on S^T = K Q^T so that P^T / dS^T are produced in the
// accumulator layout with the key on the row.
template <int DQ, int DV>
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(AttnArgs a) {
  constexpr int PK = DQ + 2, PVk = DV + 2, PQ = DQ + 18, PV = DV + 18, PP = 66, NQ = DQ / 16, NV = DV / 16;
  __shared__ float Ks[TK * PK];
  __shared__ float Vs[TK * PVk];
  __shared__ float Qs[TQ * PQ];   // [i][d]: k-contiguous for S^T, row-major [k=i][n=d] for dK
  __shared__ float dOs[TQ * PV];  // [i][dv]: k-contiguous for dP^T, row-major [k=i][n=dv] for dV
  __shared__ float Pt[TK * PP];   // P^T  [j][i]
  __shared__ float St[TK * PP];   // dS^T [j][i]
  __shared__ float lse_s[TQ], del_s[TQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, hv = h / a.v_head_div;
  const int j0 = blockIdx.x * TK;
  const int Dp = (a.D + 3) & ~3, Dvp = (a.Dv + 3) & ~3;
  const float* qb = a.q + (long)b * a.qsb + (long)h * a.qsh;
  const float* kb = a.k + (long)b * a.ksb + (long)h * a.ksh;
  const float* vb = a.v + (long)b * a.vsb + (long)hv * a.vsh;
  const float* gb = a.dout + (long)b * a.osb + (long)h * a.osh;
  const int fr = lane & 15, fq = lane >> 4;
  stage_tile(Ks, PK, kb, a.ksi, a.ksd, j0, a.Nk, a.D, DQ, a.k_dfast);
  stage_tile(Vs, PVk, vb, a.vsi, a.vsd, j0, a.Nk, a.Dv, DV, a.v_dfast);
  f32x4 dk[NQ], dv[NV];
#pragma unroll
  for (int t = 0; t < NQ; ++t) dk[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NV; ++t) dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  TilePrefetch<DQ> pq;
  TilePrefetch<DV> pg;
  pq.load(qb, a.qsi, a.qsd, 0, a.Nq, a.D, a.q_dfast);
  pg.load(gb, a.osi, a.osd, 0, a.Nq, a.Dv, a.o_dfast);
  for (int i0 = 0; i0 < a.Nq; i0 += TQ) {
    __syncthreads();
    pq.store(Qs, PQ, a.q_dfast);
    pg.store(dOs, PV, a.o_dfast);
    if (threadIdx.x < TQ) {
      const int i = i0 + threadIdx.x;
      lse_s[threadIdx.x] = (i < a.Nq) ? a.lse[((long)b * a.H + h) * a.Nq + i] : 0.f;
      del_s[threadIdx.x] = (i < a.Nq) ? a.delta[((long)b * a.H + h) * a.Nq + i] : 0.f;
    }
    __syncthreads();
    if (i0 + TQ < a.Nq) {
      pq.load(qb, a.qsi, a.qsd, i0 + TQ, a.Nq, a.D, a.q_dfast);
      pg.load(gb, a.osi, a.osd, i0 + TQ, a.Nq, a.Dv, a.o_dfast);
    }
    f32x4 st[4], dpt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      dpt[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    mma_rowsxrows(st, Ks, PK, wave * 16, Qs, PQ, Dp, lane);     // S^T[j][i]
    mma_rowsxrows(dpt, Vs, PVk, wave * 16, dOs, PV, Dvp, lane);  // dP^T[j][i]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ic = 16 * t + fr;
      const bool iv = (i0 + ic < a.Nq);
      const float lsev = lse_s[ic], delv = del_s[ic];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jr = wave * 16 + fq * 4 + r;
        const bool ok = iv && (j0 + jr < a.Nk);
        float p = ok ? fast_exp(st[t][r] * a.scale - lsev) : 0.f;
        float ds = p * (dpt[t][r] - delv) * a.scale;
        Pt[jr * PP + ic] = p;
        St[jr * PP + ic] = ds;
      }
    }
    __syncthreads();
    mma_rowsxcols<NV>(dv, Pt, PP, wave * 16, dOs, PV, lane);
    mma_rowsxcols<NQ>(dk, St, PP, wave * 16, Qs, PQ, lane);
  }
  float* dkb = a.dk + (long)b * a.ksb + (long)h * a.ksh;
  float* dvb = a.dv + (long)b * a.vsb + (long)hv * a.vsh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = j0 + wave * 16 + fq * 4 + r;
    if (j < a.Nk) {
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        const int d = 16 * t + fr;
        if (d < a.D) dkb[(long)j * a.ksi + (long)d * a.ksd] = dk[t][r];
      }
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int d = 16 * t + fr;
        if (d < a.Dv) {
          float* dst = &dvb[(long)j * a.vsi + (long)d * a.vsd];
          if (a.dv_atomic) atomicAdd(dst, dv[t][r]);
          else *dst = dv[t][r];
        }
      }
    }
  }
}

static void fill_args(AttnArgs& a, const cenet_attn_t* p) {
  a.q = (const float*)p->q; a.k = (const float*)p->k; a.v = (const float*)p->v; a.o = (float*)p->o; a.lse = p->lse;
  a.dout = (const float*)p->dout; a.dq = (float*)p->dq; a.dk = (float*)p->dk; a.dv = (float*)p->dv; a.delta = p->delta;
  a.qsb = p->qsb; a.qsh = p->qsh; a.qsi = p->qsi; a.qsd = p->qsd;
  a.ksb = p->ksb; a.ksh = p->ksh; a.ksi = p->ksi; a.ksd = p->ksd;
  a.vsb = p->vsb; a.vsh = p->vsh; a.vsi = p->vsi; a.vsd = p->vsd;
  a.osb = p->osb; a.osh = p->osh; a.osi = p->osi; a.osd = p->osd;
  a.B = p->B; a.H = p->H; a.Nq = p->Nq; a.Nk = p->Nk; a.D = p->D; a.Dv = p->Dv;
  a.v_head_div = p->v_head_div > 0 ? p->v_head_div : 1;
  a.q_dfast = (p->qsd == 1); a.k_dfast = (p->ksd == 1); a.v_dfast = (p->vsd == 1); a.o_dfast = (p->osd == 1);
  a.dv_atomic = (a.v_head_div > 1);
  a.scale = p->scale;
  a.finite_scores = p->finite_scores;
}

static int pick_variant(int D, int Dv) {
  if (D <= 16 && Dv <= 32) return 4;
  if (D <= 32 && Dv <= 32) return 0;
  if (D <= 32 && Dv <= 64) return 1;
  if (D <= 64 && Dv <= 64) return 2;
  if (D <= 64 && Dv <= 128) return 3;
  return -1;
}

// bf16-operand variants (attn_bf16.hip)
int cenet_flashb_fwd(const cenet_attn_t* p, hipStream_t stream);
int cenet_flashb_bwd(const cenet_attn_t* p, hipStream_t stream);

extern "C" int cenet_flash_attn_supported(int D, int Dv) { return pick_variant(D, Dv) >= 0; }
// bf16 tensors (throughput mode): the register-resident kernels of attn_bf16.hip
extern "C" int cenet_flash_attn_fwd_bf16(const cenet_attn_t* p, hipStream_t stream) {
  if (!p || !p->q || !p->k || !p->v || !p->o || !p->lse) return CENET_EINVAL;
  int rc = cenet_flashb_fwd(p, stream);
  if (rc != CENET_OK) return rc;
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_flash_attn_bwd_bf16(const cenet_attn_t* p, hipStream_t stream) {
  if (!p || !p->q || !p->k || !p->v || !p->o || !p->lse || !p->dout || !p->dq || !p->dk || !p->dv || !p->delta)
    return CENET_EINVAL;
  int rc = cenet_flashb_bwd(p, stream);
  if (rc != CENET_OK) return rc;
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_flash_attn_fwd_f32(const cenet_attn_t* p, hipStream_t stream) {
  if (!p || !p->q || !p->k || !p->v || !p->o || !p->lse) return CENET_EINVAL;
  if (p->B <= 0 || p->H <= 0 || p->Nq <= 0 || p->Nk <= 0 || p->D <= 0 || p->Dv <= 0) return CENET_EINVAL;
  AttnArgs a;
  fill_args(a, p);
  dim3 grid(cdiv(a.Nq, TQ), a.B * a.H);
  switch (pick_variant(a.D, a.Dv)) {
    case 0: CENET_LAUNCH((flash_fwd_kernel<32, 32>), grid, dim3(256), stream, a); break;
    case 1: CENET_LAUNCH((flash_fwd_kernel<32, 64>), grid, dim3(256), stream, a); break;
    case 2: CENET_LAUNCH((flash_fwd_kernel<64, 64>), grid, dim3(256), stream, a); break;
    case 3: CENET_LAUNCH((flash_fwd_kernel<64, 128>), grid, dim3(256), stream, a); break;
    case 4: CENET_LAUNCH((flash_fwd_kernel<16, 32>), grid, dim3(256), stream, a); break;
    default: return CENET_EUNSUPPORTED;
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_flash_attn_bwd_f32(const cenet_attn_t* p, hipStream_t stream) {
  if (!p || !p->q || !p->k || !p->v || !p->o || !p->lse || !p->dout || !p->dq || !p->dk || !p->dv || !p->delta)
    return CENET_EINVAL;
  AttnArgs a;
  fill_args(a, p);
  dim3 gq(cdiv(a.Nq, TQ), a.B * a.H), gk(cdiv(a.Nk, TK), a.B * a.H);
  switch (pick_variant(a.D, a.Dv)) {
    case 0:
      CENET_LAUNCH((flash_bwd_dq_kernel<32, 32>), gq, dim3(256), stream, a);
      CENET_LAUNCH((flash_bwd_dkv_kernel<32, 32>), gk, dim3(256), stream, a);
      break;
    case 1:
      CENET_LAUNCH((flash_bwd_dq_kernel<32, 64>), gq, dim3(256), stream, a);
      CENET_LAUNCH((flash_bwd_dkv_kernel<32, 64>), gk, dim3(256), stream, a);
      break;
    case 2:
      CENET_LAUNCH((flash_bwd_dq_kernel<64, 64>), gq, dim3(256), stream, a);
      CENET_LAUNCH((flash_bwd_dkv_kernel<64, 64>), gk, dim3(256), stream, a);
      break;
    case 3:
      CENET_LAUNCH((flash_bwd_dq_kernel<64, 128>), gq, dim3(256), stream, a);
      CENET_LAUNCH((flash_bwd_dkv_kernel<64, 128>), gk, dim3(256), stream, a);
      break;
    case 4:
      CENET_LAUNCH((flash_bwd_dq_kernel<16, 32>), gq, dim3(256), stream, a);
      CENET_LAUNCH((flash_bwd_dkv_kernel<16, 32>), gk, dim3(256), stream, a);
      break;
    default: return CENET_EUNSUPPORTED;
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

// ------------------------------------------------------------------------------------------------
// Row softmax over contiguous rows (materialised-attention path for head dims > 128): one wave per row.
// ------------------------------------------------------------------------------------------------
// x (scores) and dy (score gradients) are fp32 in every mode; the probabilities y and dx have the operand storage type T
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const float* __restrict__ x, T* __restrict__ y, long rows, int n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* xr = x + row * n;
  T* yr = y + row * n;
  float mx = NEG_BIG;
  for (int c = lane; c < n; c += 64) mx = fmaxf(mx, xr[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s += fast_exp(xr[c] - mx);
  s = 1.f / wave_sum(s);
  for (int c = lane; c < n; c += 64) stf(yr + c, fast_exp(xr[c] - mx) * s);
}

// dx = y * (dy - sum(dy*y))
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const T* __restrict__ y, const float* __restrict__ dy,
                                                              T* __restrict__ dx, long rows, int n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const T* yr = y + row * n;
  const float* gr = dy + row * n;
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s += ldf(yr + c) * gr[c];
  s = wave_sum(s);
  T* dr = dx + row * n;
  for (int c = lane; c < n; c += 64) stf(dr + c, ldf(yr + c) * (gr[c] - s));
}

template <typename T>
static int softmax_rows_fwd_impl(const float* x, T* y, long rows, int n, hipStream_t stream) {
  if (rows <= 0 || n <= 0) return CENET_EINVAL;
  CENET_LAUNCH((softmax_rows_fwd_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), stream, x, y, rows, n);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(softmax_rows_fwd, (const float* x, T* y, long rows, int n, hipStream_t stream), (x, y, rows, n, stream))
template <typename T>
static int softmax_rows_bwd_impl(const T* y, const float* dy, T* dx, long rows, int n, hipStream_t stream) {
  if (rows <= 0 || n <= 0) return CENET_EINVAL;
  CENET_LAUNCH((softmax_rows_bwd_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), stream, y, dy, dx, rows, n);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
CENET_TWIN(softmax_rows_bwd, (const T* y, const float* dy, T* dx, long rows, int n, hipStream_t stream), (y, dy, dx, rows, n, stream))
