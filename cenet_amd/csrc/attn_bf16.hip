// attn_bf16.hip — bf16-operand variant of the tiled attention kernels of attn.hip (throughput mode).
//
// Same algorithm, same generic strides, same fp32 HBM tensors, same fp32 online softmax / LSE / delta / accumulators;
// only the MFMA operands are rounded to bf16 on their way into LDS and every product is put in the form
//     C[16 x 16t] += A[rows, k] . B[cols, k]^T      (BOTH operands k-contiguous in LDS)
// so that each fragment is one 16-byte ds_read_b128 feeding v_mfma_f32_16x16x32_bf16.  Tiles that are contracted over
// their row index elsewhere (V in P.V, K in dS.K, Q and dO in the dK/dV products) are written to LDS twice from the
// prefetch registers: as staged and transposed.  Reference sites: pvtv2.py:101-105, nlb.py:117-138,
// multihead_diffattn.py:96-116.
#include "common.h"
#include "../../include/cenet_hip.h"

#define TQ 64
#define TK 64
#define NEG_BIG (-1.0e30f)
typedef unsigned short bf;

struct AttnArgsB {
  const float *q, *k, *v;
  float* o;
  float* lse;
  const float* dout;
  float *dq, *dk, *dv, *delta;
  long qsb, qsh, qsi, qsd, ksb, ksh, ksi, ksd, vsb, vsh, vsi, vsd, osb, osh, osi, osd;
  int B, H, Nq, Nk, D, Dv, v_head_div;
  int q_dfast, k_dfast, v_dfast, o_dfast, dv_atomic;
  float scale;
};

__device__ __forceinline__ bf f2bf(float f) { return (bf)cenet_f2bf(f); }

// Register prefetch of a [64 x COLS] fp32 tile; store() writes bf16 [row][col] (pitch p) and, if dstT, [col][row] (pitch pT)
template <int COLS>
struct TileB {
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
  __device__ __forceinline__ void store(bf* dst, int p, bf* dstT, int pT, int dfast) const {
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
      const bf v = f2bf(r[j]);
      if (dst) dst[rr * p + c] = v;
      if (dstT) dstT[c * pT + rr] = v;
    }
  }
};

// acc[t] += A[arow0 + (lane&15)][k] * B[16t + (lane&15)][k], k in [0, kdim), kdim % 32 == 0; 16-byte fragment reads
template <int NT>
__device__ __forceinline__ void mma_kc(f32x4 acc[NT], const bf* As, int pa, int arow0, const bf* Bs, int pb, int kdim, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  for (int k0 = 0; k0 < kdim; k0 += 32) {
    bf16x8 a;
    memcpy(&a, &As[(arow0 + fr) * pa + k0 + fq * 8], 16);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      bf16x8 b;
      memcpy(&b, &Bs[(16 * t + fr) * pb + k0 + fq * 8], 16);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
    }
  }
}

#define PT 72  // pitch of 64-wide k-contiguous tiles (144-byte rows)

template <int DQ, int DV>
__global__ __launch_bounds__(256) void flashb_fwd_kernel(AttnArgsB a) {
  constexpr int PQ = DQ + 8, NV = DV / 16;
  __shared__ __attribute__((aligned(16))) bf Qs[TQ * PQ];
  __shared__ __attribute__((aligned(16))) bf Ks[TK * PQ];
  __shared__ __attribute__((aligned(16))) bf Vt[DV * PT];  // [dv][key]
  __shared__ __attribute__((aligned(16))) bf Ps[TQ * PT];  // [row][key]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, hv = h / a.v_head_div;
  const int i0 = blockIdx.x * TQ;
  const int Dk = (a.D + 31) & ~31;
  const float* qb = a.q + (long)b * a.qsb + (long)h * a.qsh;
  const float* kb = a.k + (long)b * a.ksb + (long)h * a.ksh;
  const float* vb = a.v + (long)b * a.vsb + (long)hv * a.vsh;
  {
    TileB<DQ> tq;
    tq.load(qb, a.qsi, a.qsd, i0, a.Nq, a.D, a.q_dfast);
    tq.store(Qs, PQ, nullptr, 0, a.q_dfast);
  }
  f32x4 o[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m[r] = NEG_BIG;
    l[r] = 0.f;
  }
  const int fr = lane & 15, fq = lane >> 4;
  TileB<DQ> pk;
  TileB<DV> pv;
  pk.load(kb, a.ksi, a.ksd, 0, a.Nk, a.D, a.k_dfast);
  pv.load(vb, a.vsi, a.vsd, 0, a.Nk, a.Dv, a.v_dfast);
  for (int j0 = 0; j0 < a.Nk; j0 += TK) {
    __syncthreads();
    pk.store(Ks, PQ, nullptr, 0, a.k_dfast);
    pv.store(nullptr, 0, Vt, PT, a.v_dfast);
    __syncthreads();
    if (j0 + TK < a.Nk) {
      pk.load(kb, a.ksi, a.ksd, j0 + TK, a.Nk, a.D, a.k_dfast);
      pv.load(vb, a.vsi, a.vsd, j0 + TK, a.Nk, a.Dv, a.v_dfast);
    }
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_kc<4>(s, Qs, PQ, wave * 16, Ks, PQ, Dk, lane);
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = NEG_BIG;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v = (j0 + 16 * t + fr < a.Nk) ? s[t][r] * a.scale : NEG_BIG;
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
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ps[(wave * 16 + fq * 4 + r) * PT + 16 * t + fr] = f2bf(s[t][r]);
    __syncthreads();
    mma_kc<NV>(o, Ps, PT, wave * 16, Vt, PT, 64, lane);
  }
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

template <int DQ, int DV>
__global__ __launch_bounds__(256) void flashb_bwd_dq_kernel(AttnArgsB a) {
  constexpr int PQ = DQ + 8, PV = DV + 8, NQ = DQ / 16;
  __shared__ __attribute__((aligned(16))) bf Qs[TQ * PQ];
  __shared__ __attribute__((aligned(16))) bf Ks[TK * PQ];   // [j][d]
  __shared__ __attribute__((aligned(16))) bf Kt[DQ * PT];   // [d][j]
  __shared__ __attribute__((aligned(16))) bf Vs[TK * PV];   // [j][dv]
  __shared__ __attribute__((aligned(16))) bf dOs[TQ * PV];  // [i][dv]
  __shared__ __attribute__((aligned(16))) bf Ss[TQ * PT];   // dS [i][j]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, hv = h / a.v_head_div;
  const int i0 = blockIdx.x * TQ;
  const int Dk = (a.D + 31) & ~31, Dvk = (a.Dv + 31) & ~31;
  const float* qb = a.q + (long)b * a.qsb + (long)h * a.qsh;
  const float* kb = a.k + (long)b * a.ksb + (long)h * a.ksh;
  const float* vb = a.v + (long)b * a.vsb + (long)hv * a.vsh;
  const float* ob = a.o + (long)b * a.osb + (long)h * a.osh;
  const float* gb = a.dout + (long)b * a.osb + (long)h * a.osh;
  const int fr = lane & 15, fq = lane >> 4;
  {
    TileB<DQ> tq;
    tq.load(qb, a.qsi, a.qsd, i0, a.Nq, a.D, a.q_dfast);
    tq.store(Qs, PQ, nullptr, 0, a.q_dfast);
    TileB<DV> tg;
    tg.load(gb, a.osi, a.osd, i0, a.Nq, a.Dv, a.o_dfast);
    tg.store(dOs, PV, nullptr, 0, a.o_dfast);
  }
  // delta = rowsum(dO * O) in fp32 straight from HBM (rows fq*4+r of this wave; 16 lanes split the row)
  float dl[4], ls[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + wave * 16 + fq * 4 + r;
    float sacc = 0.f;
    if (i < a.Nq)
      for (int d = fr; d < a.Dv; d += 16) sacc += gb[(long)i * a.osi + (long)d * a.osd] * ob[(long)i * a.osi + (long)d * a.osd];
#pragma unroll
    for (int o_ = 1; o_ < 16; o_ <<= 1) sacc += __shfl_xor(sacc, o_);
    dl[r] = sacc;
    ls[r] = (i < a.Nq) ? a.lse[((long)b * a.H + h) * a.Nq + i] : 0.f;
    if (fr == 0 && i < a.Nq) a.delta[((long)b * a.H + h) * a.Nq + i] = sacc;
  }
  f32x4 dq[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  TileB<DQ> pk;
  TileB<DV> pv;
  pk.load(kb, a.ksi, a.ksd, 0, a.Nk, a.D, a.k_dfast);
  pv.load(vb, a.vsi, a.vsd, 0, a.Nk, a.Dv, a.v_dfast);
  for (int j0 = 0; j0 < a.Nk; j0 += TK) {
    __syncthreads();
    pk.store(Ks, PQ, Kt, PT, a.k_dfast);
    pv.store(Vs, PV, nullptr, 0, a.v_dfast);
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
    mma_kc<4>(s, Qs, PQ, wave * 16, Ks, PQ, Dk, lane);
    mma_kc<4>(dp, dOs, PV, wave * 16, Vs, PV, Dvk, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = (j0 + 16 * t + fr < a.Nk) ? fast_exp(s[t][r] * a.scale - ls[r]) : 0.f;
        Ss[(wave * 16 + fq * 4 + r) * PT + 16 * t + fr] = f2bf(p * (dp[t][r] - dl[r]) * a.scale);
      }
    __syncthreads();
    mma_kc<NQ>(dq, Ss, PT, wave * 16, Kt, PT, 64, lane);
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

template <int DQ, int DV>
__global__ __launch_bounds__(256) void flashb_bwd_dkv_kernel(AttnArgsB a) {
  constexpr int PQ = DQ + 8, PV = DV + 8, NQ = DQ / 16, NV = DV / 16;
  __shared__ __attribute__((aligned(16))) bf Ks[TK * PQ];    // [j][d]
  __shared__ __attribute__((aligned(16))) bf Vs[TK * PV];    // [j][dv]
  __shared__ __attribute__((aligned(16))) bf Qs[TQ * PQ];    // [i][d]
  __shared__ __attribute__((aligned(16))) bf Qt[DQ * PT];    // [d][i]
  __shared__ __attribute__((aligned(16))) bf dOs[TQ * PV];   // [i][dv]
  __shared__ __attribute__((aligned(16))) bf dOt[DV * PT];   // [dv][i]
  __shared__ __attribute__((aligned(16))) bf Pt[TK * PT];    // P^T  [j][i]
  __shared__ __attribute__((aligned(16))) bf St[TK * PT];    // dS^T [j][i]
  __shared__ float lse_s[TQ], del_s[TQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, hv = h / a.v_head_div;
  const int j0 = blockIdx.x * TK;
  const int Dk = (a.D + 31) & ~31, Dvk = (a.Dv + 31) & ~31;
  const float* qb = a.q + (long)b * a.qsb + (long)h * a.qsh;
  const float* kb = a.k + (long)b * a.ksb + (long)h * a.ksh;
  const float* vb = a.v + (long)b * a.vsb + (long)hv * a.vsh;
  const float* gb = a.dout + (long)b * a.osb + (long)h * a.osh;
  const int fr = lane & 15, fq = lane >> 4;
  {
    TileB<DQ> tk;
    tk.load(kb, a.ksi, a.ksd, j0, a.Nk, a.D, a.k_dfast);
    tk.store(Ks, PQ, nullptr, 0, a.k_dfast);
    TileB<DV> tv;
    tv.load(vb, a.vsi, a.vsd, j0, a.Nk, a.Dv, a.v_dfast);
    tv.store(Vs, PV, nullptr, 0, a.v_dfast);
  }
  f32x4 dk[NQ], dv[NV];
#pragma unroll
  for (int t = 0; t < NQ; ++t) dk[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NV; ++t) dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  TileB<DQ> pq;
  TileB<DV> pg;
  pq.load(qb, a.qsi, a.qsd, 0, a.Nq, a.D, a.q_dfast);
  pg.load(gb, a.osi, a.osd, 0, a.Nq, a.Dv, a.o_dfast);
  for (int i0 = 0; i0 < a.Nq; i0 += TQ) {
    __syncthreads();
    pq.store(Qs, PQ, Qt, PT, a.q_dfast);
    pg.store(dOs, PV, dOt, PT, a.o_dfast);
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
    mma_kc<4>(st, Ks, PQ, wave * 16, Qs, PQ, Dk, lane);     // S^T[j][i]
    mma_kc<4>(dpt, Vs, PV, wave * 16, dOs, PV, Dvk, lane);  // dP^T[j][i]
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
        Pt[jr * PT + ic] = f2bf(p);
        St[jr * PT + ic] = f2bf(p * (dpt[t][r] - delv) * a.scale);
      }
    }
    __syncthreads();
    mma_kc<NV>(dv, Pt, PT, wave * 16, dOt, PT, 64, lane);
    mma_kc<NQ>(dk, St, PT, wave * 16, Qt, PT, 64, lane);
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

static void fill_b(AttnArgsB& a, const cenet_attn_t* p) {
  a.q = p->q; a.k = p->k; a.v = p->v; a.o = p->o; a.lse = p->lse;
  a.dout = p->dout; a.dq = p->dq; a.dk = p->dk; a.dv = p->dv; a.delta = p->delta;
  a.qsb = p->qsb; a.qsh = p->qsh; a.qsi = p->qsi; a.qsd = p->qsd;
  a.ksb = p->ksb; a.ksh = p->ksh; a.ksi = p->ksi; a.ksd = p->ksd;
  a.vsb = p->vsb; a.vsh = p->vsh; a.vsi = p->vsi; a.vsd = p->vsd;
  a.osb = p->osb; a.osh = p->osh; a.osi = p->osi; a.osd = p->osd;
  a.B = p->B; a.H = p->H; a.Nq = p->Nq; a.Nk = p->Nk; a.D = p->D; a.Dv = p->Dv;
  a.v_head_div = p->v_head_div > 0 ? p->v_head_div : 1;
  a.q_dfast = (p->qsd == 1); a.k_dfast = (p->ksd == 1); a.v_dfast = (p->vsd == 1); a.o_dfast = (p->osd == 1);
  a.dv_atomic = (a.v_head_div > 1);
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
  dim3 grid(cdiv(a.Nq, TQ), a.B * a.H);
  switch (pick_b(a.D, a.Dv)) {
    case 0: CENET_LAUNCH((flashb_fwd_kernel<32, 32>), grid, dim3(256), stream, a); break;
    case 1: CENET_LAUNCH((flashb_fwd_kernel<32, 64>), grid, dim3(256), stream, a); break;
    case 2: CENET_LAUNCH((flashb_fwd_kernel<64, 64>), grid, dim3(256), stream, a); break;
    case 3: CENET_LAUNCH((flashb_fwd_kernel<64, 128>), grid, dim3(256), stream, a); break;
    default: return CENET_EUNSUPPORTED;
  }
  return CENET_OK;
}
int cenet_flashb_bwd(const cenet_attn_t* p, hipStream_t stream) {
  AttnArgsB a;
  fill_b(a, p);
  dim3 gq(cdiv(a.Nq, TQ), a.B * a.H), gk(cdiv(a.Nk, TK), a.B * a.H);
#define CENET_BWD(DQv, DVv)                                                             \
  CENET_LAUNCH((flashb_bwd_dq_kernel<DQv, DVv>), gq, dim3(256), stream, a);            \
  CENET_LAUNCH((flashb_bwd_dkv_kernel<DQv, DVv>), gk, dim3(256), stream, a);
  switch (pick_b(a.D, a.Dv)) {
    case 0: CENET_BWD(32, 32) break;
    case 1: CENET_BWD(32, 64) break;
    case 2: CENET_BWD(64, 64) break;
    case 3: CENET_BWD(64, 128) break;
    default: return CENET_EUNSUPPORTED;
  }
#undef CENET_BWD
  return CENET_OK;
}
