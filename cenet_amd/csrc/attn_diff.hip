// attn_diff.hip — differential attention of the DSEB skip blocks on bf16 tensors (throughput mode), forward and backward.
//
// Reference: multihead_diffattn.py:83-109 — q, k [B, N, 2H, hd], v [B, N, H, 2hd]; softmax head 2h+s (s = 0, 1) attends with
// q_{2h+s}, k_{2h+s} over the SHARED value head h:  U[b, 2h+s] = softmax(q k^T / sqrt(hd)) v_h   ([B, 2H, N, 2hd]).
// DSEB-56^2 (ACDC: N = 3136, hd = 16, 8 softmax heads, B = 32) is the largest attention problem of the step: 2.5 G scores
// per pass, 96 MFMA FLOP per score — the exponentials and the softmax algebra (VALU), not the matrix cores, bound it.
//
// Design (wave64, v_mfma_f32_32x32x16_bf16; C/D: lane = column, 16 registers = rows (r&3) + 8(r>>2) + 4(lane>>5)):
//   * no LDS, no barriers: a wave owns 32-row tiles of the side that stays fixed (queries in fwd / dQ, keys in dK/dV) for BOTH
//     softmax heads of a pair, and streams 32-row tiles of the other side as MFMA fragments straight from L2 (a (b, pair)'s
//     K / V working set is a few hundred KB and every workgroup of the pair — placed on one XCD — re-reads it);
//   * scores are computed transposed w.r.t. the fixed side (S^T = K Q^T in fwd / dQ, S = Q K^T in dK/dV), so the fixed index is
//     the MFMA column = the lane: softmax state (max, sum, lse, delta) is one scalar per lane, a row reduction is 15 in-lane
//     ops + one exchange with lane ^ 32, and the probability registers ARE the B operand of the next product (k-step s takes
//     registers 8s..8s+7 = rows 16s + 8(j>>2) + 4(lane>>5) + (j&3)); the matching A operand is read with two 8-byte loads
//     from a TRANSPOSED copy of the streamed tensor ([B, E, N]: the caller passes q^T, k^T, v^T, dU^T next to the row-major
//     tensors — four small transposes per step instead of scattering 2-byte elements through LDS per tile);
//   * hd = 16 is exactly one K = 16 step of the 32x32x16 MFMA (no zero padding of the contraction as with K = 32 tiles);
//     hd = 8 runs zero-padded to 16, hd = 32 as two k-steps; value width 2hd = 32 / 64 = one / two 32-row tiles;
//   * the shared value head: both softmax heads of a pair accumulate dV in the same registers — no atomics anywhere;
//   * base-2 exponentials with scale*log2(e) folded into one fma per score; the running maximum is rescaled only when some
//     lane of the wave actually raised it (wave-uniform branch).
#include "common.h"
#include "../../include/cenet_hip.h"

typedef bf16_t bf;
#define DA_NEG (-1.0e30f)
#define DA_LOG2E 1.4426950408889634f

#ifdef CENET_HOSTSIM_BUILD
__device__ __forceinline__ float da_exp2(float x) { return exp2f(x); }
__device__ __forceinline__ bool da_any(bool v) {
  int x = v ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x |= __shfl_xor(x, o);
  return x != 0;
}
#else
__device__ __forceinline__ float da_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ bool da_any(bool v) { return __any(v); }
#endif

struct DiffArgs {
  const bf *q, *k, *v;     // row-major: q, k [B, N, 2H*hd] ; v [B, N, H*2hd]
  const bf *qt, *kt, *vt;  // transposed: [B, 2H*hd, N], [B, 2H*hd, N], [B, H*2hd, N]
  bf* U;                   // [B, 2H, N, 2hd]
  float* lse;              // [B, 2H, N]  (natural log)
  const bf *dU, *dUt;      // [B, 2H, N, 2hd], [B, 2H, 2hd, N]
  bf *dq, *dk, *dv;        // like q, k, v
  float* delta;            // [B, 2H, N]
  int B, H, N, hd;
  float scale;
};

// 8 consecutive bf16 of one row as an MFMA fragment; `ok` false -> zeros
__device__ __forceinline__ bf16x8 da_ld8(const bf* p, bool ok) {
  bf16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ok) memcpy(&f, p, 16);
  return f;
}
// fragment in the accumulator-row order of k-step s, from a transposed tensor row: elements [4hh .. 4hh+3] and [8 + 4hh .. +3]
// of the 16 columns that start at p; columns >= lim (counted from p) read as zero (ragged last tile), as does !ok
__device__ __forceinline__ bf16x8 da_ldperm(const bf* p, int hh, int lim, bool ok) {
  unsigned long long w0 = 0, w1 = 0;
  if (ok) {
    if (lim >= 16) {
      memcpy(&w0, p + 4 * hh, 8);
      memcpy(&w1, p + 8 + 4 * hh, 8);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (4 * hh + e < lim) w0 |= (unsigned long long)p[4 * hh + e] << (16 * e);
        if (8 + 4 * hh + e < lim) w1 |= (unsigned long long)p[8 + 4 * hh + e] << (16 * e);
      }
    }
  }
  bf16x8 f;
  memcpy(&f, &w0, 8);
  memcpy((char*)&f + 8, &w1, 8);
  return f;
}
__device__ __forceinline__ bf16x8 da_pack8(const f32x16& x, int s) {  // registers 8s .. 8s+7 -> one B fragment
  unsigned u[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) u[j] = cenet_pack_bf2(x[8 * s + 2 * j], x[8 * s + 2 * j + 1]);
  bf16x8 f;
  memcpy(&f, u, 16);
  return f;
}
__device__ __forceinline__ int da_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }
__device__ __forceinline__ f32x16 da_zero() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
#define DA_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------------------------------------------
// forward: wave = QT tiles of 32 queries x both softmax heads of pair h; grid (ceil(N / (128 QT)), B*H)
// ---------------------------------------------------------------------------------------------------------------------------
template <int HDP, int QT>
__global__ __launch_bounds__(256) void dattn_fwd_kernel(DiffArgs a) {
  constexpr int NKS = HDP / 16, DVP = 2 * HDP, NDT = DVP / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
  const cenet_bid bid = cenet_xcd_block();
  const int b = bid.y / a.H, h = bid.y - b * a.H;
  const int q0 = (bid.x * 4 + wave) * (32 * QT);
  if (q0 >= a.N) return;  // wave-uniform; the kernel has no barriers
  const int N = a.N, hd = a.hd, dv = 2 * hd, E = 2 * a.H * hd;
  const float c = a.scale * DA_LOG2E;
  const bf* qb = a.q + (long)b * N * E;
  const bf* kb = a.k + (long)b * N * E;
  const bf* vtb = a.vt + ((long)b * a.H + h) * dv * (long)N;  // rows = value features of head h, columns = keys

  bf16x8 qf[QT][2][NKS];
  f32x16 O[QT][2][NDT];
  float m[QT][2], l[QT][2];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qi = q0 + 32 * t + r < N ? q0 + 32 * t + r : N - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        qf[t][s][ks] = da_ld8(qb + (long)qi * E + (2 * h + s) * hd + 16 * ks + 8 * hh, 16 * ks + 8 * hh < hd);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) O[t][s][dt] = da_zero();
      m[t][s] = DA_NEG;
      l[t][s] = 0.f;
    }
  }
  for (int k0 = 0; k0 < N; k0 += 32) {
    const int kr = k0 + r < N ? k0 + r : N - 1;
    const int klim = N - k0;  // valid keys of this tile (>= 32 except in the last one)
    bf16x8 kf[2][NKS], vf[NDT][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        kf[s][ks] = da_ld8(kb + (long)kr * E + (2 * h + s) * hd + 16 * ks + 8 * hh, 16 * ks + 8 * hh < hd);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        vf[dt][s2] = da_ldperm(vtb + (long)(32 * dt + r) * N + k0 + 16 * s2, hh, klim - 16 * s2, 32 * dt + r < dv);
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x16 S = da_zero();
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) S = DA_MFMA(kf[s][ks], qf[t][s][ks], S);  // S^T[key][query]
        if (klim < 32) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (da_row(i, hh) >= klim) S[i] = DA_NEG;
        }
        float mx = S[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, S[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (da_any(mx > m[t][s])) {  // some query of the wave raised its maximum: rescale (wave-uniform branch)
          const float mn = fmaxf(m[t][s], mx);
          const float alpha = da_exp2((m[t][s] - mn) * c);
          m[t][s] = mn;
          l[t][s] *= alpha;
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[t][s][dt][i] *= alpha;
        }
        const float mc = m[t][s] * c;
        float rs = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = da_exp2(S[i] * c - mc);
          S[i] = p;
          rs += p;
        }
        l[t][s] += rs;
        const bf16x8 p0 = da_pack8(S, 0), p1 = da_pack8(S, 1);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {  // O^T[value feature][query] += V^T[feature][key] P^T[key][query]
          O[t][s][dt] = DA_MFMA(vf[dt][0], p0, O[t][s][dt]);
          O[t][s][dt] = DA_MFMA(vf[dt][1], p1, O[t][s][dt]);
        }
      }
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qi = q0 + 32 * t + r;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float lt = l[t][s] + __shfl_xor(l[t][s], 32);
      if (qi < N) {
        const float inv = 1.f / lt;
        bf* up = a.U + (((long)b * 2 * a.H + 2 * h + s) * N + qi) * dv;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int f0 = 32 * dt + 8 * g + 4 * hh;  // four consecutive value features
            if (f0 < dv) {
              float o[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) o[i] = O[t][s][dt][4 * g + i] * inv;
              st4v(up + f0, o);
            }
          }
        if (hh == 0) a.lse[((long)b * 2 * a.H + 2 * h + s) * N + qi] = m[t][s] * a.scale + logf(lt);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward, dQ (and delta = rowsum(dU * U)): wave = 32 queries x both softmax heads; streams key tiles
// ---------------------------------------------------------------------------------------------------------------------------
template <int HDP>
__global__ __launch_bounds__(256) void dattn_bwd_dq_kernel(DiffArgs a) {
  constexpr int NKS = HDP / 16, DVP = 2 * HDP, NKD = DVP / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
  const cenet_bid bid = cenet_xcd_block();
  const int b = bid.y / a.H, h = bid.y - b * a.H;
  const int q0 = (bid.x * 4 + wave) * 32;
  if (q0 >= a.N) return;
  const int N = a.N, hd = a.hd, dv = 2 * hd, E = 2 * a.H * hd;
  const float c = a.scale * DA_LOG2E;
  const bf* qb = a.q + (long)b * N * E;
  const bf* kb = a.k + (long)b * N * E;
  const bf* vb = a.v + (long)b * N * (a.H * dv) + h * dv;
  const bf* ktb = a.kt + (long)b * E * (long)N;
  const int qi = q0 + r < N ? q0 + r : N - 1;

  bf16x8 qf[2][NKS], gf[2][NKD];
  float lse2[2], dl[2];
  f32x16 dq[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const long hrow = ((long)b * 2 * a.H + 2 * h + s) * N + qi;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      qf[s][ks] = da_ld8(qb + (long)qi * E + (2 * h + s) * hd + 16 * ks + 8 * hh, 16 * ks + 8 * hh < hd);
    float acc = 0.f;
#pragma unroll
    for (int kd = 0; kd < NKD; ++kd) {
      const bool ok = 16 * kd + 8 * hh < dv;
      gf[s][kd] = da_ld8(a.dU + hrow * dv + 16 * kd + 8 * hh, ok);
      const bf16x8 uf = da_ld8(a.U + hrow * dv + 16 * kd + 8 * hh, ok);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += cenet_bf2f((unsigned short)gf[s][kd][j]) * cenet_bf2f((unsigned short)uf[j]);
    }
    acc += __shfl_xor(acc, 32);
    dl[s] = acc;
    lse2[s] = a.lse[hrow] * DA_LOG2E;
    if (hh == 0 && q0 + r < N) a.delta[hrow] = acc;
    dq[s] = da_zero();
  }
  for (int k0 = 0; k0 < N; k0 += 32) {
    const int kr = k0 + r < N ? k0 + r : N - 1;
    const int klim = N - k0;
    bf16x8 vf[NKD];
#pragma unroll
    for (int kd = 0; kd < NKD; ++kd) vf[kd] = da_ld8(vb + (long)kr * (a.H * dv) + 16 * kd + 8 * hh, 16 * kd + 8 * hh < dv);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x16 S = da_zero(), dP = da_zero();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8 kf = da_ld8(kb + (long)kr * E + (2 * h + s) * hd + 16 * ks + 8 * hh, 16 * ks + 8 * hh < hd);
        S = DA_MFMA(kf, qf[s][ks], S);  // S^T[key][query]
      }
#pragma unroll
      for (int kd = 0; kd < NKD; ++kd) dP = DA_MFMA(vf[kd], gf[s][kd], dP);  // dP^T[key][query] = V dU^T
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float p = da_exp2(S[i] * c - lse2[s]);
        if (klim < 32 && da_row(i, hh) >= klim) p = 0.f;
        S[i] = p * (dP[i] - dl[s]);
      }
      // dQ^T[d][query] += K^T[d][key] dS^T[key][query]   (rows d >= hd carry zero fragments)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 ktf = da_ldperm(ktb + (long)((2 * h + s) * hd + r) * N + k0 + 16 * s2, hh, klim - 16 * s2, r < hd);
        dq[s] = DA_MFMA(ktf, da_pack8(S, s2), dq[s]);
      }
    }
  }
  if (q0 + r < N) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf* dp = a.dq + ((long)b * N + qi) * E + (2 * h + s) * hd;
#pragma unroll
      for (int g = 0; g < HDP / 8; ++g) {
        const int d0 = 8 * g + 4 * hh;
        if (d0 < hd) {
          float o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = dq[s][4 * g + i] * a.scale;
          st4v(dp + d0, o);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward, dK / dV: wave = 32 keys x both softmax heads (one shared dV accumulator); streams query tiles
// ---------------------------------------------------------------------------------------------------------------------------
template <int HDP>
__global__ __launch_bounds__(256) void dattn_bwd_dkv_kernel(DiffArgs a) {
  constexpr int NKS = HDP / 16, DVP = 2 * HDP, NKD = DVP / 16, NDT = DVP / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
  const cenet_bid bid = cenet_xcd_block();
  const int b = bid.y / a.H, h = bid.y - b * a.H;
  const int k0 = (bid.x * 4 + wave) * 32;
  if (k0 >= a.N) return;
  const int N = a.N, hd = a.hd, dv = 2 * hd, E = 2 * a.H * hd;
  const float c = a.scale * DA_LOG2E;
  const bf* qb = a.q + (long)b * N * E;
  const bf* kb = a.k + (long)b * N * E;
  const bf* vb = a.v + (long)b * N * (a.H * dv) + h * dv;
  const bf* qtb = a.qt + (long)b * E * (long)N;
  const int ki = k0 + r < N ? k0 + r : N - 1;

  bf16x8 kfB[2][NKS], vfB[NKD];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      kfB[s][ks] = da_ld8(kb + (long)ki * E + (2 * h + s) * hd + 16 * ks + 8 * hh, 16 * ks + 8 * hh < hd);
#pragma unroll
  for (int kd = 0; kd < NKD; ++kd) vfB[kd] = da_ld8(vb + (long)ki * (a.H * dv) + 16 * kd + 8 * hh, 16 * kd + 8 * hh < dv);
  f32x16 dK[2], dV[NDT];
  dK[0] = dK[1] = da_zero();
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) dV[dt] = da_zero();

  for (int q0 = 0; q0 < N; q0 += 32) {
    const int qr = q0 + r < N ? q0 + r : N - 1;
    const int qlim = N - q0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const long hbase = ((long)b * 2 * a.H + 2 * h + s) * N;
      f32x16 S = da_zero(), dP = da_zero();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8 qfA = da_ld8(qb + (long)qr * E + (2 * h + s) * hd + 16 * ks + 8 * hh, 16 * ks + 8 * hh < hd);
        S = DA_MFMA(qfA, kfB[s][ks], S);  // S[query][key]
      }
#pragma unroll
      for (int kd = 0; kd < NKD; ++kd) {
        const bf16x8 gfA = da_ld8(a.dU + (hbase + qr) * dv + 16 * kd + 8 * hh, 16 * kd + 8 * hh < dv);
        dP = DA_MFMA(gfA, vfB[kd], dP);  // dP[query][key] = dU V^T
      }
      // per-row (query) softmax statistics: rows 8g + 4hh + 0..3
      f32x16 dS;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float l4[4], d4[4];
        const int qq = q0 + 8 * g + 4 * hh;
        if (qq + 3 < N) {
          memcpy(l4, a.lse + hbase + qq, 16);
          memcpy(d4, a.delta + hbase + qq, 16);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            l4[i] = qq + i < N ? a.lse[hbase + qq + i] : 1.0e30f;  // rows beyond N: p = 0
            d4[i] = qq + i < N ? a.delta[hbase + qq + i] : 0.f;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = da_exp2(S[4 * g + i] * c - l4[i] * DA_LOG2E);
          S[4 * g + i] = p;
          dS[4 * g + i] = p * (dP[4 * g + i] - d4[i]);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pB = da_pack8(S, s2), dsB = da_pack8(dS, s2);
        // dV^T[feature][key] += dU^T[feature][query] P[query][key]
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const bf16x8 gtf = da_ldperm(a.dUt + (((long)b * 2 * a.H + 2 * h + s) * dv + 32 * dt + r) * N + q0 + 16 * s2, hh,
                                       qlim - 16 * s2, 32 * dt + r < dv);
          dV[dt] = DA_MFMA(gtf, pB, dV[dt]);
        }
        // dK^T[d][key] += Q^T[d][query] dS[query][key]
        const bf16x8 qtf = da_ldperm(qtb + (long)((2 * h + s) * hd + r) * N + q0 + 16 * s2, hh, qlim - 16 * s2, r < hd);
        dK[s] = DA_MFMA(qtf, dsB, dK[s]);
      }
    }
  }
  if (k0 + r < N) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf* dp = a.dk + ((long)b * N + ki) * E + (2 * h + s) * hd;
#pragma unroll
      for (int g = 0; g < HDP / 8; ++g) {
        const int d0 = 8 * g + 4 * hh;
        if (d0 < hd) {
          float o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = dK[s][4 * g + i] * a.scale;
          st4v(dp + d0, o);
        }
      }
    }
    bf* dvp = a.dv + ((long)b * N + ki) * (a.H * dv) + h * dv;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f0 = 32 * dt + 8 * g + 4 * hh;
        if (f0 < dv) {
          float o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = dV[dt][4 * g + i];
          st4v(dvp + f0, o);
        }
      }
  }
}

// head dims served: 8 and 16 (one k-step), 32 (two); every tensor 8-byte aligned (hd % 4 == 0 makes every row offset so)
extern "C" int cenet_diffattn_heads_supported(int hd, int N) { return (hd == 8 || hd == 16 || hd == 32) && N >= 1; }

static int da_fill(DiffArgs& a, const cenet_diffattn_t* p) {
  if (!p || !p->q || !p->k || !p->vt || !p->U || !p->lse) return CENET_EINVAL;
  if (p->B <= 0 || p->H <= 0 || p->N <= 0 || !cenet_diffattn_heads_supported(p->hd, p->N)) return CENET_EUNSUPPORTED;
  a.q = (const bf*)p->q; a.k = (const bf*)p->k; a.v = (const bf*)p->v;
  a.qt = (const bf*)p->qt; a.kt = (const bf*)p->kt; a.vt = (const bf*)p->vt;
  a.U = (bf*)p->U; a.lse = p->lse; a.dU = (const bf*)p->dU; a.dUt = (const bf*)p->dUt;
  a.dq = (bf*)p->dq; a.dk = (bf*)p->dk; a.dv = (bf*)p->dv; a.delta = p->delta;
  a.B = p->B; a.H = p->H; a.N = p->N; a.hd = p->hd; a.scale = p->scale;
  const uintptr_t m = (uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.qt | (uintptr_t)a.kt | (uintptr_t)a.vt |
                      (uintptr_t)a.U | (uintptr_t)a.dU | (uintptr_t)a.dUt | (uintptr_t)a.dq | (uintptr_t)a.dk | (uintptr_t)a.dv;
  if (m & 15) return CENET_EINVAL;  // fragments are 16-byte loads
  if (a.N & 3) return CENET_EUNSUPPORTED;  // 8-byte runs of the transposed tensors need N % 4 == 0
  return CENET_OK;
}

extern "C" int cenet_diffattn_heads_fwd_bf16(const cenet_diffattn_t* p, hipStream_t stream) {
  DiffArgs a;
  const int rc = da_fill(a, p);
  if (rc != CENET_OK) return rc;
  const dim3 bh(1, a.B * a.H);
  if (a.hd <= 16) {
    if (a.N >= 1024) CENET_LAUNCH((dattn_fwd_kernel<16, 2>), dim3(cdiv(a.N, 256), bh.y), dim3(256), stream, a);
    else CENET_LAUNCH((dattn_fwd_kernel<16, 1>), dim3(cdiv(a.N, 128), bh.y), dim3(256), stream, a);
  } else {
    CENET_LAUNCH((dattn_fwd_kernel<32, 1>), dim3(cdiv(a.N, 128), bh.y), dim3(256), stream, a);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_diffattn_heads_bwd_bf16(const cenet_diffattn_t* p, hipStream_t stream) {
  DiffArgs a;
  const int rc = da_fill(a, p);
  if (rc != CENET_OK) return rc;
  if (!a.v || !a.qt || !a.kt || !a.dU || !a.dUt || !a.dq || !a.dk || !a.dv || !a.delta) return CENET_EINVAL;
  const dim3 grid(cdiv(a.N, 128), a.B * a.H);
  if (a.hd <= 16) {
    CENET_LAUNCH((dattn_bwd_dq_kernel<16>), grid, dim3(256), stream, a);
    CENET_LAUNCH((dattn_bwd_dkv_kernel<16>), grid, dim3(256), stream, a);
  } else {
    CENET_LAUNCH((dattn_bwd_dq_kernel<32>), grid, dim3(256), stream, a);
    CENET_LAUNCH((dattn_bwd_dkv_kernel<32>), grid, dim3(256), stream, a);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
