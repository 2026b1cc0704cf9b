// attn_diff.hip — differential attention of the DSEB skip blocks on bf16 tensors (throughput mode), forward and backward.
//
// Reference: multihead_diffattn.py:83-109 — q, k [B, N, 2H, hd], v [B, N, H, 2hd]; softmax head 2h+s (s = 0, 1) attends with
// q_{2h+s}, k_{2h+s} over the SHARED value head h:  U[b, 2h+s] = softmax(q k^T / sqrt(hd)) v_h   ([B, 2H, N, 2hd]).
// DSEB-56^2 (ACDC: N = 3136, hd = 16, 8 softmax heads, B = 32) is the largest attention problem of the step: 2.5 G scores
// per pass, 96 MFMA FLOP per score — the exponentials and the softmax algebra (VALU), not the matrix cores, bound it.
//
// Design (wave64, v_mfma_f32_32x32x16_bf16; C/D: lane = column, 16 registers = rows (r&3) + 8(r>>2) + 4(lane>>5)):
//   * a wave owns 32-row tiles of the side that stays fixed (queries in fwd / dQ, keys in dK/dV) for BOTH softmax heads of
//     a pair; the four waves of a workgroup share the streamed side: each 32-row tile is fetched once per workgroup with
//     16-byte loads (register prefetch one tile ahead) into a double-buffered ROW-MAJOR bf16 image in LDS, one barrier per
//     tile;
//   * scores are computed transposed w.r.t. the fixed side (S^T = K Q^T in fwd / dQ, S = Q K^T in dK/dV), so the fixed index is
//     the MFMA column = the lane: softmax state (max, sum, lse, delta) is one scalar per lane, a row reduction is 15 in-lane
//     ops + one exchange with lane ^ 32, and the probability registers ARE the B operand of the next product (k-step s takes
//     registers 8s..8s+7 = rows 16s + 8(j>>2) + 4(lane>>5) + (j&3));
//   * the A operand of that next product runs along the token index of the streamed tile: it is read from the same row-major
//     image with the transposing LDS read (ds_read_b64_tr_b16: a 16-lane group gets a 4-row x 16-column block column-major),
//     two per fragment — no transposed copy in HBM, no 2-byte scatter into LDS;
//   * hd = 16 is exactly one K = 16 step of the 32x32x16 MFMA (no zero padding of the contraction as with K = 32 tiles);
//     hd = 8 runs zero-padded to 16, hd = 32 as two k-steps; value width 2hd = 32 / 64 = one / two 32-row tiles;
//   * the shared value head: both softmax heads of a pair accumulate dV in the same registers — no atomics anywhere;
//   * in dK/dV the softmax statistics belong to the ROWS (queries) of the score tile; instead of 2 x 16 loads and subtractions
//     per lane they ride in the contraction: the dQ kernel leaves lse / scale and delta split into three bf16 terms each
//     (24 significant bits) per (head, query), and one extra k-step against a constant -1 fragment gives S - lse/scale and
//     dP - delta straight out of the MFMA;
//   * base-2 exponentials with scale*log2(e) folded into one multiply per score; the running maximum of the forward pass is
//     rescaled only when some lane of the wave actually raised it (wave-uniform branch).
#include "common.h"
#include "../../include/cenet_hip.h"
#include <cstdlib>

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
typedef short da_s4 __attribute__((ext_vector_type(4)));
#endif
// max of three; built with -fno-honor-nans (cenet_amd/build.py) the compiler emits v_max3_f32 without the canonicalising
// v_max it otherwise puts in front of fmaxf on MFMA results.  (NOT inline asm: hipcc pads no MFMA -> VALU wait states for
// an asm statement that reads accumulator registers, cdna_hip_programming.md §5.7 — the asm form read stale scores.)
__device__ __forceinline__ float da_max3(float a, float b, float c3) { return fmaxf(fmaxf(a, b), c3); }
// wave-level ordering point between LDS stores and loads of the SAME wave
__device__ __forceinline__ void da_wave_sync() {
#ifdef CENET_HOSTSIM_BUILD
  hipsim::wave_barrier();
#else
  __builtin_amdgcn_wave_barrier();
#endif
}

struct DiffArgs {
  const bf *q, *k, *v;  // row-major: q, k [B, N, 2H*hd] ; v [B, N, H*2hd]
  bf* U;                // [B, 2H, N, 2hd]
  float* lse;           // [B, 2H, N]  (natural log)
  const bf* dU;         // [B, 2H, N, 2hd]
  bf *dq, *dk, *dv;     // like q, k, v
  bf* aug;              // [B, 2H, N, 16]: lse/scale as 3 bf16 terms at [0..2], delta at [8..10] (written by dQ, read by dK/dV)
  int B, H, N, hd;
  float scale;
  int bmul;             // batch stride of q / k / v / dq / dk / dv in units of their own [N, row] block (1: dense)
};

// ---- LDS images -------------------------------------------------------------------------------------------------------------
// One staged 16-byte chunk of a thread: 8 consecutive elements of one global row -> 8 consecutive elements of one image row.
struct DaChunk {
  const bf* g;   // global address of the chunk in tile 0 (nullptr: the chunk lies in the zero padding of the image)
  int lrow;      // row inside the 32-row tile
  int loff;      // element offset inside the image
  unsigned fill; // low word of the 16-byte fill pattern for rows beyond the tensor (rest zero)
};
__device__ __forceinline__ void da_fetch(const DaChunk& c, long tile_step, int rows_left, unsigned (&r)[4]) {
  r[0] = c.fill;
  r[1] = r[2] = r[3] = 0u;
  if (c.g && c.lrow < rows_left) memcpy(r, c.g + tile_step, 16);
  else if (!c.g) r[0] = 0u;
}
// row-major fragment: 8 consecutive columns [col + 8hh, +8) of image row `row`
__device__ __forceinline__ bf16x8 da_rm(const bf* img, int pitch, int row, int col, int hh) {
  bf16x8 f;
  memcpy(&f, img + row * pitch + col + 8 * hh, 16);
  return f;
}
// transposed fragment for k-step s2: lane (r = lane & 31, hh) gets column C0 + r of image rows 16 s2 + 4hh + {0..3} and
// 16 s2 + 8 + 4hh + {0..3} — the accumulator-row order of da_pack8.  Device: two ds_read_b64_tr_b16 (per 16-lane group a
// 4-row x 16-column block: lane 4q+p supplies the address of row q, columns 4p..4p+3, lane i receives column i).
__device__ __forceinline__ bf16x8 da_tr(const bf* img, int pitch, int s2, int C0, int lane) {
  const int hh = lane >> 5;
  bf16x8 f;
#ifdef CENET_HOSTSIM_BUILD
  const int r = lane & 31;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (short)img[(16 * s2 + 8 * (j >> 2) + 4 * hh + (j & 3)) * pitch + C0 + r];
#else
  const int g2 = (lane >> 4) & 1, i = lane & 15, q = i >> 2, p = i & 3;
  const bf* a0 = img + (16 * s2 + 4 * hh + q) * pitch + C0 + 16 * g2 + 4 * p;
  const da_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((da_s4 __attribute__((address_space(3)))*)a0);
  const da_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((da_s4 __attribute__((address_space(3)))*)(a0 + 8 * pitch));
  f[0] = lo[0], f[1] = lo[1], f[2] = lo[2], f[3] = lo[3];
  f[4] = hi[0], f[5] = hi[1], f[6] = hi[2], f[7] = hi[3];
#endif
  return f;
}
__device__ __forceinline__ bf16x8 da_ld8(const bf* p, bool ok) {
  bf16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ok) memcpy(&f, p, 16);
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
struct da_true { static constexpr bool value = true; };
struct da_false { static constexpr bool value = false; };
#define DA_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// x as hi + mid + lo with bf16 terms (24 significant bits)
__device__ __forceinline__ void da_split3(float x, unsigned short* o) {
  const unsigned h = cenet_f2bf(x);
  const float r1 = x - cenet_bf2f(h);
  const unsigned m = cenet_f2bf(r1);
  const unsigned l = cenet_f2bf(r1 - cenet_bf2f(m));
  o[0] = (unsigned short)h, o[1] = (unsigned short)m, o[2] = (unsigned short)l;
}

// image geometry (elements): K / Q image = both heads of the pair side by side, V / dU image = one value head
template <int HDP>
struct DaGeo {
  static constexpr int DVP = 2 * HDP;
  static constexpr int KP = 2 * HDP + 8;   // pitch of the K / Q image (16-byte row reads conflict-free)
  static constexpr int VPT = DVP;          // pitch of a value image that is only read transposed (forward)
  static constexpr int VPR = DVP + 8;      // pitch of a value image that is also read by rows
  static constexpr int CK = 2 * HDP / 8, CV = DVP / 8;  // 16-byte chunks per row
};

// ---------------------------------------------------------------------------------------------------------------------------
// forward: wave = QT tiles of 32 queries x both softmax heads of pair h; workgroup = 4 waves; grid (ceil(N / (128 QT)), B*H)
// ---------------------------------------------------------------------------------------------------------------------------
// ONE: a single softmax over BOTH halves of the pair's 2*HDP query / key columns (one head of dimension 2*HDP = value width):
// plain attention with head dimension 64 through the same tiles — the two half-products add into one score tile.  Tensors:
// q, k, v [B, N, H*2HDP], U [B, H, N, 2HDP], lse / aug rows [B, H, N].
template <int HDP, int QT, int MINB, bool ONE>
__global__ __launch_bounds__(256, MINB) void dattn_fwd_kernel(DiffArgs a) {
  typedef DaGeo<HDP> G;
  constexpr int NKS = HDP / 16, NDT = G::DVP / 32, NH = ONE ? 1 : 2;
  constexpr int KIMG = 32 * G::KP, VIMG = 32 * G::VPT + 64, IMG = KIMG + VIMG;  // (+64: transposed reads of padded rows)
  constexpr int NCH = 32 * (G::CK + G::CV), CPT = (NCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf lds[2 * IMG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const cenet_bid bid = cenet_xcd_block();
  const int b = bid.y / a.H, h = bid.y - b * a.H;
  const int q0 = (bid.x * 4 + wave) * (32 * QT);
  const int N = a.N, hd = a.hd, dv = 2 * hd, E = 2 * a.H * hd;
  const float c = a.scale * DA_LOG2E;
  const bf* qb = a.q + (long)b * a.bmul * N * E;
  const bf* kb = a.k + (long)b * a.bmul * N * E;
  const bf* vb = a.v + (long)b * a.bmul * N * (a.H * dv) + h * dv;

  // this thread's staged chunks (fixed for the whole key loop)
  DaChunk ch[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int id = tid + 256 * j;
    ch[j].fill = 0u;
    if (id >= NCH) {
      ch[j].g = nullptr;
      ch[j].lrow = 0;
      ch[j].loff = -1;
    } else if (id < 32 * G::CK) {
      const int row = id / G::CK, cc = id - row * G::CK, s = cc / (HDP / 8), c8 = cc - s * (HDP / 8);
      ch[j].lrow = row;
      ch[j].loff = row * G::KP + s * HDP + 8 * c8;
      ch[j].g = 8 * c8 < hd ? kb + (long)row * E + (2 * h + s) * hd + 8 * c8 : nullptr;
    } else {
      const int id2 = id - 32 * G::CK, row = id2 / G::CV, c8 = id2 - row * G::CV;
      ch[j].lrow = row;
      ch[j].loff = KIMG + row * G::VPT + 8 * c8;
      ch[j].g = 8 * c8 < dv ? vb + (long)row * (a.H * dv) + 8 * c8 : nullptr;
    }
  }
  unsigned pre[CPT][4];
  auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < CPT; ++j)
      da_fetch(ch[j], (long)k0 * (ch[j].loff < KIMG ? E : a.H * dv), N - k0, pre[j]);
  };
  auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < CPT; ++j)
      if (ch[j].loff >= 0) memcpy(lds + buf * IMG + ch[j].loff, pre[j], 16);
  };

  bf16x8 qf[QT][2][NKS];
  f32x16 O[QT][NH][NDT];
  float m[QT][NH], l[QT][NH];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qi = q0 + 32 * t + r < N ? q0 + 32 * t + r : N - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        qf[t][s][ks] = da_ld8(qb + (long)qi * E + (2 * h + s) * hd + 16 * ks + 8 * hh, 16 * ks + 8 * hh < hd);
    }
#pragma unroll
    for (int s = 0; s < NH; ++s) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) O[t][s][dt] = da_zero();
      m[t][s] = DA_NEG;
      l[t][s] = 0.f;
    }
  }
  fetch(0);
  stage(0);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < N; k0 += 32) {
    const bool more = k0 + 32 < N;
    if (more) fetch(k0 + 32);
    const bf* Kimg = lds + cur * IMG;
    const bf* Vimg = Kimg + KIMG;
    const int klim = N - k0;  // valid keys of this tile (>= 32 except in the last one)
    auto tile = [&](auto masked) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked)::value;
      // HDP <= 32: the tile's K / V fragments are read once and shared by both heads (and all QT query tiles).  HDP = 64 has
      // 128 accumulator registers per wave: fragments are read where they are used (LATE) — holding them across the softmax
      // cost 640 bytes of scratch per lane.
      constexpr bool LATE = HDP > 32;
      bf16x8 kf[LATE ? 1 : 2][NKS], vf[LATE ? 1 : NDT][2];
      if (!LATE) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) kf[LATE ? 0 : s][ks] = da_rm(Kimg, G::KP, r, s * HDP + 16 * ks, hh);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) vf[LATE ? 0 : dt][s2] = da_tr(Vimg, G::VPT, s2, 32 * dt, lane);
      }
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int s = 0; s < NH; ++s) {
          f32x16 S = da_zero();
#pragma unroll
          for (int sp = (ONE ? 0 : s); sp <= (ONE ? 1 : s); ++sp)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
              S = DA_MFMA(LATE ? da_rm(Kimg, G::KP, r, sp * HDP + 16 * ks, hh) : kf[LATE ? 0 : sp][ks], qf[t][sp][ks], S);  // S^T[key][query]
          if (MASKED) {  // ragged last tile only (its own instantiation of the tile body: no per-score selects elsewhere)
#pragma unroll
            for (int i = 0; i < 16; ++i)
              if (da_row(i, hh) >= klim) S[i] = DA_NEG;
          }
          float mx = da_max3(S[0], S[1], S[2]);
#pragma unroll
          for (int i = 3; i < 15; i += 2) mx = da_max3(mx, S[i], S[i + 1]);
          mx = fmaxf(mx, S[15]);
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
            O[t][s][dt] = DA_MFMA(LATE ? da_tr(Vimg, G::VPT, 0, 32 * dt, lane) : vf[LATE ? 0 : dt][0], p0, O[t][s][dt]);
            O[t][s][dt] = DA_MFMA(LATE ? da_tr(Vimg, G::VPT, 1, 32 * dt, lane) : vf[LATE ? 0 : dt][1], p1, O[t][s][dt]);
          }
        }
    };
    if (q0 < N) {  // wave-uniform: a wave past the last query only helps with the staging
      if (klim < 32) tile(da_true());
      else tile(da_false());
    }
    if (more) stage(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (q0 >= N) return;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qi = q0 + 32 * t + r;
#pragma unroll
    for (int s = 0; s < NH; ++s) {
      const float lt = l[t][s] + __shfl_xor(l[t][s], 32);
      const long hrow = ((long)b * NH * a.H + NH * h + s) * N + qi;
      if (qi < N) {
        const float inv = 1.f / lt;
        bf* up = a.U + hrow * dv;
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
        if (hh == 0) a.lse[hrow] = m[t][s] * a.scale + logf(lt);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward, dQ (+ delta = rowsum(dU * U) and the bf16 statistics rows for dK/dV): wave = 32 queries x both softmax heads
// ---------------------------------------------------------------------------------------------------------------------------
// (HDP <= 16: three workgroups per CU — 168 registers, nothing spilled; the kernel waits on dependent MFMA / exponential chains, a
// third wave per SIMD covers more of them: 0.78 -> 0.74 ms at DSEB-56x56, same box)
template <int HDP, bool ONE>
__global__ __launch_bounds__(256, HDP > 32 ? 1 : (HDP <= 16 ? 3 : 2)) void dattn_bwd_dq_kernel(DiffArgs a) {
  typedef DaGeo<HDP> G;
  constexpr int NKS = HDP / 16, NKD = G::DVP / 16, NH = ONE ? 1 : 2;
  constexpr int NFT = (HDP + 31) / 32;  // 32-feature accumulator tiles per head (HDP = 64: two)
  constexpr int KIMG = 32 * G::KP + 64, VIMG = 32 * G::VPR, IMG = KIMG + VIMG;
  constexpr int NCH = 32 * (G::CK + G::CV), CPT = (NCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf lds[2 * IMG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const cenet_bid bid = cenet_xcd_block();
  const int b = bid.y / a.H, h = bid.y - b * a.H;
  const int q0 = (bid.x * 4 + wave) * 32;
  const int N = a.N, hd = a.hd, dv = 2 * hd, E = 2 * a.H * hd;
  const float c = a.scale * DA_LOG2E;
  const bf* qb = a.q + (long)b * a.bmul * N * E;
  const bf* kb = a.k + (long)b * a.bmul * N * E;
  const bf* vb = a.v + (long)b * a.bmul * N * (a.H * dv) + h * dv;
  const int qi = q0 + r < N ? q0 + r : N - 1;

  DaChunk ch[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int id = tid + 256 * j;
    ch[j].fill = 0u;
    if (id >= NCH) {
      ch[j].g = nullptr;
      ch[j].lrow = 0;
      ch[j].loff = -1;
    } else if (id < 32 * G::CK) {
      const int row = id / G::CK, cc = id - row * G::CK, s = cc / (HDP / 8), c8 = cc - s * (HDP / 8);
      ch[j].lrow = row;
      ch[j].loff = row * G::KP + s * HDP + 8 * c8;
      ch[j].g = 8 * c8 < hd ? kb + (long)row * E + (2 * h + s) * hd + 8 * c8 : nullptr;
    } else {
      const int id2 = id - 32 * G::CK, row = id2 / G::CV, c8 = id2 - row * G::CV;
      ch[j].lrow = row;
      ch[j].loff = KIMG + row * G::VPR + 8 * c8;
      ch[j].g = 8 * c8 < dv ? vb + (long)row * (a.H * dv) + 8 * c8 : nullptr;
    }
  }
  unsigned pre[CPT][4];
  auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < CPT; ++j)
      da_fetch(ch[j], (long)k0 * (ch[j].loff < KIMG ? E : a.H * dv), N - k0, pre[j]);
  };
  auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < CPT; ++j)
      if (ch[j].loff >= 0) memcpy(lds + buf * IMG + ch[j].loff, pre[j], 16);
  };

  bf16x8 qf[2][NKS], gf[NH][NKD];
  float lse2[NH], dl[NH];
  f32x16 dq[2][NFT];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      qf[s][ks] = da_ld8(qb + (long)qi * E + (2 * h + s) * hd + 16 * ks + 8 * hh, 16 * ks + 8 * hh < hd);
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) dq[s][ft] = da_zero();
  }
#pragma unroll
  for (int s = 0; s < NH; ++s) {
    const long hrow = ((long)b * NH * a.H + NH * h + s) * N + qi;
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
    const float lse = a.lse[hrow];
    lse2[s] = lse * DA_LOG2E;
    if (q0 + r < N) {  // statistics rows for the dK/dV kernel: lane half 0 writes lse / scale, half 1 writes delta
      unsigned short row8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      da_split3(hh == 0 ? lse / a.scale : acc, row8);
      memcpy(a.aug + hrow * 16 + 8 * hh, row8, 16);
    }
  }
  fetch(0);
  stage(0);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < N; k0 += 32) {
    const bool more = k0 + 32 < N;
    if (more) fetch(k0 + 32);
    const bf* Kimg = lds + cur * IMG;
    const bf* Vimg = Kimg + KIMG;
    const int klim = N - k0;
    auto tile = [&](auto masked) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked)::value;
      bf16x8 vf[NKD];
#pragma unroll
      for (int kd = 0; kd < NKD; ++kd) vf[kd] = da_rm(Vimg, G::VPR, r, 16 * kd, hh);
#pragma unroll
      for (int s = 0; s < NH; ++s) {
        // the query is the MFMA column: its delta is a lane constant, so the dP chain can start from -delta (dP - delta comes out
        // of the accumulator: one VALU instruction per score less).  Not at HDP = 16: a zero accumulator is an inline constant of
        // the first MFMA, a -delta one is 16 live registers more, and that instance sits on its 168-register cap (30 spilled).
        constexpr bool INITD = HDP > 16;
        f32x16 S = da_zero(), dP;
#pragma unroll
        for (int i = 0; i < 16; ++i) dP[i] = INITD ? -dl[s] : 0.f;
#pragma unroll
        for (int sp = (ONE ? 0 : s); sp <= (ONE ? 1 : s); ++sp)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) S = DA_MFMA(da_rm(Kimg, G::KP, r, sp * HDP + 16 * ks, hh), qf[sp][ks], S);
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd) dP = DA_MFMA(vf[kd], gf[s][kd], dP);  // dP^T[key][query] = V dU^T - delta
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float p = da_exp2(S[i] * c - lse2[s]);
          if (MASKED && da_row(i, hh) >= klim) p = 0.f;
          S[i] = INITD ? p * dP[i] : p * (dP[i] - dl[s]);
        }
        // dQ^T[d][query] += K^T[d][key] dS^T[key][query]; accumulator rows >= hd (the other head's columns, padding) are
        // never stored
#pragma unroll
        for (int sp = (ONE ? 0 : s); sp <= (ONE ? 1 : s); ++sp)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 dsB = da_pack8(S, s2);
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft)
              dq[sp][ft] = DA_MFMA(da_tr(Kimg, G::KP, s2, sp * HDP + 32 * ft, lane), dsB, dq[sp][ft]);
          }
      }
    };
    if (q0 < N) {
      if (klim < 32) tile(da_true());
      else tile(da_false());
    }
    if (more) stage(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (q0 + r < N) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf* dp = a.dq + ((long)b * a.bmul * N + qi) * E + (2 * h + s) * hd;
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int g = 0; g < (HDP < 32 ? HDP / 8 : 4); ++g) {
          const int d0 = 32 * ft + 8 * g + 4 * hh;
          if (d0 < hd) {
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = dq[s][ft][4 * g + i] * a.scale;
            st4v(dp + d0, o);
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward, dK / dV: wave = 32 keys x both softmax heads (one shared dV accumulator); streams query tiles
// ---------------------------------------------------------------------------------------------------------------------------
template <int HDP, bool ONE>
__global__ __launch_bounds__(256, HDP > 32 ? 1 : 2) void dattn_bwd_dkv_kernel(DiffArgs a) {
  typedef DaGeo<HDP> G;
  constexpr int NKS = HDP / 16, NKD = G::DVP / 16, NDT = G::DVP / 32, NH = ONE ? 1 : 2;
  constexpr int NFT = (HDP + 31) / 32;  // 32-feature accumulator tiles of dK per head
  constexpr int AP = 40;  // statistics image: [query][head s: lse(8) delta(8)] + pad
  constexpr int QIMG = 32 * G::KP + 64, GIMG = 32 * G::VPR + 64, AIMG = 32 * AP, IMG = QIMG + NH * GIMG + AIMG;
  constexpr int NCH = 32 * (G::CK + NH * G::CV + 2 * NH), CPT = (NCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf lds[2 * IMG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const cenet_bid bid = cenet_xcd_block();
  const int b = bid.y / a.H, h = bid.y - b * a.H;
  const int k0 = (bid.x * 4 + wave) * 32;
  const int N = a.N, hd = a.hd, dv = 2 * hd, E = 2 * a.H * hd;
  const float c = a.scale * DA_LOG2E;
  const bf* qb = a.q + (long)b * a.bmul * N * E;
  const bf* kb = a.k + (long)b * a.bmul * N * E;
  const bf* vb = a.v + (long)b * a.bmul * N * (a.H * dv) + h * dv;
  const long hb0 = ((long)b * NH * a.H + NH * h) * N;  // (b, head 2h) row base of dU / aug; head 2h+1 is N rows further
  const int ki = k0 + r < N ? k0 + r : N - 1;

  DaChunk ch[CPT];
  long gstep[CPT];  // elements per 32-row tile step of the chunk's tensor
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    int id = tid + 256 * j;
    ch[j].fill = 0u;
    gstep[j] = 0;
    if (id >= NCH) {
      ch[j].g = nullptr;
      ch[j].lrow = 0;
      ch[j].loff = -1;
    } else if (id < 32 * G::CK) {  // Q rows, both heads
      const int row = id / G::CK, cc = id - row * G::CK, s = cc / (HDP / 8), c8 = cc - s * (HDP / 8);
      ch[j].lrow = row;
      ch[j].loff = row * G::KP + s * HDP + 8 * c8;
      ch[j].g = 8 * c8 < hd ? qb + (long)row * E + (2 * h + s) * hd + 8 * c8 : nullptr;
      gstep[j] = 32L * E;
    } else if ((id -= 32 * G::CK) < NH * 32 * G::CV) {  // dU rows of head s
      const int s = id / (32 * G::CV), id2 = id - s * 32 * G::CV, row = id2 / G::CV, c8 = id2 - row * G::CV;
      ch[j].lrow = row;
      ch[j].loff = QIMG + s * GIMG + row * G::VPR + 8 * c8;
      ch[j].g = 8 * c8 < dv ? a.dU + (hb0 + (long)s * N + row) * dv + 8 * c8 : nullptr;
      gstep[j] = 32L * dv;
    } else {  // statistics rows: chunk (row, s, kind): kind 0 = lse / scale, 1 = delta
      id -= NH * 32 * G::CV;
      const int row = id / (2 * NH), s = (id - row * 2 * NH) >> 1, kind = id & 1;
      ch[j].lrow = row;
      ch[j].loff = QIMG + NH * GIMG + row * AP + s * 16 + 8 * kind;
      ch[j].g = a.aug + (hb0 + (long)s * N + row) * 16 + 8 * kind;
      ch[j].fill = kind == 0 ? 0x7149u : 0u;  // rows beyond N: lse / scale = 1e30 -> probability 0
      gstep[j] = 32L * 16;
    }
  }
  unsigned pre[CPT][4];
  auto fetch = [&](int q0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < CPT; ++j) da_fetch(ch[j], (q0 >> 5) * gstep[j], N - q0, pre[j]);
  };
  auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < CPT; ++j)
      if (ch[j].loff >= 0) memcpy(lds + buf * IMG + ch[j].loff, pre[j], 16);
  };

  bf16x8 kfB[2][NKS], vfB[NKD];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      kfB[s][ks] = da_ld8(kb + (long)ki * E + (2 * h + s) * hd + 16 * ks + 8 * hh, 16 * ks + 8 * hh < hd);
#pragma unroll
  for (int kd = 0; kd < NKD; ++kd) vfB[kd] = da_ld8(vb + (long)ki * (a.H * dv) + 16 * kd + 8 * hh, 16 * kd + 8 * hh < dv);
  // constant B fragment of the statistics k-step: -1 at k = 0, 1, 2 (lane half 0), zero elsewhere
  bf16x8 negB = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hh == 0) negB[0] = negB[1] = negB[2] = (short)0xBF80;
  f32x16 dK[2][NFT], dV[NDT];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) dK[s][ft] = da_zero();
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) dV[dt] = da_zero();

  fetch(0);
  stage(0);
  __syncthreads();
  int cur = 0;
  for (int q0 = 0; q0 < N; q0 += 32) {
    const bool more = q0 + 32 < N;
    if (more) fetch(q0 + 32);
    const bf* Qimg = lds + cur * IMG;
    const bf* Aimg = Qimg + QIMG + NH * GIMG;
    if (k0 < N) {
#pragma unroll
      for (int s = 0; s < NH; ++s) {
        const bf* Gimg = Qimg + QIMG + s * GIMG;
        f32x16 S = da_zero(), dP = da_zero();
#pragma unroll
        for (int sp = (ONE ? 0 : s); sp <= (ONE ? 1 : s); ++sp)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) S = DA_MFMA(da_rm(Qimg, G::KP, r, sp * HDP + 16 * ks, hh), kfB[sp][ks], S);  // S[q][key]
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd) dP = DA_MFMA(da_rm(Gimg, G::VPR, r, 16 * kd, hh), vfB[kd], dP);  // dP[q][key] = dU V^T
        // row (query) statistics through the contraction: S -= lse / scale, dP -= delta
        bf16x8 al = {0, 0, 0, 0, 0, 0, 0, 0}, ad = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hh == 0) {
          memcpy(&al, Aimg + r * AP + s * 16, 16);
          memcpy(&ad, Aimg + r * AP + s * 16 + 8, 16);
        }
        S = DA_MFMA(al, negB, S);
        dP = DA_MFMA(ad, negB, dP);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = da_exp2(S[i] * c);
          S[i] = p;
          dP[i] = p * dP[i];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pB = da_pack8(S, s2), dsB = da_pack8(dP, s2);
          // dV^T[feature][key] += dU^T[feature][query] P[query][key]
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) dV[dt] = DA_MFMA(da_tr(Gimg, G::VPR, s2, 32 * dt, lane), pB, dV[dt]);
          // dK^T[d][key] += Q^T[d][query] dS[query][key]   (accumulator rows >= hd are never stored)
#pragma unroll
          for (int sp = (ONE ? 0 : s); sp <= (ONE ? 1 : s); ++sp)
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft)
              dK[sp][ft] = DA_MFMA(da_tr(Qimg, G::KP, s2, sp * HDP + 32 * ft, lane), dsB, dK[sp][ft]);
        }
      }
    }
    if (more) stage(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (k0 + r < N) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf* dp = a.dk + ((long)b * a.bmul * N + ki) * E + (2 * h + s) * hd;
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int g = 0; g < (HDP < 32 ? HDP / 8 : 4); ++g) {
          const int d0 = 32 * ft + 8 * g + 4 * hh;
          if (d0 < hd) {
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = dK[s][ft][4 * g + i] * a.scale;
            st4v(dp + d0, o);
          }
        }
    }
    bf* dvp = a.dv + ((long)b * a.bmul * N + ki) * (a.H * dv) + h * dv;
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

// head dims served: 8 and 16 (one k-step), 32 (two), 64 (four k-steps, two 32-feature accumulator tiles per head: the
// 64x64-token level of the 512x512 three-scale preset, skin.sh:93-94)
extern "C" int cenet_diffattn_heads_supported(int hd, int N) { return (hd == 8 || hd == 16 || hd == 32 || hd == 64) && N >= 1; }
extern "C" long cenet_diffattn_heads_ws_bytes(int B, int H, int N) { return (long)B * 2 * H * N * 16 * 2; }

static int da_fill(DiffArgs& a, const cenet_diffattn_t* p) {
  if (!p || !p->q || !p->k || !p->v || !p->U || !p->lse) return CENET_EINVAL;
  if (p->B <= 0 || p->H <= 0 || p->N <= 0 || !cenet_diffattn_heads_supported(p->hd, p->N)) return CENET_EUNSUPPORTED;
  a.q = (const bf*)p->q; a.k = (const bf*)p->k; a.v = (const bf*)p->v;
  a.U = (bf*)p->U; a.lse = p->lse; a.dU = (const bf*)p->dU;
  a.dq = (bf*)p->dq; a.dk = (bf*)p->dk; a.dv = (bf*)p->dv; a.aug = (bf*)p->ws;
  a.B = p->B; a.H = p->H; a.N = p->N; a.hd = p->hd; a.scale = p->scale;
  a.bmul = p->batch_mul > 0 ? p->batch_mul : 1;
  const uintptr_t m = (uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.U | (uintptr_t)a.dU | (uintptr_t)a.dq |
                      (uintptr_t)a.dk | (uintptr_t)a.dv | (uintptr_t)a.aug;
  if (m & 15) return CENET_EINVAL;  // rows are moved in 16-byte chunks
  return CENET_OK;
}

extern "C" int cenet_diffattn_heads_fwd_bf16(const cenet_diffattn_t* p, hipStream_t stream) {
  DiffArgs a;
  const int rc = da_fill(a, p);
  if (rc != CENET_OK) return rc;
  const int bh = a.B * a.H;
  if (a.hd <= 16) {
    // one 32-query tile per wave at three workgroups per CU (two tiles per wave need 256 registers and still spill)
    static const int v = getenv("CENET_DATTN_QT2") ? 2 : 1;
    if (v == 2 && a.N >= 1024) CENET_LAUNCH((dattn_fwd_kernel<16, 2, 2, false>), dim3(cdiv(a.N, 256), bh), dim3(256), stream, a);
    else CENET_LAUNCH((dattn_fwd_kernel<16, 1, 3, false>), dim3(cdiv(a.N, 128), bh), dim3(256), stream, a);
  } else if (a.hd <= 32) {
    CENET_LAUNCH((dattn_fwd_kernel<32, 1, 2, false>), dim3(cdiv(a.N, 128), bh), dim3(256), stream, a);
  } else {
    CENET_LAUNCH((dattn_fwd_kernel<64, 1, 1, false>), dim3(cdiv(a.N, 128), bh), dim3(256), stream, a);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_diffattn_heads_bwd_bf16(const cenet_diffattn_t* p, hipStream_t stream) {
  DiffArgs a;
  const int rc = da_fill(a, p);
  if (rc != CENET_OK) return rc;
  if (!a.dU || !a.dq || !a.dk || !a.dv || !a.aug) return CENET_EINVAL;
  const dim3 grid(cdiv(a.N, 128), a.B * a.H);
  if (a.hd <= 16) {
    CENET_LAUNCH((dattn_bwd_dq_kernel<16, false>), grid, dim3(256), stream, a);
    CENET_LAUNCH((dattn_bwd_dkv_kernel<16, false>), grid, dim3(256), stream, a);
  } else if (a.hd <= 32) {
    CENET_LAUNCH((dattn_bwd_dq_kernel<32, false>), grid, dim3(256), stream, a);
    CENET_LAUNCH((dattn_bwd_dkv_kernel<32, false>), grid, dim3(256), stream, a);
  } else {
    CENET_LAUNCH((dattn_bwd_dq_kernel<64, false>), grid, dim3(256), stream, a);
    CENET_LAUNCH((dattn_bwd_dkv_kernel<64, false>), grid, dim3(256), stream, a);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

// ---- plain self-attention with head dimension 64 or 128 through the pair tiles (ONE softmax over both halves) ----------------
// q, k, v [B, N, H*hd] (token-major), U / dU [B, H, N, hd], lse [B, H, N]; ws: cenet_attn64_ws_bytes.  The caller passes
// hd = 64 or 128; inside, a head is a "pair" of two hd/2-wide halves (DiffArgs.hd = 32 / 64).  Used by the Non-local blocks of
// the 56x56 (C = 64, N = 3136) and 28x28 (C = 128, N = 784) decoder levels (nlb.py:117-138).
extern "C" long cenet_attn64_ws_bytes(int B, int H, int N) { return (long)B * H * N * 16 * 2; }

static int da_fill64(DiffArgs& a, const cenet_diffattn_t* p) {
  if (!p || !p->q || !p->k || !p->v || !p->U || !p->lse) return CENET_EINVAL;
  if (p->B <= 0 || p->H <= 0 || p->N <= 0 || (p->hd != 64 && p->hd != 128)) return CENET_EUNSUPPORTED;
  a.q = (const bf*)p->q; a.k = (const bf*)p->k; a.v = (const bf*)p->v;
  a.U = (bf*)p->U; a.lse = p->lse; a.dU = (const bf*)p->dU;
  a.dq = (bf*)p->dq; a.dk = (bf*)p->dk; a.dv = (bf*)p->dv; a.aug = (bf*)p->ws;
  a.B = p->B; a.H = p->H; a.N = p->N; a.hd = p->hd / 2; a.scale = p->scale;  // a head = a "pair" of two hd/2-wide halves
  a.bmul = p->batch_mul > 0 ? p->batch_mul : 1;
  const uintptr_t m = (uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.U | (uintptr_t)a.dU | (uintptr_t)a.dq |
                      (uintptr_t)a.dk | (uintptr_t)a.dv | (uintptr_t)a.aug;
  if (m & 15) return CENET_EINVAL;
  return CENET_OK;
}

extern "C" int cenet_attn64_fwd_bf16(const cenet_diffattn_t* p, hipStream_t stream) {
  DiffArgs a;
  const int rc = da_fill64(a, p);
  if (rc != CENET_OK) return rc;
  if (a.hd == 32) CENET_LAUNCH((dattn_fwd_kernel<32, 1, 2, true>), dim3(cdiv(a.N, 128), a.B * a.H), dim3(256), stream, a);
  else CENET_LAUNCH((dattn_fwd_kernel<64, 1, 1, true>), dim3(cdiv(a.N, 128), a.B * a.H), dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_attn64_bwd_bf16(const cenet_diffattn_t* p, hipStream_t stream) {
  DiffArgs a;
  const int rc = da_fill64(a, p);
  if (rc != CENET_OK) return rc;
  if (!a.dU || !a.dq || !a.dk || !a.dv || !a.aug) return CENET_EINVAL;
  const dim3 grid(cdiv(a.N, 128), a.B * a.H);
  if (a.hd == 32) {
    CENET_LAUNCH((dattn_bwd_dq_kernel<32, true>), grid, dim3(256), stream, a);
    CENET_LAUNCH((dattn_bwd_dkv_kernel<32, true>), grid, dim3(256), stream, a);
  } else {
    CENET_LAUNCH((dattn_bwd_dq_kernel<64, true>), grid, dim3(256), stream, a);
    CENET_LAUNCH((dattn_bwd_dkv_kernel<64, true>), grid, dim3(256), stream, a);
  }
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Spatial-reduction attention (pvtv2.py:88-109), backward, bf16: head dimension 64, at most 64 keys (the 7x7 = 49 keys the
// reduction convs leave at every stage) under 49 ... 3136 queries.  ONE kernel instead of the dQ + dK/dV pair of
// attn_bf16.hip: the whole key / value set of a (batch, head) sits in LDS, a wave owns 32-query tiles and computes, per tile,
//   phase A (query = MFMA column, as dattn_bwd_dq):  S^T, dP^T -> dS^T -> dQ^T ; delta and lse as per-lane scalars;
//   phase B (key = MFMA column, as dattn_bwd_dkv):    S, dP recomputed from the tile's Q / dO images (two MFMA groups:
//            cheaper than transposing dS through LDS) -> dV^T += dO^T P, dK^T += Q^T dS, accumulated in registers over the wave's
//            tiles; the four waves meet in LDS and the workgroup adds one 64 x 128 fp32 block into dkv.
// q, o, dout, dq [B, Nq, C], kv [B, Nk, 2C] (k = columns 0..C, v = C..2C), head h = columns 64h..64h+63, lse [B, H, Nq]
// (natural log of the scaled scores, as the forward kernel leaves it), dkv [B, Nk, 2C] fp32, zero-filled by the caller.
// ---------------------------------------------------------------------------------------------------------------------------
struct SraArgs {
  const bf *q, *kv, *o, *dout;
  const float* lse;
  bf* dq;
  float* dkv;
  bf* dkv_bf;  // non-null: ONE workgroup per (batch, head) stores dK/dV as bf16 directly (no zero fill, atomics or cast pass)
  int B, H, Nq, Nk, C, tiles;  // tiles: 32-query tiles per wave
  float scale;
};

// NKB = 64-key blocks resident in LDS; DQ / DKV = which gradients this instance produces:
//   <1, true, true>   the whole backward of a (batch, head) with <= 64 keys (the 224x224 presets: 49 keys);
//   <4, true, false>  dQ over up to 256 resident keys (512x512 inputs: 256 keys under 4 096 .. 16 384 queries);
//   <1, false, true>  dK / dV of the 64-key block blockIdx.z (the same 256-key problems, one block of keys per workgroup:
//                     the softmax statistics come from the saved lse, so a block of keys needs nothing from the others).
template <int NKB, bool DQ, bool DKV>
__global__ __launch_bounds__(256, DKV ? 1 : 2) void sra_bwd_kernel(SraArgs a) {
  constexpr int KP = 72, AP = 40;                     // image pitches (elements)
  constexpr int KVIMG = NKB * 64 * KP, QIMG = 32 * KP + 64, AIMG = 32 * AP;
  constexpr int WIMG = 2 * QIMG + AIMG;               // per-wave: Q image, dO image, statistics image
  constexpr int LDS_EL = (DQ ? 2 * KVIMG : 0) + (DKV ? 4 * WIMG : 0);
  static_assert(!DKV || LDS_EL * 2 >= 64 * 128 * 4, "the dK/dV reduction block reuses the images");
  static_assert(NKB == 1 || !DKV, "dK / dV accumulators exist for one 64-key block");
  __shared__ __attribute__((aligned(16))) bf lds[LDS_EL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const cenet_bid bid = cenet_xcd_block();
  const int b = bid.y / a.H, h = bid.y - b * a.H;
  const int key0 = DQ ? 0 : 64 * bid.z;               // first key of this workgroup's block (dK / dV instances)
  const int C = a.C, Nq = a.Nq;
  const int Nk = DQ ? a.Nk : (a.Nk - key0 < 64 ? a.Nk - key0 : 64);  // keys this workgroup holds
  const float c = a.scale * DA_LOG2E;
  const bf* qb = a.q + (long)b * Nq * C + h * 64;
  const bf* ob = a.o + (long)b * Nq * C + h * 64;
  const bf* gb = a.dout + (long)b * Nq * C + h * 64;
  const bf* kb = a.kv + ((long)b * a.Nk + key0) * 2 * C + h * 64;
  const bf* vb = kb + C;
  bf* Kimg = lds;
  bf* Vimg = lds + KVIMG;
  bf* Qimg = lds + (DQ ? 2 * KVIMG : 0) + wave * WIMG;
  bf* Gimg = Qimg + QIMG;
  bf* Aimg = Gimg + QIMG;
  if (DQ) {
    // K and V of this (batch, head): NKB * 64 rows x 8 chunks each, rows >= Nk zero
    for (int id = tid; id < 2 * NKB * 64 * 8; id += 256) {
      const int which = id / (NKB * 512), rem = id - which * (NKB * 512), row = rem >> 3, c8 = rem & 7;
      unsigned v4[4] = {0u, 0u, 0u, 0u};
      if (row < Nk) memcpy(v4, (which ? vb : kb) + (long)row * 2 * C + 8 * c8, 16);
      memcpy((which ? Vimg : Kimg) + row * KP + 8 * c8, v4, 16);
    }
  }
  // B fragments of K^T / V^T for phase B: key = 32 kt + r
  bf16x8 kfB[2][4], vfB[2][4];
  if (DKV) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int key = 32 * kt + r;
        kfB[kt][ks] = da_ld8(kb + (long)(key < Nk ? key : 0) * 2 * C + 16 * ks + 8 * hh, key < Nk);
        vfB[kt][ks] = da_ld8(vb + (long)(key < Nk ? key : 0) * 2 * C + 16 * ks + 8 * hh, key < Nk);
      }
  }
  bf16x8 negB = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hh == 0) negB[0] = negB[1] = negB[2] = (short)0xBF80;
  f32x16 dK[2][2], dV[2][2];  // [key tile][feature tile]
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) dK[kt][dt] = dV[kt][dt] = da_zero();
  __syncthreads();

  const int q00 = (bid.x * 4 + wave) * 32 * a.tiles;
  for (int t = 0; t < a.tiles; ++t) {
    const int q0 = q00 + 32 * t;
    if (q0 >= Nq) break;  // (wave-uniform)
    const int qi = q0 + r < Nq ? q0 + r : Nq - 1;
    const bool qok = q0 + r < Nq;
    // ---- this tile's rows: B fragments for phase A, images for phase B ----
    bf16x8 qf[4], gf[4];
    float acc = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = da_ld8(qb + (long)qi * C + 16 * ks + 8 * hh, true);
      gf[ks] = da_ld8(gb + (long)qi * C + 16 * ks + 8 * hh, true);
      const bf16x8 uf = da_ld8(ob + (long)qi * C + 16 * ks + 8 * hh, true);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += cenet_bf2f((unsigned short)gf[ks][j]) * cenet_bf2f((unsigned short)uf[j]);
      if (DKV) {
        bf16x8 qz = qf[ks], gz = gf[ks];
        if (!qok) qz = gz = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        memcpy(Qimg + r * KP + 16 * ks + 8 * hh, &qz, 16);
        memcpy(Gimg + r * KP + 16 * ks + 8 * hh, &gz, 16);
      }
    }
    acc += __shfl_xor(acc, 32);
    const float delta = acc;
    const float lse = a.lse[((long)b * a.H + h) * Nq + qi];
    const float lse2 = lse * DA_LOG2E;
    if (DKV) {  // statistics rows: lane half 0 writes lse / scale, half 1 writes delta (rows beyond Nq: lse / scale = 1e30 -> P = 0)
      unsigned short row8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      da_split3(hh == 0 ? (qok ? lse / a.scale : 1.0e30f) : (qok ? delta : 0.f), row8);
      memcpy(Aimg + r * AP + 8 * hh, row8, 16);
    }
    // ---- phase A: dQ ----
    f32x16 dq[2] = {da_zero(), da_zero()};
#pragma unroll
    for (int kt = 0; kt < (DQ ? 2 * NKB : 0); ++kt) {
      if (32 * kt >= Nk) continue;  // (uniform)
      const bf* Kt = Kimg + 32 * kt * KP;
      const bf* Vt = Vimg + 32 * kt * KP;
      f32x16 S = da_zero(), dP = da_zero();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        S = DA_MFMA(da_rm(Kt, KP, r, 16 * ks, hh), qf[ks], S);
        dP = DA_MFMA(da_rm(Vt, KP, r, 16 * ks, hh), gf[ks], dP);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float pv = da_exp2(S[i] * c - lse2);
        if (32 * kt + da_row(i, hh) >= Nk) pv = 0.f;
        S[i] = pv * (dP[i] - delta);
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) dq[dt] = DA_MFMA(da_tr(Kt, KP, s2, 32 * dt, lane), da_pack8(S, s2), dq[dt]);
    }
    if (DQ && qok) {
      bf* dp = a.dq + ((long)b * Nq + qi) * C + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float o4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) o4[i] = dq[dt][4 * g + i] * a.scale;
          st4v(dp + 32 * dt + 8 * g + 4 * hh, o4);
        }
    }
    if (!DKV) continue;
    // ---- phase B: dK, dV (images written above by this wave only: no workgroup barrier needed) ----
    // A wave's LDS instructions execute in program order, so its own stores above are visible to its loads below; the
    // compiler is told not to move LDS accesses across this point (the host checker runs lanes as fibers: its sync point).
    da_wave_sync();
    bf16x8 al = {0, 0, 0, 0, 0, 0, 0, 0}, ad = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hh == 0) {
      memcpy(&al, Aimg + r * AP, 16);
      memcpy(&ad, Aimg + r * AP + 8, 16);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (32 * kt >= Nk) continue;
      f32x16 S = da_zero(), dP = da_zero();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        S = DA_MFMA(da_rm(Qimg, KP, r, 16 * ks, hh), kfB[kt][ks], S);   // S[query][key]
        dP = DA_MFMA(da_rm(Gimg, KP, r, 16 * ks, hh), vfB[kt][ks], dP);  // dP[query][key]
      }
      S = DA_MFMA(al, negB, S);
      dP = DA_MFMA(ad, negB, dP);
      const bool kok = 32 * kt + r < Nk;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = kok ? da_exp2(S[i] * c) : 0.f;
        S[i] = pv;
        dP[i] = pv * dP[i];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pB = da_pack8(S, s2), dsB = da_pack8(dP, s2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dV[kt][dt] = DA_MFMA(da_tr(Gimg, KP, s2, 32 * dt, lane), pB, dV[kt][dt]);
          dK[kt][dt] = DA_MFMA(da_tr(Qimg, KP, s2, 32 * dt, lane), dsB, dK[kt][dt]);
        }
      }
    }
  }
  if (!DKV) return;
  // ---- the four waves meet in LDS: red[key][0..63] = dK (unscaled), red[key][64..127] = dV ----
  // red[feature 0..127][key 0..63] (key fastest: a wave's 32 lanes hit 32 consecutive banks); the waves take turns, so
  // plain read-modify-writes suffice
  float* red = (float*)lds;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = 32 * kt + r, f = 32 * dt + da_row(i, hh);
            if (w == 0) {
              red[f * 64 + key] = dK[kt][dt][i];
              red[(64 + f) * 64 + key] = dV[kt][dt][i];
            } else {
              red[f * 64 + key] += dK[kt][dt][i];
              red[(64 + f) * 64 + key] += dV[kt][dt][i];
            }
          }
    }
    __syncthreads();
  }
  if (a.dkv_bf) {  // this workgroup saw every query of its (batch, head)
    bf* dkvb = a.dkv_bf + ((long)b * a.Nk + key0) * 2 * C + h * 64;
    for (int i = tid; i < Nk * 128; i += 256) {
      const int key = i >> 7, f = i & 127;
      const float v = red[f * 64 + key];
      if (f < 64) stf(dkvb + (long)key * 2 * C + f, v * a.scale);
      else stf(dkvb + (long)key * 2 * C + C + (f - 64), v);
    }
    return;
  }
  float* dkv = a.dkv + ((long)b * a.Nk + key0) * 2 * C + h * 64;
  for (int i = tid; i < Nk * 128; i += 256) {
    const int key = i >> 7, f = i & 127;  // 128 consecutive lanes = one key's 64 + 64 features: contiguous runs of 256 B
    const float v = red[f * 64 + key];
    if (f < 64) atomicAdd(&dkv[(long)key * 2 * C + f], v * a.scale);
    else atomicAdd(&dkv[(long)key * 2 * C + C + (f - 64)], v);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Spatial-reduction attention FORWARD with the key / value set resident (same problem class as sra_bwd_kernel: head dimension
// 64, at most 64 keys): no key loop and no online softmax — a wave takes 32-query tiles, S^T = K Q^T for both key tiles in one
// go, the softmax over the <= 64 keys of a query is 32 registers + one exchange with lane ^ 32, and O^T = V^T P^T.  The streamed
// kernel (attn_bf16.hip, flashc_fwd) spent ~20 us per launch on 13-26 MB; this one is bound by reading q and writing o.
// q, o [B, Nq, 64 H]; kv [B, Nk, 128 H]; lse [B, H, Nq] = natural log of the sum of exp(scale * scores) (what the backward
// kernels expect).
// ---------------------------------------------------------------------------------------------------------------------------
struct SraFwdArgs {
  const bf *q, *kv;
  bf* o;
  float* lse;
  int B, H, Nq, Nk, C, tiles;
  float scale;
};

__global__ __launch_bounds__(256, 2) void sra_fwd_kernel(SraFwdArgs a) {
  constexpr int KP = 72, KVIMG = 64 * KP;
  __shared__ __attribute__((aligned(16))) bf lds[2 * KVIMG + 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const cenet_bid bid = cenet_xcd_block();
  const int b = bid.y / a.H, h = bid.y - b * a.H;
  const int C = a.C, Nq = a.Nq, Nk = a.Nk;
  const float c = a.scale * DA_LOG2E;
  const bf* qb = a.q + (long)b * Nq * C + h * 64;
  bf* ob = a.o + (long)b * Nq * C + h * 64;
  const bf* kb = a.kv + (long)b * Nk * 2 * C + h * 64;
  const bf* vb = kb + C;
  bf* Kimg = lds;
  bf* Vimg = lds + KVIMG;
  for (int id = tid; id < 2 * 64 * 8; id += 256) {
    const int which = id >> 9, rem = id & 511, row = rem >> 3, c8 = rem & 7;
    unsigned v4[4] = {0u, 0u, 0u, 0u};
    if (row < Nk) memcpy(v4, (which ? vb : kb) + (long)row * 2 * C + 8 * c8, 16);
    memcpy((which ? Vimg : Kimg) + row * KP + 8 * c8, v4, 16);
  }
  __syncthreads();
  const int q00 = (bid.x * 4 + wave) * 32 * a.tiles;
  for (int t = 0; t < a.tiles; ++t) {
    const int q0 = q00 + 32 * t;
    if (q0 >= Nq) break;  // (wave-uniform)
    const int qi = q0 + r < Nq ? q0 + r : Nq - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = da_ld8(qb + (long)qi * C + 16 * ks + 8 * hh, true);
    f32x16 S[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      S[kt] = da_zero();
      if (32 * kt < Nk) {  // (uniform)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) S[kt] = DA_MFMA(da_rm(Kimg + 32 * kt * KP, KP, r, 16 * ks, hh), qf[ks], S[kt]);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (32 * kt + da_row(i, hh) >= Nk) S[kt][i] = DA_NEG;
    }
    float mx = DA_NEG;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, S[kt][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mc = mx * c;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = da_exp2(S[kt][i] * c - mc);  // masked keys: exp2(-huge) = 0
        S[kt][i] = pv;
        l += pv;
      }
    l += __shfl_xor(l, 32);
    f32x16 O[2] = {da_zero(), da_zero()};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (32 * kt >= Nk) continue;
      const bf* Vt = Vimg + 32 * kt * KP;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pB = da_pack8(S[kt], s2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) O[dt] = DA_MFMA(da_tr(Vt, KP, s2, 32 * dt, lane), pB, O[dt]);
      }
    }
    if (q0 + r < Nq) {
      const float inv = 1.f / l;
      bf* op = ob + (long)qi * C;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float o4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) o4[i] = O[dt][4 * g + i] * inv;
          st4v(op + 32 * dt + 8 * g + 4 * hh, o4);
        }
      if (hh == 0) a.lse[((long)b * a.H + h) * Nq + qi] = mx * a.scale + logf(l);
    }
  }
}

extern "C" int cenet_sra_attn_fwd_bf16(const bf* q, const bf* kv, bf* o, float* lse, int B, int H, int Nq, int Nk, float scale,
                                       hipStream_t stream) {
  if (!q || !kv || !o || !lse || B <= 0 || H <= 0 || Nq <= 0) return CENET_EINVAL;
  if (!(Nk >= 1 && Nk <= 64)) return CENET_EUNSUPPORTED;
  if ((((uintptr_t)q | (uintptr_t)kv | (uintptr_t)o) & 15) != 0) return CENET_EINVAL;
  SraFwdArgs a;
  a.q = q; a.kv = kv; a.o = o; a.lse = lse; a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.C = 64 * H; a.scale = scale;
  // two workgroups per CU: ~512 workgroups fill the chip once
  static const char* e = getenv("CENET_SRA_FWD_TILES");  // measurement aid
  int tiles = 1;
  while (tiles < 8 && (long)cdiv(Nq, 128 * tiles) * B * H > 1024) ++tiles;
  if (e) tiles = atoi(e);
  a.tiles = tiles < 1 ? 1 : tiles;
  CENET_LAUNCH(sra_fwd_kernel, dim3(cdiv(Nq, 128 * a.tiles), B * H), dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_sra_attn_bwd_supported(int hd, int Nk) { return hd == 64 && Nk >= 1 && Nk <= 64; }
// 64 < Nk <= 256 (the 512x512 presets: 256 keys): cenet_sra_attn_bwd_bf16 runs as two launches, dQ over all keys resident in
// LDS and dK / dV per 64-key block; the forward of these problems is the tiled kernel's (same lse convention)
extern "C" int cenet_sra_attn_bwd_blocks_supported(int hd, int Nk) { return hd == 64 && Nk > 64 && Nk <= 256; }

// 32-query tiles per wave: the kernel runs one workgroup per CU (484 registers), so more than 256 workgroups is a second
// round; measured best (B = 32): 4 tiles at 3136 queries (42 us), 2 at 784 / 196 (31 us), 1 at 49 (25 us)
static int sra_tiles(int B, int H, int Nq) {
  static const char* e = getenv("CENET_SRA_TILES");
  int tiles = 1;
  while (tiles < 8 && (long)cdiv(Nq, 128 * tiles) * B * H > 256) ++tiles;
  if (e) tiles = atoi(e);
  return tiles < 1 ? 1 : tiles;
}
// 1 when the launch for this problem puts every query of a (batch, head) in ONE workgroup: dK/dV then need no cross-workgroup
// sum and cenet_sra_attn_bwd_direct_bf16 stores them as bf16 (the 14x14 and 7x7 stages of the ACDC preset)
extern "C" int cenet_sra_attn_bwd_direct_supported(int B, int H, int Nq, int Nk) {
  return cenet_sra_attn_bwd_supported(64, Nk) && B > 0 && H > 0 && Nq > 0 && cdiv(Nq, 128 * sra_tiles(B, H, Nq)) == 1;
}

static int sra_bwd_launch(const bf* q, const bf* kv, const bf* o, const bf* dout, const float* lse, bf* dq, float* dkv, bf* dkv_bf,
                          int B, int H, int Nq, int Nk, float scale, hipStream_t stream) {
  if (!q || !kv || !o || !dout || !lse || !dq || (!dkv && !dkv_bf) || B <= 0 || H <= 0 || Nq <= 0) return CENET_EINVAL;
  const bool blocks = cenet_sra_attn_bwd_blocks_supported(64, Nk);
  if (!cenet_sra_attn_bwd_supported(64, Nk) && !blocks) return CENET_EUNSUPPORTED;
  if ((((uintptr_t)q | (uintptr_t)kv | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq) & 15) != 0) return CENET_EINVAL;
  SraArgs a;
  a.q = q; a.kv = kv; a.o = o; a.dout = dout; a.lse = lse; a.dq = dq; a.dkv = dkv; a.dkv_bf = dkv_bf;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.C = 64 * H; a.scale = scale;
  a.tiles = sra_tiles(B, H, Nq);
  if (blocks) {
    if (dkv_bf) return CENET_EUNSUPPORTED;  // dK / dV of a key block are summed over the query slices: fp32 accumulator
    // dQ: two workgroups per CU (no dK / dV accumulators), ~512 workgroups fill the chip once
    int tq = 1;
    while (tq < 8 && (long)cdiv(Nq, 128 * tq) * B * H > 512) ++tq;
    a.tiles = tq;
    CENET_LAUNCH((sra_bwd_kernel<4, true, false>), dim3(cdiv(Nq, 128 * tq), B * H), dim3(256), stream, a);
    const int nkb = cdiv(Nk, 64);
    int tk = 1;
    while (tk < 16 && (long)cdiv(Nq, 128 * tk) * B * H * nkb > 256) ++tk;
    a.tiles = tk;
    CENET_LAUNCH((sra_bwd_kernel<1, false, true>), dim3(cdiv(Nq, 128 * tk), B * H, nkb), dim3(256), stream, a);
    CENET_CHECK_LAUNCH();
    return CENET_OK;
  }
  if (dkv_bf && cdiv(Nq, 128 * a.tiles) != 1) return CENET_EUNSUPPORTED;
  CENET_LAUNCH((sra_bwd_kernel<1, true, true>), dim3(cdiv(Nq, 128 * a.tiles), B * H), dim3(256), stream, a);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_sra_attn_bwd_bf16(const bf* q, const bf* kv, const bf* o, const bf* dout, const float* lse, bf* dq, float* dkv,
                                       int B, int H, int Nq, int Nk, float scale, hipStream_t stream) {
  if (!dkv) return CENET_EINVAL;
  return sra_bwd_launch(q, kv, o, dout, lse, dq, dkv, nullptr, B, H, Nq, Nk, scale, stream);
}

extern "C" int cenet_sra_attn_bwd_direct_bf16(const bf* q, const bf* kv, const bf* o, const bf* dout, const float* lse, bf* dq,
                                              bf* dkv, int B, int H, int Nq, int Nk, float scale, hipStream_t stream) {
  if (!dkv) return CENET_EINVAL;
  return sra_bwd_launch(q, kv, o, dout, lse, dq, nullptr, dkv, B, H, Nq, Nk, scale, stream);
}
