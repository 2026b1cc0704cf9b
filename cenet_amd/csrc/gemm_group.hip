// gemm_group.hip — GROUPED weight-gradient GEMM (bf16 operands, fp32 accumulate into the gradient arena).
//
// The weight gradients of a training step, dW_i += dY_i^T X_i (Linear: pvtv2.py:41,45,60-63,90,98,106; 1x1 conv:
// cfam.py:149,158,299,302, nlb.py:106-115,142, blocks.py:320, dseb.py:164), feed nothing but the optimizer.  Launched one by one
// (round 2: ~150 launches per step, the `roofline` kernel of bench.py) each of them is a long reduction over few output tiles
// that has to be split over hundreds of workgroups to fill the chip — every split then adds its whole fp32 tile atomically
// (4.6x write amplification, profiles/r02_traffic.json) and every launch pays its own ramp and tail.  Here the autograd
// Functions only RECORD (dY, X, dW) and a whole backward segment is reduced by ONE launch:
//   * work item = (problem, output tile 64x64, K slice); the problems of a launch share the chip, so a problem is split only
//     as far as the WHOLE group needs to fill it (slices of >= ~32 K-steps instead of 2 - 6);
//   * the K loop is gemm_ring.h's LDS-DMA ring (4 stages of 64, one barrier per step);
//   * unsplit tiles are added to the arena by their only owner with plain read-modify-writes (no atomics, deterministic);
//   * split tiles go to a workspace in accumulator layout (16-byte stores) and ONE fold launch per group sums the slices in
//     a fixed order and adds the result — a two-pass reduction, deterministic as well;
//   * the bias gradient (row sums of dY^T) rides in the same pass (first tile column), as in the ring kernel.
// The problem table travels in the kernel arguments (64 bytes per problem, <= 56 problems per launch): no device-side
// table, no host->device copy, nothing for a hipGraph capture to get wrong.
#include "gemm_ring.h"
#include <cstdio>
#include <cstdlib>

#define GRP_MAXP 56

struct GroupProb {  // 64 bytes
  const bf16_t* A;
  const bf16_t* B;
  float* C;
  float* asum;
  int lda, ldb, skbA, skbB;  // elements
  int K;                     // | unal << 28 | atomic << 29
  int ws_tile0;              // first workspace slot of this problem (splits > 1): slot = ws_tile0 + tile * splits + split
  unsigned short M, N, nkb, splits;
};
struct GroupArgs {
  int nprob, nitems;
  int item0[GRP_MAXP + 1];  // first work item of problem i (item0[nprob] = nitems)
  int fold0[GRP_MAXP + 1];  // first fold item (= output tile of a split problem) of problem i
  GroupProb p[GRP_MAXP];
};
static_assert(sizeof(GroupProb) == 64, "problem descriptor");
static_assert(sizeof(GroupArgs) <= 4096, "the table travels in the kernel arguments");

__device__ __forceinline__ int grp_find(const int* first, int n, int L) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= L) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// C[m0.., n0..] += the BM x BN tile held as acc[i][j] (fragment (i, j) of the wave's quadrant, 4 consecutive rows per lane):
// each wave transposes one 16-row strip at a time through LDS so that lanes run along a row of C (>= 128 contiguous bytes per
// wave instruction).  `atomic`: another problem of the launch adds into the same C (a parameter used twice); otherwise this
// workgroup is the tile's only writer in the launch and the add is a plain load / add / store.
template <int BM, int BN>
__device__ __forceinline__ void grp_add_tile(f32x4 (&acc)[BM / 32][BN / 32], float* cstrip, float* C, int ldc, int M, int N, int m0,
                                             int n0, int wave, int lane, bool atomic) {
  constexpr int MI = BM / 32, NJ = BN / 32, WN = BN / 2;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  float* strip = cstrip + wave * (16 * (WN + 1));
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) strip[(fq * 4 + r) * (WN + 1) + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    const int row0 = m0 + wm * (BM / 2) + i * 16, col0 = n0 + wn * WN;
    for (int idx = lane; idx < 16 * WN; idx += 64) {
      const int r = idx / WN, c = idx - r * WN;
      if (row0 + r < M && col0 + c < N) {
        float* p = &C[(long)(row0 + r) * ldc + col0 + c];
        const float v = strip[r * (WN + 1) + c];
        if (atomic) atomicAdd(p, v);
        else *p += v;
      }
    }
  }
}

template <bool AKF, bool BKF, int BM, int BN, int NS>
__global__ __launch_bounds__(256, (NS * (BM + BN) * 128 <= 80 * 1024 ? 2 : 1)) void gemm_group_kernel(GroupArgs ga,
                                                                                                   float* __restrict__ ws) {
  constexpr int MI = BM / 32, NJ = BN / 32;
  constexpr int ABYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int G = (BM + BN) / 32;
  constexpr int TILE_FLOATS = BM * BN + BM;  // a partial tile in accumulator layout + its row sums
  static_assert(NS >= 2 && NS <= 4 && G * (NS - 2) <= 63, "ring depth");
  static_assert(NS * STAGE >= 4 * 16 * (BN / 2 + 1) * 4, "the epilogue strip reuses the ring");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef CENET_HOSTSIM_BUILD
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int wm = wave >> 1, wn = wave & 1;
  int L = blockIdx.x;
  {
    // XCD-aware order, block-cyclic: workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Runs of 8
    // consecutive items (tiles of one K slice, which share operand rows) go to ONE XCD, and the runs are dealt over the XCDs
    // in turn — contiguous eighths of the item list (as in the single-problem GEMM) would hand each XCD a different mix of
    // problems, and the XCDs of a grouped launch would finish far apart.
    const int T = gridDim.x, full = T & ~63;
    if (L < full) {
      const int q = L >> 6, r = L & 63;
      L = (q << 6) + ((r & 7) << 3) + (r >> 3);
    }
  }
  const int pi = grp_find(ga.item0, ga.nprob, L);
  const GroupProb& P = ga.p[pi];
  const int M = P.M, N = P.N, K = P.K & 0x0FFFFFFF, nkb = P.nkb, splits = P.splits;
  const bool unal = (P.K >> 28) & 1, atomic = (P.K >> 29) & 1;
  const int tiles_n = (N + BN - 1) / BN, tiles = ((M + BM - 1) / BM) * tiles_n;
  const int local = L - ga.item0[pi];
  const int split = local / tiles, tile = local - split * tiles;
  const int by = tile / tiles_n, bx = tile - by * tiles_n;
  const int m0 = by * BM, n0 = bx * BN;

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ktiles = (K + 63) / 64;
  const int total = nkb * ktiles;
  const int chunk = (total + splits - 1) / splits;
  const int it0 = split * chunk;
  const int it1 = (it0 + chunk < total) ? it0 + chunk : total;
  const int T = it1 - it0;

  const long lda = P.lda, ldb = P.ldb;
  const RingSrc<MI> sa = ring_src<AKF, BM>(P.A, lda, m0, M, wave, lane);
  const RingSrc<NJ> sb = ring_src<BKF, BN>(P.B, ldb, n0, N, wave, lane);
  const long kstepA = AKF ? 1 : lda, kstepB = BKF ? 1 : ldb;
  const bf16_t* a_end = P.A + (long)(nkb - 1) * P.skbA + (AKF ? (long)(M - 1) * lda + K : (long)(K - 1) * lda + M);
  const bf16_t* b_end = P.B + (long)(nkb - 1) * P.skbB + (BKF ? (long)(N - 1) * ldb + K : (long)(K - 1) * ldb + N);

  auto issue = [&](int it, int buf) __attribute__((always_inline)) {
    const int kb = it / ktiles;
    const int k0 = (it - kb * ktiles) * 64;
    const int klim = K - k0;
    const bool tail = klim < 64;
    unsigned char* img = lds + buf * STAGE;
    ring_issue<AKF, BM>(sa, (long)kb * P.skbA + (long)k0 * kstepA, klim, tail, img, wave, lane, unal, a_end);
    ring_issue<BKF, BN>(sb, (long)kb * P.skbB + (long)k0 * kstepB, klim, tail, img + ABYTES, wave, lane, unal, b_end);
  };
  const bool kmask = (AKF || BKF) && (K & 7) != 0;
  const bool do_asum = P.asum != nullptr && bx == 0 && wn == 0;
  float rsum[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) rsum[i] = 0.f;

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < T) issue(it0 + s, s);

  int cur = 0;
  for (int t = 0; t < T; ++t) {
    const int young = T - 1 - t;
    if (NS >= 4 && young >= 2) ring_wait_vm<G * 2 <= 63 ? G * 2 : 0>();
    else if (NS >= 3 && young >= 1) ring_wait_vm<G>();
    else ring_wait_vm<0>();
    ring_barrier();
    if (t + NS - 1 < T) issue(it0 + t + NS - 1, cur == 0 ? NS - 1 : cur - 1);
    const unsigned char* Ai = lds + cur * STAGE;
    const unsigned char* Bi = Ai + ABYTES;
    int klim_t = 64;
    if (kmask) {
      const int it = it0 + t;
      klim_t = K - (it - (it / ktiles) * ktiles) * 64;
    }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      bf16x8 a[MI], b[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        a[i] = AKF ? ring_frag_kf(Ai, wm * (BM / 2) + i * 16, kc, lane) : ring_frag_rf<BM>(Ai, wm * (BM / 2) + i * 16, kc, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        b[j] = BKF ? ring_frag_kf(Bi, wn * (BN / 2) + j * 16, kc, lane) : ring_frag_rf<BN>(Bi, wn * (BN / 2) + j * 16, kc, lane);
      if (klim_t < 64) {
        if (AKF) {
#pragma unroll
          for (int i = 0; i < MI; ++i) a[i] = ring_mask_k(a[i], kc, lane, klim_t);
        }
        if (BKF) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) b[j] = ring_mask_k(b[j], kc, lane, klim_t);
        }
      }
      if (do_asum) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) rsum[i] += cenet_bf2f((unsigned short)a[i][j]);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  if (do_asum) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float v = rsum[i];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      rsum[i] = v;  // lanes 0..15: row m0 + wm * (BM / 2) + i * 16 + lane
    }
  }
  if (splits > 1) {
    // a K slice: the raw accumulators (and row sums) go to this item's workspace slot; gemm_group_fold_kernel adds them up
    float* slot = ws + (long)(P.ws_tile0 + tile * splits + split) * TILE_FLOATS;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) memcpy(slot + ((i * NJ + j) * 256 + tid) * 4, &acc[i][j], 16);
    if (bx == 0 && wn == 0 && lane < 16) {
#pragma unroll
      for (int i = 0; i < MI; ++i) slot[BM * BN + wm * (BM / 2) + i * 16 + lane] = rsum[i];
    }
    return;
  }
  if (do_asum && lane < 16) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = m0 + wm * (BM / 2) + i * 16 + lane;
      if (row < M) {
        if (atomic) atomicAdd(&P.asum[row], rsum[i]);
        else P.asum[row] += rsum[i];
      }
    }
  }
  __syncthreads();  // the ring is free: it becomes the transpose strip
  grp_add_tile<BM, BN>(acc, (float*)lds, P.C, N, M, N, m0, n0, wave, lane, atomic);
}

// second pass of the split problems: blockIdx.y = i picks the i-th 16-row strip of each wave's quadrant, so a tile is folded by
// BM / 32 workgroups (one per output tile left most of the chip idle behind a serial chain of partial-tile reads); each sums
// its rows of the tile's K slices in slice order and adds the result
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_group_fold_kernel(GroupArgs ga, const float* __restrict__ ws) {
  constexpr int NJ = BN / 32, WN = BN / 2, TILE_FLOATS = BM * BN + BM;
  __shared__ float strip_all[4 * 16 * (WN + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  const int L = blockIdx.x, i = blockIdx.y;
  const int pi = grp_find(ga.fold0, ga.nprob, L);
  const GroupProb& P = ga.p[pi];
  const int M = P.M, N = P.N, splits = P.splits;
  const bool atomic = (P.K >> 29) & 1;
  const int tiles_n = (N + BN - 1) / BN;
  const int tile = L - ga.fold0[pi];
  const int by = tile / tiles_n, bx = tile - by * tiles_n;
  const float* base = ws + (long)(P.ws_tile0 + tile * splits) * TILE_FLOATS;
  f32x4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float rs = 0.f;
  const bool do_rs = bx == 0 && P.asum && i == 0 && tid < BM;
  // four slices' loads in flight, added in slice order (a run-time trip count with a dependent add per iteration put one memory
  // round trip between consecutive slices)
  int s = 0;
  for (; s + 4 <= splits; s += 4) {
    f32x4 v[4][NJ];
    float r4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float* slot = base + (long)(s + k) * TILE_FLOATS;
#pragma unroll
      for (int j = 0; j < NJ; ++j) memcpy(&v[k][j], slot + ((i * NJ + j) * 256 + tid) * 4, 16);
      if (do_rs) r4[k] = slot[BM * BN + tid];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] += v[k][j];
      rs += r4[k];
    }
  }
  for (; s < splits; ++s) {
    const float* slot = base + (long)s * TILE_FLOATS;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x4 v;
      memcpy(&v, slot + ((i * NJ + j) * 256 + tid) * 4, 16);
      acc[j] += v;
    }
    if (do_rs) rs += slot[BM * BN + tid];
  }
  if (do_rs && by * BM + tid < M) {
    if (atomic) atomicAdd(&P.asum[by * BM + tid], rs);
    else P.asum[by * BM + tid] += rs;
  }
  // the strip transposition of grp_add_tile for this i alone
  float* strip = strip_all + wave * (16 * (WN + 1));
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) strip[(fq * 4 + r) * (WN + 1) + j * 16 + fr] = acc[j][r];
  __syncthreads();
  const int row0 = by * BM + wm * (BM / 2) + i * 16, col0 = bx * BN + wn * WN;
  for (int idx = lane; idx < 16 * WN; idx += 64) {
    const int r = idx / WN, c = idx - r * WN;
    if (row0 + r < M && col0 + c < N) {
      float* p = &P.C[(long)(row0 + r) * N + col0 + c];
      const float v = strip[r * (WN + 1) + c];
      if (atomic) atomicAdd(p, v);
      else *p += v;
    }
  }
}

// ---- host side: tile shape, plan (K slices per problem), table, launches ------------------------------------------------
struct GrpPlan {
  int splits, tiles, steps;
};
struct GrpTile {
  int bm, bn, ns;
};
static long grp_env(const char* name, long dflt) {
  const char* v = getenv(name);
  return v ? atol(v) : dflt;
}
// Tile shape per problem class, from a sweep over the weight-gradient sets of a training step (tools/wgrad_bench.py, B = 32):
// outputs with both sides >= 128 (stages 2 - 4, the DSEB projections, the upper decoder levels) run fastest on 128x128 tiles
// with a 2-stage ring (half the operand bytes through LDS per FLOP: 234 vs 381 us for stage 2, 297 vs 477 us for stage 3);
// skinny outputs (stage 1: 64 .. 512 x 64) keep the 64x64 tile with its 4-stage ring (304 vs 385 us).
static GrpTile grp_tile(int cls) {
  GrpTile t = cls ? GrpTile{128, 128, 2} : GrpTile{64, 64, 4};
  const char* v = getenv("CENET_GROUP_TILE");  // "128x128x2" etc. for every problem: measurement aid
  if (v) sscanf(v, "%dx%dx%d", &t.bm, &t.bn, &t.ns);
  return t;
}
static int grp_class(const cenet_wgrad_prob_t& q) { return (q.M >= 128 && q.N >= 128) ? 1 : 0; }
// One launch = up to GRP_MAXP problems of one orientation and one tile class.  Every item should reduce about `depth` K-steps:
// deep enough to amortise the ring's fill and the epilogue, shallow enough that the launch has a few items per workgroup slot.
static void grp_plan(const cenet_wgrad_prob_t* p, int n, GrpTile t, GrpPlan* out) {
  const long min_depth = grp_env("CENET_GROUP_DEPTH", 32), slots = grp_env("CENET_GROUP_ITEMS", 1536);  // (tuning aids)
  long work = 0;
  for (int i = 0; i < n; ++i) {
    out[i].tiles = cdiv(p[i].M, t.bm) * cdiv(p[i].N, t.bn);
    out[i].steps = p[i].nkb * cdiv(p[i].K, 64);
    work += (long)out[i].tiles * out[i].steps;
  }
  long depth = (work + slots - 1) / slots;
  if (depth < min_depth) depth = min_depth;
  for (int i = 0; i < n; ++i) {
    // a 2-way split doubles the tile's trips through the workspace for little balance: slice only reductions of >= 3 depths
    long s = out[i].steps / depth;
    if (s < 3) s = 1;
    if (s > out[i].steps / 8) s = out[i].steps / 8;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    out[i].splits = (int)s;
  }
}
static bool grp_ok(const cenet_wgrad_prob_t& q) {
  return q.A && q.B && q.C && q.M >= 1 && q.N >= 1 && q.K >= 1 && q.nkb >= 1 && q.M <= 65535 && q.N <= 65535 && q.nkb <= 65535 &&
         q.K < (1 << 28) && q.lda >= 0 && q.ldb >= 0 && q.skbA >= 0 && q.skbB >= 0 && q.lda < (1L << 31) && q.ldb < (1L << 31) &&
         q.skbA < (1L << 31) && q.skbB < (1L << 31) && q.akf == q.bkf;
}
// the partition of a call into launches: orientation (0: row-fast operands, 1: k-fast) x tile class, in chunks of GRP_MAXP
// problems in their given order; fn(sel, m, orientation, tile) returns false to stop
template <typename F>
static void grp_for_each_launch(const cenet_wgrad_prob_t* p, int n, F fn) {
  for (int o = 0; o < 2; ++o)
    for (int c = 0; c < 2; ++c) {
      cenet_wgrad_prob_t sel[GRP_MAXP];
      int m = 0;
      for (int i = 0; i < n; ++i) {
        if ((p[i].akf != 0) != (o != 0) || grp_class(p[i]) != c) continue;
        sel[m++] = p[i];
        if (m == GRP_MAXP) {
          if (!fn(sel, m, o, grp_tile(c))) return;
          m = 0;
        }
      }
      if (m && !fn(sel, m, o, grp_tile(c))) return;
    }
}

template <bool KF, int BM, int BN, int NS>
static void grp_launch(const GroupArgs& ga, float* ws, int items, int folds, int phase, hipStream_t stream) {
  if (phase != 2) CENET_LAUNCH((gemm_group_kernel<KF, KF, BM, BN, NS>), dim3(items), dim3(256), stream, ga, ws);
  if (folds && phase != 1)
    CENET_LAUNCH((gemm_group_fold_kernel<BM, BN>), dim3(folds, BM / 32), dim3(256), stream, ga, (const float*)ws);
}
template <bool KF>
static int grp_launch_tile(const GroupArgs& ga, GrpTile t, float* ws, int items, int folds, int phase, hipStream_t stream) {
  if (t.bm == 64 && t.bn == 64 && t.ns == 4) grp_launch<KF, 64, 64, 4>(ga, ws, items, folds, phase, stream);
  else if (t.bm == 128 && t.bn == 128 && t.ns == 2) grp_launch<KF, 128, 128, 2>(ga, ws, items, folds, phase, stream);
  else if (t.bm == 128 && t.bn == 128 && t.ns == 3) grp_launch<KF, 128, 128, 3>(ga, ws, items, folds, phase, stream);
  else if (t.bm == 128 && t.bn == 64 && t.ns == 3) grp_launch<KF, 128, 64, 3>(ga, ws, items, folds, phase, stream);
  else if (t.bm == 64 && t.bn == 128 && t.ns == 3) grp_launch<KF, 64, 128, 3>(ga, ws, items, folds, phase, stream);
  else return CENET_EUNSUPPORTED;
  return CENET_OK;
}

extern "C" long cenet_wgrad_group_ws_floats(const cenet_wgrad_prob_t* p, int n) {
  if (!p || n <= 0) return 0;
  long floats = 0;
  grp_for_each_launch(p, n, [&](const cenet_wgrad_prob_t* sel, int m, int, GrpTile t) {
    GrpPlan plan[GRP_MAXP];
    grp_plan(sel, m, t, plan);
    for (int i = 0; i < m; ++i)
      if (plan[i].splits > 1) floats += (long)plan[i].tiles * plan[i].splits * ((long)t.bm * t.bn + t.bm);
    return true;
  });
  return floats;
}

/* The partition a cenet_wgrad_group_bf16 call makes of its problems: launch[i] = index of the grouped launch problem i goes to (in
 * the order the launches are issued), bm / bn / ns [i] = that launch's tile and ring depth (kernel instance
 * gemm_group_kernel<akf, akf, bm, bn, ns>).  For measurement code that brackets the launches one by one (bench.py). */
extern "C" int cenet_wgrad_group_plan(const cenet_wgrad_prob_t* p, int n, int* launch, int* bm, int* bn, int* ns) {
  if (!p || n <= 0 || !launch || !bm || !bn || !ns) return CENET_EINVAL;
  for (int i = 0; i < n; ++i) {
    if (!grp_ok(p[i])) return CENET_EUNSUPPORTED;
    launch[i] = -1;
  }
  int li = 0;
  for (int o = 0; o < 2; ++o)
    for (int c = 0; c < 2; ++c) {
      int m = 0;
      const GrpTile t = grp_tile(c);
      for (int i = 0; i < n; ++i) {
        if ((p[i].akf != 0) != (o != 0) || grp_class(p[i]) != c) continue;
        launch[i] = li, bm[i] = t.bm, bn[i] = t.bn, ns[i] = t.ns;
        if (++m == GRP_MAXP) ++li, m = 0;
      }
      if (m) ++li;
    }
  return CENET_OK;
}

// phase 0: the whole reduction; 1: the K-slice launches only; 2: the fold launches only (1 then 2 on the same arguments = 0;
// lets a profiler-free measurement bracket the two kernels separately, bench.py)
static int wgrad_group_impl(const cenet_wgrad_prob_t* p, int n, float* ws, long ws_floats, int phase, hipStream_t stream) {
  if (!p || n <= 0 || phase < 0 || phase > 2) return CENET_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!grp_ok(p[i])) return p[i].akf != p[i].bkf ? CENET_EUNSUPPORTED : CENET_EINVAL;
  {
    // dry pass: the whole call is planned and its workspace need checked BEFORE the first launch (the launches add into the
    // gradient arena: an error after some of them would leave the gradients half-updated)
    const long need = cenet_wgrad_group_ws_floats(p, n);
    if (need > ws_floats || (need > 0 && !ws)) return CENET_EINVAL;
  }
  long ws_used = 0;  // floats handed out so far: every launch gets its own stretch of the workspace
  int rc = CENET_OK;
  grp_for_each_launch(p, n, [&](const cenet_wgrad_prob_t* sel, int m, int o, GrpTile t) {
    const long tile_floats = (long)t.bm * t.bn + t.bm;
    GrpPlan plan[GRP_MAXP];
    grp_plan(sel, m, t, plan);
    // longest items first: the items of a launch are started in list order, so the deep reductions begin at once and the
    // shallow ones fill the tail (stable insertion sort on steps per item)
    int ord[GRP_MAXP];
    for (int i = 0; i < m; ++i) ord[i] = i;
    auto depth_of = [&](int i) { return cdiv(plan[i].steps, plan[i].splits); };
    for (int i = 1; i < m; ++i) {
      const int v = ord[i];
      int j = i;
      while (j > 0 && depth_of(ord[j - 1]) < depth_of(v)) ord[j] = ord[j - 1], --j;
      ord[j] = v;
    }
    GroupArgs ga;
    memset(&ga, 0, sizeof ga);
    ga.nprob = m;
    int items = 0, folds = 0;
    float* wsl = ws ? ws + ws_used : nullptr;
    long used = 0;
    for (int i = 0; i < m; ++i) {
      const cenet_wgrad_prob_t& q = sel[ord[i]];
      const GrpPlan& pl = plan[ord[i]];
      GroupProb& d = ga.p[i];
      d.A = (const bf16_t*)q.A;
      d.B = (const bf16_t*)q.B;
      d.C = q.C;
      d.asum = q.asum;
      d.lda = (int)q.lda; d.ldb = (int)q.ldb; d.skbA = (int)q.skbA; d.skbB = (int)q.skbB;
      d.M = (unsigned short)q.M; d.N = (unsigned short)q.N; d.nkb = (unsigned short)q.nkb;
      d.splits = (unsigned short)pl.splits;
      // 16-byte chunks on 16-byte boundaries, or the 2-byte-aligned LDS-DMA form with hand-fetched final chunks (gemm.hip)
      auto e8 = [](long v) { return (v & 7) == 0; };
      const bool a_al = e8(q.skbA) && (((uintptr_t)q.A & 15) == 0) && e8(q.lda) && e8(q.akf ? q.K : q.M);
      const bool b_al = e8(q.skbB) && (((uintptr_t)q.B & 15) == 0) && e8(q.ldb) && e8(q.bkf ? q.K : q.N);
      int same_c = 0, same_s = 0;  // another problem of this call adds into the same C / asum (a parameter used twice): atomics
      for (int j = 0; j < n; ++j) same_c += p[j].C == q.C, same_s += (q.asum && p[j].asum == q.asum);
      const bool dup = same_c > 1 || same_s > 1;
      d.K = q.K | ((a_al && b_al) ? 0 : (1 << 28)) | (dup ? (1 << 29) : 0);
      ga.item0[i] = items;
      ga.fold0[i] = folds;
      items += pl.tiles * pl.splits;
      if (pl.splits > 1) {
        d.ws_tile0 = (int)(used / tile_floats);
        used += (long)pl.tiles * pl.splits * tile_floats;
        folds += pl.tiles;
      }
    }
    for (int i = m; i <= GRP_MAXP; ++i) ga.item0[i] = items, ga.fold0[i] = folds;
    ga.nitems = items;
    ws_used += used;
    if (ws_used > ws_floats || (used && !ws)) {
      rc = CENET_EINVAL;
      return false;
    }
    rc = o ? grp_launch_tile<true>(ga, t, wsl, items, folds, phase, stream) : grp_launch_tile<false>(ga, t, wsl, items, folds, phase, stream);
    return rc == CENET_OK;
  });
  if (rc != CENET_OK) return rc;
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
extern "C" int cenet_wgrad_group_bf16(const cenet_wgrad_prob_t* p, int n, float* ws, long ws_floats, hipStream_t stream) {
  return wgrad_group_impl(p, n, ws, ws_floats, 0, stream);
}
extern "C" int cenet_wgrad_group_phase_bf16(const cenet_wgrad_prob_t* p, int n, float* ws, long ws_floats, int phase,
                                            hipStream_t stream) {
  return wgrad_group_impl(p, n, ws, ws_floats, phase, stream);
}
