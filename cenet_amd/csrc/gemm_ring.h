// gemm_ring.h — bf16 GEMM for plain (non-gathering) operands whose fast axis is contiguous and 16-byte aligned: the
// operand tiles go HBM -> LDS with global_load_lds_dwordx4 (no staging registers) into a ring of NS stage buffers, so a
// workgroup keeps NS - 1 K-steps of 64 in flight; one raw s_barrier per K-step, counted s_waitcnt vmcnt(N).
//
// The contractions of this network are short (K = 64 .. 1280 per workgroup, or split-K chunks of that size), so the
// register-prefetch kernel of gemm_core.h (one K-step in flight, two barriers per step) spends most of its time waiting
// for the first bytes; measured in isolation every shape of a training step ran 3 - 30x above its HBM / MFMA bound.
//
// Operand orientations (template flags, any combination):
//   k-fast (KF): element (row, k) at row * ld + k.        LDS image [rows][64 k], 128-byte rows, 16-byte chunk c of
//                row r stored at chunk c ^ ((r >> 1) & 7): fragment reads (ds_read_b128, 16 rows x 16 bytes) conflict-free.
//   row-fast (RF): element (row, k) at k * ld + row.      LDS image [64 k][BX rows]; fragments by ds_read_b64_tr_b16
//                (4 k-rows x 16 columns per 16-lane group), chunk c of k-row k stored at c ^ key(k) (rf_key below).
// An LDS-DMA writes wave-uniform base + lane * 16, so the images are lane-linear and the swizzle is applied to the
// per-lane SOURCE address (cdna_hip_programming.md rule 21).
//
// Rows beyond M / N are clamped to valid rows (their products land in accumulator rows the epilogue never stores); k
// beyond K reads a 16-byte zero block.  Everything else (epilogue, split-K, batching, XCD order) is gemm_core.h's.
#pragma once
#include "gemm_core.h"

typedef short ring_s4 __attribute__((ext_vector_type(4)));

// swizzle keys
__device__ __forceinline__ int kf_key(int row) { return (row >> 1) & 7; }
template <int BX>
__device__ __forceinline__ int rf_key(int k) {
  // the 32 lanes of one ds_read_b64_tr_b16 pass touch k-rows {0..3, 8..11} (+4 for the second read) x 2 chunks
  return BX == 128 ? 2 * (k & 3) + 8 * ((k >> 3) & 1) : 2 * ((k >> 1) & 1) + 4 * ((k >> 3) & 1);
}

// fragment of a k-fast image: rows r0 + (lane & 15), k = 32 kc + 8 (lane >> 4) .. + 7
__device__ __forceinline__ bf16x8 ring_frag_kf(const unsigned char* img, int r0, int kc, int lane) {
  const int row = r0 + (lane & 15), c = kc * 4 + (lane >> 4);
  bf16x8 f;
  memcpy(&f, img + (row * 8 + (c ^ kf_key(row))) * 16, 16);
  return f;
}
// fragment of a row-fast image [64][BX]: same element set, gathered by two transposing reads
template <int BX>
__device__ __forceinline__ bf16x8 ring_frag_rf(const unsigned char* img, int r0, int kc, int lane) {
  constexpr int CH = BX / 8;
  const int g = lane >> 4, i = lane & 15;
  bf16x8 f;
#ifdef CENET_HOSTSIM_BUILD
  const int x = r0 + i;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kc * 32 + 8 * g + j;
    const unsigned short* p = (const unsigned short*)(img + (k * CH + ((x >> 3) ^ rf_key<BX>(k))) * 16) + (x & 7);
    f[j] = (short)*p;
  }
#else
  const int q = i >> 2, p = i & 3;
  const int x = r0 + 4 * p;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = kc * 32 + 8 * g + 4 * h + q;
    const unsigned char* a = img + (k * CH + ((x >> 3) ^ rf_key<BX>(k))) * 16 + (x & 7) * 2;
    const ring_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ring_s4 __attribute__((address_space(3)))*)a);
    f[4 * h + 0] = v[0], f[4 * h + 1] = v[1], f[4 * h + 2] = v[2], f[4 * h + 3] = v[3];
  }
#endif
  return f;
}

// per-thread source description of one operand: NI = BX / 32 LDS-DMA instructions per thread and tile
template <int NI>
struct RingSrc {
  const bf16_t* p[NI];  // source of this lane's chunk at k0 = 0 (row / chunk clamped into the operand)
  int kofs[NI];         // k of the chunk's first element relative to the tile's k0
};

template <bool KF, int BX>
__device__ __forceinline__ RingSrc<BX / 32> ring_src(const bf16_t* base, long ld, int x0, int X, int wave, int lane) {
  constexpr int NI = BX / 32, CH = BX / 8;
  RingSrc<NI> s;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int S = (j * 4 + wave) * 64 + lane;  // linear 16-byte slot of the image
    if (KF) {
      const int row = S >> 3, c = (S & 7) ^ kf_key(row);
      int r = x0 + row;
      r = r < X ? r : X - 1;
      s.p[j] = base + (long)r * ld + 8 * c;
      s.kofs[j] = 8 * c;
    } else {
      const int k = S / CH, c = (S % CH) ^ rf_key<BX>(k);
      int x = x0 + 8 * c;
      x = x < X ? x : x0;  // a chunk that starts inside may run past X: those columns / rows are never stored
      s.p[j] = base + (long)k * ld + x;
      s.kofs[j] = k;
    }
  }
  return s;
}

// A chunk whose 16 bytes would cross the end of the operand (unaligned operands only: the last row's last chunk): fetch the
// 8 elements that END at `end` with ordinary loads, shift the valid ones to the front, zero the rest, store to the slot.
__device__ __forceinline__ void ring_fix_tail_chunk(const bf16_t* src, const bf16_t* end, unsigned char* slot) {
  const int d = (int)((src + 8) - end);  // 1 .. 7 elements beyond the end
  unsigned short in[8], out[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) in[j] = (end - 8)[j];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    unsigned short v = 0;
#pragma unroll
    for (int t = 1; t < 8; ++t)
      if (t == d && j + t < 8) v = in[j + t];
    out[j] = v;
  }
  memcpy(slot, out, 16);
}

template <bool KF, int BX>
__device__ __forceinline__ void ring_issue(const RingSrc<BX / 32>& s, long koff, int klim, bool tail, unsigned char* img, int wave,
                                           int lane, bool unal, const bf16_t* end) {
  constexpr int NI = BX / 32;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const bf16_t* src = s.p[j] + koff;
    const bool zero = tail && s.kofs[j] >= klim;
    unsigned char* dst = img + (j * 4 + wave) * 1024;
    if (unal) {  // (wave-uniform)
      const bool cross = !zero && src + 8 > end;
      if (!cross) ring_glds16(zero ? (const void*)ring_zero16 : (const void*)src, dst, lane);  // other lanes: exec-masked DMA
      else ring_fix_tail_chunk(src, end, dst + 16 * lane);
    } else {
      ring_glds16(zero ? (const void*)ring_zero16 : (const void*)src, dst, lane);
    }
  }
}

// k-fast fragment of a K-tail tile whose K is not a multiple of 8: elements k >= klim hold the next row's data -> zero
__device__ __forceinline__ bf16x8 ring_mask_k(bf16x8 f, int kc, int lane, int klim) {
  const int nv = klim - (kc * 32 + 8 * (lane >> 4));  // valid elements of this lane's 8
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j >= nv) f[j] = 0;
  return f;
}

template <bool AKF, bool BKF, int BM, int BN, int NS, bool SWAP>
__global__ __launch_bounds__(256, (NS * (BM + BN) * 128 <= 80 * 1024 ? 2 : 1)) void gemm_ring_kernel(GemmArgs g) {
  constexpr int MI = BM / 32, NJ = BN / 32;
  constexpr int ABYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int G = (BM + BN) / 32;  // LDS-DMA instructions per thread and tile
  static_assert(NS >= 2 && NS <= 4 && G * (NS - 2) <= 63, "ring depth");
  static_assert(NS * STAGE >= 4 * 16 * (BN / 2 + 1) * 4, "the epilogue strip reuses the ring");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE];  // ONE array: stages, then the epilogue strip

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef CENET_HOSTSIM_BUILD
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int wm = wave >> 1, wn = wave & 1;
  int bx, by, bz;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    const int T = gx * gy * (int)gridDim.z;
    int L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if (T >= 64) {  // XCD-aware order (see gemm_kernel)
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

  const int ktiles = (g.K + 63) / 64;
  const int total = g.nkb * ktiles;
  const int chunk = (total + g.splits - 1) / g.splits;
  const int it0 = split * chunk;
  const int it1 = (it0 + chunk < total) ? it0 + chunk : total;
  const int T = it1 - it0;

  const bf16_t* baseA = (const bf16_t*)g.A.ptr + (long)bo * g.A.sb + (long)bi * g.A.sb2;
  const bf16_t* baseB = (const bf16_t*)g.B.ptr + (long)bo * g.B.sb + (long)bi * g.B.sb2;
  // A(m, k): KF -> m * sr + k ; RF -> k * sc + m.    B(k, n): KF -> n * sc + k ; RF -> k * sr + n.
  const long lda = AKF ? g.A.sr : g.A.sc, ldb = BKF ? g.B.sc : g.B.sr;
  const RingSrc<MI> sa = ring_src<AKF, BM>(baseA, lda, m0, g.M, wave, lane);
  const RingSrc<NJ> sb = ring_src<BKF, BN>(baseB, ldb, n0, g.N, wave, lane);
  const long kstepA = AKF ? 1 : lda, kstepB = BKF ? 1 : ldb;

  auto issue = [&](int it, int buf) __attribute__((always_inline)) {
    const int kb = it / ktiles;
    const int k0 = (it - kb * ktiles) * 64;
    const int klim = g.K - k0;
    const bool tail = klim < 64;
    unsigned char* img = lds + buf * STAGE;
    ring_issue<AKF, BM>(sa, (long)kb * g.A.skb + (long)k0 * kstepA, klim, tail, img, wave, lane, g.ring_unal, (const bf16_t*)g.a_end);
    ring_issue<BKF, BN>(sb, (long)kb * g.B.skb + (long)k0 * kstepB, klim, tail, img + ABYTES, wave, lane, g.ring_unal,
                        (const bf16_t*)g.b_end);
  };
  const bool kmask = (AKF || BKF) && (g.K & 7) != 0;  // K tails inside a 16-byte chunk of a k-fast operand
  // bias gradient riding in the weight-gradient pass: row sums of A, taken from the fragments of the first column of tiles
  const bool do_asum = g.E.asum != nullptr && bx == 0 && wn == 0;
  float rsum[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) rsum[i] = 0.f;

  // prologue: NS - 1 tiles in flight
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < T) issue(it0 + s, s);

  int cur = 0;
  for (int t = 0; t < T; ++t) {
    // tile t has landed once at most min(NS - 2, T - 1 - t) younger tiles are still outstanding
    const int young = T - 1 - t;
    if (NS >= 4 && young >= 2) ring_wait_vm<G * 2 <= 63 ? G * 2 : 0>();
    else if (NS >= 3 && young >= 1) ring_wait_vm<G>();
    else ring_wait_vm<0>();
    ring_barrier();  // every wave's part of tile t is in LDS; every wave is done reading the buffer refilled below
    if (t + NS - 1 < T) issue(it0 + t + NS - 1, cur == 0 ? NS - 1 : cur - 1);
    const unsigned char* Ai = lds + cur * STAGE;
    const unsigned char* Bi = Ai + ABYTES;
    int klim_t = 64;
    if (kmask) {
      const int it = it0 + t;
      klim_t = g.K - (it - (it / ktiles) * ktiles) * 64;
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
      if (klim_t < 64) {  // (wave-uniform; only the last K tile of an unaligned K)
        if (AKF) {
#pragma unroll
          for (int i = 0; i < MI; ++i) a[i] = ring_mask_k(a[i], kc, lane, klim_t);
        }
        if (BKF) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) b[j] = ring_mask_k(b[j], kc, lane, klim_t);
        }
      }
      if (do_asum) {  // (wave-uniform)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) rsum[i] += cenet_bf2f((unsigned short)a[i][j]);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
          acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  if (do_asum) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float v = rsum[i];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int row = m0 + wm * (BM / 2) + i * 16 + lane;
      if (lane < 16 && row < g.M) atomicAdd(&g.E.asum[row], v);
    }
  }
  __syncthreads();  // the ring is free: the atomic epilogue uses it as its transpose strip
  static_assert(NS * STAGE >= (BM > BN ? BM : BN) * ((BM > BN ? BN : BM) + 8) * 2 && NS * STAGE >= BM * (BN + 8) * 2,
                "the staged epilogue image reuses the ring");
  gemm_epilogue<bf16_t, BM, BN, SWAP, true>(g, acc, (float*)lds, m0, n0, bo, bi, batch, wave, lane);
}

// ---- streaming variant (round 4): reductions of ONE or TWO K-steps (K = 64 / 128, aligned operands, no split) -----------------
// For K <= 128 the ring above has nothing to pipeline: a workgroup issues its tile's loads, waits out the whole first-byte
// latency, does 1 - 2 MFMA steps, stores and exits; a launch is then ~3 rounds of (launch ramp + latency + store) whatever the
// byte count (16 us for 26 MB: 1.6 TB/s).  Here workgroups are PERSISTENT over 64x64 output tiles and the ring runs over TILES:
// while tile i is multiplied and stored, the operands of tile i + 1 are already in flight into the other buffer set.  One
// barrier per tile; the wait for the next tile's loads sits BEFORE the current tile's stores are issued, so a store's
// acknowledgement is never waited for (vmcnt counts loads and stores in issue order).  Epilogue = the direct (unstaged) one.
template <bool AKF, bool BKF, bool SWAP, int T>
__global__ __launch_bounds__(256, 2) void gemm_ring_stream_kernel(GemmArgs g, int tiles_x, int tiles_y, int total_tiles) {
  constexpr int BM = 64, BN = 64, MI = 2, NJ = 2;
  constexpr int ABYTES = BM * 128, STAGE = (BM + BN) * 128;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * T * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef CENET_HOSTSIM_BUILD
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int wm = wave >> 1, wn = wave & 1;
  const long lda = AKF ? g.A.sr : g.A.sc, ldb = BKF ? g.B.sc : g.B.sr;
  const long kstepA = AKF ? 1 : lda, kstepB = BKF ? 1 : ldb;
  // virtual tile index -> tile: each XCD owns a contiguous range of the tile list (workgroups are dealt round-robin over the XCDs)
  auto tile_of = [&](int vt) -> int {
    if ((gridDim.x & 7) == 0 && total_tiles >= 64) {
      const int per = total_tiles >> 3, rem = total_tiles & 7, xcd = vt & 7, idx = vt >> 3;
      return xcd * per + (xcd < rem ? xcd : rem) + idx;
    }
    return vt;
  };
  struct Where { int m0, n0, bo, bi, batch; };
  auto where = [&](int L) -> Where {
    Where w;
    const int bx = L % tiles_x, t = L / tiles_x, by = t % tiles_y;
    w.batch = t / tiles_y;
    w.bo = w.batch / g.nb_inner;
    w.bi = w.batch - w.bo * g.nb_inner;
    w.m0 = by * BM;
    w.n0 = bx * BN;
    return w;
  };
  auto issue = [&](const Where& w, int set) __attribute__((always_inline)) {
    const bf16_t* baseA = (const bf16_t*)g.A.ptr + (long)w.bo * g.A.sb + (long)w.bi * g.A.sb2;
    const bf16_t* baseB = (const bf16_t*)g.B.ptr + (long)w.bo * g.B.sb + (long)w.bi * g.B.sb2;
    const RingSrc<MI> sa = ring_src<AKF, BM>(baseA, lda, w.m0, g.M, wave, lane);
    const RingSrc<NJ> sb = ring_src<BKF, BN>(baseB, ldb, w.n0, g.N, wave, lane);
#pragma unroll
    for (int kt = 0; kt < T; ++kt) {
      unsigned char* img = lds + (set * T + kt) * STAGE;
      ring_issue<AKF, BM>(sa, (long)kt * 64 * kstepA, 64, false, img, wave, lane, false, nullptr);
      ring_issue<BKF, BN>(sb, (long)kt * 64 * kstepB, 64, false, img + ABYTES, wave, lane, false, nullptr);
    }
  };
  int vt = blockIdx.x;
  if (vt >= total_tiles) return;
  Where cur = where(tile_of(vt));
  issue(cur, 0);
  ring_wait_vm<0>();
  int set = 0;
  for (; vt < total_tiles; vt += gridDim.x, set ^= 1) {
    ring_barrier();  // every wave's part of this tile has landed (each waited for its own); everybody is done with the other set
    const int nvt = vt + gridDim.x;
    Where nxt = cur;
    if (nvt < total_tiles) {
      nxt = where(tile_of(nvt));
      issue(nxt, set ^ 1);
    }
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < T; ++kt) {
      const unsigned char* Ai = lds + (set * T + kt) * STAGE;
      const unsigned char* Bi = Ai + ABYTES;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        bf16x8 a[MI], b[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
          a[i] = AKF ? ring_frag_kf(Ai, wm * 32 + i * 16, kc, lane) : ring_frag_rf<BM>(Ai, wm * 32 + i * 16, kc, lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          b[j] = BKF ? ring_frag_kf(Bi, wn * 32 + j * 16, kc, lane) : ring_frag_rf<BN>(Bi, wn * 32 + j * 16, kc, lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int i = 0; i < MI; ++i)
            acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    ring_wait_vm<0>();  // the next tile's loads (issued a whole tile ago) and the PREVIOUS tile's stores; not this tile's stores
    gemm_epilogue<bf16_t, BM, BN, SWAP, false>(g, acc, (float*)nullptr, cur.m0, cur.n0, cur.bo, cur.bi, cur.batch, wave, lane);
    cur = nxt;
  }
}

template <bool AKF, bool BKF, bool SWAP>
static int launch_ring_stream(const GemmArgs& g, int nbatch, hipStream_t stream) {
  const int tx = cdiv(g.N, 64), ty = cdiv(g.M, 64);
  const long total = (long)tx * ty * nbatch;
  if (total > (1L << 30)) return CENET_EUNSUPPORTED;
  static const char* ge = getenv("CENET_STREAM_WGS");  // measurement aid
  long wgs = ge ? atol(ge) : 1024;  // (measured on the whole step: 256 / 512 / 768 / 1024 / 2048 -> 21.6 / 21.55 / 21.50 / 21.46 / 21.47 ms)
  if (wgs > total) wgs = total;
  if (wgs >= 8) wgs &= ~7L;
  if (g.K == 64) CENET_LAUNCH((gemm_ring_stream_kernel<AKF, BKF, SWAP, 1>), dim3((unsigned)wgs), dim3(256), stream, g, tx, ty, (int)total);
  else if (g.K == 128) CENET_LAUNCH((gemm_ring_stream_kernel<AKF, BKF, SWAP, 2>), dim3((unsigned)wgs), dim3(256), stream, g, tx, ty, (int)total);
  else return CENET_EUNSUPPORTED;
  return CENET_OK;
}

template <bool AKF, bool BKF, bool SWAP>
static int launch_ring(const GemmArgs& g, int bm, int bn, int nbatch, hipStream_t stream) {
  dim3 grid(cdiv(g.N, bn), cdiv(g.M, bm), nbatch * g.splits);
  if (grid.y > 65535 || grid.z > 65535) return CENET_EUNSUPPORTED;
  if (bm == 128 && bn == 128) {
    CENET_LAUNCH((gemm_ring_kernel<AKF, BKF, 128, 128, 2, SWAP>), grid, dim3(256), stream, g);
    return CENET_OK;
  }
  if (bm == 64 && bn == 64) {
    CENET_LAUNCH((gemm_ring_kernel<AKF, BKF, 64, 64, 4, SWAP>), grid, dim3(256), stream, g);
    return CENET_OK;
  }
  if (bm == 128 && bn == 64) {  // skinny outputs: the long side's operand is re-read once per tile of the short side only
    CENET_LAUNCH((gemm_ring_kernel<AKF, BKF, 128, 64, 3, SWAP>), grid, dim3(256), stream, g);
    return CENET_OK;
  }
  if (bm == 64 && bn == 128) {
    CENET_LAUNCH((gemm_ring_kernel<AKF, BKF, 64, 128, 3, SWAP>), grid, dim3(256), stream, g);
    return CENET_OK;
  }
  return CENET_EUNSUPPORTED;
}
// bm == 0: the streaming variant (K = 64 / 128, 64x64 tiles)
#define CENET_RING_INSTANCE(NAME, AKF, BKF)                                                       \
  int NAME(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream) {        \
    if (bm == 0)                                                                                  \
      return swap ? launch_ring_stream<AKF, BKF, true>(g, nbatch, stream)                         \
                  : launch_ring_stream<AKF, BKF, false>(g, nbatch, stream);                       \
    return swap ? launch_ring<AKF, BKF, true>(g, bm, bn, nbatch, stream)                          \
                : launch_ring<AKF, BKF, false>(g, bm, bn, nbatch, stream);                        \
  }
int cenet_gemm_launch_ring_kk(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream);
int cenet_gemm_launch_ring_kr(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream);
int cenet_gemm_launch_ring_rk(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream);
int cenet_gemm_launch_ring_rr(const GemmArgs& g, int bm, int bn, int nbatch, bool swap, hipStream_t stream);
