// gemm.hip — C-ABI entry of the GEMM / implicit-GEMM core: argument checks, tile and split-K selection, dispatch to the
// per-type instantiations (gemm_inst_*.hip).  The kernel itself is in gemm_core.h.
#include "gemm_ring.h"
#include <cstdio>
#include <cstdlib>

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

// name of the kernel instance the last cenet_gemm_* call on this thread launched, spelled as rocprofv3 prints it
// (measurement aid: bench.py groups its live per-launch timings by this name)
static thread_local char g_last_kernel[96] = "";
extern "C" const char* cenet_gemm_last_kernel(void) { return g_last_kernel; }
static inline const char* tf(bool b) { return b ? "true" : "false"; }

// asum fallback for contractions the ring kernel does not take: asum[m] += sum over (kb, k) of A[kb*skb + m*sr + k*sc]
template <typename GT>
__global__ __launch_bounds__(256) void gemm_asum_kernel(const GT* __restrict__ A, long sr, long sc, long skb, int M, int K, int nkb,
                                                       float* __restrict__ asum) {
  __shared__ float red[16];
  if (sr == 1 && sc != 1) {  // rows contiguous: thread = row, grid.y slices the reduction
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const long total = (long)nkb * K;
    float s = 0.f;
    for (long e = blockIdx.y; e < total; e += gridDim.y) {
      const long kb = e / K, k = e - kb * K;
      s += ldf(A + kb * skb + k * sc + m);
    }
    atomicAdd(&asum[m], s);
  } else {  // workgroup = row, threads along k
    const int m = blockIdx.x;
    const long total = (long)nkb * K;
    float s = 0.f;
    for (long e = blockIdx.y * 256 + threadIdx.x; e < total; e += (long)gridDim.y * 256) {
      const long kb = e / K, k = e - kb * K;
      s += ldf(A + kb * skb + (long)m * sr + k * sc);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(&asum[m], s);
  }
}
extern "C" int cenet_col_sum_acc_f32(const float* a, float* out_acc, long R, int C, hipStream_t stream);
extern "C" int cenet_col_sum_acc_bf16(const bf16_t* a, float* out_acc, long R, int C, hipStream_t stream);
extern "C" int cenet_chan_dot_acc_f32(const float* a, long sab, const float* b, long sbb, float* out_acc, int B, int C, int HW,
                                      hipStream_t stream);
extern "C" int cenet_chan_dot_acc_bf16(const bf16_t* a, long sab, const bf16_t* b, long sbb, float* out_acc, int B, int C, int HW,
                                       hipStream_t stream);

template <typename GT>
static void launch_asum(const cenet_mat_t* A, int M, int K, int nkb, float* asum, hipStream_t stream) {
  // the two layouts the network produces have tuned reductions of their own (elementwise.hip): token-major dY [K, M]
  // (column sums) and per-image channel planes dY [nkb][M][K] (plane sums)
  if (A->sr == 1 && A->sc == M && nkb == 1) {
    if (sizeof(GT) == 2) cenet_col_sum_acc_bf16((const bf16_t*)A->ptr, asum, K, M, stream);
    else cenet_col_sum_acc_f32((const float*)A->ptr, asum, K, M, stream);
    return;
  }
  if (A->sc == 1 && A->sr == K && (nkb == 1 || A->skb >= (long)M * K)) {
    if (sizeof(GT) == 2) cenet_chan_dot_acc_bf16((const bf16_t*)A->ptr, A->skb, nullptr, 0, asum, nkb, M, K, stream);
    else cenet_chan_dot_acc_f32((const float*)A->ptr, A->skb, nullptr, 0, asum, nkb, M, K, stream);
    return;
  }
  const long total = (long)nkb * K;
  if (A->sr == 1 && A->sc != 1) {
    int ys = (int)((total + 511) / 512);
    if (ys > 256) ys = 256;
    CENET_LAUNCH((gemm_asum_kernel<GT>), dim3(cdiv(M, 256), ys), dim3(256), stream, (const GT*)A->ptr, A->sr, A->sc, A->skb, M, K,
                 nkb, asum);
  } else {
    int ys = (int)((total + 8191) / 8192);
    if (ys > 64) ys = 64;
    CENET_LAUNCH((gemm_asum_kernel<GT>), dim3(M, ys), dim3(256), stream, (const GT*)A->ptr, A->sr, A->sc, A->skb, M, K, nkb, asum);
  }
}

static inline bool m4(long v) { return (v & 3) == 0; }
static inline bool m2(long v) { return (v & 1) == 0; }

// esz: element size of A / B (and of C / R unless the epilogue is atomic): 4 = fp32 (exact MFMA chain), 2 = bf16
static int gemm_dispatch(const cenet_mat_t* A, const cenet_mat_t* B, const cenet_epi_t* E, int M, int N, int K, int nbatch,
                         int nb_inner, int nkb, int splits, int esz, hipStream_t stream) {
  if (!A || !B || !E || !A->ptr || !B->ptr || !E->C) return CENET_EINVAL;
  if (M <= 0 || N <= 0 || K <= 0 || nbatch <= 0 || nb_inner <= 0 || nkb <= 0 || splits <= 0) return CENET_EINVAL;
  if (A->mode != 0) return CENET_EUNSUPPORTED;
  if (splits > 1 && !E->atomic) return CENET_EINVAL;
  if (E->atomic == 2 && (splits > 1 || E->cmode || E->asum)) return CENET_EINVAL;  // fp32 STORE of an unsplit product
  if (E->atomic && (E->bias || E->R || E->act != ACT_NONE || E->bscale)) return CENET_EINVAL;
  if (E->asum && (!E->atomic || nbatch != 1 || A->kinner != 0)) return CENET_EINVAL;
  const bool bf = esz == 2;
  const uintptr_t qmask = 4 * esz - 1;  // a quad of elements: 16 bytes (fp32) / 8 bytes (bf16)
  auto alq = [&](const void* p) { return ((uintptr_t)p & qmask) == 0; };
  GemmArgs g;
  g.A = *A;
  g.B = *B;
  g.E = *E;
  g.M = M; g.N = N; g.K = K; g.nkb = nkb; g.splits = splits; g.nb_inner = nb_inner;
  // quad staging: k-contiguous plain operand, every row/batch offset a multiple of 4 elements, K % 4 == 0
  g.avec = (A->kfast && A->kinner == 0 && A->sc == 1 && m4(A->sr) && m4(A->sb) && m4(A->sb2) && m4(A->skb) && m4(K) &&
            alq(A->ptr)) ? 4 : 0;
  g.bvec = (B->mode == 0 && B->kfast && B->kinner == 0 && B->sr == 1 && m4(B->sc) && m4(B->sb) && m4(B->sb2) && m4(B->skb) &&
            m4(K) && alq(B->ptr)) ? 4 : 0;
  // bf16: 16-byte accesses (8 elements) where every offset is a multiple of 8 elements
  auto m8 = [](long v) { return (v & 7) == 0; };
  if (bf && g.avec && m8(A->sr) && m8(A->sb) && m8(A->sb2) && m8(A->skb) && m8(K) && (((uintptr_t)A->ptr & 15) == 0)) g.avec = 8;
  if (bf && g.bvec && m8(B->sc) && m8(B->sb) && m8(B->sb2) && m8(B->skb) && m8(K) && (((uintptr_t)B->ptr & 15) == 0)) g.bvec = 8;
  // bf16 row-contiguous operands: two adjacent rows per 4-byte load (even extents and offsets)
  g.apair = bf && !A->kfast && A->kinner == 0 && A->sr == 1 && m2(A->sc) && m2(A->sb) && m2(A->sb2) && m2(A->skb) && m2(M) &&
            (((uintptr_t)A->ptr & 3) == 0);
  g.bpair = bf && B->mode == 0 && !B->kfast && B->kinner == 0 && B->sc == 1 && m2(B->sr) && m2(B->sb) && m2(B->sb2) &&
            m2(B->skb) && m2(N) && (((uintptr_t)B->ptr & 3) == 0);
  // row-major C: lanes own 4 consecutive columns (quad epilogue). Atomic epilogues keep the un-swapped layout: one
  // atomic instruction then covers 4 rows x 16 consecutive floats (64-byte segments) instead of 16 rows x 4 strided.
  const bool swap = (E->scc == 1) && !E->atomic && !E->cmode;
  if (swap)
    g.cvec = E->scc == 1 && m4(E->scr) && m4(E->scb) && m4(E->scb2) && alq(E->C) &&
             (!E->R || (E->src == 1 && m4(E->srr) && m4(E->srb) && m4(E->srb2) && alq(E->R)));
  else
    g.cvec = E->scr == 1 && m4(E->scc) && m4(E->scb) && m4(E->scb2) && alq(E->C) &&
             (!E->R || (E->srr == 1 && m4(E->src) && m4(E->srb) && m4(E->srb2) && alq(E->R)));
  {
    static const bool no_stage = getenv("CENET_GEMM_NO_STAGE") != nullptr;  // measurement aid
    const long so = swap ? E->scr : E->scc;
    g.cvec8 = !no_stage && bf && g.cvec && (so % 8) == 0 && (E->scb % 8) == 0 && (E->scb2 % 8) == 0 && (((uintptr_t)E->C & 15) == 0);
  }
  int bm, bn;
  pick_tile(M, N, nbatch, splits, &bm, &bn);
  if (B->mode == 1 && B->kfast && bm == 32) bn = 64;  // weight-gradient view: keep the per-thread gather list short
  // LDS-DMA ring kernel (gemm_ring.h): bf16, both operands plain with a contiguous, 16-byte aligned fast axis
  int akf = -1, bkf = -1;
  g.ring_unal = 0;
  g.a_end = g.b_end = nullptr;
  if (bf && B->mode == 0 && A->kinner == 0 && B->kinner == 0) {
    auto e8 = [](long v) { return (v & 7) == 0; };
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    akf = A->sc == 1 ? 1 : (A->sr == 1 ? 0 : -1);
    bkf = B->sr == 1 ? 1 : (B->sc == 1 ? 0 : -1);
    if (akf >= 0 && bkf >= 0 && A->sb >= 0 && A->sb2 >= 0 && A->skb >= 0 && B->sb >= 0 && B->sb2 >= 0 && B->skb >= 0) {
      const long lda = akf ? A->sr : A->sc, ldb = bkf ? B->sc : B->sr;
      const bool a_al = e8(A->sb) && e8(A->sb2) && e8(A->skb) && al16(A->ptr) && e8(lda) && e8(akf ? K : M);
      const bool b_al = e8(B->sb) && e8(B->sb2) && e8(B->skb) && al16(B->ptr) && e8(ldb) && e8(bkf ? K : N);
      g.ring_unal = !(a_al && b_al);
      const long bo = (nbatch - 1) / nb_inner, bi = (nbatch < nb_inner ? nbatch : nb_inner) - 1;
      const long a_last = bo * A->sb + bi * A->sb2 + (long)(nkb - 1) * A->skb + (akf ? (long)(M - 1) * lda + K : (long)(K - 1) * lda + M);
      const long b_last = bo * B->sb + bi * B->sb2 + (long)(nkb - 1) * B->skb + (bkf ? (long)(N - 1) * ldb + K : (long)(K - 1) * ldb + N);
      g.a_end = (const unsigned short*)A->ptr + a_last;
      g.b_end = (const unsigned short*)B->ptr + b_last;
      if (lda < 0 || ldb < 0 || a_last < 8 || b_last < 8) akf = bkf = -1;
    } else {
      akf = bkf = -1;
    }
  }
  static const bool ring_off = getenv("CENET_GEMM_NO_RING") != nullptr;
  const bool ring = !ring_off && akf >= 0 && bkf >= 0 && !E->cmode && E->act == ACT_NONE && M >= 48 && N >= 48;
  if (ring) {
    // tile choice, from a sweep of every GEMM shape of a training step over the four tiles (tools/gemm_sweep.sh): the
    // 64x64 tile with its 4-stage ring wins wherever the reduction is long (the step's contractions are latency-bound:
    // pipeline depth beats tile size); the 2-stage 128x128 tile only pays for short reductions with enough tiles to fill
    // the chip; big skinny weight gradients take a 3-stage rectangular tile (fewer re-reads of the long operand, half
    // the atomic traffic).
    static const char* force = getenv("CENET_RING_TILE");  // "128x64" etc.: measurement aid
    if (force) {
      sscanf(force, "%dx%d", &bm, &bn);
    } else if (E->atomic) {
      bm = bn = 64;
      if ((long)K * nkb >= 4096 && (long)M * N >= 131072) {
        if (N >= 128) bn = 128;
        else bm = 128;
      }
    } else {
      // (tried: a 2-stage 64x64 instance with 32 KB of LDS — five workgroups per CU — for the reductions of one or two
      // K-steps, K <= 128: the step got 1 % slower, K <= 64 and K <= 256 likewise)
      bm = M >= 96 ? 128 : 64;
      bn = N >= 96 ? 128 : 64;
      const long want = K >= 320 ? 400 : 256;
      auto wgs = [&](int a, int b) { return (long)cdiv(M, a) * cdiv(N, b) * nbatch; };
      if (wgs(bm, bn) < want && bm == 128 && bn == 128) (M >= N ? bm : bn) = 64;
      if (wgs(bm, bn) < want && (bm == 128 || bn == 128)) bm = bn = 64;
      static const char* fb = getenv("CENET_RING_TILE_BATCHED");  // measurement aid: per-image (nbatch >= 8) launches only
      if (fb && nbatch >= 8) sscanf(fb, "%dx%d", &bm, &bn);
    }
  }
  const int kstep = ring ? 64 : BK;
  if (E->atomic && !E->cmode && splits > 1) {
    // split-K: every split adds the whole MxN tile with float atomics (~1.3 TB/s chip-wide), and one workgroup's K loop is a
    // serial chain of ~0.9 us (plain) / ~2 us (gathering) iterations, so with s splits
    //     time ~ iters / s * t_iter  +  s * tiles * tile_bytes / 1.3 TB/s
    // which is least at s = sqrt(iters * t_iter / per-split atomic time); never more workgroups than fit on the chip at once
    const long tiles = (long)cdiv(M, bm) * cdiv(N, bn) * nbatch;
    const long iters = (long)nkb * cdiv(K, kstep);
    const double t_iter = ring ? 0.5 : ((B->mode == 1) ? 2.0 : 0.9);
    const double t_atom = (double)tiles * bm * bn * 4.0 / 1.3e6;  // us per split
    long s = (long)(__builtin_sqrt((double)iters * t_iter / t_atom) + 0.5);
    const long slots = (bm * bn >= 128 * 128) ? 512 : (bm * bn >= 128 * 64 ? 768 : 1024);
    if (s * tiles > slots) s = slots / tiles;
    if (s > iters / 2) s = iters / 2;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    static const char* sm = getenv("CENET_RING_SPLIT_MUL");  // measurement aid: scales the split count of the ring kernel
    if (ring && sm) {
      s = (long)(s * atof(sm) + 0.5);
      if (s > iters / 2) s = iters / 2;
      if (s < 1) s = 1;
    }
    splits = (int)s;
    g.splits = splits;
  }
  const bool im = B->mode != 0;
  int rc;
  // one or two K-steps, nothing to pipeline inside a tile: persistent workgroups with the ring over TILES (gemm_ring.h)
  static const bool stream_off = getenv("CENET_GEMM_NO_STREAM") != nullptr;
  const bool rstream = ring && !stream_off && !E->atomic && nkb == 1 && splits <= 1 && !g.ring_unal && (K == 64 || K == 128) &&
                       (long)cdiv(M, 64) * cdiv(N, 64) * nbatch >= 768;
  if (rstream) {
    snprintf(g_last_kernel, sizeof g_last_kernel, "gemm_ring_stream_kernel<%s, %s, %s, %d>", tf(akf), tf(bkf), tf(swap), K / 64);
    rc = akf ? (bkf ? cenet_gemm_launch_ring_kk(g, 0, 0, nbatch, swap, stream) : cenet_gemm_launch_ring_kr(g, 0, 0, nbatch, swap, stream))
             : (bkf ? cenet_gemm_launch_ring_rk(g, 0, 0, nbatch, swap, stream) : cenet_gemm_launch_ring_rr(g, 0, 0, nbatch, swap, stream));
  } else if (ring) {
    snprintf(g_last_kernel, sizeof g_last_kernel, "gemm_ring_kernel<%s, %s, %d, %d, %d, %s>", tf(akf), tf(bkf), bm, bn,
             bm == 128 && bn == 128 ? 2 : (bm == 64 && bn == 64 ? 4 : 3), tf(swap));
    rc = akf ? (bkf ? cenet_gemm_launch_ring_kk(g, bm, bn, nbatch, swap, stream) : cenet_gemm_launch_ring_kr(g, bm, bn, nbatch, swap, stream))
             : (bkf ? cenet_gemm_launch_ring_rk(g, bm, bn, nbatch, swap, stream) : cenet_gemm_launch_ring_rr(g, bm, bn, nbatch, swap, stream));
  } else if (bf) {
    if (E->asum) {
      launch_asum<unsigned short>(A, M, K, nkb, E->asum, stream);
      g.E.asum = nullptr;
    }
    // plain operands with a reduction of at least two 64-steps: K step 64 halves the barriers and the serial
    // load -> LDS -> MFMA round trips of the (mostly latency-bound) mid-size contractions
    // (measured: a gain for the 64x64 / 32x64 tiles that these small launches get, a loss for the 128-wide tiles, whose
    // doubled prefetch registers drop them to one workgroup per CU, and for the split-K weight gradients)
    // with bf16 tensors in HBM every tile but 32x256 has a K-step-64 instance; taken when the chunk a workgroup reduces
    // is at least two 64-steps long
    const long kchunk = ((long)nkb * K) / (splits > 0 ? splits : 1);
    const bool k64 = !im && K >= 128 && kchunk >= 128 && bm <= 64 && bn == 64;  // (the 128-wide K-64 instances spill)
    snprintf(g_last_kernel, sizeof g_last_kernel, "gemm_kernel<unsigned short, %d, %d, %s, %s, %d>", bm, bn, tf(im), tf(swap),
             k64 ? 64 : 32);
    rc = im ? cenet_gemm_launch_bf16_im2col(g, bm, bn, nbatch, swap, stream)
            : (k64 ? cenet_gemm_launch_bf16_plain_k64(g, bm, bn, nbatch, swap, stream)
                   : cenet_gemm_launch_bf16_plain(g, bm, bn, nbatch, swap, stream));
  }
  else {
    if (E->asum) {
      launch_asum<float>(A, M, K, nkb, E->asum, stream);
      g.E.asum = nullptr;
    }
    snprintf(g_last_kernel, sizeof g_last_kernel, "gemm_kernel<float, %d, %d, %s, %s, 32>", bm, bn, tf(im), tf(swap));
    rc = im ? cenet_gemm_launch_f32_im2col(g, bm, bn, nbatch, swap, stream) : cenet_gemm_launch_f32_plain(g, bm, bn, nbatch, swap, stream);
  }
  if (rc != CENET_OK) return rc;
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_gemm_f32(const cenet_mat_t* A, const cenet_mat_t* B, const cenet_epi_t* E, int M, int N, int K,
                              int nbatch, int nb_inner, int nkb, int splits, hipStream_t stream) {
  return gemm_dispatch(A, B, E, M, N, K, nbatch, nb_inner, nkb, splits, 4, stream);
}
extern "C" int cenet_gemm_bf16(const cenet_mat_t* A, const cenet_mat_t* B, const cenet_epi_t* E, int M, int N, int K,
                               int nbatch, int nb_inner, int nkb, int splits, hipStream_t stream) {
  return gemm_dispatch(A, B, E, M, N, K, nbatch, nb_inner, nkb, splits, 2, stream);
}
