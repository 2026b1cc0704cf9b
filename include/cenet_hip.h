/* cenet_hip.h — C ABI of libcenet_hip.so, the MI355X (gfx950) kernel library behind cenet_amd.networks.CENet.
 *
 * The reference (xmindflow/cenet) is pure PyTorch: its "FFI" for this path is the ATen operator set its
 * nn.Modules call.  Each entry point below replaces the ATen op group cited next to it (file:line relative to
 * /root/reference/src/); cenet_amd/ops.py binds them with ctypes and wraps them in torch.autograd.Functions,
 * and cenet_amd/networks mirrors the reference's nn.Module tree on top.
 *
 * Conventions: raw device pointers; no allocation, no ownership transfer and no host synchronisation inside;
 * workspaces are passed in; every call takes the HIP stream to launch on and is re-entrant (called from the
 * autograd worker thread); return 0 on success, non-zero error code otherwise (the Python wrapper raises).
 * Tensors are fp32 ("_f32" entry points: the parity mode) or, for activations, bf16 ("_bf16" twins at the end of this header:
 * the throughput mode).  "NCHW" tensors are addressed as ptr[b*sb + c*HW + y*W + x] so that a
 * channel slice of a larger tensor is (ptr + lo*HW, sb = Ctotal*HW).  "TOK" tensors are [B, N, C] contiguous.
 * Functions whose name ends in _acc ADD into their gradient outputs (the flat gradient arena is zeroed once
 * per step), using float atomics where several workgroups share a destination.
 */
#ifndef CENET_HIP_H
#define CENET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef CENET_HOSTSIM_BUILD
typedef struct ihipStream_t* cenet_stream_t;
#else
typedef void* cenet_stream_t;
#endif

/* ---- GEMM / implicit-GEMM convolution (gemm.hip) ------------------------------------------------------- */
/* Logical matrix operand. mode 0: elem(r,c) = ptr[b*sb + kb*skb + r*sr + c*sc].
 * mode 1: on-the-fly im2col view of an image tensor: with patch index e=(ci,ky,kx) and pixel index p=(py,px)
 *   (e,p) = patch_is_row ? (r,c) : (c,r);
 *   !transposed: iy = py*stride - pad + ky*dil           (conv forward / weight-gradient gather)
 *    transposed: iy = (py + pad - ky*dil)/stride if divisible   (data-gradient gather)
 *   elem = in-range ? ptr[b*sb + kb*skb + ci*sci + iy*sy + ix*sx] : 0.
 * kfast: 1 if consecutive k are adjacent in memory (picks the coalesced staging pattern). */
typedef struct {
  const void* ptr;           /* fp32 or bf16 elements (the `dtype` argument of cenet_gemm); strides count ELEMENTS */
  long sb, sb2, skb, sr, sc; /* batch z -> (z / nb_inner, z % nb_inner): offset = (z/nb_inner)*sb + (z%nb_inner)*sb2 */
  long sk_outer;             /* mode 0 with kinner > 0: k -> (k / kinner)*sk_outer + (k % kinner)*(sc or sr) */
  int kinner;
  int mode, kfast;
  int patch_is_row, transposed;
  int KH, KW, Pw, Hs, Ws, stride, pad, dil;
  long sci, sy, sx;
} cenet_mat_t;

/* Epilogue: v = alpha*acc; atomic ? C += v : C = bscale[b]*act(v + bias) + R.  C and R have the operands' element type,
 * except that an ATOMIC epilogue (split-K, scatter-add) always adds into an fp32 C; bias / bscale are always fp32. */
typedef struct {
  void* C;
  long scb, scb2, scr, scc;
  const float* bias;
  int bias_on_row;
  int act;
  float slope;
  const float* bscale;
  const void* R;
  long srb, srb2, srr, src;
  int atomic;
  float alpha;
  /* cmode 1: col2im scatter — row = patch index (ci,ky,kx), col = output pixel (py,px);
   * C address = b*scb + ci*csci + (py*cstride - cpad + ky)*csy + (px*cstride - cpad + kx)*csx, skipped when out of
   * [0,cHs)x[0,cWs). Used for the data-gradient of strided convolutions (atomic when patches overlap). */
  int cmode, cKH, cKW, cPw, cHs, cWs, cstride, cpad;
  long csci, csy, csx;
  /* > 0: bscale is indexed by (row / bscale_rows) instead of by batch -- a [B, n, K] activation run as ONE flat GEMM of
   * B*n rows with the per-sample DropPath scale of pvtv2.py:117-118 still applied per sample (bscale_rows = n). */
  int bscale_rows;
  /* atomic (weight-gradient) contractions only: asum[m] += sum_k A[m, k] over the whole reduction (all k-batches) -- the
   * bias gradient that goes with the weight gradient dW = dY^T X (A = dY^T), folded into the pass that already reads dY
   * instead of a second sweep (aten::sum of the linear / 1x1-convolution backward).  fp32, may be NULL. */
  float* asum;
} cenet_epi_t;

/* Replaces aten::addmm/mm/bmm/convolution(+_backward) — pvtv2.py:41,45,90,98,106,164; cfam.py:149,158,299,302;
 * nlb.py:106-115,142; blocks.py:178,211,320; dseb.py:164; unet.py:156-197; multihead_diffattn.py:79-81,126. */
int cenet_gemm_f32(const cenet_mat_t* A, const cenet_mat_t* B, const cenet_epi_t* E, int M, int N, int K,
                   int nbatch, int nb_inner, int nkb, int splits, cenet_stream_t stream);
/* The same contraction on bf16 tensors (throughput mode): A, B (and C / R unless E->atomic) hold bf16 elements, products
 * run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; weights are passed as their bf16 shadow copy (cenet_sgd_step_shadow_f32
 * / cenet_cast_f32_to_bf16). */
int cenet_gemm_bf16(const cenet_mat_t* A, const cenet_mat_t* B, const cenet_epi_t* E, int M, int N, int K,
                    int nbatch, int nb_inner, int nkb, int splits, cenet_stream_t stream);
/* Measurement aid: the kernel instance the last cenet_gemm_* call on this thread launched, spelled as rocprofv3 prints it
 * ("gemm_ring_kernel<false, false, 64, 64, 4, false>").  bench.py groups its live per-launch timings by this name. */
const char* cenet_gemm_last_kernel(void);

/* GROUPED weight gradients (gemm_group.hip): the parameter gradients that autograd computes one aten::mm / aten::convolution_
 * backward at a time for every Linear (pvtv2.py:41,45,60-63,90,98,106; multihead_diffattn.py:57-60) and 1x1 convolution
 * (cfam.py:128-144,283-288; nlb.py:66-92; blocks.py:303-311; dseb.py:111) of a backward segment, as ONE launch per <= 56
 * problems.  Problem i:  C_i[m, n] += sum over kb < nkb, k < K of A_i(m, k) * B_i(k, n)   and   asum_i[m] += sum of A_i(m, k)
 * (the bias gradient; may be NULL), with  A(m, k) = akf ? A[kb*skbA + m*lda + k] : A[kb*skbA + k*lda + m],
 * B(k, n) = bkf ? B[kb*skbB + n*ldb + k] : B[kb*skbB + k*ldb + n];  A, B bf16 (dY and the layer input), C fp32 row-major
 * [M, N] (a slot of the gradient arena), strides in elements.  akf == bkf is required (token-major Linear: both 0; NCHW 1x1
 * convolution: both 1), else CENET_EUNSUPPORTED.  Long reductions are cut into K slices whose partial tiles go to `ws`
 * (cenet_wgrad_group_ws_floats(p, n) floats, may be 0) and are summed in slice order by a second launch: no float atomics
 * unless two problems of a call share a C, results independent of scheduling. */
typedef struct {
  const void* A;
  const void* B;
  float* C;
  float* asum;
  long lda, ldb, skbA, skbB;
  int M, N, K, nkb;
  int akf, bkf;
} cenet_wgrad_prob_t;
long cenet_wgrad_group_ws_floats(const cenet_wgrad_prob_t* p, int n);
int cenet_wgrad_group_bf16(const cenet_wgrad_prob_t* p, int n, float* ws, long ws_floats, cenet_stream_t stream);
/* Measurement aid: the same reduction issued in two calls — phase 1 launches the K-slice kernels only, phase 2 the fold
 * kernels only (phase 0 = cenet_wgrad_group_bf16) — so that bench.py can bracket the two kernel symbols with HIP events. */
/* the partition cenet_wgrad_group_bf16 makes of its problems: launch[i] = index of the grouped launch problem i goes to (issue
 * order), bm / bn / ns [i] = that launch's tile and ring depth (gemm_group_kernel<akf, akf, bm, bn, ns>) */
int cenet_wgrad_group_plan(const cenet_wgrad_prob_t* p, int n, int* launch, int* bm, int* bn, int* ns);
int cenet_wgrad_group_phase_bf16(const cenet_wgrad_prob_t* p, int n, float* ws, long ws_floats, int phase, cenet_stream_t stream);

/* Direct ("LDS halo") stride-1 same-padded convolution on bf16 tensors (throughput mode, conv_direct.hip): replaces
 * aten::convolution(+ data-gradient) for out.py:41-49,59 (5x5 32->32, 3x3 64->64, 3x3 64->32 and their dgrads).
 * x, y: bf16; w: the fp32 master weight.  dgrad = 1: x is dY [B,Cin,H,W], w is the ORIGINAL forward weight [Cin,Cout,k,k],
 * y is dX [B,Cout,H,W]. */
int cenet_conv_direct_supported(int Cin, int Cout, int k, int stride, int pad);
int cenet_conv_direct_bf16(const unsigned short* x, const float* w, unsigned short* y, int B, int Cin, int Cout, int H, int W,
                           int k, int dgrad, cenet_stream_t stream);
/* Weight gradient of the same convolutions in the same mode (replaces the implicit-GEMM wgrad of unet.py:156-197,
 * blocks.py:211, out.py:41-49 for 5x5 32->32 and 3x3 64->64 / 64->32): dw_acc[Cout,Cin,k,k] += dY (*) X.
 * ws: cenet_conv_wgrad_direct_ws_floats(Cin, Cout, k) floats of scratch (per-workgroup partial sums). */
/* One-channel input (the network image) into Cout <= 32 channels, stride 1, pad k/2, k in {1, 3, 5}, bf16 tensors
 * (conv_c1.hip): the residual block of the output head that reads x (out.py:41-44; unet.py:156-197 conv1 / conv3).
 * Forward y = w (*) x and the weight gradient dw_acc += dy (*) x; the image itself needs no gradient. */
int cenet_conv_c1_supported(int Cin, int Cout, int k, int stride, int pad);
/* `groups` independent square bias-free 1x1 convolutions over G <= 40 channels each (conv_c1.hip, bf16): y[b, j*G + o, p] =
 * sum_i W[j][o][i] x[b, j*G + i, p]  (transpose != 0: W[j][i][o], the data gradient).  The pointwise convs of the dilated
 * SepConvBN branches and the pooled-branch conv of cfam.py:208-219 at channel counts where a GEMM tile is mostly padding. */
/* 1x1 convolution from 64 channels to Cout <= 16 channels with bias, bf16 tensors (conv_c1.hip): the last layer of the
 * segmentation head (unet.py:200-217 UnetOutBlock, out.py:49).  x [B,64,HW], W bf16 [Cout,64] (shadow of the fp32 weight),
 * y / dy [B,Cout,HW]; the weight-gradient entry adds into fp32 dW [Cout,64] and dbias [Cout] (Cout in {2, 4, 9}). */
int cenet_pw_fewout_supported(int Cin, int Cout);
int cenet_pw_fewout_wgrad_supported(int Cin, int Cout);
int cenet_pw_fewout_fwd_bf16(const unsigned short* x, const unsigned short* W, const float* bias, unsigned short* y, int B, int Cin,
    int Cout, long HW, cenet_stream_t stream);
int cenet_pw_fewout_dgrad_bf16(const unsigned short* dy, const unsigned short* W, unsigned short* dx, int B, int Cin, int Cout,
    long HW, cenet_stream_t stream);
int cenet_pw_fewout_wgrad_bf16(const unsigned short* x, const unsigned short* dy, float* dW_acc, float* dbias_acc, int B, int Cin,
    int Cout, long HW, cenet_stream_t stream);
int cenet_pw_small_supported(int G);
int cenet_pw_small_bf16(const unsigned short* x, const unsigned short* W, unsigned short* y, int B, int groups, int G, long HW,
    int transpose, cenet_stream_t stream);
int cenet_conv_c1_fwd_bf16(const unsigned short* x, const float* w, unsigned short* y, int B, int Cout, int H, int W, int k,
    cenet_stream_t stream);
int cenet_conv_c1_wgrad_bf16(const unsigned short* x, const unsigned short* dy, float* dw_acc, int B, int Cout, int H, int W,
    int k, cenet_stream_t stream);
int cenet_conv_wgrad_direct_supported(int Cin, int Cout, int k, int stride, int pad);
long cenet_conv_wgrad_direct_ws_floats(int Cin, int Cout, int k);
int cenet_conv_wgrad_direct_bf16(const unsigned short* x, const unsigned short* dy, float* dw_acc, float* ws, int B, int Cin,
                                 int Cout, int H, int W, int k, cenet_stream_t stream);

/* ---- attention (attn.hip) -------------------------------------------------------------------------------- */
/* Element (b,h,i,d) of Q = q[b*qsb + h*qsh + i*qsi + d*qsd]; same for K (Nk rows), V (head h / v_head_div,
 * width Dv) and O / dO (strides os*).  lse, delta: [B,H,Nq] fp32.  q, k, v, o, dout, dq, dk, dv are fp32 (`_f32` entry points)
 * or bf16 (`_bf16`).  When v_head_div > 1 the backward ADDS into dv with atomics (caller zero-fills dv); with bf16 tensors
 * atomic accumulation needs `dkv_f32`: dk and dv then point to fp32 buffers laid out like k and v. */
typedef struct {
  const void *q, *k, *v;
  void* o;
  float* lse;
  const void* dout;
  void *dq, *dk, *dv;
  float* delta;
  long qsb, qsh, qsi, qsd, ksb, ksh, ksi, ksd, vsb, vsh, vsi, vsd, osb, osh, osi, osd;
  int B, H, Nq, Nk, D, Dv, v_head_div;
  float scale;
  int dkv_zeroed; /* backward only: caller guarantees dk / dv are zero-filled, so the library may slice the query range
                     over workgroups and accumulate dK / dV atomically (few keys under many queries) */
  int dkv_f32;    /* bf16 entry points only: dk / dv are fp32 accumulators */
  int finite_scores; /* cenet_flash_attn_fwd_f32 only (multihead_diffattn.py:95-106): q is scaled BEFORE the product (`q *= self.scaling`)
                        and the scores pass through torch.nan_to_num (NaN -> 0, +-inf -> +-FLT_MAX) before the softmax, so a q.k
                        product that overflows fp32 gives the reference's finite output.  The bf16 entry points and the backward
                        ignore it (overflowing scores are outside the supported domain there: INTEGRATION.md) */
} cenet_attn_t;
/* Replaces q@k^T -> softmax -> @v (pvtv2.py:101-105; nlb.py:117-138; multihead_diffattn.py:96-116) and backward.
 * Supported head dims: D<=64 with Dv<=128 (cenet_flash_attn_supported). */
int cenet_flash_attn_supported(int D, int Dv);
int cenet_flash_attn_fwd_f32(const cenet_attn_t* p, cenet_stream_t stream);
int cenet_flash_attn_bwd_f32(const cenet_attn_t* p, cenet_stream_t stream);
int cenet_flash_attn_fwd_bf16(const cenet_attn_t* p, cenet_stream_t stream);
int cenet_flash_attn_bwd_bf16(const cenet_attn_t* p, cenet_stream_t stream);
/* Differential attention of the DSEB skip blocks on bf16 tensors (attn_diff.hip; multihead_diffattn.py:83-109): softmax head
 * 2h+s (s = 0, 1) attends with q_{2h+s}, k_{2h+s} over the SHARED value head h; U[b, 2h+s] = softmax(q k^T * scale) v_h.
 * q, k [B, N, 2H*hd], v [B, N, H*2hd] (token-major, as the projections produce them), U, dU [B, 2H, N, 2hd], lse [B, 2H, N]
 * fp32; dq, dk, dv laid out like q, k, v.  ws: cenet_diffattn_heads_ws_bytes(B, H, N) bytes of scratch the backward uses to
 * pass the softmax statistics between its two kernels.  hd in {8, 16, 32}; all pointers 16-byte aligned. */
typedef struct {
  const void *q, *k, *v;
  void* U;
  float* lse;
  const void* dU;
  void *dq, *dk, *dv;
  void* ws;
  int B, H, N, hd;
  float scale;
  int batch_mul; /* 0 / 1: dense tensors.  m > 1: q / k / v (and dq / dk / dv) of image b start m rows-blocks apart, i.e. at
                  * element b * m * N * row_len of their pointers — q | k | v stacked per image as [B, 3, N, C] (the Non-local
                  * block's merged theta / phi / g projection, nlb.py:117-119, m = 3) are read and written in place */
} cenet_diffattn_t;
int cenet_diffattn_heads_supported(int hd, int N);
long cenet_diffattn_heads_ws_bytes(int B, int H, int N);
int cenet_diffattn_heads_fwd_bf16(const cenet_diffattn_t* p, cenet_stream_t stream);
int cenet_diffattn_heads_bwd_bf16(const cenet_diffattn_t* p, cenet_stream_t stream);
/* Plain self-attention with head dimension 64 or 128 on bf16 tensors through the same tiles (one softmax over both hd/2-column
 * halves): q, k, v [B, N, H*hd] token-major, U / dU [B, H, N, hd], lse [B, H, N], hd = 64 | 128, ws of cenet_attn64_ws_bytes.
 * Replaces the score / softmax / value products of the Non-local block at the 56x56 (C = 64) and 28x28 (C = 128) decoder
 * levels (nlb.py:117-138). */
long cenet_attn64_ws_bytes(int B, int H, int N);
int cenet_attn64_fwd_bf16(const cenet_diffattn_t* p, cenet_stream_t stream);
int cenet_attn64_bwd_bf16(const cenet_diffattn_t* p, cenet_stream_t stream);
/* Spatial-reduction attention backward (pvtv2.py:88-109) on bf16 tensors in ONE kernel: head dimension 64, Nk <= 64 keys
 * resident in LDS.  q, o, dout, dq [B, Nq, 64 H]; kv [B, Nk, 128 H] (k | v); lse [B, H, Nq] as the forward kernel left it;
 * dkv [B, Nk, 128 H] fp32, ZERO-FILLED by the caller (workgroups add their partial sums atomically). */
int cenet_sra_attn_bwd_supported(int hd, int Nk);
/* 64 < Nk <= 256 keys of head dimension 64 (pvtv2.py:92-105 at 512x512 inputs): cenet_sra_attn_bwd_bf16 then runs as two
 * launches — dQ with all keys resident, dK / dV per 64-key block (the saved lse makes key blocks independent). */
int cenet_sra_attn_bwd_blocks_supported(int hd, int Nk);
/* The forward of the same problem class (head dimension 64, Nk <= 64, cenet_sra_attn_bwd_supported) with the keys / values
 * resident: o [B, Nq, 64 H], lse [B, H, Nq] (natural log, as the backward entries expect). */
int cenet_sra_attn_fwd_bf16(const unsigned short* q, const unsigned short* kv, unsigned short* o, float* lse, int B, int H, int Nq,
    int Nk, float scale, cenet_stream_t stream);
int cenet_sra_attn_bwd_bf16(const unsigned short* q, const unsigned short* kv, const unsigned short* o,
    const unsigned short* dout, const float* lse, unsigned short* dq, float* dkv, int B, int H, int Nq, int Nk, float scale,
    cenet_stream_t stream);
/* The same kernel where its launch puts all queries of a (batch, head) in ONE workgroup (cenet_sra_attn_bwd_direct_supported:
 * the 14x14 and 7x7 stages at B = 32): dkv [B, Nk, 128 H] is written as bf16 directly — no zero fill, no atomics, no cast. */
int cenet_sra_attn_bwd_direct_supported(int B, int H, int Nq, int Nk);
int cenet_sra_attn_bwd_direct_bf16(const unsigned short* q, const unsigned short* kv, const unsigned short* o,
    const unsigned short* dout, const float* lse, unsigned short* dq, unsigned short* dkv, int B, int H, int Nq, int Nk,
    float scale, cenet_stream_t stream);
/* Row softmax for the materialised path (head dims > 128): aten::_softmax(+_backward_data).  Scores x and score gradients
 * dy are fp32 in both forms; the probabilities y and dx have the storage type of the entry point. */
int cenet_softmax_rows_fwd_f32(const float* x, float* y, long rows, int n, cenet_stream_t stream);
int cenet_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, long rows, int n, cenet_stream_t stream);
int cenet_softmax_rows_fwd_bf16(const float* x, unsigned short* y, long rows, int n, cenet_stream_t stream);
int cenet_softmax_rows_bwd_bf16(const unsigned short* y, const float* dy, unsigned short* dx, long rows, int n,
                                cenet_stream_t stream);

/* ---- normalisation (norm.hip) ----------------------------------------------------------------------------- */
/* aten::native_layer_norm(+_backward) — pvtv2.py:117,124,166,69,221-245. x,y [rows,C]. */
int cenet_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                            int rows, int C, float eps, cenet_stream_t stream);
int cenet_layernorm_bwd_acc_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                float* dx, float* dgamma_acc, float* dbeta_acc, int rows, int C, cenet_stream_t stream);
/* the same with dx = LN-backward(dy) + dx_add: the gradient of a residual connection that by-passed the LayerNorm
 * (x + f(LN(x)), pvtv2.py:141-142) is folded in here instead of a separate aten::add; dx_add may be NULL. */
int cenet_layernorm_bwd_add_acc_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                    const float* dx_add, float* dx, float* dgamma_acc, float* dbeta_acc, int rows, int C,
                                    cenet_stream_t stream);
/* aten::native_batch_norm(+_backward), training mode — cfam.py:22-32,92,250; blocks.py:151,161,212,307; nlb.py:81;
 * unet.py:175-197.  ws: CENET_BN_WS_FLOATS(C) floats of scratch (per-split partial sums, no zero-fill needed).
 * stats writes batch mean / biased var and updates the running
 * buffers (momentum, unbiased var) and num_batches_tracked when given. */
#define CENET_BN_WS_FLOATS(C) (2L * (C) * 256)
int cenet_bn_stats_f32(const float* x, long sb, int B, int C, int HW, float* ws, float* mean, float* var,
                       float* running_mean, float* running_var, float momentum, long* num_batches_tracked,
                       cenet_stream_t stream);
int cenet_bn_apply_f32(const float* x, long sxb, float* y, long syb, const float* mean, const float* var, float eps,
                       const float* gamma, const float* beta, int act, float slope, int B, int C, int HW,
                       cenet_stream_t stream);
/* Train-mode BatchNorm forward in one call (two launches): statistics + normalisation (+ activation); writes y, the batch
 * mean / var the backward pass needs, and the running statistics (nn.BatchNorm2d.forward in training mode). */
int cenet_bn_train_fwd_f32(const float* x, long sxb, float* y, long syb, float* ws, float* mean, float* var,
    float* running_mean, float* running_var, float momentum, long* num_batches_tracked, float eps, const float* gamma,
    const float* beta, int act, float slope, int B, int C, int HW, int nbt_count, cenet_stream_t stream);
/* nbt_count (1 <= nbt_count <= C): the launch covers the channels of that many nn.BatchNorm2d modules whose parameters and
 * buffers lie back to back (the three dilated SepConvBN branches of cfam.py:208-212 in one launch); num_batches_tracked[0 ..
 * nbt_count) are each incremented.  1 for a single module. */
int cenet_bn_bwd_acc_f32(const float* dy, long sgb, const float* x, long sxb, float* dx, long sdb, const float* mean,
                         const float* var, float eps, const float* gamma, const float* beta, int act, float slope, int B,
                         int C, int HW, float* ws, float* dgamma_acc, float* dbeta_acc, cenet_stream_t stream);
/* ... + dx_add (batch stride sab, may be NULL): dx = BatchNormBackward(dy) + dx_add — the gradient of the residual connection
 * around the normalised branch (cfam.py:365-374) added by the kernel that writes dx (round 4) */
int cenet_bn_bwd_add_acc_f32(const float* dy, long sgb, const float* x, long sxb, float* dx, long sdb, const float* dx_add,
                             long sab, const float* mean, const float* var, float eps, const float* gamma, const float* beta,
                             int act, float slope, int B, int C, int HW, float* ws, float* dgamma_acc, float* dbeta_acc,
                             cenet_stream_t stream);

/* BatchNorm1d on a [B, C] fp32 matrix with 2 <= B <= 64 (the CCU gate, cfam.py:251-264: one value per image and channel), training
 * mode, forward and backward in ONE launch each (a thread owns a channel; round 4).  Forward: zn, batch mean / biased variance, running
 * statistics (unbiased variance) and the batch counter (NULL: not wanted); backward: dz written, dgamma / dbeta ADDED into. */
int cenet_bn1d_supported(int B);
int cenet_bn1d_train_fwd_f32(const float* z, float* zn, float* mean, float* var, float* running_mean, float* running_var,
                             float momentum, long* num_batches_tracked, float eps, const float* gamma, const float* beta, int B,
                             int C, cenet_stream_t stream);
int cenet_bn1d_bwd_acc_f32(const float* dy, const float* z, float* dz, const float* mean, const float* var, float eps,
                           const float* gamma, float* dgamma_acc, float* dbeta_acc, int B, int C, cenet_stream_t stream);

/* ---- depthwise 3x3 conv (dwconv.hip) — aten::convolution(groups=C)(+_backward) ------------------------------ */
/* pvtv2.py:359-370 (token layout), cfam.py:132-140, blocks.py:142-150,305 (NCHW). y = conv(x)+bias; a = act(y) if a. */
int cenet_dwconv3x3_nchw_f32(const float* x, long sxb, const float* w, const float* bias, float* y, long syb, float* a,
                             long sab, int B, int C, int H, int W, int dil, int flip, int act, float slope,
                             cenet_stream_t stream);
int cenet_dwconv3x3_tok_f32(const float* x, const float* w, const float* bias, float* y, float* a, int B, int C, int H, int W,
                            int flip, int act, float slope, cenet_stream_t stream);
int cenet_dwconv3x3_wgrad_nchw_acc_f32(const float* x, long sxb, const float* dy, long sgb, float* dw_acc, float* dbias_acc,
                                       int B, int C, int H, int W, int dil, cenet_stream_t stream);
int cenet_dwconv3x3_wgrad_tok_acc_f32(const float* x, const float* dy, float* dw_acc, float* dbias_acc, int B, int C, int H,
                                      int W, cenet_stream_t stream);

/* LayerNorm backward on bf16 rows with DEFERRED affine gradients (round 4): instead of 2 C float atomics per workgroup, each
 * workgroup writes one row of part[cenet_layernorm_bwd_part_rows(rows, C)][2 C]; cenet_ln_fold_group adds the column sums of up to
 * any number of such buffers (all LayerNorms of a backward segment) into their dgamma / dbeta with one launch per 48. */
int cenet_layernorm_bwd_part_rows(int rows, int C);
int cenet_layernorm_bwd_add_part_bf16(const unsigned short* dy, const unsigned short* x, const float* gamma, const float* mean,
                                      const float* rstd, const unsigned short* dx_add, unsigned short* dx, float* part, int rows,
                                      int C, cenet_stream_t stream);
/* the same, also writing dxs = bscale[row / rows_per_sample] * dx: the gradient the DropPath-scaled branch upstream wants
 * (pvtv2.py:141-149) comes out of the kernel that produces dx instead of a scale pass (both NULL: plain) */
int cenet_layernorm_bwd_add_part_scaled_bf16(const unsigned short* dy, const unsigned short* x, const float* gamma,
                                             const float* mean, const float* rstd, const unsigned short* dx_add,
                                             unsigned short* dx, float* part, const float* bscale, int rows_per_sample,
                                             unsigned short* dxs, int rows, int C, cenet_stream_t stream);
int cenet_ln_fold_group(const void* const* part, float* const* dgamma_acc, float* const* dbeta_acc, const int* nrows, const int* C,
                        int n, cenet_stream_t stream);

/* ---- fused PVTv2 MLP half (pvt_mlp.hip) — pvtv2.py:40-47,145-149,364-370 on bf16 tokens [B, H*W, C] --------------------- */
/* y = x + s_b (fc2(GELU(DW3x3(fc1(LayerNorm(x))) + bd)) + b2) in ONE launch: replaces aten::native_layer_norm + addmm +
 * convolution(groups) + gelu + addmm + mul + add.  w1 [HD, C] / w2 [C, HD] are bf16, everything else fp32; bscale [B] (DropPath
 * keep / keep_prob per sample) may be NULL.  xn_out / mean_out / rstd_out / h_out / a_out (all or none): the LayerNorm output
 * and statistics, the fc1 output and s_b * the GELU output, stored for the backward kernels.  CENET_EUNSUPPORTED unless
 * cenet_pvt_mlp_supported(C, HD, H, W) (C in {64, 128}, HD % 64 == 0, W % 14 == 0, H % 7 == 0 or H % 8 == 0). */
int cenet_pvt_mlp_supported(int C, int HD, int H, int W);
int cenet_pvt_mlp_fwd_bf16(const unsigned short* x, const float* ln_g, const float* ln_b, float eps, const unsigned short* w1,
                           const float* b1, const float* wd, const float* bd, const unsigned short* w2, const float* b2,
                           const float* bscale, unsigned short* y, unsigned short* xn_out, float* mean_out, float* rstd_out,
                           unsigned short* h_out, unsigned short* a_out, int B, int H, int W, int C, int HD, cenet_stream_t stream);

/* backward of cenet_pvt_mlp_fwd_bf16 from its saved tensors in two launches (+ a fold of the LayerNorm affine gradients): replaces
 * mul (DropPath scale) + mm (fc2 data gradient) + gelu_backward + convolution_backward(groups) + mm (fc1 data gradient) +
 * native_layer_norm_backward + add (residual).  gu, dh [B, H*W, HD]: the gradient of the depthwise conv's output (scratch) and of
 * fc1's output (operand of the fc1 weight gradient dW1 = dh^T xn; the fc2 weight gradient is dW2 = g^T a with the forward's saved a,
 * which already carries s_b); dx = g + LayerNormBackward(dh . W1); dwd / dbd / dln_g / dln_b and db2 (fc2 bias gradient =
 * column sums of s_b g; may be NULL) are ADDED into.
 * ws: cenet_pvt_mlp_bwd_ws_floats(B, H, W, C) floats. */
long cenet_pvt_mlp_bwd_ws_floats(int B, int H, int W, int C);
int cenet_pvt_mlp_bwd_bf16(const unsigned short* g, const float* bscale, const unsigned short* w1, const unsigned short* w2,
                           const float* wd, const float* bd, const unsigned short* h, const unsigned short* x, const float* ln_g,
                           const float* mean, const float* rstd, unsigned short* gu, unsigned short* dh, unsigned short* dx,
                           const float* up_scale, unsigned short* dxs, float* dwd_acc, float* dbd_acc, float* dln_g_acc,
                           float* dln_b_acc, float* db2_acc, float* ws, int B, int H, int W, int C, int HD, cenet_stream_t stream);

/* ---- tail of the output head's image branch, fused (res_tail.hip, round 4) — out.py:60,70 over unet.py:201-214 on bf16 maps -------
 * out = w[c] * MaxPool2d(2,2)(LeakyReLU(BN2(x2) + BN3(x3))) with x2 = conv2's output, x3 = the 1x1 shortcut conv's output
 * ([B, C, H, W], H even, W % 8 == 0) and the BatchNorms' batch statistics given (cenet_bn_stats_*): replaces bn_apply x 2 + add +
 * LeakyReLU + max_pool2d + mul forward (one launch; the pre-pool activation is never stored) and max_pool2d_backward +
 * leaky_relu_backward + native_batch_norm_backward x 2 backward (two launches; the activation is recomputed from x2, x3).
 * dgamma / dbeta of both norms and dw (gradient of w) are ADDED into (any may be NULL); ws: cenet_res_tail_bwd_ws_floats(C). */
int cenet_res_tail_supported(int H, int W);
int cenet_res_tail_fwd_bf16(const unsigned short* x2, const unsigned short* x3, const float* mean2, const float* var2,
                            const float* gamma2, const float* beta2, float eps2, const float* mean3, const float* var3,
                            const float* gamma3, const float* beta3, float eps3, const float* w, float slope, unsigned short* out,
                            int B, int C, int H, int W, cenet_stream_t stream);
long cenet_res_tail_bwd_ws_floats(int C);
int cenet_res_tail_bwd_bf16(const unsigned short* g, const unsigned short* x2, const unsigned short* x3, const float* mean2,
                            const float* var2, const float* gamma2, const float* beta2, float eps2, const float* mean3,
                            const float* var3, const float* gamma3, const float* beta3, float eps3, const float* w, float slope,
                            unsigned short* dx2, unsigned short* dx3, float* dgamma2_acc, float* dbeta2_acc, float* dgamma3_acc,
                            float* dbeta3_acc, float* dw_acc, float* ws, int B, int C, int H, int W, cenet_stream_t stream);

/* The same with the shortcut branch NOT materialised (one-channel network input: unet.py conv3 is a 1x1 conv 1 -> C, x3 = w3[c] * img):
 * BN3(x3) is an affine map of the image whose coefficients follow from the image's batch mean / variance (img_mean, img_var: one float
 * each, cenet_bn_stats_* of the [B, 1, H, W] image).  Replaces, besides the above, the shortcut conv, its BatchNorm statistics pass and
 * its weight-gradient kernel.  The forward kernel updates BatchNorm3's running statistics / batch counter (rmean3, rvar3, nbt3, momentum
 * mom3; NULL: not wanted); the backward ADDS dw3 = the shortcut weight's gradient (it reaches the output through eps only). */
int cenet_res_tail_img_fwd_bf16(const unsigned short* x2, const unsigned short* img, const float* mean2, const float* var2,
                                const float* gamma2, const float* beta2, float eps2, const float* img_mean, const float* img_var,
                                const float* w3, const float* gamma3, const float* beta3, float eps3, float* rmean3, float* rvar3,
                                long* nbt3, float mom3, const float* w, float slope, unsigned short* out, int B, int C, int H, int W,
                                cenet_stream_t stream);
int cenet_res_tail_img_bwd_bf16(const unsigned short* g, const unsigned short* x2, const unsigned short* img, const float* mean2,
                                const float* var2, const float* gamma2, const float* beta2, float eps2, const float* img_mean,
                                const float* img_var, const float* w3, const float* gamma3, const float* beta3, float eps3,
                                const float* w, float slope, unsigned short* dx2, float* dgamma2_acc, float* dbeta2_acc,
                                float* dgamma3_acc, float* dbeta3_acc, float* dw3_acc, float* dw_acc, float* ws, int B, int C, int H,
                                int W, cenet_stream_t stream);

/* n <= 4 bias-free, activation-free depthwise 3x3 convs (flip = 1: their data gradients) / weight gradients of bf16 NCHW channel slices
 * in ONE launch — the three dilated SepConvBN branches of a CFAM block (cfam.py:208-212, blocks.py:142-150).  Branch i: x[i] (batch
 * stride sxb[i]) -> y[i] (syb[i]), C[i] channels, dilation dil[i], all on H x W maps of B images.  CENET_EUNSUPPORTED unless every
 * branch takes the plane-in-LDS path (H W % 4 == 0 and the padded plane fits): the caller then launches the branches one by one. */
int cenet_dwconv3x3_nchw_multi_bf16(const unsigned short* const* x, const long* sxb, const float* const* w, unsigned short* const* y,
                                    const long* syb, const int* C, const int* dil, int n, int B, int H, int W, int flip,
                                    cenet_stream_t stream);
int cenet_dwconv3x3_wgrad_nchw_multi_bf16(const unsigned short* const* x, const long* sxb, const unsigned short* const* dy,
                                          const long* sgb, float* const* dw_acc, const int* C, const int* dil, int n, int B, int H,
                                          int W, cenet_stream_t stream);

/* ---- resampling (resample.hip) ---------------------------------------------------------------------------- */
/* aten::upsample_bilinear2d(+_backward) — dseb.py:67-68; cfam.py:217,232; blocks.py:210; out.py:74 */
int cenet_bilinear_fwd_f32(const float* x, long sxb, float* y, long syb, int B, int C, int Hi, int Wi, int Ho, int Wo,
                           float scale_h, float scale_w, int align_corners, cenet_stream_t stream);
int cenet_bilinear_bwd_f32(const float* dy, long sgb, float* dx, long sdb, int B, int C, int Hi, int Wi, int Ho, int Wo,
                           float scale_h, float scale_w, int align_corners, cenet_stream_t stream);
/* dx = backward(dy) + dx_add (laid out like dx): the resampled tensor has further consumers (dseb.py:63-76, 153-160) whose
 * gradients arrive as one addend instead of aten::add launches (round 4) */
int cenet_bilinear_bwd_add_f32(const float* dy, long sgb, float* dx, long sdb, const float* dx_add, int B, int C, int Hi, int Wi,
                               int Ho, int Wo, float scale_h, float scale_w, int align_corners, cenet_stream_t stream);
/* aten::upsample_nearest2d(+_backward) x2 — blocks.py:304 */
int cenet_nearest2x_fwd_f32(const float* x, long sxb, float* y, long syb, int B, int C, int Hi, int Wi, cenet_stream_t stream);
int cenet_nearest2x_bwd_f32(const float* dy, long sgb, float* dx, long sdb, int B, int C, int Hi, int Wi, cenet_stream_t stream);
/* aten::_adaptive_avg_pool2d(+_backward) — cfam.py:213 */
int cenet_adaptive_avgpool_fwd_f32(const float* x, long sxb, float* y, long syb, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                   cenet_stream_t stream);
int cenet_adaptive_avgpool_bwd_f32(const float* dy, long sgb, float* dx, long sdb, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                   cenet_stream_t stream);
/* aten::max_pool2d_with_indices(2,2)(+_backward) fused with the per-channel scale of out.py:70 */
int cenet_maxpool2_fwd_f32(const float* x, float* y, long syb, const float* scale, int B, int C, int Hi, int Wi,
                           cenet_stream_t stream);
int cenet_maxpool2_bwd_acc_f32(const float* x, const float* dy, long sgb, float* dx, const float* scale, float* dscale_acc,
                               int B, int C, int Hi, int Wi, cenet_stream_t stream);

/* ---- CCU / SRM statistics gates (stats.hip) — cfam.py:251-264, 93-101 --------------------------------------- */
int cenet_ccu_stats_fwd_f32(const float* x, const float* fc1, const float* fc2, float* u, int* amax, float* z, int B, int C,
                            int HW, cenet_stream_t stream);
int cenet_gate_chan_fwd_f32(const float* x, const float* g, float* y, int BC, int HW, cenet_stream_t stream);
int cenet_gate_chan_bwd_reduce_f32(const float* x, const float* dy, const float* g, float* dg, int BC, int HW,
                                   cenet_stream_t stream);
int cenet_ccu_bwd_apply_acc_f32(const float* x, const float* dy, const float* g, const float* dz, const float* u,
                                const int* amax, const float* fc1, const float* fc2, float* dfc1_acc, float* dfc2_acc,
                                float* dx, int B, int C, int HW, cenet_stream_t stream);
/* ... + dx_add (like x, may be NULL): the gradient of x's other consumer (the MCA shortcut, cfam.py:298-303), added by the kernel
 * that writes dx (round 4) */
int cenet_ccu_bwd_apply_add_acc_f32(const float* x, const float* dy, const float* g, const float* dz, const float* u,
                                    const int* amax, const float* fc1, const float* fc2, float* dfc1_acc, float* dfc2_acc,
                                    float* dx, const float* dx_add, int B, int C, int HW, cenet_stream_t stream);
int cenet_srm_stats_fwd_f32(const float* x, float* u, int* amax, int B, int C, int HW, cenet_stream_t stream);
int cenet_srm_conv_fwd_f32(const float* u, const float* pwc, const float* dwc, float* f, int B, int H, int W,
                           cenet_stream_t stream);
int cenet_srm_conv_bwd_acc_f32(const float* u, const float* df, const float* pwc, const float* dwc, float* du, float* dpwc_acc,
                               float* ddwc_acc, int B, int H, int W, cenet_stream_t stream);
int cenet_gate_pix_fwd_f32(const float* x, const float* f, float* y, int B, int C, int HW, cenet_stream_t stream);
int cenet_gate_pix_bwd_reduce_f32(const float* x, const float* dy, const float* f, float* df, int B, int C, int HW,
                                  cenet_stream_t stream);
int cenet_srm_bwd_apply_f32(const float* x, const float* dy, const float* f, const float* u, const float* du, const int* amax,
                            float* dx, int B, int C, int HW, cenet_stream_t stream);

/* ---- glue (elementwise.hip) --------------------------------------------------------------------------------- */
int cenet_transpose_f32(const float* x, long sxb, float* y, long syb, int B, int R, int Cc, cenet_stream_t stream);
/* y = x^T + add (add laid out like y, batch stride syb): backward of a layout change whose source has a second consumer
 * (pvtv2.py:320-321: a stage's token output feeds the decoder as NCHW and the next stage's patch embedding as tokens) */
int cenet_transpose_add_f32(const float* x, long sxb, float* y, long syb, const float* add, int B, int R, int Cc,
                            cenet_stream_t stream);
int cenet_copy_batched_f32(const float* x, long sxb, float* y, long syb, int B, long n, int accumulate, cenet_stream_t stream);
/* torch.cat(parts, dim=1) of up to four contiguous NCHW tensors [B, c_j, HW] into joined [B, sum c_j, HW] (split == 0), or
 * its backward: the channel slices of joined copied out into the parts (split == 1) — one launch either way.  Unused
 * trailing parts: pointer NULL, c_j = 0.  Reference call sites: networks/cenet/modules/cfam.py:238 (torch.cat of the four
 * MultiOrderDWConv branches), modules/dseb.py:156, out.py:63. */
int cenet_cat_channels_f32(float* p0, float* p1, float* p2, float* p3, int c0, int c1, int c2, int c3, float* joined, int B,
                           long HW, int split, cenet_stream_t stream);
/* The split with addends (round 4): p_j = channel slice j of joined + a_j (a_j NULL: the slice alone).  Backward of a concat whose
 * inputs have further consumers (dseb.py:156-164: `dec` also feeds decoders.py:96's residual add, `skip` the mixer's residual):
 * the gradients that reached the inputs along those paths are added by this launch instead of one aten::add each. */
int cenet_split_channels_add_f32(float* p0, float* p1, float* p2, float* p3, const float* a0, const float* a1, const float* a2,
                                 const float* a3, int c0, int c1, int c2, int c3, const float* joined, int B, long HW,
                                 cenet_stream_t stream);
/* Space-to-depth of a token map for the kernel == stride spatial-reduction conv (reference networks/cenet/pvtv2.py:93-95,
 * `self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)` applied to x.permute(0,2,1).reshape(B,C,H,W)):
 * inverse == 0 gathers tok [B, Ho*S, Wo*S, C] into patch rows [B*Ho*Wo, C*S*S] (k = (c, ky, kx), the weight's own order);
 * inverse == 1 scatters patch rows back to the token map (the data gradient). S in {2, 4, 8}. */
int cenet_patch_tok_f32(const float* src, float* dst, int B, int Ho, int Wo, int C, int S, int inverse, cenet_stream_t stream);
/* Overlapping K x K patches of a token-layout map (K = 3; the stride-2 pad-1 patch-embedding convs, pvtv2.py:164):
 * inverse = 0: src tok [B, H*W, C] -> dst rows [B*Ho*Wo, C*K*K] (k = (c, ky, kx), zeros outside the map);
 * inverse = 1: src rows -> dst tok, every pixel summing the row entries that cover it (the conv's data gradient). */
int cenet_im2col_tok_f32(const float* src, float* dst, int B, int H, int W, int C, int K, int stride, int pad, int inverse,
    cenet_stream_t stream);
int cenet_scale_batch_f32(const float* x, const float* s, float* y, int B, long n, cenet_stream_t stream);
int cenet_act_fwd_f32(const float* x, float* y, long n, int act, float slope, cenet_stream_t stream);
int cenet_act_bwd_f32(const float* pre, const float* dy, float* dx, long n, int act, float slope, cenet_stream_t stream);
int cenet_silu_mul_fwd_f32(const float* a, const float* b, float* y, long n, cenet_stream_t stream);
int cenet_silu_mul_bwd_f32(const float* a, const float* b, const float* dy, float* da, float* db, long n, cenet_stream_t stream);
int cenet_mix_fwd_f32(const float* x, const float* p, const float* w, float* z, long n, cenet_stream_t stream);
int cenet_mix_bwd_acc_f32(const float* x, const float* p, const float* w, const float* dz, float* dx, float* dp, float* dw_acc,
                          long n, cenet_stream_t stream);
int cenet_scale_residual_fwd_f32(const float* x, const float* y, const float* ls, float* out, int B, int C, int HW,
                                 cenet_stream_t stream);
int cenet_scale_chan_f32(const float* g, const float* ls, float* out, int B, int C, int HW, cenet_stream_t stream);
int cenet_chan_dot_acc_f32(const float* a, long sab, const float* b, long sbb, float* out_acc, int B, int C, int HW,
                           cenet_stream_t stream);
int cenet_col_sum_acc_f32(const float* a, float* out_acc, long R, int C, cenet_stream_t stream);
int cenet_add_act_fwd_f32(const float* a, const float* b, float* out, long n, int act, float slope, cenet_stream_t stream);
int cenet_lrelu_bwd_from_out_f32(const float* out, const float* dy, float* dx, long n, float slope, cenet_stream_t stream);
/* dseb.py:40-50,63-76,156-163: z = ycoef*y + w[c]*edge(y, r_0..r_{n-1}) + diff*y (r_s NULL = scale 1.0; diff NULL = 0) */
int cenet_dseb_combine_fwd_f32(const float* y, const float* r0, const float* r1, const float* r2, int n, const float* w,
                               const float* diff, float ycoef, float* z, int B, int C, int HW, cenet_stream_t stream);
int cenet_dseb_combine_bwd_acc_f32(const float* y, const float* r0, const float* r1, const float* r2, int n, const float* w,
                                   const float* diff, float ycoef, const float* dz, float* dy, float* dr0, float* dr1, float* dr2,
                                   float* ddiff, float* dw_acc, int B, int C, int HW, cenet_stream_t stream);
/* multihead_diffattn.py:112-123: lambda, A1-lambda*A2 combine, RMSNorm(2hd, eps, no affine), *(1-lambda_init) */
int cenet_diffattn_lambda_fwd_f32(const float* q1, const float* k1, const float* q2, const float* k2, float lambda_init,
                                  float* lam3, int hd, cenet_stream_t stream);
int cenet_diffattn_lambda_bwd_acc_f32(const float* q1, const float* k1, const float* q2, const float* k2, const float* lam3,
                                      const float* dlam, float* dq1, float* dk1, float* dq2, float* dk2, int hd,
                                      cenet_stream_t stream);
int cenet_diffattn_combine_fwd_f32(const float* U, const float* lam3, float* out, int B, int H, int N, int dv, float eps,
                                   float post, cenet_stream_t stream);
int cenet_diffattn_combine_bwd_acc_f32(const float* U, const float* lam3, const float* dout, float* dU, float* dlam_acc, int B,
                                       int H, int N, int dv, float eps, float post, cenet_stream_t stream);

/* ---- loss + optimiser (loss_optim.hip) ---------------------------------------------------------------------- */
/* utils/core.py:44-80,161-188: loss = w_dice*Dice(softmax(logits), onehot(labels)) + w_ce*CE.
 * acc: CENET_LOSS_ACC_FLOATS floats of workspace (replicated partial sums); its first 3K+1 floats are what bwd reads. */
#define CENET_LOSS_ACC_FLOATS 16384
int cenet_dice_ce_fwd_f32(const float* logits, const float* labels, float* acc, float* loss, int B, int K, int HW, float w_dice,
                          float w_ce, cenet_stream_t stream);
int cenet_dice_ce_bwd_f32(const float* logits, const float* labels, const float* acc, const float* gout, float* dlogits, int B,
                          int K, int HW, float w_dice, float w_ce, cenet_stream_t stream);
/* The same with BoundaryDoULoss (utils/core.py:83-131, the loss of scripts/acdc.sh:63 / synapse.sh:67) as a third term:
 * loss = w_dice*Dice + w_ce*CE + w_bd*BoundaryDoU; labels [B,H,W] (class ids as floats). */
int cenet_seg_loss_fwd_f32(const float* logits, const float* labels, float* acc, float* loss, int B, int K, int H, int W,
                           float w_dice, float w_ce, float w_bd, cenet_stream_t stream);
int cenet_seg_loss_bwd_f32(const float* logits, const float* labels, const float* acc, const float* gout, float* dlogits,
                           int B, int K, int H, int W, float w_dice, float w_ce, float w_bd, cenet_stream_t stream);
/* Evaluation (main_acdc.py:218-231 val(), metrics_eval.py:24-34,46-49): pred[B,HW] = argmax over the K
 * logit planes (class ids as floats, may be NULL), counts[(K+1)*3] (uint32, zero-filled by the call; may be NULL together with
 * labels) = per class {|pred==c & gt==c|, |pred==c|, |gt==c|} and, in row K, the same for the binary masks pred>0 / gt>0. */
int cenet_argmax_counts_f32(const float* logits, const float* labels, float* pred, unsigned* counts, int B, int K, int HW,
                            cenet_stream_t stream);
/* Surface-distance metrics of metrics_eval.py:9-21 (`calculate_metric_percase`: medpy.metric.binary.hd95 / assd, medpy==0.5.2,
 * requirements.txt:7; medpy is a third-party dependency absent from the reference tree, its published algorithm is restated).
 * border[D*H*W] = mask XOR binary_erosion(mask) with the 6-neighbourhood and background outside the volume (0/1 bytes). */
int cenet_surface_border_u8(const unsigned char* mask, unsigned char* border, int D, int H, int W, cenet_stream_t stream);
/* out[i] = min(out[i], min_j |a_i - b_j|^2): a [na,3], b [nb,3] int32 voxel coordinates (z,y,x), unit spacing as the
 * reference's calls use; out [na] int32 pre-filled by the caller with INT_MAX. The distance transform of medpy's
 * __surface_distances sampled at the other surface, kept in exact integer arithmetic (host takes the square root). */
int cenet_min_sqdist_i32(const int* a, int na, const int* b, int nb, int* out, cenet_stream_t stream);
/* torch.optim.SGD(momentum, weight_decay) over a flat arena; hyper5 (device) = [lr, momentum, wd, grad_scale, first_step] */
int cenet_sgd_step_f32(float* p, const float* g, float* buf, const float* hyper5, long n, cenet_stream_t stream);
int cenet_zero_f32(float* p, long n, cenet_stream_t stream);
/* zero-fill of any byte range (bf16 activations with an odd element count, views starting on an odd element): the buffers
 * torch.zeros / Tensor.zero_() would give the reference's autograd (e.g. the absent branch of a channel split) */
int cenet_zero_bytes(void* p, long nbytes, cenet_stream_t stream);

/* ---- bf16 twins (throughput mode) ------------------------------------------------------------------------------------
 * Every `cenet_<op>_f32` entry point above that takes ACTIVATION tensors has a twin `cenet_<op>_bf16` with the same
 * semantics and argument order in which the activation / activation-gradient pointers are bf16 (`unsigned short`, bf16 bit
 * patterns) while parameters, statistics (mean / rstd / var / lse / u / z / f ...), workspaces and parameter gradients stay
 * fp32.  Arithmetic is fp32 in both; the reference sites are those of the fp32 declaration.  (GEMM, attention, direct conv
 * and row softmax declare their bf16 forms next to the fp32 ones.) */
/* attn.hip */
int cenet_softmax_rows_fwd_bf16(const float* x, unsigned short* y, long rows, int n, cenet_stream_t stream);
int cenet_softmax_rows_bwd_bf16(const unsigned short* y, const float* dy, unsigned short* dx, long rows, int n,
    cenet_stream_t stream);
/* dwconv.hip */
int cenet_dwconv3x3_nchw_bf16(const unsigned short* x, long sxb, const float* w, const float* bias, unsigned short* y, long
    syb, unsigned short* a, long sab, int B, int C, int H, int W, int dil, int flip, int act, float slope, cenet_stream_t
    stream);
int cenet_dwconv3x3_tok_bf16(const unsigned short* x, const float* w, const float* bias, unsigned short* y, unsigned short*
    a, int B, int C, int H, int W, int flip, int act, float slope, cenet_stream_t stream);
int cenet_dwconv3x3_wgrad_nchw_acc_bf16(const unsigned short* x, long sxb, const unsigned short* dy, long sgb, float*
    dw_acc, float* dbias_acc, int B, int C, int H, int W, int dil, cenet_stream_t stream);
int cenet_dwconv3x3_wgrad_tok_acc_bf16(const unsigned short* x, const unsigned short* dy, float* dw_acc, float* dbias_acc,
    int B, int C, int H, int W, cenet_stream_t stream);
/* backward of act(DW3x3(x) + bias) on bf16 tokens from the saved INPUT (pvtv2.py:42-43,359-370; the pre-activation is not
 * stored): gu = g * act'(conv(x) + bias), dw_acc += gu (*) x, dbias_acc += sum gu.  C % 8 == 0, 16-byte aligned tensors. */
int cenet_dwconv3x3_tok_bwd_pre_bf16(const unsigned short* x, const unsigned short* g, const float* w, const float* bias,
    unsigned short* gu, float* dw_acc, float* dbias_acc, int B, int C, int H, int W, int act, float slope,
    cenet_stream_t stream);
/* elementwise.hip */
int cenet_transpose_bf16(const unsigned short* x, long sxb, unsigned short* y, long syb, int B, int R, int Cc,
    cenet_stream_t stream);
int cenet_transpose_add_bf16(const unsigned short* x, long sxb, unsigned short* y, long syb, const unsigned short* add, int B,
    int R, int Cc, cenet_stream_t stream);
int cenet_copy_batched_bf16(const unsigned short* x, long sxb, unsigned short* y, long syb, int B, long n, int accumulate,
    cenet_stream_t stream);
int cenet_cat_channels_bf16(unsigned short* p0, unsigned short* p1, unsigned short* p2, unsigned short* p3, int c0, int c1,
    int c2, int c3, unsigned short* joined, int B, long HW, int split, cenet_stream_t stream);
int cenet_split_channels_add_bf16(unsigned short* p0, unsigned short* p1, unsigned short* p2, unsigned short* p3,
    const unsigned short* a0, const unsigned short* a1, const unsigned short* a2, const unsigned short* a3, int c0, int c1, int c2,
    int c3, const unsigned short* joined, int B, long HW, cenet_stream_t stream);
int cenet_patch_tok_bf16(const unsigned short* src, unsigned short* dst, int B, int Ho, int Wo, int C, int S, int inverse,
    cenet_stream_t stream);
int cenet_im2col_tok_bf16(const unsigned short* src, unsigned short* dst, int B, int H, int W, int C, int K, int stride,
    int pad, int inverse, cenet_stream_t stream);
int cenet_scale_batch_bf16(const unsigned short* x, const float* s, unsigned short* y, int B, long n, cenet_stream_t
    stream);
int cenet_act_fwd_bf16(const unsigned short* x, unsigned short* y, long n, int act, float slope, cenet_stream_t stream);
int cenet_act_bwd_bf16(const unsigned short* pre, const unsigned short* dy, unsigned short* dx, long n, int act, float
    slope, cenet_stream_t stream);
int cenet_silu_mul_fwd_bf16(const unsigned short* a, const unsigned short* b, unsigned short* y, long n, cenet_stream_t
    stream);
int cenet_silu_mul_bwd_bf16(const unsigned short* a, const unsigned short* b, const unsigned short* dy, unsigned short* da,
    unsigned short* db, long n, cenet_stream_t stream);
int cenet_mix_fwd_bf16(const unsigned short* x, const unsigned short* p, const float* w, unsigned short* z, long n,
    cenet_stream_t stream);
int cenet_mix_bwd_acc_bf16(const unsigned short* x, const unsigned short* p, const float* w, const unsigned short* dz,
    unsigned short* dx, unsigned short* dp, float* dw_acc, long n, cenet_stream_t stream);
int cenet_scale_residual_fwd_bf16(const unsigned short* x, const unsigned short* y, const float* ls, unsigned short* out,
    int B, int C, int HW, cenet_stream_t stream);
int cenet_scale_chan_bf16(const unsigned short* g, const float* ls, unsigned short* out, int B, int C, int HW,
    cenet_stream_t stream);
int cenet_chan_dot_acc_bf16(const unsigned short* a, long sab, const unsigned short* b, long sbb, float* out_acc, int B, int
    C, int HW, cenet_stream_t stream);
int cenet_col_sum_acc_bf16(const unsigned short* a, float* out_acc, long R, int C, cenet_stream_t stream);
int cenet_add_act_fwd_bf16(const unsigned short* a, const unsigned short* b, unsigned short* out, long n, int act, float
    slope, cenet_stream_t stream);
int cenet_lrelu_bwd_from_out_bf16(const unsigned short* out, const unsigned short* dy, unsigned short* dx, long n, float
    slope, cenet_stream_t stream);
int cenet_dseb_combine_fwd_bf16(const unsigned short* y, const unsigned short* r0, const unsigned short* r1, const unsigned
    short* r2, int n, const float* w, const unsigned short* diff, float ycoef, unsigned short* z, int B, int C, int HW,
    cenet_stream_t stream);
int cenet_dseb_combine_bwd_acc_bf16(const unsigned short* y, const unsigned short* r0, const unsigned short* r1, const
    unsigned short* r2, int n, const float* w, const unsigned short* diff, float ycoef, const unsigned short* dz, unsigned
    short* dy, unsigned short* dr0, unsigned short* dr1, unsigned short* dr2, unsigned short* ddiff, float* dw_acc, int B,
    int C, int HW, cenet_stream_t stream);
int cenet_diffattn_combine_fwd_bf16(const unsigned short* U, const float* lam3, unsigned short* out, int B, int H, int N,
    int dv, float eps, float post, cenet_stream_t stream);
int cenet_diffattn_combine_bwd_acc_bf16(const unsigned short* U, const float* lam3, const unsigned short* dout, unsigned
    short* dU, float* dlam_acc, int B, int H, int N, int dv, float eps, float post, cenet_stream_t stream);
/* loss_optim.hip */
int cenet_seg_loss_fwd_bf16(const unsigned short* logits, const float* labels, float* acc, float* loss, int B, int K, int H,
    int W, float w_dice, float w_ce, float w_bd, cenet_stream_t stream);
int cenet_seg_loss_bwd_bf16(const unsigned short* logits, const float* labels, const float* acc, const float* gout, unsigned
    short* dlogits, int B, int K, int H, int W, float w_dice, float w_ce, float w_bd, cenet_stream_t stream);
int cenet_argmax_counts_bf16(const unsigned short* logits, const float* labels, float* pred, unsigned* counts, int B, int K,
    int HW, cenet_stream_t stream);
/* norm.hip */
int cenet_layernorm_fwd_bf16(const unsigned short* x, const float* gamma, const float* beta, unsigned short* y, float* mean,
    float* rstd, int rows, int C, float eps, cenet_stream_t stream);
int cenet_layernorm_bwd_add_acc_bf16(const unsigned short* dy, const unsigned short* x, const float* gamma, const float*
    mean, const float* rstd, const unsigned short* dx_add, unsigned short* dx, float* dgamma_acc, float* dbeta_acc, int
    rows, int C, cenet_stream_t stream);
int cenet_layernorm_bwd_acc_bf16(const unsigned short* dy, const unsigned short* x, const float* gamma, const float* mean,
    const float* rstd, unsigned short* dx, float* dgamma_acc, float* dbeta_acc, int rows, int C, cenet_stream_t stream);
int cenet_bn_stats_bf16(const unsigned short* x, long sb, int B, int C, int HW, float* ws, float* mean, float* var, float*
    running_mean, float* running_var, float momentum, long* num_batches_tracked, cenet_stream_t stream);
int cenet_bn_apply_bf16(const unsigned short* x, long sxb, unsigned short* y, long syb, const float* mean, const float* var,
    float eps, const float* gamma, const float* beta, int act, float slope, int B, int C, int HW, cenet_stream_t stream);
int cenet_bn_train_fwd_bf16(const unsigned short* x, long sxb, unsigned short* y, long syb, float* ws, float* mean, float* var,
    float* running_mean, float* running_var, float momentum, long* num_batches_tracked, float eps, const float* gamma,
    const float* beta, int act, float slope, int B, int C, int HW, int nbt_count, cenet_stream_t stream);
int cenet_bn_bwd_acc_bf16(const unsigned short* dy, long sgb, const unsigned short* x, long sxb, unsigned short* dx, long
    sdb, const float* mean, const float* var, float eps, const float* gamma, const float* beta, int act, float slope, int B,
    int C, int HW, float* ws, float* dgamma_acc, float* dbeta_acc, cenet_stream_t stream);
int cenet_bn_bwd_add_acc_bf16(const unsigned short* dy, long sgb, const unsigned short* x, long sxb, unsigned short* dx, long sdb,
    const unsigned short* dx_add, long sab, const float* mean, const float* var, float eps, const float* gamma, const float* beta,
    int act, float slope, int B, int C, int HW, float* ws, float* dgamma_acc, float* dbeta_acc, cenet_stream_t stream);
/* resample.hip */
int cenet_bilinear_fwd_bf16(const unsigned short* x, long sxb, unsigned short* y, long syb, int B, int C, int Hi, int Wi,
    int Ho, int Wo, float scale_h, float scale_w, int align_corners, cenet_stream_t stream);
int cenet_bilinear_bwd_bf16(const unsigned short* dy, long sgb, unsigned short* dx, long sdb, int B, int C, int Hi, int Wi,
    int Ho, int Wo, float scale_h, float scale_w, int align_corners, cenet_stream_t stream);
int cenet_bilinear_bwd_add_bf16(const unsigned short* dy, long sgb, unsigned short* dx, long sdb, const unsigned short* dx_add,
    int B, int C, int Hi, int Wi, int Ho, int Wo, float scale_h, float scale_w, int align_corners, cenet_stream_t stream);
int cenet_nearest2x_fwd_bf16(const unsigned short* x, long sxb, unsigned short* y, long syb, int B, int C, int Hi, int Wi,
    cenet_stream_t stream);
int cenet_nearest2x_bwd_bf16(const unsigned short* dy, long sgb, unsigned short* dx, long sdb, int B, int C, int Hi, int Wi,
    cenet_stream_t stream);
int cenet_adaptive_avgpool_fwd_bf16(const unsigned short* x, long sxb, unsigned short* y, long syb, int B, int C, int Hi,
    int Wi, int Ho, int Wo, cenet_stream_t stream);
int cenet_adaptive_avgpool_bwd_bf16(const unsigned short* dy, long sgb, unsigned short* dx, long sdb, int B, int C, int Hi,
    int Wi, int Ho, int Wo, cenet_stream_t stream);
int cenet_maxpool2_fwd_bf16(const unsigned short* x, unsigned short* y, long syb, const float* scale, int B, int C, int Hi,
    int Wi, cenet_stream_t stream);
int cenet_maxpool2_bwd_acc_bf16(const unsigned short* x, const unsigned short* dy, long sgb, unsigned short* dx, const
    float* scale, float* dscale_acc, int B, int C, int Hi, int Wi, cenet_stream_t stream);
/* stats.hip */
int cenet_ccu_stats_fwd_bf16(const unsigned short* x, const float* fc1, const float* fc2, float* u, int* amax, float* z, int
    B, int C, int HW, cenet_stream_t stream);
int cenet_gate_chan_fwd_bf16(const unsigned short* x, const float* g, unsigned short* y, int BC, int HW, cenet_stream_t
    stream);
int cenet_gate_chan_bwd_reduce_bf16(const unsigned short* x, const unsigned short* dy, const float* g, float* dg, int BC,
    int HW, cenet_stream_t stream);
int cenet_ccu_bwd_apply_acc_bf16(const unsigned short* x, const unsigned short* dy, const float* g, const float* dz, const
    float* u, const int* amax, const float* fc1, const float* fc2, float* dfc1_acc, float* dfc2_acc, unsigned short* dx, int
    B, int C, int HW, cenet_stream_t stream);
int cenet_ccu_bwd_apply_add_acc_bf16(const unsigned short* x, const unsigned short* dy, const float* g, const float* dz, const
    float* u, const int* amax, const float* fc1, const float* fc2, float* dfc1_acc, float* dfc2_acc, unsigned short* dx, const
    unsigned short* dx_add, int B, int C, int HW, cenet_stream_t stream);
int cenet_srm_stats_fwd_bf16(const unsigned short* x, float* u, int* amax, int B, int C, int HW, cenet_stream_t stream);
int cenet_gate_pix_fwd_bf16(const unsigned short* x, const float* f, unsigned short* y, int B, int C, int HW, cenet_stream_t
    stream);
int cenet_gate_pix_bwd_reduce_bf16(const unsigned short* x, const unsigned short* dy, const float* f, float* df, int B, int
    C, int HW, cenet_stream_t stream);
int cenet_srm_bwd_apply_bf16(const unsigned short* x, const unsigned short* dy, const float* f, const float* u, const float*
    du, const int* amax, unsigned short* dx, int B, int C, int HW, cenet_stream_t stream);

/* bf16 shadow of the fp32 master parameters (the weight operand of cenet_gemm_bf16): the fused SGD step can rewrite it in the
 * same pass (shadow_bf16 may be NULL), and whole buffers can be converted either way. */
int cenet_sgd_step_shadow_f32(float* p, const float* g, float* buf, const float* hyper5, long n, unsigned short* shadow_bf16,
                              cenet_stream_t stream);
int cenet_cast_f32_to_bf16(const float* x, unsigned short* y, long n, cenet_stream_t stream);
int cenet_cast_bf16_to_f32(const unsigned short* x, float* y, long n, cenet_stream_t stream);
/* y = bf16(x) and x = 0 in one pass: x is a persistent fp32 accumulator that kernels add into atomically (the dK / dV of the
 * spatial-reduction attention backward, pvtv2.py:88-109), left zero for its next use instead of a fill before every use */
int cenet_cast_clear_f32_to_bf16(float* x, unsigned short* y, long n, cenet_stream_t stream);
/* rows of N columns (N % 4 == 0): y = bf16(x + bias[col]); x = 0 — the tail of a split-K Linear (pvtv2.py:93-95 spatial-reduction
 * conv as a GEMM over K = C s^2) whose partial products were added atomically into the zero-at-rest accumulator x */
int cenet_cast_clear_bias_f32_to_bf16(float* x, unsigned short* y, const float* bias, int N, long n, cenet_stream_t stream);

/* ---- channel-local fused chains (chanloc.hip, round 5) --------------------------------------------------------------------
 * One workgroup owns one channel over the whole batch: BatchNorm's batch statistics are workgroup reductions, nothing is
 * stored between the steps of the chain, no float atomics (results independent of scheduling).
 * EUCB front, blocks.py:297-321 up to the 1x1 conv: y = LeakyReLU(BatchNorm_train(DW3x3(nearest_x2(x)))); x [B, C, H, W] (batch
 * stride sxb), y [B, C, 2H, 2W] (batch stride syb), w [C][9] (no bias).  Forward writes the batch mean / biased variance of the
 * conv output and updates the running statistics (unbiased variance; NULL: not wanted) and the batch counter.  Backward takes
 * the gradient g of y and writes dx; dw / dgamma / dbeta are ADDED into.  cenet_eucb_supported: the channel's source planes
 * (B*H*W elements of esize bytes) plus one group of gradient planes fit a workgroup's LDS. */
int cenet_eucb_supported(int B, int H, int W, int esize);
int cenet_eucb_fwd_f32(const float* x, long sxb, const float* w, const float* gamma, const float* beta, float eps, float slope,
                       float* y, long syb, float* mean, float* var, float* running_mean, float* running_var, float momentum,
                       long* num_batches_tracked, int B, int C, int H, int W, cenet_stream_t stream);
int cenet_eucb_bwd_acc_f32(const float* g, long sgb, const float* x, long sxb, const float* w, const float* gamma,
                           const float* beta, float eps, float slope, const float* mean, const float* var, float* dx, long sdb,
                           float* dw_acc, float* dgamma_acc, float* dbeta_acc, int B, int C, int H, int W,
                           cenet_stream_t stream);
int cenet_eucb_fwd_bf16(const unsigned short* x, long sxb, const float* w, const float* gamma, const float* beta, float eps,
                        float slope, unsigned short* y, long syb, float* mean, float* var, float* running_mean,
                        float* running_var, float momentum, long* num_batches_tracked, int B, int C, int H, int W,
                        cenet_stream_t stream);
int cenet_eucb_bwd_acc_bf16(const unsigned short* g, long sgb, const unsigned short* x, long sxb, const float* w,
                            const float* gamma, const float* beta, float eps, float slope, const float* mean, const float* var,
                            unsigned short* dx, long sdb, float* dw_acc, float* dgamma_acc, float* dbeta_acc, int B, int C,
                            int H, int W, cenet_stream_t stream);

/* CFAM "mid" chain, cfam.py:368-372 around nlb.py:141-148 (p_raw = the Non-local block's output conv, m = that block's input, x0 =
 * the CFAM block's input; all [B, C, HW] contiguous):  p = BatchNorm_nl(p_raw); z = (1 - w) m + w p; x1 = x0 + ls[c] z;
 * y2 = BatchNorm_2(x1).  One launch each way (workgroup = channel over the batch; cenet_chanloc_supported(B, HW) says whether a
 * channel is small enough).  Forward writes x1, y2, both batch means / biased variances, updates both running statistics and
 * counters (NULL: not wanted).  Backward: g_y2 and (optional) g_x1 in; d_p_raw, d_m, d_x0 out; the six parameter gradients are
 * ADDED into (dw, a scalar, with one float atomic per channel). */
int cenet_chanloc_supported(int B, int HW);
int cenet_cfam_mid_fwd_f32(const float* p_raw, const float* m, const float* x0, float* x1, float* y2, const float* gamma_p,
                           const float* beta_p, float eps_p, float* mean_p, float* var_p, float* rmean_p, float* rvar_p,
                           float mom_p, long* nbt_p, const float* w, const float* ls, const float* gamma_2, const float* beta_2,
                           float eps_2, float* mean_2, float* var_2, float* rmean_2, float* rvar_2, float mom_2, long* nbt_2,
                           int B, int C, int HW, cenet_stream_t stream);
int cenet_cfam_mid_bwd_acc_f32(const float* g_y2, const float* g_x1, const float* p_raw, const float* m, const float* x1,
                               float* d_p_raw, float* d_m, float* d_x0, const float* gamma_p, const float* beta_p, float eps_p,
                               const float* mean_p, const float* var_p, const float* w, const float* ls, const float* gamma_2,
                               float eps_2, const float* mean_2, const float* var_2, float* dgamma_p_acc, float* dbeta_p_acc,
                               float* dw_acc, float* dls_acc, float* dgamma_2_acc, float* dbeta_2_acc, int B, int C, int HW,
                               cenet_stream_t stream);
int cenet_cfam_mid_fwd_bf16(const unsigned short* p_raw, const unsigned short* m, const unsigned short* x0, unsigned short* x1,
                            unsigned short* y2, const float* gamma_p, const float* beta_p, float eps_p, float* mean_p,
                            float* var_p, float* rmean_p, float* rvar_p, float mom_p, long* nbt_p, const float* w, const float* ls,
                            const float* gamma_2, const float* beta_2, float eps_2, float* mean_2, float* var_2, float* rmean_2,
                            float* rvar_2, float mom_2, long* nbt_2, int B, int C, int HW, cenet_stream_t stream);
int cenet_cfam_mid_bwd_acc_bf16(const unsigned short* g_y2, const unsigned short* g_x1, const unsigned short* p_raw,
                                const unsigned short* m, const unsigned short* x1, unsigned short* d_p_raw, unsigned short* d_m,
                                unsigned short* d_x0, const float* gamma_p, const float* beta_p, float eps_p, const float* mean_p,
                                const float* var_p, const float* w, const float* ls, const float* gamma_2, float eps_2,
                                const float* mean_2, const float* var_2, float* dgamma_p_acc, float* dbeta_p_acc, float* dw_acc,
                                float* dls_acc, float* dgamma_2_acc, float* dbeta_2_acc, int B, int C, int HW,
                                cenet_stream_t stream);

/* Dilated depthwise branches of MultiOrderDWConv with their BatchNorm, cfam.py:227-241 over blocks.py:169-177, one launch each
 * way (workgroup = channel over the batch; cenet_chanloc_supported(B, H*W)):
 *   v[:, j*g + i] = ReLU(BatchNorm_train(DW3x3 with dilation dil[j] and weights w[j] ([g][9], no bias) of x[:, j*g + i])), j < nb <= 3;
 *   rest = x[:, nb*g : nb*g + p] (the pooled branch's slice; p may be 0).
 * x [B, nb*g + p, H, W] (batch stride sxb), v [B, nb*g, H, W] (svb), rest [B, p, H, W] (srb); gamma / beta / mean / var / running
 * statistics cover the nb*g channels, num_batches_tracked holds nb counters (NULL: not wanted).
 * Backward: g_v, g_rest and g_add (another gradient of x, [B, nb*g + p, H, W], may be NULL) in; dx = their sum through the chain;
 * dw_acc[j] / dgamma / dbeta are ADDED into (B*H*W <= 8192: the channel's planes and gradient planes sit in LDS). */
int cenet_dwbn_fwd_f32(const float* x, long sxb, const float* const* w, const int* dil, int nb, int g, int p, float* v, long svb,
                       float* rest, long srb, const float* gamma, const float* beta, float eps, float* mean, float* var,
                       float* running_mean, float* running_var, float momentum, long* num_batches_tracked, int B, int H, int W,
                       cenet_stream_t stream);
int cenet_dwbn_bwd_acc_f32(const float* g_v, long sgb, const float* g_rest, long srb, const float* g_add, long sab, const float* x,
                           long sxb, const float* const* w, const int* dil, int nb, int g, int p, const float* gamma,
                           const float* beta, float eps, const float* mean, const float* var, float* dx, long sdb,
                           float* const* dw_acc, float* dgamma_acc, float* dbeta_acc, int B, int H, int W, cenet_stream_t stream);
int cenet_dwbn_fwd_bf16(const unsigned short* x, long sxb, const float* const* w, const int* dil, int nb, int g, int p,
                        unsigned short* v, long svb, unsigned short* rest, long srb, const float* gamma, const float* beta,
                        float eps, float* mean, float* var, float* running_mean, float* running_var, float momentum,
                        long* num_batches_tracked, int B, int H, int W, cenet_stream_t stream);
int cenet_dwbn_bwd_acc_bf16(const unsigned short* g_v, long sgb, const unsigned short* g_rest, long srb,
                            const unsigned short* g_add, long sab, const unsigned short* x, long sxb, const float* const* w,
                            const int* dil, int nb, int g, int p, const float* gamma, const float* beta, float eps,
                            const float* mean, const float* var, unsigned short* dx, long sdb, float* const* dw_acc,
                            float* dgamma_acc, float* dbeta_acc, int B, int H, int W, cenet_stream_t stream);

/* CFAM front, cfam.py:366 over cfam.py:251-264, one launch each way (workgroup = channel over the batch, one wave per image for the
 * per-image statistics; cenet_chanloc_supported(B, HW), B <= 256):  y1 = BatchNorm_1(x0);  u[b, c] = [max, mean, biased std] of
 * y1[b, c];  z = fc2 . relu(fc1 u) (fc1 [C][3][3], fc2 [C][3]);  zn = BatchNorm1d_train(z) over the batch (gamma_d == NULL: zn = z,
 * the reference's batch-of-one rule);  xs = y1 * sigmoid(zn).  All tensors [B, C, HW] contiguous; u [B, C, 3], amax / z / zn
 * [B, C] are saved for the backward pass.  Backward: g_xs (gradient of xs), g_y1 (of y1: the MCA shortcut; may be NULL), g_tap (of
 * x0 itself: the residual around the block; may be NULL) in, dx0 out; parameter gradients ADDED into. */
int cenet_cfam_front_fwd_f32(const float* x0, float* y1, float* xs, const float* gamma1, const float* beta1, float eps1,
                             float* mean1, float* var1, float* rmean1, float* rvar1, float mom1, long* nbt1, const float* fc1,
                             const float* fc2, const float* gamma_d, const float* beta_d, float eps_d, float* mean_d, float* var_d,
                             float* rmean_d, float* rvar_d, float mom_d, long* nbt_d, float* u, int* amax, float* z, float* zn,
                             int B, int C, int HW, cenet_stream_t stream);
int cenet_cfam_front_bwd_acc_f32(const float* g_xs, const float* g_y1, const float* g_tap, const float* x0, float* dx0,
                                 const float* gamma1, const float* beta1, float eps1, const float* mean1, const float* var1,
                                 const float* fc1, const float* fc2, const float* gamma_d, float eps_d, const float* mean_d,
                                 const float* var_d, const float* u, const int* amax, const float* z, const float* zn,
                                 float* dgamma1_acc, float* dbeta1_acc, float* dfc1_acc, float* dfc2_acc, float* dgamma_d_acc,
                                 float* dbeta_d_acc, int B, int C, int HW, cenet_stream_t stream);
int cenet_cfam_front_fwd_bf16(const unsigned short* x0, unsigned short* y1, unsigned short* xs, const float* gamma1,
                              const float* beta1, float eps1, float* mean1, float* var1, float* rmean1, float* rvar1, float mom1,
                              long* nbt1, const float* fc1, const float* fc2, const float* gamma_d, const float* beta_d,
                              float eps_d, float* mean_d, float* var_d, float* rmean_d, float* rvar_d, float mom_d, long* nbt_d,
                              float* u, int* amax, float* z, float* zn, int B, int C, int HW, cenet_stream_t stream);
int cenet_cfam_front_bwd_acc_bf16(const unsigned short* g_xs, const unsigned short* g_y1, const unsigned short* g_tap,
                                  const unsigned short* x0, unsigned short* dx0, const float* gamma1, const float* beta1,
                                  float eps1, const float* mean1, const float* var1, const float* fc1, const float* fc2,
                                  const float* gamma_d, float eps_d, const float* mean_d, const float* var_d, const float* u,
                                  const int* amax, const float* z, const float* zn, float* dgamma1_acc, float* dbeta1_acc,
                                  float* dfc1_acc, float* dfc2_acc, float* dgamma_d_acc, float* dbeta_d_acc, int B, int C, int HW,
                                  cenet_stream_t stream);

/* Depthwise 3x3 (+bias) + activation with NO BatchNorm, cfam.py:150-151 (the CFAM Mlp's conv + GELU), one launch each way
 * (workgroup = channel over the batch, B*H*W <= 8192 and cenet_chanloc_supported(B, H*W)): y = act(DW3x3_dil(x) + bias); x, y
 * [B, C, H, W] contiguous, w [C][9], bias [C] or NULL.  The pre-activation is not stored; backward recomputes it from x and
 * writes dx; dw / dbias (may be NULL) are ADDED into. */
int cenet_dwact_fwd_f32(const float* x, const float* w, const float* bias, float* y, int act, float slope, int dil, int B, int C,
                        int H, int W, cenet_stream_t stream);
int cenet_dwact_bwd_acc_f32(const float* g, const float* x, const float* w, const float* bias, float* dx, float* dw_acc,
                            float* dbias_acc, int act, float slope, int dil, int B, int C, int H, int W, cenet_stream_t stream);
int cenet_dwact_fwd_bf16(const unsigned short* x, const float* w, const float* bias, unsigned short* y, int act, float slope,
                         int dil, int B, int C, int H, int W, cenet_stream_t stream);
int cenet_dwact_bwd_acc_bf16(const unsigned short* g, const unsigned short* x, const float* w, const float* bias,
                             unsigned short* dx, float* dw_acc, float* dbias_acc, int act, float slope, int dil, int B, int C,
                             int H, int W, cenet_stream_t stream);

/* Pooled branch of MultiOrderDWConv, cfam.py:212-218,231-232, two launches each way at every decoder level:
 *   y = bilinear_{(H, W), align_corners = False}(bilinear_{x7, align_corners = True}(LeakyReLU(BatchNorm_train(Conv1x1_{PxP}(AdaptiveAvgPool_7x7(x))))))
 * x [B, P, H, W] (batch stride sxb: a channel slice of a wider tensor), y likewise (syb); P <= 32, B <= 64, H, W <= 64.
 * RH [H, 7] / RW [W, 7]: the two resamplings composed into one linear map per axis (R_{49 -> H} R_{7 -> 49}; host-built with
 * the coordinate rule of cenet_bilinear_fwd).  pooled, t [B, P, 49] fp32 are written for the backward pass, which takes g (the
 * gradient of y, batch stride sgb) and a [B, P, 49] fp32 workspace, writes dx [B, P, H, W] (batch stride sdb) and ADDS the
 * conv-weight [P, P] and BatchNorm gradients. */
int cenet_pool_branch_fwd_f32(const float* x, long sxb, const float* wc, const float* gamma, const float* beta, float eps,
                              float slope, const float* RH, const float* RW, float* y, long syb, float* pooled, float* t,
                              float* mean, float* var, float* running_mean, float* running_var, float momentum,
                              long* num_batches_tracked, int B, int P, int H, int W, cenet_stream_t stream);
int cenet_pool_branch_bwd_acc_f32(const float* g, long sgb, const float* wc, const float* gamma, const float* beta, float eps,
                                  float slope, const float* RH, const float* RW, const float* pooled, const float* t,
                                  const float* mean, const float* var, float* dt_ws, float* dx, long sdb, float* dwc_acc,
                                  float* dgamma_acc, float* dbeta_acc, int B, int P, int H, int W, cenet_stream_t stream);
int cenet_pool_branch_fwd_bf16(const unsigned short* x, long sxb, const float* wc, const float* gamma, const float* beta,
                               float eps, float slope, const float* RH, const float* RW, unsigned short* y, long syb,
                               float* pooled, float* t, float* mean, float* var, float* running_mean, float* running_var,
                               float momentum, long* num_batches_tracked, int B, int P, int H, int W, cenet_stream_t stream);
int cenet_pool_branch_bwd_acc_bf16(const unsigned short* g, long sgb, const float* wc, const float* gamma, const float* beta,
                                   float eps, float slope, const float* RH, const float* RW, const float* pooled, const float* t,
                                   const float* mean, const float* var, float* dt_ws, unsigned short* dx, long sdb,
                                   float* dwc_acc, float* dgamma_acc, float* dbeta_acc, int B, int P, int H, int W,
                                   cenet_stream_t stream);

/* Self-test of the wave reductions every kernel's block reductions rest on (common.h wave_sum / wave_max / wave_min_i on the
 * DPP path): wave w reduces x[64 w .. 64 w + 64) -> sums[w], maxs[w], mins[w] = min over lanes of (int)(1024 x) + lane, as lane 0
 * holds them, and the same three at [nwaves + w] as lane 37 holds them (each array: 2 nwaves entries). */
int cenet_selftest_wave_reduce(const float* x, float* sums, float* maxs, int* mins, int nwaves, cenet_stream_t stream);

/* SRM tail without separate activation / BatchNorm launches (stats.hip, round 5; cfam.py:93-101 after the channel statistics,
 * training mode): cenet_srm_conv_gelu_fwd_f32 writes f = pwc(u) + dwc(u), fa = GELU(f) and cenet_srm_parts(B, H, W) (count, mean, M2)
 * triples; cenet_gate_pix_bn_fwd_* folds them (exact in any order), writes y = x * sigmoid(fb) with fb = BatchNorm(fa) ([B, HW],
 * also stored), the batch mean / biased variance, and updates the running statistics; cenet_srm_conv_bn_bwd_acc_f32 takes dfb (the
 * gradient of fb, from cenet_gate_pix_bwd_reduce_*) and does BatchNorm backward, GELU' and the conv backward (du [B, 3, H, W]; the
 * conv and BatchNorm parameter gradients ADDED into; part2_ws: 2 * cenet_srm_parts floats).  H * W <= 4096. */
int cenet_srm_fused_supported(int B, int H, int W);
int cenet_srm_parts(int B, int H, int W);
int cenet_srm_conv_gelu_fwd_f32(const float* u, const float* pwc, const float* dwc, float* f, float* fa, float* part, int B, int H,
                                int W, cenet_stream_t stream);
int cenet_gate_pix_bn_fwd_f32(const float* x, const float* fa, const float* part, int G, float* fb, float* y, const float* gamma,
                              const float* beta, float eps, float* mean, float* var, float* running_mean, float* running_var,
                              float momentum, long* num_batches_tracked, int B, int C, int HW, cenet_stream_t stream);
int cenet_gate_pix_bn_fwd_bf16(const unsigned short* x, const float* fa, const float* part, int G, float* fb, unsigned short* y,
                               const float* gamma, const float* beta, float eps, float* mean, float* var, float* running_mean,
                               float* running_var, float momentum, long* num_batches_tracked, int B, int C, int HW,
                               cenet_stream_t stream);
int cenet_srm_conv_bn_bwd_acc_f32(const float* u, const float* dfb, const float* fa, const float* f, const float* mean,
                                  const float* var, float eps, const float* gamma, const float* pwc, const float* dwc,
                                  float* part2_ws, float* du, float* dpwc_acc, float* ddwc_acc, float* dgamma_acc, float* dbeta_acc,
                                  int B, int H, int W, cenet_stream_t stream);

/* The ACDC training-time augmentation on the device (augment.hip, round 5; replaces RandomGenerator.__call__,
 * src/datasets/dataset_acdc.py:32-48 with random_rot_flip :15-22 and random_rotate :25-29, for a whole batch).  The training
 * slices live in two device pools (pool_img fp32, pool_lab uint8, same element offsets).  Per sample b:
 *   tab[8 b ..]  = offset, H, W, mode (0 none / 1 quarter turns + flip / 2 rotation), k, axis, resize flag, 0
 *   dp[10 b ..]  = rotation matrix m00 m01 m10 m11, offset0, offset1 (scipy.ndimage.rotate's, mode 2 only);
 *                  pole^(Ha-1), pole^(Wa-1) with pole = sqrt(3) - 2 and (Ha, Wa) the size after stage 1; (Ha-1)/(OH-1), (Wa-1)/(OW-1)
 * all computed by the host in fp64 (cenet_amd/data.py DeviceAugmenter.draw, which consumes `random` / `np.random` in the
 * reference's order).  stage_img (fp64) / stage_lab (uint8): B * stage_stride elements of scratch, stage_stride >= max_h * max_w
 * >= every sample's H * W.  out_img [B, 1, OH, OW] fp32, out_lab [B, OH, OW] fp32 (class indices).  Labels are bit-identical to
 * the reference's, images to fp64 rounding of the same operation order. */
int cenet_augment_acdc(const float* pool_img, const unsigned char* pool_lab, const long* tab, const double* dp, double* stage_img,
                       unsigned char* stage_lab, long stage_stride, int max_h, int max_w, float* out_img, float* out_lab, int B,
                       int OH, int OW, cenet_stream_t stream);

/* LayerNorm of a split-K accumulator (norm.hip, round 5; pvtv2.py:93-95,99-100: sr conv -> norm): acc [rows, C] fp32 holds the
 * conv's products (zero-at-rest accumulator of a split-K GEMM); xpre = bf16(acc + bias) (the rows the LayerNorm backward reads),
 * y = LayerNorm(xpre) * gamma + beta, mean / rstd [rows]; acc is left zero.  bias may be NULL.  C % 4 == 0, C <= 512. */
int cenet_layernorm_fwd_acc_bf16(float* acc, const float* bias, unsigned short* xpre, const float* gamma, const float* beta,
                                 unsigned short* y, float* mean, float* rstd, int rows, int C, float eps, cenet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CENET_HIP_H */
