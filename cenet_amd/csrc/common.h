// common.h — shared device helpers for the CENet gfx950 kernels.
//
// Built two ways:
//   * hipcc --offload-arch=gfx950  -> libcenet_hip.so  (THE product; the only library cenet_amd loads)
//   * g++ -DCENET_HOSTSIM_BUILD    -> tests/hostsim/build/libcenet_sim.so (kernel-logic checker used by
//     tests only, see tests/hostsim/hipsim.h; lets the same sources run under ASan/UBSan without a GPU)
#pragma once

#ifdef CENET_HOSTSIM_BUILD
#include "hipsim.h"
#else
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define CENET_LAUNCH(kernel, grid, block, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), 0, (stream), __VA_ARGS__)
#endif

#define CENET_WAVE 64

// XCD-aware block numbering.  Workgroups are dealt round-robin over the 8 XCDs by linear id (MI355X_MICROARCH.md "Workgroup
// dispatch"), each XCD with its own L2, so spatial neighbours of a 3-D grid land on different L2s and every shared halo row
// is fetched from HBM once per XCD.  Re-number so that each XCD owns a contiguous range of the grid (x fastest).  Speed only.
struct cenet_bid {
  int x, y, z;
};
__device__ __forceinline__ cenet_bid cenet_xcd_block() {
  const int gx = gridDim.x, gy = gridDim.y;
  const int T = gx * gy * (int)gridDim.z;
  int L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  if (T >= 64) {
    const int per = T >> 3, rem = T & 7, xcd = L & 7, idx = L >> 3;
    L = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  cenet_bid b;
  b.x = L % gx;
  const int t = L / gx;
  b.y = t % gy;
  b.z = t / gy;
  return b;
}

// ---- LDS-DMA (global_load_lds_dwordx4) helpers shared by gemm_ring.h and the tiled depthwise kernels -------------------------
#ifdef CENET_HOSTSIM_BUILD
static const unsigned ring_zero16[4] = {0, 0, 0, 0};
#else
__device__ __attribute__((aligned(16))) const unsigned ring_zero16[4] = {0, 0, 0, 0};
#endif

// one 16-byte LDS-DMA: lane L of the wave writes wave_base + 16 L
__device__ __forceinline__ void ring_glds16(const void* gsrc, unsigned char* wave_base, int lane) {
#ifdef CENET_HOSTSIM_BUILD
  memcpy(wave_base + 16 * lane, gsrc, 16);
#else
  (void)lane;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned*)gsrc,
                                   (__attribute__((address_space(3))) unsigned*)wave_base, 16, 0, 0);
#endif
}

template <int N>
__device__ __forceinline__ void ring_wait_vm() {
#ifndef CENET_HOSTSIM_BUILD
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void ring_barrier() {
#ifdef CENET_HOSTSIM_BUILD
  __syncthreads();
#else
  asm volatile("s_barrier" ::: "memory");
#endif
}


// fp32 -> bf16, round to nearest even.  gfx950 converts two floats per instruction (v_cvt_pk_bf16_f32); the host-side
// checker build uses the equivalent integer rounding.
#ifdef CENET_HOSTSIM_BUILD
__device__ __forceinline__ unsigned cenet_f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ unsigned cenet_pack_bf2(float lo, float hi) { return cenet_f2bf(lo) | (cenet_f2bf(hi) << 16); }
#else
typedef __bf16 cenet_bf2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cenet_pack_bf2(float lo, float hi) {
  cenet_bf2_t v = {(__bf16)lo, (__bf16)hi};
  unsigned u;
  memcpy(&u, &v, 4);
  return u;
}
// low 16 bits = bf16(f); the high half is bf16(0) = 0, so no mask is needed
__device__ __forceinline__ unsigned cenet_f2bf(float f) { return cenet_pack_bf2(f, 0.f); }
#endif

// ---- storage-type-generic element access ----------------------------------------------------------------------------
// Activation tensors are fp32 (parity mode) or bf16 (throughput mode: `_bf16` entry points); kernels are templates over the
// storage type T and do all arithmetic in fp32.  A "quad" = 4 consecutive elements = one 16-byte (fp32) or 8-byte (bf16)
// access; an "oct" = 8 consecutive bf16 = 16 bytes.
typedef unsigned short bf16_t;
__device__ __forceinline__ float cenet_bf2f(unsigned h) {
  unsigned u = h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return cenet_bf2f(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = (bf16_t)cenet_f2bf(v); }
struct f4 {
  float v[4];
};
__device__ __forceinline__ f4 ld4(const float* p) {
  f4 r;
  memcpy(r.v, p, 16);
  return r;
}
__device__ __forceinline__ f4 ld4(const bf16_t* p) {
  unsigned u[2];
  memcpy(u, p, 8);
  f4 r;
  r.v[0] = cenet_bf2f(u[0] & 0xFFFFu);
  r.v[1] = cenet_bf2f(u[0] >> 16);
  r.v[2] = cenet_bf2f(u[1] & 0xFFFFu);
  r.v[3] = cenet_bf2f(u[1] >> 16);
  return r;
}
__device__ __forceinline__ void st4(float* p, const f4& r) { memcpy(p, r.v, 16); }
__device__ __forceinline__ void st4(bf16_t* p, const f4& r) {
  const unsigned u[2] = {cenet_pack_bf2(r.v[0], r.v[1]), cenet_pack_bf2(r.v[2], r.v[3])};
  memcpy(p, u, 8);
}
// raw-array forms (float v[4]) used by kernels that keep plain arrays
__device__ __forceinline__ void ld4v(float* v, const float* p) { memcpy(v, p, 16); }
__device__ __forceinline__ void ld4v(float* v, const bf16_t* p) {
  const f4 r = ld4(p);
  v[0] = r.v[0], v[1] = r.v[1], v[2] = r.v[2], v[3] = r.v[3];
}
__device__ __forceinline__ void st4v(float* p, const float* v) { memcpy(p, v, 16); }
__device__ __forceinline__ void st4v(bf16_t* p, const float* v) {
  const unsigned u[2] = {cenet_pack_bf2(v[0], v[1]), cenet_pack_bf2(v[2], v[3])};
  memcpy(p, u, 8);
}
// V consecutive elements (V = 1, 4 or 8) -> fp32 registers and back
template <int V>
__device__ __forceinline__ void ldv(float* v, const float* p) {
  if (V == 1) v[0] = *p;
  else memcpy(v, p, 4 * V);
}
template <int V>
__device__ __forceinline__ void ldv(float* v, const bf16_t* p) {
  if (V == 1) {
    v[0] = cenet_bf2f(*p);
  } else {
    unsigned u[V / 2 > 0 ? V / 2 : 1];
    memcpy(u, p, 2 * V);
#pragma unroll
    for (int e = 0; e < V / 2; ++e) {
      v[2 * e] = cenet_bf2f(u[e] & 0xFFFFu);
      v[2 * e + 1] = cenet_bf2f(u[e] >> 16);
    }
  }
}
template <int V>
__device__ __forceinline__ void stv(float* p, const float* v) {
  if (V == 1) *p = v[0];
  else memcpy(p, v, 4 * V);
}
template <int V>
__device__ __forceinline__ void stv(bf16_t* p, const float* v) {
  if (V == 1) {
    *p = (bf16_t)cenet_f2bf(v[0]);
  } else {
    unsigned u[V / 2 > 0 ? V / 2 : 1];
#pragma unroll
    for (int e = 0; e < V / 2; ++e) u[e] = cenet_pack_bf2(v[2 * e], v[2 * e + 1]);
    memcpy(p, u, 2 * V);
  }
}
// host side: widest vector (8, 4 or 1 elements) that n elements behind these pointers allow; a null pointer allows anything
template <typename T>
static inline int vec_width(long n, const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr,
                            const void* e = nullptr, const void* f = nullptr) {
  const uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e | (uintptr_t)f;
  if ((n & 7) == 0 && (m & (8 * sizeof(T) - 1)) == 0 && sizeof(T) == 2) return 8;
  if ((n & 3) == 0 && (m & (4 * sizeof(T) - 1)) == 0) return 4;
  return 1;
}
// host side: a quad access at p is legal
template <typename T>
static inline bool quad_aligned(const void* p) {
  return (((uintptr_t)p) & (4 * sizeof(T) - 1)) == 0;
}
// One template implementation -> the two C-ABI entry points `cenet_<name>_f32` and `cenet_<name>_bf16`: PARAMS is the
// parameter list written with the storage type `T`, ARGS the forwarded argument list.
#define CENET_TWIN(NAME, PARAMS, ARGS)                                                    \
  namespace twin_f32 {                                                                    \
  typedef float T;                                                                        \
  extern "C" int cenet_##NAME##_f32 PARAMS { return NAME##_impl<T> ARGS; }                \
  }                                                                                       \
  namespace twin_bf16 {                                                                   \
  typedef bf16_t T;                                                                       \
  extern "C" int cenet_##NAME##_bf16 PARAMS { return NAME##_impl<T> ARGS; }               \
  }

#define CENET_CHECK_LAUNCH()                         \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return 100 + (int)e__;    \
  } while (0)

// error codes returned through the C ABI (0 == ok)
enum { CENET_OK = 0, CENET_EINVAL = 1, CENET_EUNSUPPORTED = 2 };

// zero-fill n floats on `stream` with a kernel (hipMemsetAsync from a captured stream was observed not to be replayed
// faithfully by hipGraph on this stack; a kernel node always is). Defined in loss_optim.hip.
int cenet_zero_async(float* p, long n, hipStream_t stream);

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// q / n without the ~40-instruction integer division, for 0 <= q < 2^20 (exact there: (q + 0.5) / n is at least 0.5 / n away from
// an integer, the float product's error is below that); inv = cenet_inv_small(n, bound on q), 0 -> the integer division
static inline float cenet_inv_small_host(int n, long qmax) { return (n > 0 && qmax < (1L << 20)) ? 1.f / (float)n : 0.f; }
__device__ __forceinline__ float cenet_inv_small(int n, long qmax) { return (n > 0 && qmax < (1L << 20)) ? 1.f / (float)n : 0.f; }
__device__ __forceinline__ int cenet_div_small(int q, int n, float inv) {
  return inv > 0.f ? (int)(((float)q + 0.5f) * inv) : q / n;
}

// activation ids shared by several kernels
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_GELU = 3, ACT_SILU = 4, ACT_SIGMOID = 5 };

// the shuffle-tree forms (the host checker's, and the reference the DPP forms below are tested against)
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max_shfl(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- wave reductions on the DPP path (round 5) -------------------------------------------------------------------------------
// __shfl_xor lowers to ds_bpermute_b32: an LDS-crossbar round trip (~60 - 100 cycles) per step, six dependent steps per wave
// reduction.  Data-parallel-primitive moves (quad_perm / row_shr / row_bcast modifiers on an ordinary VALU instruction) cost an
// issue slot each: a full 64-lane reduction is six VALU instructions, result in lane 63, broadcast with one v_readlane.  Lanes
// whose source lies outside the row (or is masked) keep `old` = the reduction's identity.  EVERY lane of the wave must be active.
#ifdef CENET_HOSTSIM_BUILD
__device__ __forceinline__ float wave_sum_dpp(float v) { return wave_sum_shfl(v); }
__device__ __forceinline__ float wave_max_dpp(float v) { return wave_max_shfl(v); }
__device__ __forceinline__ int wave_min_i_dpp(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(v, o);
    v = t < v ? t : v;
  }
  return v;
}
#else
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                               ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_f<0xb1, 0xf>(0.f, v);   // quad_perm [1, 0, 3, 2]
  v += dpp_f<0x4e, 0xf>(0.f, v);   // quad_perm [2, 3, 0, 1]
  v += dpp_f<0x114, 0xf>(0.f, v);  // row_shr 4
  v += dpp_f<0x118, 0xf>(0.f, v);  // row_shr 8   -> lanes 12 .. 15 of a row hold the row's sum
  v += dpp_f<0x142, 0xa>(0.f, v);  // row_bcast 15 into rows 1 and 3
  v += dpp_f<0x143, 0xc>(0.f, v);  // row_bcast 31 into rows 2 and 3 -> lane 63 holds the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  const float I = -3.4028235e38f;
  v = fmaxf(v, dpp_f<0xb1, 0xf>(I, v));
  v = fmaxf(v, dpp_f<0x4e, 0xf>(I, v));
  v = fmaxf(v, dpp_f<0x114, 0xf>(I, v));
  v = fmaxf(v, dpp_f<0x118, 0xf>(I, v));
  v = fmaxf(v, dpp_f<0x142, 0xa>(I, v));
  v = fmaxf(v, dpp_f<0x143, 0xc>(I, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_min_i_dpp(int v) {
  const int I = 0x7fffffff;
  int t;
  t = dpp_i<0xb1, 0xf>(I, v); v = t < v ? t : v;
  t = dpp_i<0x4e, 0xf>(I, v); v = t < v ? t : v;
  t = dpp_i<0x114, 0xf>(I, v); v = t < v ? t : v;
  t = dpp_i<0x118, 0xf>(I, v); v = t < v ? t : v;
  t = dpp_i<0x142, 0xa>(I, v); v = t < v ? t : v;
  t = dpp_i<0x143, 0xc>(I, v); v = t < v ? t : v;
  return __builtin_amdgcn_readlane(v, 63);
}
#endif

// wave_sum / wave_max of every kernel: the DPP forms (every call site runs with all 64 lanes of the wave active)
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_max_dpp(v); }

// Block-wide sum over blockDim.x threads (multiple of 64, <= 1024). `red` is >= 16 floats of LDS.
// Every thread must call; result broadcast to all threads.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// hardware exponential (v_exp_f32 on the device build; libm on the host checker)
#ifdef CENET_HOSTSIM_BUILD
__device__ __forceinline__ float fast_exp(float x) { return expf(x); }
__device__ __forceinline__ float fast_log(float x) { return logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return 1.f / x; }
#else
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_log(float x) { return __logf(x); }           // v_log_f32 * ln 2
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }  // v_rcp_f32 (1 ulp)
#endif

#ifdef CENET_HOSTSIM_BUILD
__device__ __forceinline__ float fast_exp2(float x) { return exp2f(x); }
#else
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
#endif

// GELU / GELU' of the bf16 (throughput) kernels, erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below a bf16 ulp), in the
// FEWEST vector instructions — the kernels that call them (csrc/pvt_mlp.hip P2, the depthwise tile / plane kernels, the channel-local
// DW + GELU chains) are bound by vector-instruction issue:
//   h(u) = 1/2 poly(t) exp(-u^2 / 2), t = 1 / (1 + p |u| / sqrt 2)   (= (1 - erf(|u| / sqrt 2)) / 2 = Phi(-|u|))
//   GELU(u)  = u Phi(u) = max(u, 0) - |u| h          (u >= 0: u - u h;  u < 0: u h — no sign transfer, no 0.5 u (1 + erf) chain)
//   GELU'(u) = Phi(u) + u phi(u),  Phi(u) = u >= 0 ? 1 - h : h,  phi(u) = exp(-u^2 / 2) / sqrt(2 pi): ONE exponential serves both
// 11 full-rate instructions + v_rcp_f32 + v_exp_f32 for GELU (was 15 + 2, and an IEEE division — ten instructions — in dwconv.hip).
// The fp32 (parity) kernels keep libm's erff (gelu_f / gelu_grad_f below).
__device__ __forceinline__ float gelu_as_h(float au, float e) {  // au = |u|, e = exp(-u^2 / 2)
  const float t = fast_rcp(fmaf(au, 0.3275911f * 0.70710678118654752f, 1.f));
  const float ph = t * (0.5f * 0.254829592f + t * (0.5f * -0.284496736f + t * (0.5f * 1.421413741f +
                   t * (0.5f * -1.453152027f + t * (0.5f * 1.061405429f)))));
  return ph * e;
}
__device__ __forceinline__ float gelu_as(float u) {
  const float au = fabsf(u);
  const float h = gelu_as_h(au, fast_exp2(u * u * (-0.5f * 1.4426950408889634f)));
  return fmaf(-au, h, fmaxf(u, 0.f));
}
__device__ __forceinline__ float gelu_as_grad(float u) {
  const float au = fabsf(u);
  const float e = fast_exp2(u * u * (-0.5f * 1.4426950408889634f));
  const float h = gelu_as_h(au, e);
  return fmaf(u * 0.3989422804014327f, e, u >= 0.f ? 1.f - h : h);
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float act_fwd(int act, float v, float slope) {
  switch (act) {
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_LRELU: return v > 0.f ? v : v * slope;
    case ACT_GELU: return gelu_f(v);
    case ACT_SILU: return v * sigmoid_f(v);
    case ACT_SIGMOID: return sigmoid_f(v);
    default: return v;
  }
}
// derivative of act at pre-activation v
__device__ __forceinline__ float act_bwd(int act, float v, float slope) {
  switch (act) {
    case ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case ACT_LRELU: return v > 0.f ? 1.f : slope;
    case ACT_GELU: return gelu_grad_f(v);
    case ACT_SILU: { float s = sigmoid_f(v); return s * (1.f + v * (1.f - s)); }
    case ACT_SIGMOID: { float s = sigmoid_f(v); return s * (1.f - s); }
    default: return 1.f;
  }
}
