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

// activation ids shared by several kernels
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_GELU = 3, ACT_SILU = 4, ACT_SIGMOID = 5 };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

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
#else
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
#endif

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
