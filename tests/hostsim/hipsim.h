// hipsim.h — TEST INFRASTRUCTURE ONLY.
//
// A tiny single-threaded SIMT interpreter that lets the *same* kernel sources under cenet_amd/csrc/
// be compiled with g++ and executed on the host, so that indexing / barrier / wave-shuffle / MFMA
// fragment-layout logic can be checked (and run under ASan/UBSan) in a container that has no GPU.
// It is NOT a CPU fallback: the product library (libcenet_hip.so) is built by hipcc for gfx950 only
// and never contains this file; only tests/ build and load libcenet_sim.so.
//
// Model: every thread of a workgroup is a ucontext fiber; fibers run round-robin and switch at
// __syncthreads(), wave shuffles and MFMA calls.  Wave = 64 lanes.  MFMA builtins are emulated with
// the gfx950 lane maps documented in /opt/skills/guides/cdna_hip_programming.md §3.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CENET_HOSTSIM 1

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_sim { unsigned x, y, z; };

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "hostsim"; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

namespace hipsim {
// Fiber switch.  x86-64: a hand-rolled callee-saved-register switch (glibc's swapcontext makes a sigprocmask system call per
// switch, which dominated the run time of barrier-heavy kernels); elsewhere: ucontext.
#if defined(__x86_64__)
#define HIPSIM_FAST_SWITCH 1
struct Ctx {
  void* rsp = nullptr;
};
#else
#define HIPSIM_FAST_SWITCH 0
struct Ctx {
  ucontext_t uc;
};
#endif
struct Fiber {
  Ctx ctx;
  char* stack = nullptr;
  bool done = true;
  unsigned xph = 0;  // exchanges done so far by this lane (parity picks the xbuf copy)
  uint3_sim tid;
};
struct State {
  std::vector<Fiber> fibers;
  Ctx sched;
  int cur = -1;
  int nthreads = 0;
  // block barrier
  int arrived = 0;
  unsigned gen = 0;
  // wave barriers / exchange
  int warrived[32];
  unsigned wgen[32];
  // per wave, per lane, 64 bytes — TWO copies used alternately (Fiber::xph): a lane that runs ahead into the next exchange writes
  // the other copy, and cannot reach the one after that before every lane of its wave has passed the barrier in between, i.e.
  // has finished reading this one.  One wave barrier per exchange / MFMA instead of two (half the fiber switches).
  alignas(16) unsigned char xbuf2[2][32][64][64];
  std::function<void()> body;
};
State& st();
void yield();
void wave_barrier();
void note_progress();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace hipsim

extern uint3_sim threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

static inline void __syncthreads() {
  auto& s = hipsim::st();
  unsigned g = s.gen;
  if (++s.arrived == s.nthreads) {
    s.arrived = 0;
    s.gen++;
    hipsim::note_progress();
  } else {
    while (s.gen == g) hipsim::yield();
  }
}

static inline int sim_lane() { return hipsim::st().cur & 63; }
// this lane's exchange buffer set for the exchange it is about to make (and advances its count)
static inline unsigned char (*sim_xset())[64][64] {
  auto& s = hipsim::st();
  return s.xbuf2[s.fibers[s.cur].xph++ & 1];
}
static inline int sim_wave() { return hipsim::st().cur >> 6; }

template <typename T>
static inline T sim_exchange(T v, int src_lane) {
  auto& s = hipsim::st();
  int w = sim_wave(), l = sim_lane();
  static_assert(sizeof(T) <= 64, "exchange payload too large");
  auto xb = sim_xset();
  memcpy(xb[w][l], &v, sizeof(T));
  hipsim::wave_barrier();
  int nl = s.nthreads - w * 64;
  if (nl > 64) nl = 64;
  T r = v;
  if (src_lane >= 0 && src_lane < nl) memcpy(&r, xb[w][src_lane], sizeof(T));
  return r;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  int l = sim_lane();
  int src = l ^ mask;
  if ((src / width) != (l / width)) src = l;
  return sim_exchange(v, src);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = sim_lane();
  int src = l + (int)d;
  if ((src / width) != (l / width)) src = l;
  return sim_exchange(v, src);
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
  int l = sim_lane();
  return sim_exchange(v, (l / width) * width + (src % width));
}

static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }

typedef float f32x4 __attribute__((vector_size(16)));
typedef float f32x16 __attribute__((vector_size(64)));
typedef short bf16x8 __attribute__((vector_size(16)));
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

static inline float sim_bf2f(short h) {
  uint32_t u = ((uint32_t)(uint16_t)h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// v_mfma_f32_16x16x4_f32: lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D col=l&15,row=(l>>4)*4+r.
static inline f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
  auto& s = hipsim::st();
  int w = sim_wave(), l = sim_lane();
  auto xb = sim_xset();
  float ab[2] = {a, b};
  memcpy(xb[w][l], ab, 8);
  hipsim::wave_barrier();
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, xb[w][k * 16 + row], 4);
      memcpy(&bv, xb[w][k * 16 + col] + 4, 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  return c;
}

// v_mfma_f32_16x16x32_bf16: lane l holds A[row l&15][k=8(l>>4)+j], B[k=8(l>>4)+j][col l&15], j=0..7.
static inline f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c, int, int, int) {
  auto& s = hipsim::st();
  int w = sim_wave(), l = sim_lane();
  auto xb = sim_xset();
  memcpy(xb[w][l], &a, 16);
  memcpy(xb[w][l] + 16, &b, 16);
  hipsim::wave_barrier();
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k) {
      short av, bv;
      memcpy(&av, xb[w][(k >> 3) * 16 + row] + 2 * (k & 7), 2);
      memcpy(&bv, xb[w][(k >> 3) * 16 + col] + 16 + 2 * (k & 7), 2);
      acc += sim_bf2f(av) * sim_bf2f(bv);
    }
    c[r] = acc;
  }
  return c;
}

// v_mfma_f32_32x32x16_bf16: lane l holds A[row l&31][k = 8(l>>5)+j], B[k = 8(l>>5)+j][col l&31], j = 0..7;
// D col = l&31, row = (reg&3) + 8(reg>>2) + 4(l>>5), reg = 0..15  (cdna_hip_programming.md §3)
static inline f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c, int, int, int) {
  auto& s = hipsim::st();
  int w = sim_wave(), l = sim_lane();
  auto xb = sim_xset();
  memcpy(xb[w][l], &a, 16);
  memcpy(xb[w][l] + 16, &b, 16);
  hipsim::wave_barrier();
  const int col = l & 31, h = l >> 5;
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    float acc = c[reg];
    for (int k = 0; k < 16; ++k) {
      short av, bv;
      memcpy(&av, xb[w][(k >> 3) * 32 + row] + 2 * (k & 7), 2);
      memcpy(&bv, xb[w][(k >> 3) * 32 + col] + 16 + 2 * (k & 7), 2);
      acc += sim_bf2f(av) * sim_bf2f(bv);
    }
    c[reg] = acc;
  }
  return c;
}

#define CENET_LAUNCH(kernel, grid, block, stream, ...) \
  hipsim::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
