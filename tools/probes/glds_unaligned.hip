// Does global_load_lds_dwordx4 accept source addresses that are only 2-byte aligned?  (decides whether the LDS-DMA GEMM
// can stage row-fast operands whose row pitch is not a multiple of 8 elements, e.g. NCHW planes of 49 or 196 pixels)
// hipcc --offload-arch=gfx950 -O2 -o glds_unaligned tools/probes/glds_unaligned.hip && ./glds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned short* src, unsigned short* dst, int shift) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[1024];
  const int lane = threadIdx.x;
  const unsigned short* p = src + shift + lane * 8;  // 16 bytes per lane, base misaligned by `shift` elements
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned*)p,
                                   (__attribute__((address_space(3))) unsigned*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < 8; ++j) dst[lane * 8 + j] = ((unsigned short*)lds)[lane * 8 + j];
}
int main() {
  const int n = 64 * 8 + 16;
  std::vector<unsigned short> h(n);
  for (int i = 0; i < n; ++i) h[i] = (unsigned short)(i * 7 + 3);
  unsigned short *d, *o;
  hipMalloc(&d, n * 2);
  hipMalloc(&o, 512 * 2);
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 8; ++shift) {
    hipMemset(o, 0, 1024);
    k<<<1, 64>>>(d, o, shift);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned short> r(512);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 512; ++i) bad += r[i] != h[i + shift];
    printf("shift %d elements (%2d bytes): %s, %d mismatches\n", shift, shift * 2, hipGetErrorString(e), bad);
  }
  return 0;
}
