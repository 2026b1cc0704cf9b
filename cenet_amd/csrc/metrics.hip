// metrics.hip — surface-distance metrics of the evaluation path (reference src/utils/metrics_eval.py:9-21,
// `calculate_metric_percase`: medpy.metric.binary hd95 / assd on whole volumes; medpy==0.5.2, requirements.txt:7).
// medpy's published algorithm: border = mask XOR binary_erosion(mask, 6-neighbourhood, border_value 0); distances of one
// border's voxels to the other border (its Euclidean distance transform sampled there); hd95 = 95th percentile of both
// directions together, assd = mean of the two directed means.  With unit voxel spacing (the reference passes none) every
// squared distance is an integer, so the device side stays in exact integer arithmetic and the host takes the square root.
#include "common.h"
#include "../../include/cenet_hip.h"

static inline unsigned cdiv_u(long a, long b) { return (unsigned)((a + b - 1) / b); }

// border[v] = mask[v] && !(all six face neighbours inside the volume and set)
__global__ __launch_bounds__(256) void surface_border_kernel(const unsigned char* __restrict__ mask,
                                                            unsigned char* __restrict__ border, int D, int H, int W) {
  const long n = (long)D * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    unsigned char b = 0;
    if (mask[i]) {
      const int x = (int)(i % W);
      const long t = i / W;
      const int y = (int)(t % H);
      const int z = (int)(t / H);
      const long sz = (long)H * W;
      const bool interior = x > 0 && x + 1 < W && y > 0 && y + 1 < H && z > 0 && z + 1 < D && mask[i - 1] && mask[i + 1] &&
                            mask[i - W] && mask[i + W] && mask[i - sz] && mask[i + sz];
      b = interior ? 0 : 1;
    }
    border[i] = b;
  }
}

// out[i] = min(out[i], min_j |a_i - b_j|^2) over this workgroup's slice of b; points are (z, y, x) int32 triples.
// grid (a chunks, b chunks): the b slice is staged through LDS 1024 points at a time, each thread keeps one a point.
#define MD_TILE 1024
__global__ __launch_bounds__(256) void min_sqdist_kernel(const int* __restrict__ a, int na, const int* __restrict__ b, int nb,
                                                        int b_per_block, int* __restrict__ out) {
  __shared__ int sb[MD_TILE * 3];
  const int i = blockIdx.x * 256 + threadIdx.x;
  int az = 0, ay = 0, ax = 0;
  if (i < na) az = a[i * 3], ay = a[i * 3 + 1], ax = a[i * 3 + 2];
  const int j0 = blockIdx.y * b_per_block;
  const int j1 = (j0 + b_per_block < nb) ? j0 + b_per_block : nb;
  int best = 0x7fffffff;
  for (int t0 = j0; t0 < j1; t0 += MD_TILE) {
    const int cnt = (j1 - t0 < MD_TILE) ? j1 - t0 : MD_TILE;
    __syncthreads();
    for (int k = threadIdx.x; k < cnt * 3; k += 256) sb[k] = b[(long)t0 * 3 + k];
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const int dz = az - sb[k * 3], dy = ay - sb[k * 3 + 1], dx = ax - sb[k * 3 + 2];
      const int d2 = dz * dz + dy * dy + dx * dx;
      best = d2 < best ? d2 : best;
    }
  }
  if (i < na && j0 < j1) atomicMin(&out[i], best);
}

extern "C" int cenet_surface_border_u8(const unsigned char* mask, unsigned char* border, int D, int H, int W,
                                       hipStream_t stream) {
  if (D <= 0 || H <= 0 || W <= 0) return CENET_EINVAL;
  const long n = (long)D * H * W;
  unsigned blocks = cdiv_u(n, 256);
  if (blocks > 8192) blocks = 8192;
  CENET_LAUNCH(surface_border_kernel, dim3(blocks), dim3(256), stream, mask, border, D, H, W);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}

extern "C" int cenet_min_sqdist_i32(const int* a, int na, const int* b, int nb, int* out, hipStream_t stream) {
  if (na <= 0 || nb <= 0) return CENET_EINVAL;
  // enough workgroups to fill the chip even for a few thousand surface points: split b when a alone gives too few
  const unsigned ga = cdiv_u(na, 256);
  int chunks = (int)(1024 / ga);
  const int max_chunks = (int)cdiv_u(nb, MD_TILE);
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  int per = (int)cdiv_u(nb, chunks);
  per = (int)cdiv_u(per, MD_TILE) * MD_TILE;
  chunks = (int)cdiv_u(nb, per);
  CENET_LAUNCH(min_sqdist_kernel, dim3(ga, (unsigned)chunks), dim3(256), stream, a, na, b, nb, per, out);
  CENET_CHECK_LAUNCH();
  return CENET_OK;
}
