// gemm_inst_ring_kk.hip — LDS-DMA ring GEMM (gemm_ring.h): A k-fast, B k-fast
#include "gemm_ring.h"
CENET_RING_INSTANCE(cenet_gemm_launch_ring_kk, true, true)
