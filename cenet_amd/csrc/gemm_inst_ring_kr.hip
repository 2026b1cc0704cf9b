// gemm_inst_ring_kr.hip — LDS-DMA ring GEMM (gemm_ring.h): A k-fast, B row-fast
#include "gemm_ring.h"
CENET_RING_INSTANCE(cenet_gemm_launch_ring_kr, true, false)
