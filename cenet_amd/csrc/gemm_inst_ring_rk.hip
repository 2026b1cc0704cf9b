// gemm_inst_ring_rk.hip — LDS-DMA ring GEMM (gemm_ring.h): A row-fast, B k-fast
#include "gemm_ring.h"
CENET_RING_INSTANCE(cenet_gemm_launch_ring_rk, false, true)
