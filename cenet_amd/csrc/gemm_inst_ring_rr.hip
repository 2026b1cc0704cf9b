// gemm_inst_ring_rr.hip — LDS-DMA ring GEMM (gemm_ring.h): A row-fast, B row-fast
#include "gemm_ring.h"
CENET_RING_INSTANCE(cenet_gemm_launch_ring_rr, false, false)
