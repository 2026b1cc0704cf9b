// gemm_inst_bf16_plain_k64.hip — instantiates gemm_kernel<unsigned short, *, *, false, *, 64> (see gemm_core.h)
#include "gemm_core.h"
CENET_GEMM_INSTANCE_K64(cenet_gemm_launch_bf16_plain_k64, unsigned short)
