// gemm_inst_bf16_plain.hip — instantiates gemm_kernel<unsigned short, *, *, false, *> (see gemm_core.h)
#include "gemm_core.h"
CENET_GEMM_INSTANCE(cenet_gemm_launch_bf16_plain, unsigned short, false)
