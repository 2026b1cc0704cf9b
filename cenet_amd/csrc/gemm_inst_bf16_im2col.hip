// gemm_inst_bf16_im2col.hip — instantiates gemm_kernel<unsigned short, *, *, true, *> (see gemm_core.h)
#include "gemm_core.h"
CENET_GEMM_INSTANCE(cenet_gemm_launch_bf16_im2col, unsigned short, true)
