// gemm_inst_f32_im2col.hip — instantiates gemm_kernel<float, *, *, true, *> (see gemm_core.h)
#include "gemm_core.h"
CENET_GEMM_INSTANCE(cenet_gemm_launch_f32_im2col, float, true)
