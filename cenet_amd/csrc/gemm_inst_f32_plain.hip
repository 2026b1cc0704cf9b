// gemm_inst_f32_plain.hip — instantiates gemm_kernel<float, *, *, false, *> (see gemm_core.h)
#include "gemm_core.h"
CENET_GEMM_INSTANCE(cenet_gemm_launch_f32_plain, float, false)
