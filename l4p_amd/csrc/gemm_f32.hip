#define GEMM_T float
#define GEMM_FN launch_gemm_f32
#include "gemm_launch.inc"
