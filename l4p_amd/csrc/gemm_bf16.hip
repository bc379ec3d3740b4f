#define GEMM_T bf16_t
#define GEMM_FN launch_gemm_bf16
#include "gemm_launch.inc"
