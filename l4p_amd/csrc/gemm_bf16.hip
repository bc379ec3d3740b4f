#define GEMM_T bf16_t
#define GEMM_FN launch_gemm_bf16
#define GEMM_GROUP_FN launch_gemm_group_bf16
#define GEMM_HAS_8P 1
#include "gemm8p.hpp"
#include "gemm4w.hpp"
#include "conv3_halo.hpp"
#include "gemm_launch.inc"
