// The 16-bit GEMM / conv kernels once more on IEEE half (L4P_F16: the reference's "16-mixed" is fp16 autocast).
#define GEMM_T f16_t
#define GEMM_FN launch_gemm_f16
#define GEMM_GROUP_FN launch_gemm_group_f16
#define GEMM_HAS_8P 1
#include "gemm8p.hpp"
#ifdef L4P_PROBE_KERNELS  // (measured, not adopted: only in a PROBES=1 build)
#include "gemm4w.hpp"
#endif
#include "conv3_halo.hpp"
#include "gemm_skinny.hpp"
#include "gemm_launch.inc"
