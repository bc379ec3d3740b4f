#!/bin/bash
# Ablation / trace builds of the 64-rows-per-wave attention kernel: tools/probes/attn64_var_<name>
cd "$(dirname "$0")/../.."
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I include -I l4p_amd/csrc "${@:2}" tools/probes/attn64_probe.hip -o tools/probes/attn64_var_$1 & }
build base
build trace -DATTN64_TRACE
build noload -DATTN64_DBG_NOLOAD
build nosoftmax -DATTN64_DBG_NOSOFTMAX
build noldsread -DATTN64_DBG_NOLDSREAD
build nobarrier -DATTN64_DBG_NOBARRIER
build nosm_nolds -DATTN64_DBG_NOSOFTMAX -DATTN64_DBG_NOLDSREAD
build nosm_nolds_noload -DATTN64_DBG_NOSOFTMAX -DATTN64_DBG_NOLDSREAD -DATTN64_DBG_NOLOAD
build pre2 -DATTN64_PRE=2
build pre6 -DATTN64_PRE=6
build pre8 -DATTN64_PRE=8
build mfma_only -DATTN64_DBG_NOSOFTMAX -DATTN64_DBG_NOLDSREAD -DATTN64_DBG_NOLOAD -DATTN64_DBG_NOBARRIER
wait
ls tools/probes/attn64_var_*
