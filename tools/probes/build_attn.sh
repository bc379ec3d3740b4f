#!/bin/bash
# Ablation builds of the attention kernel (tools/probes/attn_variants.hip): tools/probes/attn_var_<name>
cd "$(dirname "$0")/../.."
set -e
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I include -I l4p_amd/csrc "${@:2}" tools/probes/attn_variants.hip -o tools/probes/attn_var_$1 & }
build base
build noload -DATTN_DBG_NOLOAD
build noexp -DATTN_DBG_NOEXP
build nomax -DATTN_DBG_NOMAX
build nosoftmax -DATTN_DBG_NOSOFTMAX -DATTN_DBG_NOMAX
build noldsread -DATTN_DBG_NOLDSREAD
build nobarrier -DATTN_DBG_NOBARRIER
build nosm_nolds -DATTN_DBG_NOSOFTMAX -DATTN_DBG_NOMAX -DATTN_DBG_NOLDSREAD
build nosm_nolds_noload -DATTN_DBG_NOSOFTMAX -DATTN_DBG_NOMAX -DATTN_DBG_NOLDSREAD -DATTN_DBG_NOLOAD
build nosm_noload -DATTN_DBG_NOSOFTMAX -DATTN_DBG_NOMAX -DATTN_DBG_NOLOAD
wait
ls tools/probes/attn_var_*
