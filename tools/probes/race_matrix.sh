#!/bin/bash
# After the fix: the library's pointmap kernel beside the strongest aggressors.
O=gpurun_out/race
mkdir -p $O
run() { name=$1; shift; echo "== $name: $*" ; ( timeout 900 env "$@" ) > $O/$name.txt 2>&1; grep -E "launches whose|Error|error|rep " $O/$name.txt | cut -c1-300; }
P="python tools/probes/race_probe.py"
run fixed_vs_mfma_f16 A=1 $P --victim real --aggressor mfma:2 --reps 10
run fixed_vs_mfma_bf16 A=1 $P --victim real --aggressor mfma:0 --reps 10
run fixed_vs_tracker A=1 $P --victim real --aggressor tracker --reps 20
run orig_asm_vs_mfma_f16 A=1 $P --victim asm:orig --aggressor mfma:2 --reps 4
