#!/bin/bash
# PMC counters of the context kernel alone (tools/probes/ctx_time.py): where a step's ~1000 cycles go
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_WAVE32_INSTS"; do
  rm -rf /tmp/pmc_ctx
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_ctx -- python $GRAFT_REPO_ROOT/tools/probes/ctx_time.py > /dev/null 2>/tmp/pmc_err.txt || { echo "set [$set] failed: $(tail -2 /tmp/pmc_err.txt)"; continue; }
  python - "$set" <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/pmc_ctx/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter file for', sys.argv[1]); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if 't2i_ctx' in r['Kernel_Name']:
        agg[(r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', '?'))][r['Counter_Name']].append(float(r['Counter_Value']))
for g, d in sorted(agg.items(), key=lambda kv: int(kv[0]) if kv[0].isdigit() else 0):
    print('grid', g, {k: round(sum(v) / len(v)) for k, v in d.items()}, 'launches', len(next(iter(d.values()))))
PY
done
