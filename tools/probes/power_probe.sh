#!/bin/bash
# samples rocm-smi (socket power, sclk, power cap) while the c3 bench loop runs: is the step power-limited?
cd "$(dirname "$0")/../.."
rocm-smi --showmaxpower --showpowercap 2>/dev/null | grep -E "Power|power" | head -4
( python bench.py --steps ${POWER_STEPS:-400} --warmup 5 --no-prof --no-cpu-baseline > /tmp/bench.out 2>/dev/null ) &
BP=$!
while kill -0 $BP 2>/dev/null; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 2; done | sort | uniq -c | sort -k1,1nr | head -12
tail -1 /tmp/bench.out | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
