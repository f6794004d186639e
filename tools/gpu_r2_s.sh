#!/bin/bash
# is the slow start of the headline loop a clock ramp?  same loop, different warm-up lengths; clocks sampled meanwhile
cd "$(dirname "$0")/.."
O=gpurun_out/r2s2; mkdir -p $O
show='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
print("value %.1f M/s  ms_per_step %.4f  kernel_ms %.4f  pipelined %.4f  one_in_flight %.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_pipelined"], r["ms_per_step_one_in_flight"]))'
for cfg in "3 20" "30 20" "100 20" "30 100" "3 20" "30 20" "300 100"; do
  set -- $cfg
  echo "== depth 3 warmup $1 steps $2" | tee -a $O/ab.txt
  timeout 300 python bench.py --msm-only --pipeline-depth 3 --warmup $1 --steps $2 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
done
# clocks and power under the sustained loop
( for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/smi.txt &
timeout 300 python bench.py --msm-only --pipeline-depth 3 --warmup 300 --steps 3000 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
wait
cat $O/smi.txt | cut -c1-220
