#!/bin/bash
# run-to-run spread of the headline loop: depth 2 / 3, default steps, fresh processes
cd "$(dirname "$0")/.."
O=gpurun_out/r2q; mkdir -p $O
show='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
print("value %.1f M/s  ms_per_step %.4f  kernel_ms %.4f  pipelined %.4f  one_in_flight %.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_pipelined"], r["ms_per_step_one_in_flight"]))'
for rep in 1 2 3 4; do
  for d in 3 2; do
    echo "== depth $d steps 20" | tee -a $O/ab.txt
    timeout 300 python bench.py --msm-only --pipeline-depth $d 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
  done
done
for q in 4 16; do
  echo "== depth 3, GPU_MAX_HW_QUEUES=$q" | tee -a $O/ab.txt
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --msm-only --pipeline-depth 3 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
done
echo "== depth 3 steps 100" | tee -a $O/ab.txt
timeout 300 python bench.py --msm-only --pipeline-depth 3 --steps 100 --warmup 10 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
