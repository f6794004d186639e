#!/bin/bash
# does the GPU need sustained load before the short timed region?  W=5, K=20 with 0 / 50 / 150 / 400 commitments before the warm-up
cd "$(dirname "$0")/.."
O=gpurun_out/r2w; mkdir -p $O
show='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
print("value %.1f M/s  ms_per_step %.4f  kernel_ms %.4f  pipelined %.4f  one_in_flight %.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_pipelined"], r["ms_per_step_one_in_flight"]))'
for rep in 1 2; do
  for st in 0 50 150 400; do
    echo "== settle $st, warmup 5 steps 20" | tee -a $O/ab.txt
    timeout 300 python bench.py --msm-only --settle-steps $st --warmup 5 --steps 20 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
  done
done
