#!/bin/bash
# NOTE: the knob / build variant this script exercises was an experiment of round 2 that was measured and NOT kept (profiles/r02_msm_three_in_flight_ab.txt); the script is the record of how it was run.
# how many commitments in flight?  the headline loop as the driver runs it (warm-up 5, 20 steps) and in steady state
cd "$(dirname "$0")/.."
O=gpurun_out/r2u; mkdir -p $O
show='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
print("value %.1f M/s  ms_per_step %.4f  kernel_ms %.4f  pipelined %.4f  one_in_flight %.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_pipelined"], r["ms_per_step_one_in_flight"]))'
for rep in 1 2; do
  for d in 3 4 5 2; do
    echo "== depth $d, warmup 5 steps 20" | tee -a $O/ab.txt
    timeout 300 python bench.py --msm-only --pipeline-depth $d --warmup 5 --steps 20 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
  done
done
for d in 3 4 5; do
  echo "== depth $d, warmup 30 steps 100" | tee -a $O/ab.txt
  timeout 300 python bench.py --msm-only --pipeline-depth $d --warmup 30 --steps 100 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
done
