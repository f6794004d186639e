#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2fb; mkdir -p $O
show='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
print("value %.1f M/s  ms_per_step %.4f  kernel_ms %.4f  pipelined %.4f  one_in_flight %.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_pipelined"], r["ms_per_step_one_in_flight"]))'
for rep in 1 2; do
  for fb in 6 7; do
    echo "== fine bits $fb, warmup 5 steps 20" | tee -a $O/ab.txt
    PLK_MSM_FINE_BITS=$fb timeout 300 python bench.py --msm-only --warmup 5 --steps 20 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
    echo "== fine bits $fb, warmup 5 steps 100" | tee -a $O/ab.txt
    PLK_MSM_FINE_BITS=$fb timeout 300 python bench.py --msm-only --warmup 5 --steps 100 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
  done
done
