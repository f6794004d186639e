#!/bin/bash
# NOTE: the knob / build variant this script exercises was an experiment of round 2 that was measured and NOT kept (profiles/r02_msm_three_in_flight_ab.txt); the script is the record of how it was run.
# accumulations ordered across slots: parity, then the headline loop at the default step count
cd "$(dirname "$0")/.."
O=gpurun_out/r2r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sharded_prove.py -m gpu -x -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 300 python tools/msm_fuzz.py 100 77 2>&1 | tail -1 | tee $O/fuzz.txt
show='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
print("value %.1f M/s  ms_per_step %.4f  kernel_ms %.4f  pipelined %.4f  one_in_flight %.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_pipelined"], r["ms_per_step_one_in_flight"]))'
for rep in 1 2 3; do
  for d in 3 2; do
    echo "== depth $d steps 20" | tee -a $O/ab.txt
    timeout 300 python bench.py --msm-only --pipeline-depth $d 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
  done
done
echo "== depth 3 steps 100" | tee -a $O/ab.txt
timeout 300 python bench.py --msm-only --pipeline-depth 3 --steps 100 --warmup 10 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
timeout 300 python tools/prove_probe.py 20 5 2>&1 | grep prove | tee -a $O/ab.txt
