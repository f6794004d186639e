#!/bin/bash
# NOTE: the knob / build variant this script exercises was an experiment of round 2 that was measured and NOT kept (profiles/r02_msm_three_in_flight_ab.txt); the script is the record of how it was run.
# high-priority side stream per slot for the short MSM kernels: parity, then the headline loop as the driver runs it
cd "$(dirname "$0")/.."
O=gpurun_out/r2t2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sharded_prove.py -m gpu -x -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 300 python tools/msm_fuzz.py 100 78 2>&1 | tail -1 | tee $O/fuzz.txt
show='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
print("value %.1f M/s  ms_per_step %.4f  kernel_ms %.4f  pipelined %.4f  one_in_flight %.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_pipelined"], r["ms_per_step_one_in_flight"]))'
for rep in 1 2 3; do
  for side in 1 0; do
    echo "== side stream $side, depth 3, warmup 5 steps 20" | tee -a $O/ab.txt
    PLK_MSM_SIDE_STREAM=$side timeout 300 python bench.py --msm-only --warmup 5 --steps 20 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
  done
done
for side in 1 0; do
  echo "== side stream $side, depth 3, warmup 30 steps 100" | tee -a $O/ab.txt
  PLK_MSM_SIDE_STREAM=$side timeout 300 python bench.py --msm-only --warmup 30 --steps 100 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
  echo "== side stream $side, depth 2, warmup 5 steps 20" | tee -a $O/ab.txt
  PLK_MSM_SIDE_STREAM=$side timeout 300 python bench.py --msm-only --pipeline-depth 2 --warmup 5 --steps 20 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
  echo "== side stream $side, prove" | tee -a $O/ab.txt
  PLK_MSM_SIDE_STREAM=$side timeout 300 python tools/prove_probe.py 20 5 2>&1 | grep prove | tail -3 | tee -a $O/ab.txt
done
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_side -o side -- python $GRAFT_REPO_ROOT/bench.py --msm-only --warmup 5 --steps 20 > /dev/null 2>&1
