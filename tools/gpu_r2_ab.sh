#!/bin/bash
# A/B of two builds in one call: the working tree against ab_old/ (a built export of an earlier commit), interleaved
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r2ab; mkdir -p $O
show='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
print("value %.1f M/s  ms_per_step %.4f  kernel_ms %.4f  pipelined %.4f  one_in_flight %.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_pipelined"], r["ms_per_step_one_in_flight"]))'
cd $R; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -1 | tee $O/pytest.txt
timeout 300 python tools/msm_fuzz.py 120 80 2>&1 | tail -1 | tee $O/fuzz.txt
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = new ]; then cd $R; else cd $R/ab_old; fi
    echo "== $v, warmup 5 steps 20" | tee -a $O/ab.txt
    timeout 300 python bench.py --msm-only --warmup 5 --steps 20 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
  done
done
for v in new old; do
  if [ $v = new ]; then cd $R; else cd $R/ab_old; fi
  echo "== $v, warmup 5 steps 100" | tee -a $O/ab.txt
  timeout 300 python bench.py --msm-only --warmup 5 --steps 100 2>/dev/null | python -c "$show" | tee -a $O/ab.txt
  echo "== $v, prove" | tee -a $O/ab.txt
  timeout 300 python tools/prove_probe.py 20 5 2>&1 | grep prove | tail -3 | tee -a $O/ab.txt
done
