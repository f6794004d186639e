#!/bin/bash
# three commitments in flight: parity (kernels + sharded + fuzz), then depth 2 vs 3 in the bench loop, interleaved
cd "$(dirname "$0")/.."
O=gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sharded_prove.py tests/test_gpu_prove.py -m gpu -x -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/msm_fuzz.py 150 33 2>&1 | tail -1 | tee $O/fuzz.txt
for rep in 1 2 3; do
  for d in 2 3; do
    echo "== depth $d" | tee -a $O/ab.txt
    timeout 300 python bench.py --msm-only --pipeline-depth $d --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.1f M/s  ms_per_step %.4f  kernel_ms %.4f  pipelined %.4f  one_in_flight %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_pipelined'], r['ms_per_step_one_in_flight']))" | tee -a $O/ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_pipe3 -o pipe3 -- python $GRAFT_REPO_ROOT/bench.py --msm-only --pipeline-depth 3 --steps 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof_pipe3 -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} $O/pipe3_kernel_stats.csv; head -12 $O/pipe3_kernel_stats.csv | cut -c1-150
