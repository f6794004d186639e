#!/bin/bash
# round 2, GPU call B: correctness of the reworked MSM reduction (64 / 128 buckets per task, bin folding, 5-role window sums)
# and of the lazy NTT butterfly, then A/B timings: fine bits 6 vs 7, scalar distributions, NTT sizes, prove kernel profile
cd "$(dirname "$0")/.."
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prove.py -m gpu -x -q ) > $O/pytest_fb6.log 2>&1; tail -3 $O/pytest_fb6.log
( time PLK_MSM_FINE_BITS=7 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k msm ) > $O/pytest_fb7.log 2>&1; tail -3 $O/pytest_fb7.log
for fb in 6 7; do
  echo "== fine bits $fb" >> $O/ab.txt
  PLK_MSM_FINE_BITS=$fb timeout 300 python tools/msm_pipeline_probe.py >> $O/ab.txt 2>&1
  PLK_MSM_FINE_BITS=$fb timeout 300 python tools/msm_dist_probe.py >> $O/ab.txt 2>&1
  PLK_MSM_FINE_BITS=$fb timeout 300 python tools/prove_probe.py 20 3 >> $O/ab.txt 2>&1
done
for l in 20 22 24; do timeout 120 python tools/ntt_probe.py $l 20 >> $O/ab.txt 2>&1; done
cat $O/ab.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_prove -o prove -- python tools/prove_probe.py 20 6 > $O/prove_prof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_pipe -o pipe -- python bench.py --msm-only --steps 20 > $O/bench_pipe.log 2>&1
tail -1 $O/bench_pipe.log | cut -c1-400
