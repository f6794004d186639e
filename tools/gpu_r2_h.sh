#!/bin/bash
# A/B: msm_accumulate compiled for one wave per SIMD (298 registers, no spill, one workgroup per CU) against the default
# (256 registers, two workgroups per CU); HBM traffic counters of the default kernel
cd "$(dirname "$0")/.."
O=gpurun_out/r2h; mkdir -p $O; R=$PWD
export TMPDIR=/tmp
for v in default onewave default onewave; do
  echo "== $v" >> $O/ab.txt
  if [ $v = onewave ]; then export PLK_MSM_ONE_WAVE=1; else unset PLK_MSM_ONE_WAVE; fi
  timeout 300 python tools/msm_pipeline_probe.py 2>&1 | grep timing >> $O/ab.txt
  timeout 300 python tools/prove_probe.py 20 3 2>&1 | grep prove | tail -2 >> $O/ab.txt
done
unset PLK_MSM_ONE_WAVE
cat $O/ab.txt
PLK_MSM_ONE_WAVE=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "msm" 2>&1 | tail -2
for c in FETCH_SIZE WRITE_SIZE; do bash tools/pmc_kernel.sh msm_accumulate $c -- python $R/bench.py --msm-only --pipeline-depth 1 --steps 5 --warmup 1 >> $O/pmc_traffic.txt 2>&1; done
cat $O/pmc_traffic.txt
