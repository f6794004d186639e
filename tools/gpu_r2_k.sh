#!/bin/bash
# round 2, GPU call K: window sums as twelve plain tree sums (bit roles): MSM parity, latency of one commitment, prove
cd "$(dirname "$0")/.."
O=gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "msm or batch or flight" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
PLK_MSM_FINE_BITS=7 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "trapdoor or distributions or matches_oracle" 2>&1 | tail -2
timeout 300 python tools/msm_pipeline_probe.py 2>&1 | grep timing | tee $O/msm.txt
timeout 300 python tools/msm_dist_probe.py 2>&1 | grep ms | tee -a $O/msm.txt
timeout 300 python tools/prove_probe.py 20 4 2>&1 | grep prove | tee $O/prove.txt
timeout 300 python tools/prove_probe.py 22 2 2>&1 | grep prove | tee -a $O/prove.txt
