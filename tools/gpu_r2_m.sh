#!/bin/bash
# SQ counters of the two VALU-bound kernels (separate rocprofv3 --pmc passes, kernel-trace only) + the general-tau SRS test
cd "$(dirname "$0")/.."
O=gpurun_out/r2m; mkdir -p $O; R=$PWD
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "general_tau or golden" 2>&1 | tail -2
SET="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
bash tools/pmc_kernel.sh msm_accumulate "$SET" -- python $R/bench.py --msm-only --pipeline-depth 1 --steps 5 --warmup 1 > $O/pmc_sq.txt 2>&1
bash tools/pmc_kernel.sh ntt_pass "$SET" -- python $R/tools/ntt_probe.py 22 3 >> $O/pmc_sq.txt 2>&1
bash tools/pmc_kernel.sh msm_task_reduce "$SET" -- python $R/bench.py --msm-only --pipeline-depth 1 --steps 5 --warmup 1 >> $O/pmc_sq.txt 2>&1
cat $O/pmc_sq.txt
