#!/bin/bash
# HBM traffic of msm_accumulate on the final round-2 code: FETCH_SIZE and WRITE_SIZE in separate passes (kernel-trace only)
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r2pmc; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc_kernel.sh msm_accumulate "$c" -- python $R/bench.py --msm-only --pipeline-depth 1 --steps 8 --warmup 2 --settle-steps 0 2>&1 | tee -a $O/pmc.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc_kernel.sh msm_partition "$c" -- python $R/bench.py --msm-only --pipeline-depth 1 --steps 8 --warmup 2 --settle-steps 0 2>&1 | tee -a $O/pmc.txt
done
