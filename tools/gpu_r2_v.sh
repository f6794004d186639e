#!/bin/bash
# NOTE: the knob / build variant this script exercises was an experiment of round 2 that was measured and NOT kept (profiles/r02_msm_three_in_flight_ab.txt); the script is the record of how it was run.
# prover batches as smaller groups in flight (PLK_COMMIT_GROUP)
cd "$(dirname "$0")/.."
O=gpurun_out/r2v; mkdir -p $O
for rep in 1 2; do
  for g in 0 2 1; do
    echo "== PLK_COMMIT_GROUP=$g" | tee -a $O/ab.txt
    PLK_COMMIT_GROUP=$g timeout 300 python tools/prove_probe.py 20 6 2>&1 | grep prove | tail -4 | tee -a $O/ab.txt
  done
done
PLK_COMMIT_GROUP=2 timeout 600 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 | tee -a $O/ab.txt
