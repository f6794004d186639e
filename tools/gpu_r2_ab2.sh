#!/bin/bash
# A/B of two builds in one call (working tree against ab_old/): NTT timings and prove, interleaved; parity first
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r2ab2; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_large.py tests/test_gpu_rounds.py -m gpu -x -q -p no:cacheprovider -k "ntt or lde or prove or round or trace or grand" 2>&1 | tail -1 | tee $O/pytest.txt
timeout 300 python tools/prove_fuzz.py 30 81 2>&1 | tail -1 | tee $O/fuzz.txt
for rep in 1 2; do
  for v in new old; do
    if [ $v = new ]; then cd $R; else cd $R/ab_old; fi
    echo "== $v" | tee -a $O/ab.txt
    timeout 300 python tools/ntt_ab_probe.py 16 18 20 22 24 2>&1 | grep "2^" | tee -a $O/ab.txt
    timeout 300 python tools/prove_probe.py 20 5 2>&1 | grep prove | tail -3 | tee -a $O/ab.txt
  done
done
