#!/bin/bash
# NTT inter-pass twiddle tables: parity tests, then A/B against composing the twiddles (PLK_NTT_DIRECT=0)
cd "$(dirname "$0")/.."
O=gpurun_out/r2n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_large.py tests/test_gpu_rounds.py tests/test_gpu_prove.py -m gpu -x -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do
  echo "== tables" | tee -a $O/ab.txt; timeout 300 python tools/ntt_ab_probe.py 16 18 20 22 24 26 2>&1 | grep "2^" | tee -a $O/ab.txt
  echo "== composed (PLK_NTT_DIRECT=0)" | tee -a $O/ab.txt; PLK_NTT_DIRECT=0 timeout 300 python tools/ntt_ab_probe.py 16 18 20 22 24 26 2>&1 | grep "2^" | tee -a $O/ab.txt
done
echo "== prove tables" | tee -a $O/ab.txt; timeout 300 python tools/prove_probe.py 20 6 2>&1 | grep prove | tee -a $O/ab.txt
echo "== prove composed" | tee -a $O/ab.txt; PLK_NTT_DIRECT=0 timeout 300 python tools/prove_probe.py 20 6 2>&1 | grep prove | tee -a $O/ab.txt
