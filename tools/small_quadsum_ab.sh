#!/bin/bash
# usage: tools/small_quadsum_ab.sh — same-box A/B of the in-quad summation at the end of msm_small_accumulate (PLK_MSM_SMALL_QUADSUM = 0 never / 1 always /
# default: batches only): single commitments and proofs at 2^12 .. 2^15; then the short path at 2^16 (PLK_MSM_SMALL_MAX=65536) against the 2^20-shaped pipeline
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "msm" 2>&1 | tail -2
for q in 0 1 default; do
  if [ $q = default ]; then unset PLK_MSM_SMALL_QUADSUM; else export PLK_MSM_SMALL_QUADSUM=$q; fi
  echo "## PLK_MSM_SMALL_QUADSUM=$q"
  python tools/msm_size_probe.py 12 14 15 2>&1 | grep terms
  for L in 12 13 14 15; do python tools/prove_probe.py $L 30 2>&1 | grep over | cut -c1-64; done
done
unset PLK_MSM_SMALL_QUADSUM
for v in 65536 default; do
  if [ $v = default ]; then unset PLK_MSM_SMALL_MAX; else export PLK_MSM_SMALL_MAX=$v; fi
  echo "## 2^16, PLK_MSM_SMALL_MAX=$v"
  python tools/msm_size_probe.py 16 2>&1 | grep terms
  python tools/prove_probe.py 16 30 2>&1 | grep over | cut -c1-64
done
