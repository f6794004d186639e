#!/bin/bash
# usage: tools/owned_ab.sh — same-box A/B of the bucket-owning accumulation for evenly filled tasks (msm_accumulate<.., OWNED>; PLK_MSM_OWNED_MAX=0 = the
# equal-pieces kernel for every task, as before): commitments of 2^16 .. 2^19 terms and proofs at the 2^16 .. 2^18 domains
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "msm" 2>&1 | tail -2
for v in 0 default; do
  if [ $v = default ]; then unset PLK_MSM_OWNED_MAX; else export PLK_MSM_OWNED_MAX=$v; fi
  echo "## PLK_MSM_OWNED_MAX=$v"
  python tools/msm_size_probe.py 16 17 18 19 2>&1 | grep terms
  for L in 16 17 18; do python tools/prove_probe.py $L 30 2>&1 | grep over | cut -c1-64; done
done
