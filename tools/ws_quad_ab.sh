#!/bin/bash
# usage: tools/ws_quad_ab.sh — same-box A/B of msm_window_sums_quad (PLK_MSM_WS_QUAD=0: the lane-wise kernel): commitments of 2^16 .. 2^20 terms, proofs at 2^16 / 2^18 / 2^20
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "msm" 2>&1 | tail -2
for v in 0 default; do
  if [ $v = default ]; then unset PLK_MSM_WS_QUAD; else export PLK_MSM_WS_QUAD=$v; fi
  echo "## PLK_MSM_WS_QUAD=$v"
  python tools/msm_size_probe.py 16 18 20 2>&1 | grep terms
  for L in 16 18; do python tools/prove_probe.py $L 30 2>&1 | grep over | cut -c1-64; done
  python tools/prove_probe.py 20 12 2>&1 | grep over | cut -c1-64
done
