#!/bin/bash
# usage: tools/small_ch_ab.sh — terms per workgroup of msm_small_accumulate (PLK_MSM_SMALL_CH = 64 / 128 / 256 / default rule), same box: commitments and proofs
cd "$(dirname "$0")/.."
for c in 64 128 256 default; do
  if [ $c = default ]; then unset PLK_MSM_SMALL_CH; else export PLK_MSM_SMALL_CH=$c; fi
  echo "## PLK_MSM_SMALL_CH=$c"
  python tools/msm_size_probe.py 12 14 15 2>&1 | grep terms
  for L in 12 14 15; do python tools/prove_probe.py $L 30 2>&1 | grep over | cut -c1-64; done
done
