#!/bin/bash
# usage: tools/ntt_env_ab.sh <tag> <ENV_VAR> [sizes...]  — same-box A/B of one NTT knob (VAR=0 against VAR=1), interleaved twice; bit-exactness first
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p "$O"; V=$2; shift 2
SIZES=${@:-"18 20 22 24"}
{
for rep in 1 2; do
  for w in 0 1; do echo "## $V=$w (run $rep)"; env $V=$w python tools/ntt_ab_probe.py $SIZES 2>&1 | grep -v amdgpu.ids; done
done
} | tee "$O/ntt_${V}_ab.txt"
