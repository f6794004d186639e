#!/bin/bash
# same-box comparison of several builds (tools/ab_flags.sh / tools/ab_build.sh): accumulate kernel, NTT shapes, whole prove
# usage: tools/build_ab_probe.sh <root> [<root> ...]       ('.' = working tree)
cd "$(dirname "$0")/.."
python tools/msm_kernel_ab_probe.py "$@"
for rep in 1 2; do for r in "$@"; do
  echo "--- $r (rep $rep)"; PLK_AB_ROOT=$r python tools/ntt_ab_probe.py 20 22 2>&1 | grep "2^"; PLK_AB_ROOT=$r python tools/prove_probe.py 20 24 2>&1 | tail -1
done; done
