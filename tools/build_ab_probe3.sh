#!/bin/bash
# bucket-reduction A/B (tools/ab_flags.sh builds): commitments one at a time / three in flight, whole prove, two proofs in flight
# usage: tools/build_ab_probe3.sh <root> [<root> ...]       ('.' = working tree)
cd "$(dirname "$0")/.."
for rep in 1 2; do for r in "$@"; do
  echo "--- $r (rep $rep) $(cat $r/FLAGS 2>/dev/null)"
  PLK_AB_ROOT=$r python tools/msm_pipeline_probe.py 2>&1 | tail -2
  PLK_AB_ROOT=$r python tools/prove_probe.py 20 24 2>&1 | tail -1
  PLK_AB_ROOT=$r python tools/prove_inflight_probe.py 20 2 10 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400
done; done
