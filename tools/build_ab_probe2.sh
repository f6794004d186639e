#!/bin/bash
# per-file scheduling A/B (tools/ab_flags.sh with PLK_HIPCC_EXTRA_FILES): G1 iNTT at 2^18, NTT shapes, whole prove
cd "$(dirname "$0")/.."
for rep in 1 2; do for r in "$@"; do
  echo "--- $r (rep $rep) $(cat $r/FLAGS 2>/dev/null)"
  PLK_AB_ROOT=$r python tools/g1intt_probe.py 18 2>&1 | tail -1
  PLK_AB_ROOT=$r python tools/ntt_ab_probe.py 20 22 2>&1 | grep "2^"
  PLK_AB_ROOT=$r python tools/prove_probe.py 20 24 2>&1 | tail -1
done; done
