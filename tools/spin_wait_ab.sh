#!/bin/bash
cd "$(dirname "$0")/.."
# usage: tools/spin_wait_ab.sh — proofs at 2^10 / 2^12 / 2^16 / 2^20 with and without ROC_ACTIVE_WAIT_TIMEOUT=2000 (the HIP runtime spinning on completion signals), twice
for rep in 1 2; do
for v in default 2000; do
  if [ $v = default ]; then unset ROC_ACTIVE_WAIT_TIMEOUT; else export ROC_ACTIVE_WAIT_TIMEOUT=$v; fi
  echo -n "ROC_ACTIVE_WAIT_TIMEOUT=$v: "; for L in 10 12 16 20; do python tools/prove_probe.py $L 30 2>&1 | grep over | sed 's/.*median \([0-9.]*\) ms.*/\1/' | tr '\n' ' '; done; echo
done; done
