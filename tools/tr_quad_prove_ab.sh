#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for v in 0 1; do
  export PLK_MSM_TR_QUAD=$v
  echo -n "TR_QUAD=$v: "; for L in 16 18 20; do python tools/prove_probe.py $L 24 2>&1 | grep over | sed 's/.*median \([0-9.]*\) ms.*/\1/' | tr '\n' ' '; done; echo
done; done
