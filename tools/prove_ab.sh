#!/bin/bash
# same-box A/B of a whole proof: working tree against ab_old (tools/ab_build.sh <commit>, built in the container), sparse and dense circuit,
# interleaved.   usage (GPU box): tools/prove_ab.sh <tag> [extra env for the NEW arm, e.g. PLK_MSM_RL_LOG=5]
out=gpurun_out/$1; mkdir -p $out; f=$out/prove_ab.txt
for rep in 1 2 3; do
  for lc in 0 7; do
    echo "== rep $rep lc_terms=$lc old" >> $f; PLK_AB_ROOT=ab_old timeout 120 python tools/prove_probe.py 20 15 $lc 2>&1 | tail -1 >> $f
    echo "== rep $rep lc_terms=$lc new" >> $f; timeout 120 python tools/prove_probe.py 20 15 $lc 2>&1 | tail -1 >> $f
    if [ -n "$2" ]; then echo "== rep $rep lc_terms=$lc new $2" >> $f; env $2 timeout 120 python tools/prove_probe.py 20 15 $lc 2>&1 | tail -1 >> $f; fi
  done
done
cat $f
