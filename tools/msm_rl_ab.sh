#!/bin/bash
# same-box A/B of the lanes-per-task of the bucket reduction (PLK_MSM_RL_LOG: default = 5, or 4 for batches >= 3): prove at 2^20 and the
# one-in-flight commitment, interleaved.   usage (GPU box): tools/msm_rl_ab.sh <tag>
out=gpurun_out/$1; mkdir -p $out
for rep in 1 2 3; do
  for rl in 0 5 4; do
    echo "== rep $rep PLK_MSM_RL_LOG=$rl" >> $out/rl_ab.txt
    PLK_MSM_RL_LOG=$rl PROBE_VERIFY=1 timeout 120 python tools/prove_probe.py 20 15 2>&1 | tail -2 >> $out/rl_ab.txt
    PLK_MSM_RL_LOG=$rl timeout 120 python bench.py --msm-only --pipeline-depth 1 --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('one in flight ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])" >> $out/rl_ab.txt
  done
done
cat $out/rl_ab.txt
