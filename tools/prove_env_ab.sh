#!/bin/bash
# usage: tools/prove_env_ab.sh <tag> <ENV_VAR> <kernel-substring> [log_n=20] — same-box A/B of one knob inside a proof: VAR=0 against VAR=1, interleaved twice;
# prints the proof's median and the mean duration of the named kernel from a rocprofv3 kernel trace of each arm
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p "$O"; V=$2; K=$3; L=${4:-20}
export TMPDIR=/tmp
{
for rep in 1 2; do
  for w in 0 1; do
    echo "## $V=$w (run $rep)"
    env $V=$w python tools/prove_probe.py $L 12 2>&1 | grep "over"
    d=/tmp/pab_$$_$rep$w
    (cd /tmp && env $V=$w rocprofv3 --kernel-trace -d $d -o p -- python $OLDPWD/tools/prove_probe.py $L 6 > $d.log 2>&1)
    python tools/rocpd_stats.py $d/p_results.db 2>/dev/null | grep -i "$K" | awk -F, '{printf "   %s calls %s avg %.1f us\n", substr($1,1,60), $2, $4/1000}'
    rm -rf $d $d.log
  done
done
} | tee "$O/prove_${V}_ab.txt"
