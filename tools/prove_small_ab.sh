#!/bin/bash
# usage: tools/prove_small_ab.sh <tag> — same-box A/B of the short-commitment path inside proofs: PLK_MSM_SMALL_MAX=0 (the 2^20-shaped pipeline for every
# commitment, below 4096 terms the per-term double-and-add: rounds 1-5) against the default, domains 2^8 .. 2^16, median of 20 warm proofs each; then the kernel
# timelines of one 2^12 and one 2^16 proof (rocprofv3 --kernel-trace)
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p "$O"
export TMPDIR=/tmp
{
for L in 8 10 11 12 13 14 15 16; do
  for v in 0 default; do
    if [ $v = 0 ]; then export PLK_MSM_SMALL_MAX=0; else unset PLK_MSM_SMALL_MAX; fi
    echo -n "2^$L PLK_MSM_SMALL_MAX=$v: "; python tools/prove_probe.py $L 20 2>&1 | grep over | sed 's/.*over 20: //; s/  rounds.*//'
  done
done
unset PLK_MSM_SMALL_MAX
} | tee "$O/small_proofs_ab.txt"
for L in 12 16; do
  d=/tmp/tl_$$_$L
  (cd /tmp && rocprofv3 --kernel-trace -d $d -o p -- python $OLDPWD/tools/prove_probe.py $L 6 > $d.log 2>&1)
  python tools/timeline.py $d/p_results.db > "$O/prove_timeline_2pow$L.txt"; tail -1 "$O/prove_timeline_2pow$L.txt"
  rm -rf $d $d.log
done
