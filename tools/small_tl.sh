#!/bin/bash
# usage: tools/small_tl.sh <tag> — kernel timelines (rocprofv3 --kernel-trace): one short commitment at 2^12 / 2^14 / 2^15 terms, then the last of six warm proofs at 2^12 and 2^14
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p "$O"
export TMPDIR=/tmp
for L in 12 14 15; do
  d=/tmp/stl_$$_$L
  (cd /tmp && rocprofv3 --kernel-trace -d $d -o p -- python $OLDPWD/tools/msm_size_probe.py $L > $d.log 2>&1)
  echo "# 2^$L terms: $(grep terms $d.log)"; python tools/small_timeline.py $d/p_results.db 10 5
  rm -rf $d $d.log
done | tee "$O/msm_small_timeline.txt"
for L in 12 14; do
  d=/tmp/tl_$$_$L
  (cd /tmp && rocprofv3 --kernel-trace -d $d -o p -- python $OLDPWD/tools/prove_probe.py $L 6 > $d.log 2>&1)
  python tools/timeline.py $d/p_results.db > "$O/prove_timeline_2pow$L.txt"; tail -1 "$O/prove_timeline_2pow$L.txt"
  rm -rf $d $d.log
done
