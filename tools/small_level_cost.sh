#!/bin/bash
# usage: tools/small_level_cost.sh — cost of one tree level of the four-lane addition: msm_small_planes with 14 extra levels that add the identity
# (PLK_MSM_SMALL_EXTRA=14) against the plain kernel, from a rocprofv3 kernel trace of tools/msm_size_probe.py 12
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for x in 0 14; do
  d=/tmp/lvl_$$_$x
  (cd /tmp && PLK_MSM_SMALL_EXTRA=$x rocprofv3 --kernel-trace --stats -d $d -o p -- python $OLDPWD/tools/msm_size_probe.py 12 > $d.log 2>&1)
  echo "## PLK_MSM_SMALL_EXTRA=$x"; grep terms $d.log
  python tools/rocpd_stats.py $d/p_results.db /tmp/lvl_$x.csv > /dev/null; grep "msm_small" /tmp/lvl_$x.csv | cut -d, -f1-4 | sed 's/(.*)"/"/'
  rm -rf $d $d.log
done
