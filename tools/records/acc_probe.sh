R=$PWD; cd /tmp; export TMPDIR=/tmp
for c in 3 5; do for d in 0 1 3; do
PLK_MSM_COPIES=$c PLK_MSM_DEBUG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_${c}_$d -o p -- python $R/tools/records/msm_c_probe.py 21 > /tmp/o.log 2>&1
f=$(find /tmp/p_${c}_$d -name "*kernel_stats*")
python3 - $f $c $d <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "msm_accumulate" in r["Name"]: print("copies=%s debug=%s accumulate avg %.1f us"%(sys.argv[2], sys.argv[3], float(r["AverageNs"])/1e3))
PY
done; done
