#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-substring> "<COUNTER ...>" -- <command...>
# one rocprofv3 --pmc pass (kernel-trace only), prints the mean of each counter over the matching kernels
pat=$1; set_=$2; shift 3
R=$PWD; export TMPDIR=/tmp; d=/tmp/pmc_$$; rm -rf $d
args=(); for a in "$@"; do if [ -f "$R/$a" ]; then args+=("$R/$a"); else args+=("$a"); fi; done; set -- "${args[@]}"   # (the profiler runs from /tmp)
(cd /tmp && timeout 150 rocprofv3 --pmc $set_ --kernel-trace --output-format csv -d $d -o p -- "$@" > $d.log 2>&1)
f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no counter file; log tail:"; tail -5 $d.log; exit 1; fi
python3 - "$f" "$pat" <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"].split("(")[0]
    if sys.argv[2] in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    for c,v in d.items():
        v=sorted(v); print("%-40s %-24s n=%d min=%.4e median=%.4e mean=%.4e max=%.4e"%(k[:40], c, len(v), v[0], v[len(v)//2], sum(v)/len(v), v[-1]))
PY
rm -rf $d $d.log
