#!/bin/bash
# repeated whole-CLI timings, A/B inside one call (boxes differ by up to 2x in host speed): witness page-locked at the
# first proof ("always") or only from the second proof of a circuit object on (default)
cd "$(dirname "$0")/.."
O=gpurun_out/r2g; mkdir -p $O; D=/tmp/cli_scale
bash tools/cli_scale.sh 20 $D > $O/cli_scale.txt 2>&1; grep -E "timing|whole|export" $O/cli_scale.txt
CLI=plonkit_amd/lib/plonkit
A="prove -m $D/key.bin -c $D/circuit.r1cs -w $D/witness.wtns -p $D/p.bin -j $D/pj.json -i $D/ij.json --overwrite"
python3 - <<PY > $O/repeat.txt
import subprocess, time, statistics, os
def t(cmd, env=None, n=8):
    xs=[]
    for _ in range(n):
        s=time.perf_counter(); subprocess.run(cmd, shell=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, **(env or {}))); xs.append(time.perf_counter()-s)
    return xs
rows = (("prove, witness page-locked at once", "$CLI $A", {"PLK_HOST_REGISTER": "always"}),
        ("prove (default: from the 2nd proof)", "$CLI $A", None),
        ("prove, witness page-locked at once", "$CLI $A", {"PLK_HOST_REGISTER": "always"}),
        ("prove (default: from the 2nd proof)", "$CLI $A", None),
        ("export-verification-key", "$CLI export-verification-key -m $D/key.bin -c $D/circuit.r1cs -v $D/vk2.bin --overwrite", None),
        ("verify", "$CLI verify -p $D/p.bin -v $D/vk2.bin", None))
for tag, cmd, env in rows:
    xs=t(cmd, env)
    print("%-38s median %.3f s  min %.3f  max %.3f  (8 runs)" % (tag, statistics.median(xs), min(xs), max(xs)))
PY
cat $O/repeat.txt; nproc; cat /proc/loadavg
