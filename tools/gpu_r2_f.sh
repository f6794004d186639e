#!/bin/bash
# where does `plonkit prove` spend the 0.2 s between its last phase line and the end of the process?
cd "$(dirname "$0")/.."
O=gpurun_out/r2f; mkdir -p $O; D=/tmp/cli_scale
bash tools/cli_scale.sh 20 $D > $O/cli_scale.txt 2>&1
CLI=plonkit_amd/lib/plonkit
run() { local s=$(date +%s%N); "$@" > /dev/null 2>> $O/err.txt; local e=$(date +%s%N); python3 -c "print('%-40s %.3f s' % ('$TAG', ($e - $s) / 1e9))"; }
A="prove -m $D/key.bin -c $D/circuit.r1cs -w $D/witness.wtns -p $D/p.bin -j $D/pj.json -i $D/ij.json --overwrite"
for i in 1 2; do TAG="prove"; run $CLI $A; done
TAG="prove, witness not page-locked"; PLK_NO_HOST_REGISTER=1 run $CLI $A
TAG="prove, explicit frees"; PLK_CLI_FREE=1 PLK_CLI_TIMING=1 $CLI $A 2>&1 | grep -E "free|files"; PLK_CLI_FREE=1 run $CLI $A
TAG="export-verification-key"; run $CLI export-verification-key -m $D/key.bin -c $D/circuit.r1cs -v $D/vk2.bin --overwrite
which strace perf ltrace 2>&1 | head -3
