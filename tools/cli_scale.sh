#!/bin/bash
# whole-CLI timing at the 2^log_n domain: setup key, export a synthetic circuit in the reference's binary formats,
# export-verification-key, prove, verify — every step through plonkit_amd/lib/plonkit (C ABI only)
set -e
L=${1:-20}; D=${2:-/tmp/cli_scale}; mkdir -p $D; cd $(dirname $0)/..
CLI=plonkit_amd/lib/plonkit
t() { local s=$(date +%s%N); "$@" 2>/dev/null; local e=$(date +%s%N); printf "%-30s %8.3f s\n" "$TAG" $(python3 -c "print(($e - $s) / 1e9)"); }
python3 - <<PY
import sys, time; sys.path.insert(0, ".")
import plonkit_amd as pa
t0 = time.time(); c = pa.Circuit.synthetic((1 << $L) - 2); t1 = time.time()
open("$D/circuit.r1cs", "wb").write(c.export("r1cs")); open("$D/witness.wtns", "wb").write(c.export("wtns"))
print("%-28s %8.3f s  (+ %.3f s writing %d MB)" % ("synthetic circuit (host)", t1 - t0, time.time() - t1, (len(c.export("r1cs")) + len(c.export("wtns"))) >> 20))
PY
TAG="setup -p $L (crs_42 on GPU)"; t $CLI setup -p $L -m $D/key.bin --overwrite
TAG="export-verification-key"; t $CLI export-verification-key -m $D/key.bin -c $D/circuit.r1cs -v $D/vk.bin --overwrite
TAG="prove (whole CLI)"; PLK_CLI_TIMING=1 $CLI prove -m $D/key.bin -c $D/circuit.r1cs -w $D/witness.wtns -p $D/proof_t.bin -j $D/proof.json -i $D/public.json --overwrite 2>&1 | grep timing; t $CLI prove -m $D/key.bin -c $D/circuit.r1cs -w $D/witness.wtns -p $D/proof.bin -j $D/proof.json -i $D/public.json --overwrite
TAG="verify"; t $CLI verify -p $D/proof.bin -v $D/vk.bin
ls -la $D | awk '{print $5, $9}' | tail -6
