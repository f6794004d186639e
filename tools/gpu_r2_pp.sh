#!/bin/bash
# NOTE: the prover-side variant this script exercised (batches as pipelined 2^20-term pieces) was measured and NOT kept (profiles/r02_msm_three_in_flight_ab.txt); the plk_msm_g1 side was.
# prover commitments of >= 2^23 terms as pipelined 2^20-term pieces: parity at every tier, then the large domains
cd "$(dirname "$0")/.."
O=gpurun_out/r2pp; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_large.py tests/test_gpu_sharded_prove.py tests/test_gpu_rounds.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $O/pytest.txt
timeout 300 python tools/prove_fuzz.py 40 92 2>&1 | tail -1 | tee $O/fuzz.txt
for l in 23 24; do PROBE_VERIFY=1 timeout 400 python tools/prove_probe.py $l 2 2>&1 | grep -E "prove|verifier" | tail -2 | tee -a $O/large.txt; done
PROBE_VERIFY=1 timeout 600 python tools/prove_probe.py 26 1 2>&1 | grep -E "prove|verifier" | tail -2 | tee -a $O/large.txt
