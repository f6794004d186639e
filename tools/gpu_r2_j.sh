#!/bin/bash
# round 2, GPU call J: the point-wise kernels of rounds 2, 4, 5 on the 29-bit layer: round-level and proof-level parity, timing
cd "$(dirname "$0")/.."
O=gpurun_out/r2j; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_rounds.py tests/test_gpu_prove.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 300 python tools/prove_probe.py 20 4 2>&1 | grep prove | tee $O/prove.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_prove -o prove -- python tools/prove_probe.py 20 6 > $O/prove_prof.log 2>&1
python tools/rocpd_stats.py $O/prof_prove/prove_results.db | awk -F'",' '{print $1","$2}' | cut -c1-60,150-260 | grep -E "k_|fill_pow" | head -20
