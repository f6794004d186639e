#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_rounds.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python tools/prove_probe.py 20 5 2>&1 | grep prove | tee $O/prove.txt
