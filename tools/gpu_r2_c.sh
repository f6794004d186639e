#!/bin/bash
# round 2, GPU call C: whole -m gpu suite after the host-side rework (flat R1CS + parallel loaders, chunked transpiler,
# device-built permutation, kernel-level round tests, built-in combiner tests), then the whole-CLI timing at 2^20
cd "$(dirname "$0")/.."
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest.log 2>&1; tail -22 $O/pytest.log
bash tools/cli_scale.sh 20 /tmp/cli_scale > $O/cli_scale.txt 2>&1; cat $O/cli_scale.txt
