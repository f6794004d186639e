#!/bin/bash
# round 2, GPU call I: the two-rank control flow of bench.py on one GPU, then the default bench line with the
# BASELINE.md rows B2 / B3 / B5 added to cpu_baseline
cd "$(dirname "$0")/.."
O=gpurun_out/r2i; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_sharded_prove.py -m gpu -x -q --durations=5 ) > $O/pytest.log 2>&1; tail -12 $O/pytest.log
( time timeout 1200 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -c 4500 $O/bench.log; tail -5 $O/bench.err
