#!/bin/bash
# round 2, GPU call A: the whole -m gpu suite (with the new large shapes), the default bench line, and the rocprofv3
# kernel summary of the one-in-flight MSM command that roofline.kernel_ms refers to
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 ) > gpurun_out/r2a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -40 gpurun_out/r2a/pytest.log
( time timeout 900 python bench.py ) > gpurun_out/r2a/bench.log 2> gpurun_out/r2a/bench.err
tail -c 6000 gpurun_out/r2a/bench.log; tail -5 gpurun_out/r2a/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2a/prof_solo -o solo -- python bench.py --msm-only --pipeline-depth 1 --steps 20 > gpurun_out/r2a/bench_solo.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2a/prof_pipe -o pipe -- python bench.py --msm-only --steps 20 > gpurun_out/r2a/bench_pipe.log 2>&1
find gpurun_out/r2a -name "*kernel_stats.csv" | head; tail -2 gpurun_out/r2a/bench_solo.log
