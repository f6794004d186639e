#!/bin/bash
# end of round 2: the whole -m gpu suite, smoke, the default bench line, the rocprofv3 summaries the line refers to
# (one commitment in flight / three in flight), the prove profile, fuzz + soak
cd "$(dirname "$0")/.."
O=gpurun_out/r2z; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest.log 2>&1; tail -14 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err; tail -c 1200 $O/bench.log | head -c 600; echo
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_solo -o solo -- python bench.py --msm-only --pipeline-depth 1 --steps 20 --warmup 5 > $O/bench_solo.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_pipe -o pipe -- python bench.py --msm-only --steps 20 --warmup 5 > $O/bench_pipe.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_prove -o prove -- python tools/prove_probe.py 20 6 > $O/prove_prof.log 2>&1
for t in solo pipe prove; do python tools/rocpd_stats.py $O/prof_$t/${t}_results.db $O/${t}_kernel_stats.csv; done
grep -h "msm_accumulate" $O/solo_kernel_stats.csv $O/pipe_kernel_stats.csv | cut -c1-40,150-260
tail -1 $O/bench_solo.log | cut -c1-300
timeout 600 python tools/msm_fuzz.py 120 5 2>&1 | tail -2 | tee $O/fuzz.txt
timeout 600 python tools/prove_fuzz.py 40 9 2>&1 | tail -2 | tee -a $O/fuzz.txt
timeout 600 python tools/soak.py 2>&1 | tail -3 | tee -a $O/fuzz.txt
