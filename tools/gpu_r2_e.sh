#!/bin/bash
# round 2, GPU call E: whole -m gpu suite, whole-CLI timing (huge pages, two-phase setup on both commands), the N > 1 bench
# path with the library's own RCCL exchange (one rank, forced), HBM traffic counters of msm_accumulate, the default bench line
cd "$(dirname "$0")/.."
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
cat /sys/kernel/mm/transparent_hugepage/enabled > $O/thp.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest.log 2>&1; tail -12 $O/pytest.log
bash tools/cli_scale.sh 20 /tmp/cli_scale > $O/cli_scale.txt 2>&1; cat $O/thp.txt $O/cli_scale.txt
( time PLK_FORCE_GATHER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline ) > $O/bench_forced.log 2> $O/bench_forced.err
tail -c 1500 $O/bench_forced.log; tail -3 $O/bench_forced.err
for c in FETCH_SIZE WRITE_SIZE; do bash tools/pmc_kernel.sh msm_accumulate $c -- python bench.py --msm-only --pipeline-depth 1 --steps 5 --warmup 1 >> $O/pmc_traffic.txt 2>&1; done
cat $O/pmc_traffic.txt
( time timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -c 2500 $O/bench.log
