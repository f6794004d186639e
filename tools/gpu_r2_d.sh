#!/bin/bash
# round 2, GPU call D: poseidon-shaped proves, two-phase setup through the CLI, whole-CLI timing, and the N > 1 code path
# of bench.py exercised with ONE rank over real RCCL (PLK_FORCE_GATHER: strong-scaling leg + sharded prove)
cd "$(dirname "$0")/.."
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_sharded_prove.py tests/test_gpu_rounds.py -m gpu -x -q --durations=8 ) > $O/pytest.log 2>&1; tail -14 $O/pytest.log
bash tools/cli_scale.sh 20 /tmp/cli_scale > $O/cli_scale.txt 2>&1; cat $O/cli_scale.txt
( time PLK_FORCE_GATHER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 ) > $O/bench_forced.log 2> $O/bench_forced.err
tail -c 3000 $O/bench_forced.log; tail -5 $O/bench_forced.err
