#!/bin/bash
# usage: tools/rl_stream_ab.sh — same-box A/B of 16 lanes per task in the bucket reduction whenever other commitments are in flight (PLK_MSM_RL_STREAM=0: 32),
# three interleaved rounds of the driver's commitment stream, then commitments by size and a proof
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for v in 0 1; do
  export PLK_MSM_RL_STREAM=$v
  python bench.py --msm-only --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('RL_STREAM=$v  bench --msm-only: %.1f M/s  %.4f ms per step  sustained %.1f' % (d['value'], d['ms_per_step'], d['value_sustained']))"
done; done
for v in 0 1; do export PLK_MSM_RL_STREAM=$v; echo "## RL_STREAM=$v"; python tools/msm_size_probe.py 16 18 20 2>&1 | grep terms; python tools/prove_probe.py 20 12 2>&1 | grep over | cut -c1-60; done
