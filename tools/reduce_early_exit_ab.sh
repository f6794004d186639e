#!/bin/bash
# usage: tools/reduce_early_exit_ab.sh — same-box A/B of the early exit of reduction waves without a live task (PLK_MSM_REDUCE_EARLY_EXIT=0: they walk the tree steps
# on identities, as in rounds 1-6): commitments by size, proofs at 2^16 / 2^18 / 2^20 (three interleaved rounds), the driver's commitment stream
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "msm" 2>&1 | tail -1
for rep in 1 2 3; do
for v in 0 1; do
  export PLK_MSM_REDUCE_EARLY_EXIT=$v
  echo -n "EARLY_EXIT=$v: prove 2^16 / 2^18 / 2^20: "; for L in 16 18 20; do python tools/prove_probe.py $L 24 2>&1 | grep over | sed 's/.*median \([0-9.]*\) ms.*/\1/' | tr '\n' ' '; done; echo
done; done
for v in 0 1; do
  export PLK_MSM_REDUCE_EARLY_EXIT=$v
  echo "## EARLY_EXIT=$v"; python tools/msm_size_probe.py 16 18 20 2>&1 | grep terms
  python bench.py --msm-only --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench --msm-only: %.1f M/s  %.4f ms per step' % (d['value'], d['ms_per_step']))"
done
