#!/bin/bash
# long differential fuzz + determinism soak on the final code of round 2
cd "$(dirname "$0")/.."
O=gpurun_out/r2s; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_prove.py tests/test_gpu_sharded_prove.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1 | tee $O/pytest.txt
timeout 900 python tools/msm_fuzz.py 800 2026 2>&1 | tail -1 | tee $O/soak.txt
PLK_MSM_FINE_BITS=7 timeout 600 python tools/msm_fuzz.py 300 7 2>&1 | tail -1 | tee -a $O/soak.txt
timeout 900 python tools/prove_fuzz.py 250 2026 2>&1 | tail -1 | tee -a $O/soak.txt
timeout 600 python tools/soak.py 20 40 2>&1 | tail -1 | tee -a $O/soak.txt
timeout 600 python tools/soak.py 16 300 2>&1 | tail -1 | tee -a $O/soak.txt
