#!/bin/bash
# NOTE: the knob / build variant this script exercises was an experiment of round 2 that was measured and NOT kept (profiles/r02_msm_three_in_flight_ab.txt); the script is the record of how it was run.
# challenge-free LDEs on a second stream of the prover: parity, then A/B (PLK_NO_AUX_STREAM=1 = one stream as before)
cd "$(dirname "$0")/.."
O=gpurun_out/r2aux; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_rounds.py tests/test_gpu_sharded_prove.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.txt
timeout 300 python tools/prove_fuzz.py 60 91 2>&1 | tail -1 | tee $O/fuzz.txt
for rep in 1 2 3; do
  for v in aux one; do
    echo "== $v" | tee -a $O/ab.txt
    if [ $v = one ]; then export PLK_NO_AUX_STREAM=1; else unset PLK_NO_AUX_STREAM; fi
    timeout 300 python tools/prove_probe.py 20 6 2>&1 | grep prove | tail -4 | tee -a $O/ab.txt
  done
done
unset PLK_NO_AUX_STREAM
for v in aux one; do
  echo "== $v 2^22 / 2^18" | tee -a $O/ab.txt
  if [ $v = one ]; then export PLK_NO_AUX_STREAM=1; else unset PLK_NO_AUX_STREAM; fi
  timeout 300 python tools/prove_probe.py 22 3 2>&1 | grep prove | tail -2 | tee -a $O/ab.txt
  timeout 300 python tools/prove_probe.py 18 6 2>&1 | grep prove | tail -2 | tee -a $O/ab.txt
done
