#!/bin/bash
# usage: tools/ntt_wave_ab.sh <tag>   — same-box A/B of the wave-owned NTT passes (PLK_NTT_WAVE=0: barrier-per-round kernels), interleaved twice
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p "$O"
{
for rep in 1 2; do
  for w in 0 1; do echo "## PLK_NTT_WAVE=$w (run $rep)"; PLK_NTT_WAVE=$w python tools/ntt_ab_probe.py 16 18 20 22 24 2>&1 | grep -v amdgpu.ids; done
done
} | tee "$O/ntt_wave_ab.txt"
