#!/bin/bash
# usage: tools/g1ntt_iso_ab.sh — same-box A/B of the effectively affine window table of the G1 iNTT (PLK_G1NTT_ISO=0: XYZZ table, full additions; 1: four effectively affine entries; 2: eight, 4-bit windows), twice
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "lagrange or dump" 2>&1 | tail -2
for rep in 1 2; do for v in 0 1 2; do echo "## PLK_G1NTT_ISO=$v"; PLK_G1NTT_ISO=$v python tools/g1intt_probe.py 8 13 16 20 2>&1 | grep g1_intt; done; done
