#!/bin/bash
# usage: tools/ntt_pmc.sh <tag> [log_n]   — fresh SQ counters of the NTT passes (one rocprofv3 --pmc pass per counter group,
# kernel trace only) plus a --kernel-trace --stats pass of the same command; results under gpurun_out/<tag>/
cd "$(dirname "$0")/.."
TAG=$1; LOGN=${2:-22}
O=gpurun_out/$TAG; mkdir -p "$O"
export TMPDIR=/tmp
{
echo "# tools/ntt_pmc.sh $TAG $LOGN  ($(git rev-parse --short HEAD 2>/dev/null || echo snapshot))  python tools/ntt_probe.py $LOGN 20"
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES_EQ_64" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  echo "## $grp"
  tools/pmc_kernel.sh ntt_pass "$grp" -- python tools/ntt_probe.py $LOGN 20
done
echo "## kernel-trace --stats"
d=/tmp/nttstats_$$
(cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o n -- python $PWD/tools/ntt_probe.py $LOGN 20 > $d.log 2>&1)
python tools/rocpd_stats.py $d/n_results.db "$O/ntt_${LOGN}_kernel_stats.csv" >/dev/null 2>&1 && grep -i ntt_pass "$O/ntt_${LOGN}_kernel_stats.csv" | cut -c1-240
echo "## unprofiled"
python tools/ntt_probe.py $LOGN 50
} 2>&1 | tee "$O/ntt_pmc_$LOGN.txt"
