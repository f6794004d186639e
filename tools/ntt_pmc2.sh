#!/bin/bash
# usage: tools/ntt_pmc2.sh <tag> [log_n] — second counter set of the NTT passes: instruction fetch, VMEM / TA back-pressure, LDS FIFOs
cd "$(dirname "$0")/.."
TAG=$1; LOGN=${2:-22}
O=gpurun_out/$TAG; mkdir -p "$O"
export TMPDIR=/tmp
{
echo "# tools/ntt_pmc2.sh $TAG $LOGN  python tools/ntt_probe.py $LOGN 20   PLK_NTT_WAVE=${PLK_NTT_WAVE:-1}"
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU2" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  echo "## $grp"
  tools/pmc_kernel.sh ntt_pass "$grp" -- python tools/ntt_probe.py $LOGN 20
done
} 2>&1 | tee "$O/ntt_pmc2_$LOGN.txt"
