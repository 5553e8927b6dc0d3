#!/bin/bash
# tools/pmc_passes.sh <tag> <kernel-substring> -- <program args...>
# Where does the main kernel spend its time?  One rocprofv3 --pmc pass per counter group (never mixed with
# hip/hsa tracing), per-kernel averages printed by tools/pmc_table.py.  Development aid.
set -e -o pipefail
export TMPDIR=/tmp
tag=$1; kern=$2; shift 2; [ "$1" == "--" ] && shift
out=gpurun_out/$tag
mkdir -p $out
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  d=$out/pmc_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- "$@" > $d.log 2>&1 || { echo "pass $i ($grp) failed"; tail -5 $d.log; }
  i=$((i+1))
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS
SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TA_BUSY_avr TA_ADDR_STALLED_BY_TD_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum TCC_CYCLE_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
GROUPS
python3 tools/pmc_table.py $out "$kern"
