#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02m
mkdir -p $out
i=0
while read -r grp; do
  for mode in sell group; do
    d=$out/${mode}_pmc_$i
    if [ $mode == group ]; then export GCN_AMD_SELL=0; else export GCN_AMD_SELL=1; fi
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 tools/sweep.py --graph reddit --scale 0.125 --ks 128 --slices 4 --blocks-per-cu 32 --iters 3 > $d.log 2>&1 || { echo "pass failed"; tail -3 $d.log; }
  done
  i=$((i+1))
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_IFETCH
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
GROUPS
for mode in sell group; do
mkdir -p $out/$mode; for d in $out/${mode}_pmc_*; do [ -d $d ] && mv $d $out/$mode/pmc_${d##*_}; done
echo "=== $mode"; python3 tools/pmc_table.py $out/$mode "spmm_${mode}_kernel"
done
