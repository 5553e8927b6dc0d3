#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02u
mkdir -p $out
for n in 60000 240000; do
for order in truth communities; do
for mf in 1 0; do
GCN_AMD_PANEL_MFMA=$mf python3 tools/panel_mfma_probe.py $n $order 2>&1 | grep -v amdgpu.ids | tee -a $out/probe.log
done; done; done
GCN_AMD_PANEL_MFMA=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_mfma -- python3 tools/panel_mfma_probe.py 240000 truth > $out/prof_mfma.log 2>&1
GCN_AMD_PANEL_MFMA=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_lds -- python3 tools/panel_mfma_probe.py 240000 truth > $out/prof_lds.log 2>&1
for t in mfma lds; do echo "== kernel stats $t"; f=$(find $out/stats_$t -name "*kernel_stats.csv" | head -1); grep -E "gcn::" $f | cut -d, -f1-4 | cut -c1-150 | head -12; done | tee -a $out/probe.log
