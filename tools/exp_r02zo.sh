#!/bin/bash
# r02zo: non-temporal loads of the column / value streams in spmm_quad_kernel (after) against plain ones (before)
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02zo
mkdir -p $out
cp gcn_amd/lib/libgcnspmm.so /tmp/libgcnspmm_after.so
for which in after before after; do
cp tools/probes/_bin/libgcnspmm_before.so /tmp/libgcnspmm_before.so
cp /tmp/libgcnspmm_$which.so gcn_amd/lib/libgcnspmm.so
echo "== $which: quad kernel with values, 8 slices" | tee -a $out/log.txt
GCN_AMD_VALLESS=0 GCN_AMD_GROUP_WEIGHTED=0 python3 tools/sweep.py --graph reddit --ks 128 --slices=8 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
echo "== $which: quad kernel, unsliced, 1/16 scale (table fits L2)" | tee -a $out/log.txt
GCN_AMD_VALLESS=0 python3 tools/sweep.py --graph reddit --scale 0.0625 --ks 128 --slices=0 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
done
cp /tmp/libgcnspmm_after.so gcn_amd/lib/libgcnspmm.so
