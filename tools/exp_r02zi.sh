#!/bin/bash
# r02zi: the sliced pass WITH a value stream (spmm_quad_kernel): non-temporal partial-row stores, slice counts
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02zi
mkdir -p $out
for nt in 1 0; do
echo "== values kept (GCN_AMD_VALLESS=0), quad nt stores $nt" | tee -a $out/log.txt
GCN_AMD_VALLESS=0 GCN_AMD_QUAD_NT=$nt python3 tools/sweep.py --graph reddit --ks 128 --slices=8,12,16 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
done
