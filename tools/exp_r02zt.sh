#!/bin/bash
# r02zt: LDS ring (occupancy 4): slice counts, k = 64 / 256, chunk lengths
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02zt
mkdir -p $out
for ring in 1 0; do
echo "== ring $ring" | tee -a $out/log.txt
GCN_AMD_GROUP_RING=$ring python3 tools/sweep.py --graph reddit --ks 128 --slices=14,15,16,17,18 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
GCN_AMD_GROUP_RING=$ring python3 tools/sweep.py --graph reddit --ks 64,256 --slices=15,16 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
done
echo "== ring 1, T=256 / 1024" | tee -a $out/log.txt
for T in 256 1024; do
GCN_AMD_GROUP_T=$T GCN_AMD_GROUP_RING=1 python3 tools/sweep.py --graph reddit --ks 128 --slices=16 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
done
