#!/bin/bash
# r02z5: nt partial-row stores: slices and chunk lengths
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02z5
mkdir -p $out
for T in 512 256; do
echo "== nt stores, T=$T" | tee -a $out/log.txt
GCN_AMD_GROUP_T=$T GCN_AMD_GROUP_SC1=2 python3 tools/sweep.py --graph reddit --ks 128 --slices 14,16,18,20,24,32 --blocks-per-cu 32 2>&1 | grep -E "^128|^256" | tee -a $out/log.txt
done
echo "== nt stores, k=256 / 64" | tee -a $out/log.txt
GCN_AMD_GROUP_SC1=2 python3 tools/sweep.py --graph reddit --ks 64,256 --slices 8,16,24 --blocks-per-cu 32 2>&1 | grep -E "^64|^256" | tee -a $out/log.txt
