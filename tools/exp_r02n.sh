#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02n
mkdir -p $out
for ab in 0 1 2; do
echo "== sell ablate $ab" | tee -a $out/log.txt
GCN_AMD_SELL_ABLATE=$ab python3 tools/sweep.py --graph reddit --ks 128 --slices 8,16 --blocks-per-cu 32 2>&1 | grep "^128" | tee -a $out/log.txt
GCN_AMD_SELL_ABLATE=$ab GCN_AMD_GROUP_SC1=0 python3 tools/sweep.py --graph reddit --ks 128 --slices 16 --blocks-per-cu 32 2>&1 | grep "^128" | tee -a $out/log.txt
done
