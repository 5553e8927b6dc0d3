#!/bin/bash
# r02z2 (temporary build): what a row end costs — 0 = product, 1 = the store instruction skipped, 2 = store kept, accumulator reset skipped
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02z2
mkdir -p $out
for ab in 0 1; do
echo "== ablate $ab" | tee -a $out/log.txt
GCN_ABLATE=$ab python3 tools/sweep.py --graph reddit --ks 128 --slices 8,16 --blocks-per-cu 32 2>&1 | grep -E "^128|^256" | tee -a $out/log.txt
done
echo "== ablate 0, plain (write-back) stores" | tee -a $out/log.txt
GCN_AMD_GROUP_SC1=0 python3 tools/sweep.py --graph reddit --ks 128 --slices 8,16 --blocks-per-cu 32 2>&1 | grep -E "^128|^256" | tee -a $out/log.txt
echo "== ablate 1, plain stores" | tee -a $out/log.txt
GCN_ABLATE=1 GCN_AMD_GROUP_SC1=0 python3 tools/sweep.py --graph reddit --ks 128 --slices 8,16 --blocks-per-cu 32 2>&1 | grep -E "^128|^256" | tee -a $out/log.txt
