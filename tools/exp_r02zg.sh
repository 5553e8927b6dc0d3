#!/bin/bash
# r02zg: chunk length and slice count once the fix-up is list-based
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02zg
mkdir -p $out
for T in 256 512 1024 2048; do
echo "== T=$T" | tee -a $out/log.txt
GCN_AMD_GROUP_T=$T python3 tools/sweep.py --graph reddit --ks 128 --slices=14,15,16 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
done
