#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02v
mkdir -p $out
for T in 256 512 1024; do
echo "== group T=$T" | tee -a $out/log.txt
GCN_AMD_GROUP_T=$T python3 tools/sweep.py --graph reddit --ks 128 --slices 8,9,10,12 --blocks-per-cu 32 2>&1 | grep "^128" | tee -a $out/log.txt
done
