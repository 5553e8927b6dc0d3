#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02h
mkdir -p $out
for ab in 0 1 2 3; do
echo "== ablate $ab" | tee -a $out/ablate.log
GCN_AMD_GROUP_ABLATE=$ab python3 tools/sweep.py --graph reddit --scale 0.25 --ks 128 --slices 4,16 --blocks-per-cu 32 2>&1 | grep "^128" | tee -a $out/ablate.log
GCN_AMD_GROUP_ABLATE=$ab python3 tools/sweep.py --graph reddit --scale 1.0 --ks 128 --slices 8,16 --blocks-per-cu 32 2>&1 | grep "^128" | tee -a $out/ablate.log
done
