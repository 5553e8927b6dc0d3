#!/bin/bash
# r02zp: partial-row store policy on smaller graphs (slab of partial rows small enough to stay cached?)
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02zp
mkdir -p $out
for sc in 0.125 0.25 0.5; do
for pol in 2 0; do
echo "== scale $sc store policy $pol" | tee -a $out/log.txt
GCN_AMD_GROUP_STORE=$pol python3 tools/sweep.py --graph reddit --scale $sc --ks 128 --slices=-1 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
done; done
echo "== full scale, stream_nt by size (default)" | tee -a $out/log.txt
python3 tools/sweep.py --graph reddit --ks 128,256 --slices=-1 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
