#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02j
mkdir -p $out
for R in 4 8 16; do
echo "== round $R" | tee -a $out/round.log
GCN_AMD_SELL_ROUND=$R python3 tools/sweep.py --graph reddit --ks 128 --slices 8,12,16 --blocks-per-cu 32 2>&1 | grep "^128" | tee -a $out/round.log
done
echo "== round 16 lmax 64" | tee -a $out/round.log
GCN_AMD_SELL_ROUND=16 GCN_AMD_SELL_LMAX=64 python3 tools/sweep.py --graph reddit --ks 128 --slices 8,12 --blocks-per-cu 32 2>&1 | grep "^128" | tee -a $out/round.log
echo "== round 16 lmax 240" | tee -a $out/round.log
GCN_AMD_SELL_ROUND=16 GCN_AMD_SELL_LMAX=240 python3 tools/sweep.py --graph reddit --ks 128 --slices 8,12 --blocks-per-cu 32 2>&1 | grep "^128" | tee -a $out/round.log
