#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02l
mkdir -p $out
echo "== sell small" | tee -a $out/log.txt
python3 tools/sweep.py --graph reddit --scale 0.125 --ks 128 --slices 4,8,16 --blocks-per-cu 32 2>&1 | grep "^128" | tee -a $out/log.txt
echo "== group small" | tee -a $out/log.txt
GCN_AMD_SELL=0 python3 tools/sweep.py --graph reddit --scale 0.125 --ks 128 --slices 4,8,16 --blocks-per-cu 32 2>&1 | grep "^128" | tee -a $out/log.txt
for T in 512 1024; do
echo "== sell full T=$T" | tee -a $out/log.txt
GCN_AMD_SELL_T=$T python3 tools/sweep.py --graph reddit --ks 128 --slices 8,12 --blocks-per-cu 32 2>&1 | grep "^128" | tee -a $out/log.txt
done
