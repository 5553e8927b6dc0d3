#!/bin/bash
# r02zn: non-temporal loads of the 16-bit stream (a build with -DSTREAM_NT=1 swapped in) against plain ones
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02zn
mkdir -p $out
echo "== plain stream loads" | tee -a $out/log.txt
python3 tools/sweep.py --graph reddit --ks 128,256 --slices=-1 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
cp gcn_amd/lib/libgcnspmm.so /tmp/libgcnspmm_plain.so
cp tools/probes/_bin/libgcnspmm_streamnt.so gcn_amd/lib/libgcnspmm.so
echo "== nt stream loads" | tee -a $out/log.txt
python3 tools/sweep.py --graph reddit --ks 128,256 --slices=-1 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
cp /tmp/libgcnspmm_plain.so gcn_amd/lib/libgcnspmm.so
echo "== plain again" | tee -a $out/log.txt
python3 tools/sweep.py --graph reddit --ks 128,256 --slices=-1 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
