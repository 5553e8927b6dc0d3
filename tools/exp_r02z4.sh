#!/bin/bash
# r02z4: cache policy of the partial-row stores in the group kernel: 0 plain, 1 sc1, 2 nt, 3 sc1 nt
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02z4
mkdir -p $out
for m in 1 2 3 0; do
echo "== store mode $m" | tee -a $out/log.txt
GCN_AMD_GROUP_SC1=$m python3 tools/sweep.py --graph reddit --ks 128 --slices 8,12,16 --blocks-per-cu 32 2>&1 | grep -E "^128|^256" | tee -a $out/log.txt
done
