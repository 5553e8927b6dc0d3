#!/bin/bash
# r02zj: the group kernel WITH values (spmm_group_weighted_kernel) against the four-per-gather kernel on a matrix
# whose values are kept (GCN_AMD_VALLESS=0: nothing is detected)
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02zj
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_spmm_gpu.py tests/test_stress_gpu.py -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; }
tail -3 $out/pytest.log
for wt in 1 0; do
echo "== values kept (GCN_AMD_VALLESS=0), weighted group kernel $wt" | tee -a $out/log.txt
GCN_AMD_VALLESS=0 GCN_AMD_GROUP_WEIGHTED=$wt python3 tools/sweep.py --graph reddit --ks 64,128,256 --slices=-1,8,12,16 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
done
