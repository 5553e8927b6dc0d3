#!/bin/bash
# k <= 32 on the eight-engine kernel (spmm_group8_kernel) against the 64-column pass
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02zzi
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_spmm_gpu.py tests/test_stress_gpu.py tests/test_stress_group_gpu.py tests/test_layers_gpu.py -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for g8 in 1 0; do
echo "== group8 $g8" | tee -a $out/log.txt
GCN_AMD_GROUP8=$g8 python3 tools/sweep.py --graph reddit --ks 12,16,20,24,32 --slices=-1,8 --blocks-per-cu 32 2>&1 | grep -E "^[0-9]" | tee -a $out/log.txt
done
