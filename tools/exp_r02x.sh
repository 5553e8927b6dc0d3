#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02x
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_spmm_gpu.py tests/test_layers_gpu.py -x -q -k "group_kernel or dropout or value_free or fused" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for ov in 1 0; do
echo "== overlap $ov" | tee -a $out/log.txt
GCN_AMD_OVERLAP=$ov python3 tools/sweep.py --graph reddit --ks 128,256 --slices 8,12,16 --blocks-per-cu 32 2>&1 | grep -E "^128|^256" | tee -a $out/log.txt
done
