#!/bin/bash
# r02zk: stream stored in lane-major runs of 64 entries (one 8-byte stream load per four blocks)
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02zk
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_spmm_gpu.py tests/test_stress_gpu.py tests/test_layers_gpu.py -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; }
tail -3 $out/pytest.log
echo "== value-free" | tee -a $out/log.txt
python3 tools/sweep.py --graph reddit --ks 64,128,256 --slices=-1 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
echo "== values kept (GCN_AMD_VALLESS=0), weighted group kernel" | tee -a $out/log.txt
GCN_AMD_VALLESS=0 python3 tools/sweep.py --graph reddit --ks 128 --slices=-1,8,12 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
