#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02i
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_spmm_gpu.py -x -q -k "lockstep_kernel" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for T in 512; do
GCN_AMD_SELL_T=$T python3 tools/sweep.py --graph reddit --ks 128 --slices 8,12,16,24 --blocks-per-cu 32 > $out/sweep_T$T.log 2>&1
cat $out/sweep_T$T.log
done
GCN_AMD_GROUP_SC1=0 python3 tools/sweep.py --graph reddit --ks 128 --slices 8,16 --blocks-per-cu 32 > $out/sweep_plainstores.log 2>&1
cat $out/sweep_plainstores.log
