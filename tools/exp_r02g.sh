#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02g
mkdir -p $out
for sc in 0.125 0.25 0.5; do
python3 tools/sweep.py --graph reddit --scale $sc --ks 128 --slices 4,8,16,32 --blocks-per-cu 32 > $out/sweep_$sc.log 2>&1
cat $out/sweep_$sc.log
done
