#!/bin/bash
# BASELINE config 5: R-MAT, feat 512, orderings none / degree / RCM / Gorder(w=3) — full size (scale 24) for the first
# three, Gorder (serial host algorithm) at scales 20 and 22 with its host time stated
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02w
mkdir -p $out
python3 tools/reorder_sweep.py --rmat-scale 20 --k 512 --orders none,deg,rcm,gorder 2>&1 | grep -v amdgpu.ids | tee -a $out/reorder_rmat_k512.txt
python3 tools/reorder_sweep.py --rmat-scale 24 --k 512 --orders none,deg,rcm 2>&1 | grep -v amdgpu.ids | tee -a $out/reorder_rmat_k512.txt
python3 tools/reorder_sweep.py --rmat-scale 22 --k 512 --orders none,gorder 2>&1 | grep -v amdgpu.ids | tee -a $out/reorder_rmat_k512.txt
