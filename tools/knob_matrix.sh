#!/bin/bash
# tools/knob_matrix.sh — the group-kernel tests under each of the switches that still select another code path (round 4
# pruned the decided experiments: what is left is what a test needs): the fix-up pass of its own, k <= 32 and 33..48 on the
# 64-column pass, the (tile, block) order of the merged launch.  (GCN_AMD_GROUP_BIG=1 has a test of its own in a child
# process, test_group_kernels_with_64_bit_slice_bases: the kernel names other tests expect do not hold under it.)  Development aid.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-knobs}
mkdir -p $out
rc=0
for env in "GCN_AMD_GROUP_FUSED_FIXUP=0" "GCN_AMD_GROUP8=0" "GCN_AMD_GROUP12=0" "GCN_AMD_GROUP_SEGMENTS=1" "GCN_AMD_GROUP_SEGMENTS=5"; do
  if env $env timeout -k 10 300 python3 -m pytest tests/test_stress_group_gpu.py tests/test_spmm_gpu.py -x -q \
       -k "random or group_kernel or value_free_sliced or row_block or second_slice or widths or captured" -p no:cacheprovider > $out/log_$env.txt 2>&1; then
    echo "$env: $(tail -1 $out/log_$env.txt)" | tee -a $out/summary.txt
  else
    echo "$env: FAILED" | tee -a $out/summary.txt; tail -15 $out/log_$env.txt; rc=1
  fi
done
exit $rc
