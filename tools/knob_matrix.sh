#!/bin/bash
# tools/knob_matrix.sh — the group-kernel tests under every development knob that selects another code path
# (chunk lengths, store policies, ring / eight-engine / merged launch / weighted off).  Development aid.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-knobs}
mkdir -p $out
rc=0
for env in "GCN_AMD_GROUP_T=256" "GCN_AMD_GROUP_T=1024" "GCN_AMD_GROUP_T=2048" "GCN_AMD_GROUP_RING=0" "GCN_AMD_GROUP8=0" \
           "GCN_AMD_GROUP_MERGE_TILES=0" "GCN_AMD_GROUP_STORE=0" "GCN_AMD_GROUP_STORE=1" "GCN_AMD_GROUP_MIN_K=33" \
           "GCN_AMD_GROUP_FUSED_FIXUP=0" "GCN_AMD_GROUP12=0" "GCN_AMD_GROUP_NARROW_SLICES=0" "GCN_AMD_VALLESS_MIN_PER_COL=1"; do
  if env $env timeout -k 10 300 python3 -m pytest tests/test_stress_group_gpu.py tests/test_spmm_gpu.py -x -q \
       -k "random or group_kernel or value_free_sliced or row_block or second_slice or widths or captured" -p no:cacheprovider > $out/log_$env.txt 2>&1; then
    echo "$env: $(tail -1 $out/log_$env.txt)" | tee -a $out/summary.txt
  else
    echo "$env: FAILED" | tee -a $out/summary.txt; tail -15 $out/log_$env.txt; rc=1
  fi
done
exit $rc
