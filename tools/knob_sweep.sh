#!/bin/bash
# tools/knob_sweep.sh <tag> <ENV_KNOB> <v1,v2,...> -- <tools/sweep.py args...>
# One development knob (DESIGN.md "Development knobs") swept over its values, the kernel-time sweep of
# tools/sweep.py under each; log under gpurun_out/<tag>/log.txt.  Replaces the one-shot experiment drivers of
# round 2 (tools/exp_r02*.sh), e.g. the store-policy experiment r02z4:
#   tools/knob_sweep.sh r02z4 GCN_AMD_GROUP_STORE 1,2,0 -- --graph reddit --ks 128 --slices 8,12,16
set -e -o pipefail
export TMPDIR=/tmp
tag=$1; knob=$2; values=$3; shift 3; [ "$1" == "--" ] && shift
out=gpurun_out/$tag
mkdir -p $out
for v in ${values//,/ }; do
  echo "== $knob=$v" | tee -a $out/log.txt
  env $knob=$v python3 tools/sweep.py "$@" 2>&1 | grep -E "^#|^[0-9]+ " | tee -a $out/log.txt
done
