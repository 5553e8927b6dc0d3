#!/bin/bash
# r02z7: rank-0 share of an N-way row partition on one GPU: value-free pass forced on (threshold 0) vs the default rule
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02z7
mkdir -p $out
for thr in 96 0; do
for w in 2 4 8; do
  GCN_AMD_VALLESS_MIN_PER_COL=$thr timeout -k 10 200 python3 bench.py --no-cpu-baseline --sim-world $w --steps 30 --warmup 5 > $out/sim_${thr}_$w.json 2> $out/sim_${thr}_$w.err
  python3 - "$w" "$thr" <<'PY' | tee -a gpurun_out/r02z7/log.txt
import json, sys
w, thr = sys.argv[1], sys.argv[2]
d = json.loads(open(f"gpurun_out/r02z7/sim_{thr}_{w}.json").read().strip().splitlines()[-1])
print("threshold", thr, "sim-world", w, "ms_per_step", d["ms_per_step"], "kernel", d["roofline"]["kernel"], "slices", d["roofline"]["slices"],
      "kernel_ms", d["roofline"]["kernel_ms_avg"], "check", d["check"]["rel_err"])
PY
done; done
