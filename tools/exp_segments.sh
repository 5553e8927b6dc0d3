#!/bin/bash
# tools/exp_segments.sh — EXPERIMENT (r04): order of the (column tile, block) pairs inside the merged launch of the group
# kernels: GCN_AMD_GROUP_SEGMENTS = runs per XCD (1 = tile-major, 0 = the rule), see spmm_group.hip.  Prints the headline step and the
# weighted leg for every setting.  usage: exp_segments.sh <out dir> "<segment counts>" [bench args...]
out=$1; segs=$2; shift 2
mkdir -p $out
for seg in $segs; do
  GCN_AMD_GROUP_SEGMENTS=$seg python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $out/b_${seg}.json 2> $out/b_${seg}.err || { echo "seg $seg FAILED"; tail -2 $out/b_${seg}.err; continue; }
  python3 - $out/b_${seg}.json $seg <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
w = d["roofline"]["weighted"] or {"ms_per_spmm": float("nan")}
print(f"segments {sys.argv[2]:>2}: k={d['config']['k']} step {d['ms_per_step']:.3f} ms (main {d['roofline']['kernel_ms_avg']:.3f}, slices {d['roofline']['slices']}); weighted {w['ms_per_spmm']:.3f} ms; check {d['check']['rel_err']:.1e}")
PY
done
