#!/bin/bash
# tools/weighted_ablation.sh <tag> — where does the weighted sliced pass (values that do not factor) spend what it costs
# beyond the value-free pass?  Runs bench.py's `roofline.weighted` leg on the product and on the ablation builds of
# tools/ablate_group.sh (16: value stream cache-resident, 32: no value broadcast, 64: adds instead of FMAs, 80 = 16+64,
# 112 = all three): wrong numbers in that leg on purpose, exact costs.  Development aid.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-weighted_ablation}
mkdir -p $out
for n in ${ABLATIONS:-0 16 32 64 80 112}; do
  lib=""; [ $n != 0 ] && lib="artifacts/ablate/libgcnspmm_abl$n.so"
  GCN_AMD_LIB=$lib python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_abl$n.json 2> $out/bench_abl$n.err || { echo "abl $n failed"; tail -3 $out/bench_abl$n.err; continue; }
  python3 - $out/bench_abl$n.json $n <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
w = d["roofline"]["weighted"]
print(f"ablate {sys.argv[2]:>3}: value-free step {d['ms_per_step']:.3f} ms (main kernel {d['roofline']['kernel_ms_avg']:.3f}); weighted SpMM {w['ms_per_spmm']:.3f} ms  {w['kernel']}")
PY
done | tee $out/summary.txt
