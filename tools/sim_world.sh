#!/bin/bash
# Rehearse, on ONE GPU, the per-rank compute of an N-way row partition (rank 0's block, no collective).
mkdir -p gpurun_out
for w in 2 4 8; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --sim-world $w --steps 50 --warmup 10 > gpurun_out/sim_world_$w.json 2> gpurun_out/sim_world_$w.err
  python - "$w" <<'PY'
import json, sys
w = sys.argv[1]
d = json.load(open(f"gpurun_out/sim_world_{w}.json"))
print("sim-world", w, "ms_per_step", d["ms_per_step"], "plane_kernel_ms", d["roofline"]["kernel_ms_avg"],
      "frac", d["roofline"]["frac"], d["config"]["chunks"])
PY
done
