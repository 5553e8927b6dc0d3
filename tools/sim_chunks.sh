#!/bin/bash
# chunk-size sweep for the simulated rank of an 8-way partition and for the single GPU
mkdir -p gpurun_out
for w in 8 4 1; do for c in 384 768 1024 2048 4096; do
  if [ $w = 1 ]; then extra=""; else extra="--sim-world $w"; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline $extra --chunk $c --steps 40 --warmup 10 > gpurun_out/simc_${w}_$c.json 2> gpurun_out/simc_${w}_$c.err
  python - "$w" "$c" <<'PY'
import json, sys
w, c = sys.argv[1:3]
d = json.load(open(f"gpurun_out/simc_{w}_{c}.json"))
print("world", w, "chunk", c, "ms_per_step", d["ms_per_step"], "main_kernel_ms", d["roofline"]["kernel_ms_avg"], d["config"]["chunks"])
PY
done; done
