#!/bin/bash
# tools/products_dcsbm_lines.sh <out dir> — bench lines on the products-sized degree-corrected planted partition (k = 256):
# as handed over, in rabbit_device order, and with CsrAdjacency.autotune() choosing slices and column tile; plus the R-MAT
# products config under autotune (it must keep the default).  Development aid (round 4).
out=${1:-gpurun_out/products_dcsbm}
mkdir -p $out
for v in "--order none" "--order rabbit" "--order rabbit --autotune" "--order none --autotune"; do
  n=$(echo $v | tr -d " -")
  timeout -k 10 300 python3 bench.py --graph products-dcsbm $v --steps 10 --warmup 3 --no-cpu-baseline > $out/pd_$n.json 2> $out/pd_$n.err
  python3 - $out/pd_$n.json "$v" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("products-dcsbm", sys.argv[2], "| ms", d["ms_per_step"], "|", d["roofline"]["kernel"], "| frac", d["roofline"]["frac"], d["roofline"]["bound"],
      "| check", d["check"]["rel_err"], "rows", d["check"]["rows_per_rank"], "| autotune", d["config"].get("autotune"), "| ordering s", d["config"].get("ordering_seconds"))
PY
done
timeout -k 10 300 python3 bench.py --graph products --autotune --steps 10 --warmup 3 --no-cpu-baseline > $out/products_autotune.json 2> $out/products_autotune.err
python3 - $out/products_autotune.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("products (R-MAT, RCM) --autotune | ms", d["ms_per_step"], "|", d["roofline"]["kernel"], "| autotune", d["config"].get("autotune"))
PY
