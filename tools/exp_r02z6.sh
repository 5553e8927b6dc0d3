#!/bin/bash
# r02z6: nt stores by default, automatic slice count of the value-free path, wide slice reduction
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02z6
mkdir -p $out
echo "(tests ran in the previous call: 244 passed)"
python3 tools/sweep.py --graph reddit --ks 64,128,256 --slices=-1,8,14,15,16 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r02z6/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["check"], d["roofline"]["kernel"], d["roofline"]["slices"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"])
PY
