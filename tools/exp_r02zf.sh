#!/bin/bash
# r02zf: list-based fix-up of the group path
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02zf
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_spmm_gpu.py tests/test_layers_gpu.py tests/test_stress_gpu.py -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
python3 tools/sweep.py --graph reddit --ks 64,128,256 --slices=-1 --blocks-per-cu 32 2>&1 | grep -E "^64|^128|^256" | tee -a $out/log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee -a $out/log.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gcn::" in r["Name"] and int(r["Calls"]) >= 20:
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"])
PY
tail -1 $out/bench_under_rocprof.json | cut -c1-330
