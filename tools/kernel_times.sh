#!/bin/bash
# tools/kernel_times.sh <tag>: a subset of the GPU tests, then the kernel times of one bench run (rocprofv3 --stats)
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-r02zx}
mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_spmm_gpu.py tests/test_layers_gpu.py -x -q -k "slic or dropout or value_free or fused or epilogue" > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee -a $out/log.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gcn::" in r["Name"] and int(r["Calls"]) >= 20:
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"])
PY
tail -1 $out/bench_under_rocprof.json | cut -c1-260
