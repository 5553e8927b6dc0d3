#!/bin/bash
# tools/ab_kernel_stats.sh <tag> <ENV_NAME> <value> [<value> ...] [-- bench.py args]: kernel times (rocprofv3 --kernel-trace
# --stats) of one bench run per value of a development knob, side by side.  Everything under gpurun_out/<tag>/.
set -e -o pipefail
export TMPDIR=/tmp
tag=$1; name=$2; shift 2
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
out=gpurun_out/$tag
mkdir -p $out
for v in "${vals[@]}"; do
  export $name=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$v -- python3 bench.py --no-cpu-baseline "$@" > $out/bench_$v.json 2> $out/rocprof_$v.err
  find $out/stats_$v -name "*kernel_trace.csv" -delete
  f=$(find $out/stats_$v -name "*kernel_stats.csv" | head -1)
  echo "== $name=$v" | tee -a $out/log.txt
  python3 - "$f" $out/bench_$v.json <<'PY' | tee -a $out/log.txt
import csv, json, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if "gcn::" in r["Name"] and int(r["Calls"]) >= 10:
        print("  %-64s calls %4s avg %9.1f us min %9.1f us" % (r["Name"].replace("void ", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
print("  ms_per_step", d["ms_per_step"], "check", d["check"]["rel_err"])
PY
done
