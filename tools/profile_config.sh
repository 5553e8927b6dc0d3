#!/bin/bash
# tools/profile_config.sh <tag> [bench.py args...] — on the GPU box: one BASELINE config measured the way the bench
# line is judged: (1) bench.py itself, (2) the same command under rocprofv3 --kernel-trace --stats, (3) the HBM /
# L2 counters in separate --pmc passes restricted to this library's kernels (never with hip/hsa tracing), folded
# into gpurun_out/<tag>/pmc.json by tools/pmc_summary.py.  Everything lands under gpurun_out/<tag>/.
#   GCN_PROFILE_CPU=1   keep the CPU-baseline leg in step (1)
set -e -o pipefail
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
cpu="--no-cpu-baseline"; [ "${GCN_PROFILE_CPU:-0}" == "1" ] && cpu=""
python3 bench.py "$@" $cpu > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
tail -c 600 $out/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err || { tail -20 $out/rocprof.err; exit 1; }
find $out/stats -name "*kernel_stats.csv" -exec head -6 {} \; | cut -c1-220
find $out/stats -name "*kernel_trace.csv" -delete      # (the stats table is what is kept: gpurun_out merges back <= 64 MiB)
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$out/pmc_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gcn::" --output-format csv -d $d -- python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $d.json 2> $d.err || { tail -20 $d.err; exit 1; }
  find $d -name "*kernel_trace.csv" -delete; find $d -name "*agent_info.csv" -delete
  echo "pmc $c done"
done
python3 - "$out" <<'PY'
import json, subprocess, sys
out = sys.argv[1]
d = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][0])
kern = d["roofline"]["kernel"].split("<")[0].replace("gcn::", "")
c = d["config"]
graph = c["graph"]
subprocess.run([sys.executable, "tools/pmc_summary.py", "--glob", out + "/pmc_*", "--graph", graph, "--k", str(c["k"]), "--order", c.get("order", "none"),
                "--launches-per-spmm", str(d["roofline"]["launches_per_spmm"]), "--algorithmic-bytes-per-launch", str(d["roofline"]["algorithmic_bytes_per_launch"]),
                "--kernel", kern, "--out", out + "/pmc.json"], check=True, stdout=subprocess.DEVNULL)
p = json.load(open(out + "/pmc.json"))
print(graph, c["k"], c.get("order"), kern, "traffic GB/launch", round(p["traffic_bytes_per_launch"] / 1e9, 3), "L2 hit", p["l2_hit_rate"],
      "kernel ms", d["roofline"]["kernel_ms_avg"], "frac", d["roofline"]["frac"], d["roofline"]["bound"])
PY
