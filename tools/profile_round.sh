#!/bin/bash
# tools/profile_round.sh <tag> — on the GPU box: bench.py under rocprofv3 --kernel-trace --stats, then the
# PMC passes (one counter group per run, never with hip/hsa tracing), all under gpurun_out/<tag>/.
set -e -o pipefail
tag=${1:-rXX}
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
tail -1 $out/bench_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$out/pmc_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $d.json 2> $d.err
  echo "pmc $c done"
done
find $out -name "*kernel_stats.csv" | head -3
