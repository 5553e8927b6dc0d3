#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02b
mkdir -p $out
python3 tools/sweep.py --graph reddit --ks 128 --slices 8,12,16,24,32 --blocks-per-cu 64 --gather-width 4 > $out/sweep_full.log 2>&1
cat $out/sweep_full.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/pmc_hit -- python3 tools/sweep.py --graph reddit --ks 128 --slices 16,24 --blocks-per-cu 64 --gather-width 4 --iters 2 > $out/pmc_hit.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r02b/pmc_hit/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gcn::spmm' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    h = sum(v['TCC_HIT_sum']) / len(v['TCC_HIT_sum']); m = sum(v['TCC_MISS_sum']) / len(v['TCC_MISS_sum'])
    print(k, 'launches', len(v['TCC_HIT_sum']), 'hit rate', h / (h + m), 'miss', m)
PY
