#!/bin/bash
export TMPDIR=/tmp
for S in 8 16; do
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=gpurun_out/pmc_s${S}_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/sweep.py --graph reddit --ks 128 --slices=$S --iters 3 > $d.log 2>&1
done
done
echo done
