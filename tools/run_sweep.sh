#!/bin/bash
# tools/run_sweep.sh — the reference's run.sh (hidden size 4 over its six GraphSAINT datasets) on gcn_amd.
# ./dataset/<name>/ is used when present (GraphSAINT format); otherwise the shape-matched synthetic
# stand-in of gcn_amd/graphgen.py (no dataset ships offline).  Extra arguments go to every run.
d=${D:-4}
for g in pubmed flickr reddit ppi amazon yelp; do
  echo "=== $g (hidden $d) ==="
  python "$(dirname "$0")/profiling_gcn.py" -g $g -k $d -i ${ITERS:-30} "$@" 2>&1 | grep -v "amdgpu.ids\|UserWarning\|to_sparse_csr\|^Epoch"
done
