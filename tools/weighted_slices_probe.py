#!/usr/bin/env python3
"""tools/weighted_slices_probe.py — the weighted sliced pass (values kept: gcn_spmm_plan_set_value_factors(null, null)) of the
Reddit-shaped graph at several explicit slice counts.  Round 4: flat around the automatic 15 (10 / 12 / 13 / 14 / 15 / 16 / 17 /
18 / 20 slices: 3.14 / 3.10 / 3.09 / 3.10 / 3.05 / 3.08 / 3.06 / 3.13 / 3.17 ms).  Development aid."""
import sys, torch
sys.path.insert(0, '.')
import gcn_amd
from gcn_amd import graphgen
dev = torch.device('cuda:0')
rowptr, col, val, n = graphgen.make_graph('reddit', device=dev, seed=1)
k = 128
H = graphgen.random_features(n, k, seed=2, device=dev)
out = torch.empty((n, k), device=dev)
def timed(adj, reps=10):
    for _ in range(3): adj.matmul_raw(H, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): adj.matmul_raw(H, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for S in (10, 12, 13, 14, 15, 16, 17, 18, 20):
    adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True, slices=S)
    adj.plan
    adj.set_value_factors(None, None)
    print(f"weighted S={S}: {timed(adj):.3f} ms  {adj.main_kernel(k)}", flush=True)
    del adj
