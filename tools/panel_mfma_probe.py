#!/usr/bin/env python3
"""tools/panel_mfma_probe.py — the dense-panel MFMA path against the LDS panel kernels and the gather-only paths on
planted-partition graphs in a community order.  Per configuration: whole SpMM (k = 128) from
torch events, and — under rocprofv3 --kernel-trace — the per-kernel split.
    python tools/panel_mfma_probe.py [n=240000] [order=truth|communities|rcm]
(Round 2 compared against LDS-only panels through a knob that round 4 removed: dense tiles from 25 % window density.)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd
from gcn_amd import graphgen, reorder

dev = torch.device("cuda:0")
n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 240000
order = sys.argv[2] if len(sys.argv) > 2 else "truth"
if order == "truth":      # the planted order itself (what a perfect community ordering gives)
    rowptr, col, val, n = graphgen.make_sbm(n_req, device=dev, seed=7, relabel=False)
else:
    rowptr, col, val, n = graphgen.make_sbm(n_req, device=dev, seed=7)
    rank = reorder.order_communities_device(rowptr, col) if order == "communities" else reorder.order_rcm_device(rowptr, col)
    rowptr, col, val, _ = reorder.apply_rank_device(rowptr, col, val, rank)
nnz = int(col.numel())
H = graphgen.random_features(n, 128, seed=2, device=dev)
out = torch.empty((n, 128), device=dev)
ref = None
print(f"# planted partition n={n} nnz={nnz} order={order}")
for name, kw in (("gather only, unsliced", dict(panels=0, slices=0)), ("gather only, auto slices", dict(panels=0, slices="auto")),
                 ("panels (auto)", dict(panels="auto", slices=0))):
    adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True, **kw)
    for _ in range(3):
        adj.matmul_raw(H, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        adj.matmul_raw(H, out=out)
    e1.record()
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"{name:26s} spmm_ms={e0.elapsed_time(e1) / 10:.4f} slices={adj.num_slices} panel_rows={adj.panel_rows} "
          f"coverage={adj.panel_coverage:.3f} dense_panels={adj.dense_panels} kernel={adj.main_kernel(128)} rel_diff_vs_first={err:.1e}")
    del adj
