#!/usr/bin/env python3
"""tools/panel_probe.py — run the LDS-panel SpMM a few times on a renumbered planted-partition graph
(for rocprofv3 --pmc / --kernel-trace runs).  The Rabbit ordering is cached in gpurun_out/."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd
from gcn_amd import graphgen, reorder

dev = torch.device("cuda:0")
n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
order = sys.argv[2] if len(sys.argv) > 2 else "rcm"
panels = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rowptr, col, val, n = graphgen.make_sbm(n_req, device=dev, seed=7)
rp, ci, va = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
if order == "truth":      # the planted order itself (what a perfect community ordering would give)
    rp2, ci2, va2, n = [x.cpu().numpy() for x in graphgen.make_sbm(n_req, device=dev, seed=7, relabel=False)[:3]] + [n]
elif order == "rcm":
    rp2, ci2, va2, _ = reorder.apply_rank(rp, ci, va, reorder.order_rcm(rp, ci))
else:
    rp2, ci2, va2, _ = getattr(reorder, order)(rp, ci, va)
adj = gcn_amd.CsrAdjacency(torch.from_numpy(rp2).to(dev), torch.from_numpy(ci2).to(dev), torch.from_numpy(va2).to(dev),
                           (n, n), symmetric=True, panels=panels, slices=0)
H = graphgen.random_features(n, 128, seed=2, device=dev)
out = torch.empty((n, 128), device=dev)
for _ in range(3):
    adj.matmul_raw(H, out=out)
torch.cuda.synchronize()
adj.profile_begin(5)
for _ in range(5):
    adj.matmul_raw(H, out=out)
ms = adj.profile_end()
print(f"order={order} panels={adj.panel_rows} coverage={adj.panel_coverage:.3f} kernel_ms={sum(ms)/len(ms):.4f}")
