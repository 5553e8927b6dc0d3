#!/usr/bin/env python3
"""tools/reorder_device_bench.py — time the device reorderers (order_deg / order_rcm / CSR rewrite on the
GPU) against the host versions on the BASELINE-shaped graphs, and check that the integers agree.
    python tools/reorder_device_bench.py --graph products [--scale 1.0] [--no-host]
    python tools/reorder_device_bench.py --graph rmat --rmat-scale 24 --no-host
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import graphgen, reorder        # noqa: E402


def timed(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="products")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--rmat-scale", type=int, default=22)
    ap.add_argument("--no-host", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.graph == "rmat":
        rowptr, col, val, n = graphgen.make_rmat(args.rmat_scale, device=dev, seed=5)
        name = f"rmat{args.rmat_scale}"
    elif args.graph == "sbm":
        rowptr, col, val, n = graphgen.make_sbm(int(240000 * args.scale), device=dev, seed=7)
        name = "sbm"
    else:
        rowptr, col, val, n = graphgen.make_graph(args.graph, device=dev, seed=3 if args.graph == "products" else 1, scale=args.scale)
        name = args.graph
    nnz = int(col.numel())
    print(f"# graph {name} n={n} nnz={nnz}", flush=True)
    reorder.order_deg_device(rowptr, col)                       # warm-up (module load, hipCUB temp)
    rank_d, t_deg = timed(lambda: reorder.order_deg_device(rowptr, col, "total", True))
    (rank_r, levels), t_rcm = timed(lambda: reorder.order_rcm_device(rowptr, col, return_levels=True))
    out, t_app = timed(lambda: reorder.apply_rank_device(rowptr, col, val, rank_r))
    print(f"device: order_deg {t_deg * 1e3:.1f} ms | order_rcm {t_rcm * 1e3:.1f} ms ({levels} BFS levels) | "
          f"csr_apply_rank {t_app * 1e3:.1f} ms", flush=True)
    if not args.no_host:
        rp, ci, va = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
        t = time.perf_counter(); h_deg = reorder.order_deg(rp, ci, "total", True); t_hdeg = time.perf_counter() - t
        t = time.perf_counter(); h_rcm = reorder.order_rcm(rp, ci, True); t_hrcm = time.perf_counter() - t
        t = time.perf_counter(); h = reorder.apply_rank(rp, ci, va, h_rcm); t_happ = time.perf_counter() - t
        print(f"host:   order_deg {t_hdeg * 1e3:.1f} ms | order_rcm {t_hrcm * 1e3:.1f} ms | csr_apply_rank {t_happ * 1e3:.1f} ms")
        same = (np.array_equal(rank_d.cpu().numpy(), h_deg) and np.array_equal(rank_r.cpu().numpy(), h_rcm)
                and all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(out, h)))
        print("identical integers (rank vectors, rewritten CSR, vomp):", same)
        if not same:
            sys.exit(1)


if __name__ == "__main__":
    main()
