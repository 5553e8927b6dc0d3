#!/usr/bin/env python3
"""tools/rabbit_probe.py — Rabbit on the device (parallel incremental aggregation, csrc/rabbit_device.hip) against the
serial host Rabbit (bit-exact with the reference) on graphs WITH community structure: ordering time, number of
communities, modularity, window coverage of the LDS panels, SpMM time in each order.  Development aid.
    python tools/rabbit_probe.py --n 60000,240000 [--rmat reddit]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd                                  # noqa: E402
from gcn_amd import graphgen, reorder           # noqa: E402


def spmm_ms(rowptr, col, val, n, k=128, iters=10, panels=0):
    adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True, panels=panels)
    adj.autotune(k=k)
    B = graphgen.random_features(n, k, seed=2, device=rowptr.device)
    out = torch.empty((n, k), device=rowptr.device)
    for _ in range(3):
        adj.matmul_raw(B, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        adj.matmul_raw(B, out=out)
    e1.record()
    torch.cuda.synchronize()
    cov = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), panels="auto").panel_coverage
    return e0.elapsed_time(e1) / iters, adj.num_slices, cov


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", default="60000")
    ap.add_argument("--graph", default="sbm", help="sbm | reddit | products (R-MAT shapes have no communities to find)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--no-host", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for n in [int(x) for x in args.n.split(",")]:
        if args.graph == "sbm":
            rowptr, col, val, n = graphgen.make_sbm(n, device=dev, seed=7)
        else:
            rowptr, col, val, n = graphgen.make_graph(args.graph, device=dev, seed=1, scale=args.scale)
        nnz = int(col.numel())
        print(f"# {args.graph} n={n} nnz={nnz}", flush=True)
        ms, S, cov = spmm_ms(rowptr, col, val, n)
        print(f"order none: spmm {ms:.3f} ms (slices {S}) coverage {cov:.3f}", flush=True)
        torch.cuda.synchronize()
        reorder.order_rabbit_device(rowptr, col)                      # (first call: allocations, code load)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rank, comm, stats = reorder.order_rabbit_device(rowptr, col, return_communities=True, return_stats=True)
        torch.cuda.synchronize()
        t_dev = time.perf_counter() - t0
        assert torch.equal(torch.sort(rank).values, torch.arange(n, device=dev))
        q_dev = reorder.modularity(rowptr, col, comm)
        rp, ci, va, _ = reorder.apply_rank_device(rowptr, col, val, rank)
        ms, S, cov = spmm_ms(rp, ci, va, n)
        print(f"rabbit device: {t_dev * 1e3:.1f} ms  {stats}  Q {q_dev:.4f}  spmm {ms:.3f} ms (slices {S}) coverage {cov:.3f}", flush=True)
        t0 = time.perf_counter()
        rk2 = reorder.order_communities_device(rowptr, col)
        torch.cuda.synchronize()
        t_star = time.perf_counter() - t0
        rp, ci, va, _ = reorder.apply_rank_device(rowptr, col, val, rk2)
        ms, S, cov = spmm_ms(rp, ci, va, n)
        print(f"star merges (r01): {t_star * 1e3:.1f} ms  spmm {ms:.3f} ms (slices {S}) coverage {cov:.3f}", flush=True)
        if not args.no_host:
            rph, cih = rowptr.cpu().numpy(), col.cpu().numpy()
            t0 = time.perf_counter()
            rank_h, comm_h = reorder.order_rabbit(rph, cih, return_communities=True)
            t_host = time.perf_counter() - t0
            q_host = reorder.modularity(rowptr, col, torch.from_numpy(comm_h).to(dev))
            rp, ci, va, _ = reorder.apply_rank_device(rowptr, col, val, torch.from_numpy(rank_h).to(dev))
            ms, S, cov = spmm_ms(rp, ci, va, n)
            print(f"rabbit host: {t_host * 1e3:.1f} ms  communities {len(np.unique(comm_h))}  Q {q_host:.4f}  spmm {ms:.3f} ms "
                  f"(slices {S}) coverage {cov:.3f}   device/host time {t_host / t_dev:.1f}x  Q ratio {q_dev / q_host:.4f}", flush=True)


if __name__ == "__main__":
    main()
