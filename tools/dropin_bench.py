#!/usr/bin/env python3
"""tools/dropin_bench.py — the reference's own call sequence (gcn6.py:334-366: csr2tile on the host, buffers to the
device, flexspmm per layer) through the drop-in symbols, timed on the Reddit-shaped graph, beside the plan API.

    python tools/dropin_bench.py [--scale 1.0] [--k 128] [--iters 20]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd                              # noqa: E402
from gcn_amd import dropin, graphgen, check  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    d = torch.device("cuda:0")
    rowptr, col, val, n = graphgen.make_graph("reddit", device=d, seed=1, scale=args.scale)
    nnz, k = int(col.numel()), args.k
    t0 = time.time()
    packed = dropin.csr2tile(rowptr.cpu(), col.cpu(), val.cpu(), n, n, nnz, torch.arange(n, dtype=torch.int32))
    t_pack = time.time() - t0
    seg_rowPtr, segNzCV, segVoMap, tail, nxt, n_segs = packed
    hdr = seg_rowPtr[:9].tolist()
    dev = [t.to(d) for t in (seg_rowPtr, segNzCV, segVoMap, tail, nxt)]
    B = graphgen.random_features(n, k, seed=2, device=d)

    def run():
        return dropin.flexspmm.apply(dev[0], dev[1], dev[2], n, n, int(n_segs[0]), dev[3], dev[4], B)

    for _ in range(3):
        C = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    e0.record()
    for _ in range(args.iters):
        C = run()
    e1.record()
    torch.cuda.synchronize()
    wall = (time.time() - t0) / args.iters * 1e3
    ms = e0.elapsed_time(e1) / args.iters
    rows = torch.arange(0, n, max(1, n // 2048), device=d)
    err = check.sampled_rows_rel_err(rowptr, col, val, B, C, rows)
    adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True)
    out = torch.empty((n, k), device=d)
    for _ in range(3):
        adj.matmul_raw(B, out=out)
    e0.record()
    for _ in range(args.iters):
        adj.matmul_raw(B, out=out)
    e1.record()
    torch.cuda.synchronize()
    plan_ms = e0.elapsed_time(e1) / args.iters
    print(json.dumps({"graph": "reddit-shaped", "n": n, "nnz": nnz, "k": k, "csr2tile_host_s": round(t_pack, 2),
                      "format": "group" if hdr[0] == 0x47434E47 else "csr", "slices": hdr[1] if hdr[0] == 0x47434E47 else 0,
                      "value_free": hdr[6] if hdr[0] == 0x47434E47 else None,
                      "flexspmm_ms_gpu": round(ms, 4), "flexspmm_ms_wall": round(wall, 4),
                      "flexspmm_GFLOPs": round(2.0 * nnz * k / ms / 1e6, 1), "rel_err_vs_fp64": err,
                      "plan_api_ms": round(plan_ms, 4)}))


if __name__ == "__main__":
    main()
