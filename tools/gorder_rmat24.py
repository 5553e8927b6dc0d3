#!/usr/bin/env python3
"""tools/gorder_rmat24.py — BASELINE config 5's Gorder leg at the stated size, computed ONCE on host cores.

Gorder (RCM then the windowed priority order of order_gorder.cu:35-143, window 3 as renumber.cu:176) is an
inherently sequential algorithm whose work grows with the sum of squared degrees: on the Graph500 R-MAT graph of
scale 24 it needs the better part of an hour of one core, more than one gpurun call allows (20 min).  This tool
generates the graph on the CPU (torch CPU generators: the same integers on every machine, unlike the device
generators bench.py uses by default), runs the library's host Gorder on it and stores

    artifacts/gorder_rmat<scale>_rank.npy   rank[old] = new, int32 (git-ignored: 67 MB at scale 24; it travels to
                                            the GPU box with the snapshot)
    profiles/r03_gorder_rmat<scale>.json    graph hash, rank hash, host seconds, host CPU

`bench.py --graph rmat24 --graph-device cpu --order gorder` regenerates the same graph (checked by hash) and
loads the rank instead of recomputing it.   python tools/gorder_rmat24.py [--scale 24]"""
import argparse
import hashlib
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcn_amd import graphgen, reorder      # noqa: E402


def graph_hash(rowptr, col):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(rowptr).tobytes())
    h.update(np.ascontiguousarray(col).tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    args = ap.parse_args()
    t0 = time.time()
    rowptr, col, _val, n = graphgen.make_rmat(args.scale, device="cpu", seed=5)
    rp, ci = rowptr.numpy(), col.numpy()
    gen_s = time.time() - t0
    gh = graph_hash(rp, ci)
    print(f"graph: n={n} nnz={len(ci)} generated in {gen_s:.1f} s, sha256 {gh[:16]}", flush=True)
    t0 = time.time()
    rank = reorder.order_gorder(rp, ci, 3)
    secs = time.time() - t0
    assert np.array_equal(np.sort(rank), np.arange(n))
    r32 = rank.astype(np.int32)
    os.makedirs(os.path.join(ROOT, "artifacts"), exist_ok=True)
    np.save(os.path.join(ROOT, "artifacts", f"gorder_rmat{args.scale}_rank.npy"), r32)
    cpu = ""
    try:
        cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    meta = {"graph": f"R-MAT scale {args.scale}, edge factor 16, (.57,.19,.19,.05), seed 5, CPU generator (graphgen.make_rmat)",
            "n": int(n), "nnz": int(len(ci)), "graph_sha256": gh, "rank_sha256": hashlib.sha256(r32.tobytes()).hexdigest(),
            "gorder_host_seconds": round(secs, 1), "threads": 1, "host_cpu": cpu, "machine": platform.machine(),
            "window": 3, "torch": torch.__version__}
    json.dump(meta, open(os.path.join(ROOT, "profiles", f"r03_gorder_rmat{args.scale}.json"), "w"), indent=1)
    print(json.dumps(meta), flush=True)


if __name__ == "__main__":
    main()
