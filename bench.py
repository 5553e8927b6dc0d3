#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric: SpMM GFLOP/s (+ achieved GB/s against the rooflines that bound
the kernel) on the Reddit-shaped graph (n = 232 965, nnz ≈ 114.85 M incl. self-loops), feature width 128,
fp32, no reorder, on 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one aggregation layer over the whole graph: C = Â·H (N = 1), or, for N > 1, every rank's
row-block SpMM followed by the exchange of the layer output over xGMI (the next layer's input) — strong
scaling, the graph is fixed.  For N > 1 every rank builds ITS row block only (graphgen.make_graph_row_block →
RowShardedAdjacency.from_row_block): no rank ever holds the whole CSR.  Inputs are resident in HBM when the
timed region starts.  After the timed loop the output of the last step (N = 1) — or, on the sharded path, of one more
layer of the same pipeline run right behind the timed ones, whose input can be kept without copying tens of GB inside
the timed region — is checked on sampled rows against an fp64 evaluation (torch arithmetic, gcn_amd/check.py); the run
fails above 1e-5.  Prints ONE JSON line on rank 0.

The other BASELINE configs run through the same code and print the same line (not the headline metric; the
driver only runs the default):
`--graph products` (config 3): products-shaped graph, feat = 256, renumbered by the library's device RCM
  (the same integers as the host order_rcm) before the SpMM; `--order none|deg|rcm|gorder` overrides.
`--graph rmat24` (config 5): Graph500 R-MAT scale 24 (`--rmat-scale` for smaller ones), feat = 512,
  `--order none|deg|rcm|gorder` (Gorder is the host algorithm, window 3; its host seconds are printed).
`--graph papers100m` (config 4): n = 111 059 956, 1.616 G directed R-MAT samples, ≈ 3.3 G non-zeros in the
  whole graph, generated block by block; N = 8 ranks hold one block each, and on ONE GPU the run is rank 0's
  share of the 8-way partition (compute only), stated as such.
For graphs whose feature table is far larger than the caches (configs 3-5) `roofline.bound` is "hbm" and
`roofline.frac` is the SURVEY §8(d) figure itself (algorithmic bytes / kernel time / 8 TB/s); for the column-
sliced Reddit-shaped run the gathers are served by L2 and `bound` says "l2".
"""
import argparse
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC: RCCL / device-tensor sharing across processes needs
# this before the HIP runtime starts (it is exported on the boxes already; kept here so a bare launch works)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import gcn_amd                      # noqa: E402
from gcn_amd import graphgen        # noqa: E402
from gcn_amd.check import sampled_rows_rel_err                       # noqa: E402
from gcn_amd.dist import PipelinedAggregation, RowShardedAdjacency   # noqa: E402

# MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 measured copy); L2 -> CU ≈ 34.5 TB/s aggregate; random
# 256-byte-row gathers from an Infinity-Cache-resident table 8.6 TB/s
HBM_PEAK = 8.0e12
L2_PEAK = 34.5e12
FABRIC_GATHER_CEILING = 8.6e12
K_FEAT = {"reddit": 128, "products": 256, "rmat24": 512, "papers100m": 128, "reddit-dcsbm": 128, "products-dcsbm": 256}   # BASELINE.json configs 2-5 (+ the structured stand-ins)
ORDER = {"reddit": "none", "products": "rcm", "rmat24": "none", "papers100m": "none", "reddit-dcsbm": "none", "products-dcsbm": "none"}
TOL = 1e-5
FULL_CHECK_MAX_N = 3_000_000      # "fp64 ... on the full matrix for n <= 3 M, on a fixed random sample of 4 096 rows otherwise"
CPU_SAMPLE_WORK = 1.6e10          # nnz x k of the CPU-baseline sample: about 10 s per torch.spmm on the box's host
PAPERS_N, PAPERS_SAMPLES = 111059956, 1615685872


def algorithmic_bytes(m, nnz, k):
    # one gathered feature row + col + val per non-zero, rowptr, C store
    return nnz * (4 + 4 + 4 * k) + (m + 1) * 4 + m * k * 4


def pmc_traffic(graph, k, order, passes, kernel):
    """Fabric-side bytes per main-kernel launch from the committed PMC summaries (separate `rocprofv3 --pmc`
    passes of this same bench command, tools/pmc_summary.py) — only if a summary was taken for this graph,
    width, ordering and the kernel this run timed; else None (traffic is then 'not collected').
    → (bytes, l2_hit_rate, file) or None"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    except (OSError, ValueError):
        return None
    for e in d.get("entries", [d]):
        try:
            if (e.get("graph") == graph and e.get("k") == k and e.get("order", "none") == order
                    and e.get("launches_per_spmm") == passes and any(kernel in name for name in e.get("kernel", []))):
                return int(e["traffic_bytes_per_launch"]), e.get("l2_hit_rate"), e.get("source", "profiles/pmc_latest.json")
        except (KeyError, TypeError, ValueError):
            continue
    return None


def cpu_baseline(rowptr, col, val, n, k, H, graph):
    """pygcn's CPU path: torch.spmm(adj_sparse_coo_fp32, dense) exactly as gcn1.py:53 issues it, on this box's
    host cores (baseline only): 1 warm-up + 3 timed repetitions, median.  Bounded sample: the
    leading rows of the matrix up to CPU_SAMPLE_WORK = nnz x k (the whole Reddit-shaped graph at k = 128 is just
    below it), against the full feature matrix."""
    rp = rowptr.cpu().long()
    budget = int(CPU_SAMPLE_WORK / k)
    rows_s = n if int(rp[-1]) <= budget else max(1, int(torch.searchsorted(rp, torch.tensor(budget))) - 1)
    nnz_s = int(rp[rows_s])
    rows = torch.repeat_interleave(torch.arange(rows_s, dtype=torch.int64), rp[1:rows_s + 1] - rp[:rows_s])
    idx = torch.stack([rows, col[:nnz_s].cpu().long()])
    adj = torch.sparse_coo_tensor(idx, val[:nnz_s].cpu(), (rows_s, n))      # as utils.py:243-250 builds it
    B = H.cpu()
    times = []
    torch.spmm(adj, B)                                         # warm-up
    for _ in range(3):
        t0 = time.perf_counter()
        torch.spmm(adj, B)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    whole = rows_s == n
    return {
        "value": round(2.0 * nnz_s * k / med / 1e9, 3), "unit": "GFLOP/s",
        "cores": torch.get_num_threads(), "kind": "reference",
        "sample": f"torch.spmm(sparse_coo fp32, dense) as pygcn/gcn1.py:53, "
                  + (f"full {graph}-shaped graph" if whole else f"leading {rows_s} of {n} rows of the {graph}-shaped graph")
                  + f" (nnz={nnz_s}, k={k}), median of {len(times)} runs after 1 warm-up, "
                  f"os.cpu_count()={os.cpu_count()}, torch {torch.__version__}",
        "seconds_per_spmm": round(med, 4),
    }


def offline_gorder_rank(rowptr, col, scale):
    """the rank tools/gorder_rmat24.py computed for exactly this graph (checked by hash), or None"""
    import hashlib
    import numpy as np
    meta_f = os.path.join(ROOT, "profiles", f"r03_gorder_rmat{scale}.json")
    rank_f = os.path.join(ROOT, "artifacts", f"gorder_rmat{scale}_rank.npy")
    if not (os.path.exists(meta_f) and os.path.exists(rank_f)):
        return None
    meta = json.load(open(meta_f))
    h = hashlib.sha256()
    h.update(rowptr.cpu().numpy().tobytes())
    h.update(col.cpu().numpy().tobytes())
    rank = np.load(rank_f)
    if h.hexdigest() != meta["graph_sha256"] or hashlib.sha256(rank.tobytes()).hexdigest() != meta["rank_sha256"]:
        sys.exit("bench.py: artifacts/gorder rank does not belong to this graph (hash mismatch)")
    return rank, meta


def renumber(rowptr, col, val, order, dev, offline=0):
    """the adjacency in the numbering of `order` (device CSR rewrite); → (rowptr, col, val, seconds of the ordering,
    where it ran).  deg / rcm: the library's device versions (bit-identical to order_deg / order_rcm of the host code);
    gorder: the host algorithm (window 3, renumber.cu:176), the only form that exists."""
    from gcn_amd import reorder
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if order == "deg":
        rank, where = reorder.order_deg_device(rowptr, col, "total", True), "device"
    elif order == "rcm":
        rank, where = reorder.order_rcm_device(rowptr, col), "device"
    elif order == "rabbit":
        rank, where = reorder.order_rabbit_device(rowptr, col), "device (parallel Rabbit, rabbit_device.hip)"
    elif order == "gorder":
        pre = offline_gorder_rank(rowptr, col, offline) if offline else None
        if pre is not None:
            rank, where = torch.from_numpy(pre[0].astype("int64")).to(dev), None
            off_meta = pre[1]
        else:
            rank = torch.from_numpy(reorder.order_gorder(rowptr.cpu().numpy(), col.cpu().numpy(), 3)).to(dev)
            where = "host (1 thread)"
    else:
        raise ValueError(order)
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    if where is None:                       # loaded: report what the off-line run measured
        secs = float(off_meta["gorder_host_seconds"])
        where = (f"host (1 thread), computed off-line by tools/gorder_rmat24.py on {off_meta.get('host_cpu', '?')} "
                 f"(profiles/r03_gorder_rmat{offline}.json; rank sha256 {off_meta['rank_sha256'][:16]})")
    rp, ci, va, _vomp = reorder.apply_rank_device(rowptr, col, val, rank)
    return rp, ci, va, secs, where


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--graph", default="reddit", choices=["reddit", "products", "rmat24", "papers100m", "reddit-dcsbm", "products-dcsbm"])
    ap.add_argument("--mixing", type=float, default=0.35, help="--graph *-dcsbm: share of edge samples drawn across communities")
    ap.add_argument("--communities", type=int, default=0, help="--graph *-dcsbm: planted communities (0: 200 for reddit-, 2000 for products-)")
    ap.add_argument("--size-skew", type=float, default=-1.0, help="--graph *-dcsbm: Zipf exponent of the community sizes (default 1.0 / 0.3)")
    ap.add_argument("--autotune", action="store_true",
                    help="single GPU: let CsrAdjacency.autotune() pick slices and column tile by measurement before the timed steps")
    ap.add_argument("--slices", type=int, default=-1, help="single GPU: explicit column-slice count of the plan (-1 = automatic)")
    ap.add_argument("--k", type=int, default=0, help="feature width (0 = the width BASELINE.json names for the graph)")
    ap.add_argument("--order", default="config", choices=["config", "none", "deg", "rcm", "gorder", "rabbit"],
                    help="single GPU: renumber the graph before the SpMM (config = what BASELINE.json names: RCM for "
                         "products, none otherwise)")
    ap.add_argument("--rmat-scale", type=int, default=24, help="--graph rmat24: log2 of the vertex count (24 = config 5)")
    ap.add_argument("--graph-device", default="gpu", choices=["gpu", "cpu"],
                    help="--graph rmat24: generate the graph with the CPU generators (the same integers on every machine; a "
                         "Gorder rank computed off-line by tools/gorder_rmat24.py is then loaded instead of recomputed)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; invalidates the metric)")
    ap.add_argument("--exchange", default="all_gather", choices=["all_gather", "direct", "push"],
                    help="N > 1: one RCCL all-gather per plane and layer, a grouped send/recv to every peer, or shards pushed "
                         "into the peers' IPC-mapped buffers by the copy engines (no collective kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-shard", action="store_true",
                    help="debug: run the row-sharded pipelined path even with one rank (no collective)")
    ap.add_argument("--chunk", type=int, default=0, help="debug: chunk size in non-zeros (0 = automatic)")
    ap.add_argument("--gather-width", type=int, default=-1,
                    help="debug: non-zeros per gather instruction of the 64-column kernel (0 auto, 1, 4)")
    ap.add_argument("--blocks-per-cu", type=int, default=0, help="debug: persistent-grid blocks per CU (1..8)")
    ap.add_argument("--no-prelaid", action="store_true",
                    help="debug: N > 1 / --sim-world: every layer re-lays its input (scaled copy of B per plane and layer) "
                         "instead of writing the next layer's pre-laid input in the epilogue")
    ap.add_argument("--no-plane-streams", action="store_true",
                    help="debug: N > 1 / --sim-world: all column planes on one stream (default: one stream per plane)")
    ap.add_argument("--plane-cols", type=int, default=64,
                    help="debug: N > 1 / --sim-world: columns per plane (64: two planes pipeline their exchanges under each other's "
                         "SpMM; 128: ONE plane, one launch whose tiles re-read the stream from the Infinity Cache, one exchange behind it)")
    ap.add_argument("--sim-world", type=int, default=0,
                    help="debug: on ONE GPU, time rank 0's row block of a W-way partition (compute only, "
                         "no collective) — a rehearsal of the per-rank work at N = W, not a metric")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch N>1 with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (no CPU fallback for the product path)")
    # rehearsal on a ONE-GPU box (not a metric): GCN_AMD_BENCH_REHEARSAL=1 puts every rank on cuda:0 and
    # swaps RCCL (which refuses two ranks on one device) for gloo — same partition, same pipeline, same
    # kernels, same checks; only the transport differs
    rehearsal = os.environ.get("GCN_AMD_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")   # collective kernels ahead of compute in the HW queues
            dist.init_process_group("nccl", device_id=dev)          # nccl == RCCL on ROCm
    k = args.k or K_FEAT[args.graph]
    order = ORDER[args.graph] if args.order == "config" else args.order
    papers = args.graph == "papers100m"
    dcsbm = args.graph.endswith("-dcsbm")
    if dcsbm:                              # (degree-corrected planted partitions at the reddit- / products-shaped size)
        shape = graphgen.SHAPES[args.graph.split("-")[0]]
        ncomm = args.communities or (200 if args.graph == "reddit-dcsbm" else 2000)
        skew = args.size_skew if args.size_skew >= 0 else (1.0 if args.graph == "reddit-dcsbm" else 0.3)
        maxdeg = 20000 if args.graph == "reddit-dcsbm" else 10000
    sim = args.sim_world if (world == 1 and args.sim_world > 1) else 0
    if papers and world == 1 and not sim:
        sim = 8                                              # one GPU: rank 0's share of the 8-way partition
    sharded = world > 1 or args.force_shard or sim > 1
    part_world, part_rank = (sim, 0) if sim else (world, rank)
    if sharded and (order != "none" or args.graph in ("products", "rmat24") or dcsbm):
        sys.exit("bench.py: the row-sharded path runs the reddit / papers100m graphs un-renumbered")
    order_secs, order_where = 0.0, ""

    def make_local(rp, ci, va, shape, slices="auto"):
        return gcn_amd.CsrAdjacency(rp, ci, va, shape, chunk_nnz=args.chunk, slices=slices)

    # ---- inputs ------------------------------------------------------------------------------
    rowptr = col = val = None
    if not sharded:
        if args.graph == "rmat24" and args.graph_device == "cpu":
            rowptr, col, val, n = (t.to(dev) if torch.is_tensor(t) else t
                                   for t in graphgen.make_rmat(args.rmat_scale, device="cpu", seed=5))
        elif args.graph == "rmat24":
            rowptr, col, val, n = graphgen.make_rmat(args.rmat_scale, device=dev, seed=5)
        elif dcsbm:
            rowptr, col, val, n = graphgen.make_dcsbm(n=max(16, int(shape["n"] * args.scale)), edges=int(shape["edges"] * args.scale),
                                                      communities=ncomm, mixing=args.mixing, size_skew=skew, max_degree=maxdeg,
                                                      device=dev, seed=11)
        else:
            rowptr, col, val, n = graphgen.make_graph(args.graph, device=dev, seed=1, scale=args.scale)
        if order != "none":
            rowptr, col, val, order_secs, order_where = renumber(rowptr, col, val, order, dev,
                                                                 offline=args.rmat_scale if args.graph_device == "cpu" else 0)
            torch.cuda.empty_cache()
        nnz = int(col.numel())
        H = graphgen.random_features(n, k, seed=2, device=dev)
        torch.cuda.synchronize(dev)
        t_plan = time.perf_counter()
        adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True, chunk_nnz=args.chunk,
                                   slices="auto" if args.slices < 0 else args.slices)
        out = torch.empty((n, k), dtype=torch.float32, device=dev)
        tuned = adj.autotune(k=k) if args.autotune else None
        adj.matmul_raw(H, out=out)                           # (the first call builds what the plan builds lazily for this width)
        torch.cuda.synchronize(dev)
        plan_secs = time.perf_counter() - t_plan

        def step():
            adj.matmul_raw(H, out=out)
        local_adj, local_nnz, local_m = adj, nnz, n
        launches_per_step = 1
        collective = "none"
    else:
        if papers:
            n = int(PAPERS_N * args.scale)
            lrp, lcol, lval, n, lo, hi, deg = graphgen.make_rmat_row_block(
                n, int(PAPERS_SAMPLES * args.scale), part_world, part_rank, device=dev, seed=4)
            rows_per = (n + part_world - 1) // part_world
            bounds = [min(n, p * rows_per) for p in range(part_world + 1)]
            nnz = int(deg.sum())                             # whole graph
            u = graphgen.value_factor_from_degrees(deg)
            del deg
        else:
            lrp, lcol, lval, n, bounds, u, nnz = graphgen.make_graph_row_block(
                args.graph, part_world, part_rank, device=dev, seed=1, scale=args.scale)
        shard = RowShardedAdjacency.from_row_block(lrp, lcol, lval, bounds, part_rank, part_world, make_local,
                                                   value_factor=u, total_nnz=nnz, exchange=args.exchange,
                                                   prelaid=False if args.no_prelaid else "auto", plane_cols=args.plane_cols)
        del lcol, u
        if sim:
            shard.collective = False
        # column planes of 64: the exchange of one plane overlaps the SpMM of the next
        pipe = PipelinedAggregation(shard, k, dev, plane_cols=args.plane_cols, streams=False if args.no_plane_streams else None)

        def fill(p, buf):                 # features exist in the exchange layout only: every rank fills ITS rows,
            g = torch.Generator(device=dev)   # one exchange assembles the layer input
            g.manual_seed(1000 * (shard.rank + 1) + p)
            # (pre-laid layout: the buffer holds diag(u)·H — any values do as the input of the first layer)
            buf[shard.buffer_rows_of_local_rows(dev)] = torch.randn((shard.rows, buf.shape[1]), generator=g, device=dev)
            if sim:                       # one GPU standing in for rank 0 of W: the peers' rows are made up locally
                g.manual_seed(7 + p)
                for q in range(1, shard.world):
                    idx = shard.buffer_rows_of_rank(q, dev)
                    buf[idx] = torch.randn((idx.numel(), buf.shape[1]), generator=g, device=dev)
            elif world > 1:
                shard._exchange(buf, shard._slot(buf), None, False)
        pipe.load_padded_block(fill)
        if world > 1 or sim:
            # One stream + one operator per plane (PipelinedAggregation): the tail kernels and launch gaps of
            # one plane hide under the main kernel of the other.  Grid of 8 blocks per CU and plane: the two
            # concurrent main kernels keep every CU full and are oversubscribed together, so blocks retire
            # every few tens of microseconds and the RCCL kernels (high-priority stream) get their workgroups
            # in as they do (profiles/r01f_sim8_streams.log).
            pipe.set_local_option("set_blocks_per_cu", 8)
        launches_per_step = len(pipe.widths)

        def step():                       # layer l+1 consumes the exchanged output of layer l
            pipe.step()
        local_adj, local_nnz, local_m = shard.local, shard.local_nnz, shard.rows
        collective = shard.collective_form()
        torch.cuda.empty_cache()

    for opt_name, opt_val in (("set_gather_width", args.gather_width if args.gather_width >= 0 else None),
                              ("set_blocks_per_cu", args.blocks_per_cu if args.blocks_per_cu > 0 else None)):
        if opt_val is not None:
            (pipe.set_local_option(opt_name, opt_val) if sharded else getattr(local_adj, opt_name)(opt_val))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if sharded:
        pipe.finish()
    local_adj.profile_begin(args.steps * launches_per_step)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    if sharded:
        pipe.finish()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = local_adj.profile_end()
    if sharded:
        shard.check_exchange()                            # (push form: did a wait for a peer's flag give up?)
        # The check of the sharded path needs a layer's INPUT beside its output, i.e. a copy of the exchange buffers — tens
        # of GB for the papers100M-shaped graph, whose allocation alone took 1.6 s inside the timed loop when it was made
        # there.  So: the timed layers run undisturbed, and ONE more layer of the same pipeline, right behind them and
        # outside the clock, is the one that is checked.
        last_in = [b.clone() for b in pipe.src]
        step()
        pipe.finish()
        barrier()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the output that was timed, checked on sampled rows against fp64 (every rank its own rows) --------
    gsel = torch.Generator(device="cpu")
    gsel.manual_seed(1234 + rank)
    # BASELINE.md §3 / SURVEY §8(d): the FULL matrix for n <= 3 M (every output row of every rank), a fixed random sample of
    # rows beyond (4 096 on one GPU)
    full_check = n <= FULL_CHECK_MAX_N
    nsample = local_m if full_check else min(4096 if world == 1 else 1024, local_m)
    sel = torch.arange(local_m) if full_check else torch.randperm(local_m, generator=gsel)[:nsample].sort().values
    if not sharded:
        rel, checked = sampled_rows_rel_err(rowptr, col, val, H, out, sel, batch_nnz=1 << 22)
    else:
        op_rp, op_col, op_val = shard.as_buffer_operator()      # the layer as a map between exchange buffers
        own = shard.buffer_rows_of_local_rows(dev)
        rel, checked = 0.0, 0
        for p, w in enumerate(pipe.widths):                   # plane by plane: Â·(plane of the last layer's input)
            got = pipe.src[p].index_select(0, own)
            r_p, checked = sampled_rows_rel_err(op_rp, op_col, op_val, last_in[p], got, sel)
            rel = max(rel, r_p)
        del last_in
    verdict = torch.tensor([rel], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(verdict, op=dist.ReduceOp.MAX)        # every rank learns the worst error: all fail together
    rel_all = float(verdict.item())
    check_failed = not (rel_all <= TOL)

    # ---- roofline of the dominant kernel (the plan's main kernel), this rank's launches ---------
    # (N = 1: one timed interval per SpMM = all column passes of the main kernel; N > 1: one per
    #  64-column plane SpMM of this rank's row block)
    kp = k if not sharded else min(k, args.plane_cols)    # columns per timed SpMM
    passes = local_adj.num_passes(kp)                     # main-kernel launches per timed SpMM
    spmm_avg = sum(kernel_ms) / max(len(kernel_ms), 1) * 1e-3
    kavg = spmm_avg / passes                              # per LAUNCH, what rocprofv3 --stats averages
    cols_per_launch = kp // passes
    balg = algorithmic_bytes(local_m, local_nnz, kp) / passes
    achieved = balg / kavg if kavg > 0 else 0.0
    gathered = local_nnz * cols_per_launch * 4            # feature-row bytes the kernel pulls through L2 -> CU per launch
    kname = local_adj.main_kernel(kp)
    full_size = args.scale == 1.0 and (args.graph != "rmat24" or args.rmat_scale == 24)
    pmc = pmc_traffic(args.graph, k, order, passes, kname.split("<")[0]) if (not sharded and full_size and
                                                                             args.graph_device == "gpu") else None
    traffic, pmc_hit, pmc_src = pmc if pmc else (None, None, None)
    n_cols = n if not sharded else shard.world * shard.max_rows
    compulsory = local_nnz * 8 + (local_m + 1) * 4 + n_cols * kp * 4 + local_m * kp * 4
    # Which wall?  A column-sliced plan keeps its gathers in the XCDs' L2s (the Reddit-shaped graph: 93 % hits): the
    # bounded fraction is L2 -> CU bytes.  An unsliced plan on a table far larger than the caches gathers from HBM /
    # Infinity Cache: SURVEY §8(d)'s algorithmic bytes over the HBM peak IS the fraction there (<= 1 by construction).
    # (a table that fits the 256 MiB Infinity Cache as it is never reaches HBM either: same wall, same fraction)
    l2_bound = local_adj.num_slices > 1 or n_cols * kp * 4 <= (256 << 20)
    frac_l2 = round(gathered / kavg / L2_PEAK, 4) if kavg > 0 else None
    frac_alg = round(achieved / HBM_PEAK, 4)
    concurrent = len(pipe.widths) if (sharded and pipe.streams is not None) else 1
    if concurrent > 1:
        # the planes' main kernels run side by side on their own streams: a launch's duration includes the share of the
        # chip the other plane holds, so the fractions are taken over the LAYER (all planes' bytes / step time)
        layer_s = elapsed / args.steps
        frac_l2 = round(local_nnz * k * 4 / layer_s / L2_PEAK, 4)
        frac_alg = round(algorithmic_bytes(local_m, local_nnz, k) / layer_s / HBM_PEAK, 4)

    # the same SpMM with the matrix values kept (matrices whose values do not factor as u[r]*u[c] run this pass):
    # a second plan told to forget the factors (API, not an environment switch), a short timed loop of its own
    weighted_ms = None
    if not sharded and l2_bound and local_adj.has_value_factors and not check_failed:
        adj_w = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True, chunk_nnz=args.chunk)
        adj_w.plan                                                         # noqa: B018  (builds the plan)
        adj_w.set_value_factors(None, None)
        reps = max(3, min(args.steps, 10))
        for _ in range(2):
            adj_w.matmul_raw(H, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            adj_w.matmul_raw(H, out=out)
        e1.record()
        torch.cuda.synchronize()
        weighted_ms = {"ms_per_spmm": round(e0.elapsed_time(e1) / reps, 4), "kernel": adj_w.main_kernel(k), "spmms_timed": reps,
                       "what": "whole SpMM with the value stream kept (gcn_spmm_plan_set_value_factors(null, null)): what an "
                               "adjacency whose values are not u[r]*u[c] costs on this graph"}
        del adj_w

    if rank == 0:
        flops = 2.0 * nnz * k if not sim else 2.0 * local_nnz * k
        ord_txt = {"none": "no reorder", "deg": "degree-descending order (order_deg)", "rcm": "RCM order (order_rcm)",
                   "gorder": "Gorder (RCM then Gorder, window 3)", "rabbit": "Rabbit order (parallel Rabbit on the device)"}[order]
        gname = (f"R-MAT scale {args.rmat_scale} (Graph500 a,b,c,d = .57,.19,.19,.05, edge factor 16, seed 5, labels permuted with seed 1005"
                 + (", CPU generator" if args.graph_device == "cpu" else "") + ")") if args.graph == "rmat24" \
            else None
        if papers:
            gname = (f"papers100M-shaped R-MAT graph (graphgen.make_rmat_row_block: a,b,c,d = .57,.19,.19,.05, {PAPERS_SAMPLES} directed "
                     "samples, seed 4, labels permuted with seed 1004)")
        elif dcsbm:
            gname = (f"{args.graph.split('-')[0]}-sized degree-corrected planted-partition graph (graphgen.make_dcsbm: {ncomm} communities of "
                     f"Zipf sizes (exponent {skew}), mixing {args.mixing}, power-law degrees gamma 2.5 capped near {maxdeg}, seed 11, labels shuffled)")
        elif gname is None:
            gname = (f"{args.graph}-shaped R-MAT graph (gcn_amd.graphgen.make_graph: a,b,c,d = "
                     + ",".join(f"{x:g}" for x in graphgen.SHAPES[args.graph]["abcd"]) + f", {graphgen.SHAPES[args.graph]['edges']} distinct "
                     "undirected edges, seed 1, vertex labels randomly permuted with seed 1001, symmetrised, + I, "
                     "D^-1/2 (A+I) D^-1/2; the milder skew than SURVEY §8(d)'s .57,.19,.19,.05 is deliberate, graphgen.py:18-21)")
        line = {
            "metric": "SpMM GFLOP/s + achieved HBM GB/s, Reddit feat=128, 1/2/4/8 MI355X",
            "value": round(flops * args.steps / elapsed / 1e9, 2),
            "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL (all ranks on one GPU, gloo transport) - not a metric",
            "config": {
                "workload": (f"{gname}, n={n}, nnz={nnz} (incl. self-loops), "
                             f"feat={k}, fp32, {ord_txt}; step = C = Â·H"
                             + ("" if world == 1 else " per row block + exchange of the layer output")
                             + (f"; ONE GPU computing rank 0's row block of a {sim}-way partition (rows={local_m}, "
                                f"nnz={local_nnz}), no exchange: not the headline metric" if sim else "")
                             + ("" if args.graph == "reddit" and order == "none" and not sim else
                                "; NOT the headline config (BASELINE.json metric is quoted on reddit feat=128)")),
                "graph": args.graph, "n": n, "nnz": nnz, "k": k, "order": order,
                "parallelism": "single GPU" if world == 1 else f"1-D row partition x{world} (nnz-balanced, every rank built from its own block), {collective} per plane and layer",
                "collective": collective, "ranks_seen": world,
                "prelaid": bool(sharded and shard.prelaid),
                "chunks": f"{local_adj.num_chunks} x {local_adj.chunk_size} nnz",
            },
            "check": {"rel_err": rel_all, "tol": TOL, "rows_per_rank": int(checked), "passed": not check_failed,
                      "what": ("the layer right behind the timed ones (same pipeline, outside the clock)" if sharded else "last timed step's output")
                              + (" vs fp64 evaluation of EVERY output row (n <= 3 M: the full matrix" if full_check else
                                 " vs fp64 evaluation of sampled rows (n > 3 M") + "; torch, gcn_amd/check.py), max over ranks",
                      "rows_total": int(local_m), "full_matrix": bool(full_check)},
            "roofline": {
                "bound": "l2" if l2_bound else "hbm",
                "kernel": kname,
                "slices": local_adj.num_slices,
                # SURVEY §8(d) contract figure: one gathered feature row per non-zero charged to HBM
                "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "algorithmic_over_hbm_peak": frac_alg,
                "algorithmic_note": ("not a roofline fraction here: most gathered rows are served by L2, so the contract's byte "
                                     "count exceeds what crosses any one interface") if l2_bound else
                                    "the table is far larger than the caches: this IS the fraction of the HBM roofline (frac)",
                # the physically bounded fraction of the wall this kernel is bound by
                "frac": frac_l2 if l2_bound else frac_alg,
                "frac_of": ("L2->CU gather path: gathered feature-row bytes per launch / kernel time / 34.5 TB/s "
                            "(MI355X_MICROARCH.md §L2) - the wall the sliced kernel is bound by (texture addressers busy 85 %)")
                           if l2_bound else "HBM: SURVEY §8(d) algorithmic bytes per launch / kernel time / 8.0 TB/s",
                "frac_l2": frac_l2,
                "gathered_bytes_per_launch": int(gathered),
                "frac_fabric": None if traffic is None or kavg <= 0 else round(traffic / kavg / FABRIC_GATHER_CEILING, 4),
                "fabric_ceiling": "8.6 TB/s: random 256-byte-row gathers from an Infinity-Cache-resident table (MI355X_MICROARCH.md)",
                "frac_hbm_compulsory": round(compulsory / spmm_avg / HBM_PEAK, 4) if spmm_avg > 0 else None,
                "algorithmic_bytes_per_launch": int(balg),
                "compulsory_bytes_per_spmm": int(compulsory),
                "launches_per_spmm": passes, "columns_per_launch": cols_per_launch, "concurrent_planes": concurrent,
                "frac_note": None if concurrent == 1 else "planes run concurrently: frac / frac_l2 / algorithmic_over_hbm_peak are "
                             "taken over the whole layer (all planes' bytes / ms_per_step), kernel_ms_avg stays per plane launch",
                "kernel_ms_avg": round(kavg * 1e3, 4),
                "spmm_ms_min": round(min(kernel_ms), 4) if kernel_ms else None,
                "spmms_timed": len(kernel_ms),
                "timing": "HIP events recorded by libgcnspmm on the launch stream around the main-kernel "
                          "passes of every timed SpMM (gcn_spmm_profile_begin/_end)",
                "traffic": traffic,
                "traffic_GBps": None if traffic is None or kavg <= 0 else round(traffic / kavg / 1e9, 1),
                "traffic_over_compulsory": None if traffic is None else round(traffic / compulsory, 3),
                "l2_hit_rate": pmc_hit,
                "traffic_source": "not collected in this run" if traffic is None else
                f"committed profile of this same command and kernel, NOT measured in this run: {pmc_src} "
                "(separate rocprofv3 --pmc passes; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; L2-miss bytes incl. "
                "Infinity-Cache hits)",
                "weighted": weighted_ms,
            },
            "gflops_kernel_only": round(2.0 * local_nnz * kp / spmm_avg / 1e9, 1) if spmm_avg > 0 else None,
        }
        if order != "none":
            line["config"]["ordering_seconds"] = round(order_secs, 3)
        if not sharded:
            # the analogue of the reference's csr2tile (tile.cu:104-169, host, 0.53 s per 3.4 M non-zeros): CSR on the device
            # -> plan (value factors, column slices, 15-bit stream, cut lists) + the first SpMM; outside the timed steps
            line["config"]["plan_build_seconds"] = round(plan_secs, 3)
            if tuned is not None:
                line["config"]["autotune"] = {f"slices={s},tile={t}": round(ms, 4) for (s, t), ms in tuned.items()}
            line["config"]["ordering_ran_on"] = order_where
        if world > 1 or sim:
            # what the exchange may cost before 6x at 8 GPUs is lost: 8 ranks must finish a layer in (single-GPU step / 6)
            line["scaling_budget"] = {
                "layer_ms_this_run": round(elapsed / args.steps * 1e3, 4),
                "note": "a 6x aggregate at N = 8 needs (rank share + unhidden exchange) <= ms_per_step(N = 1) / 6; "
                        "ms_per_step(N = 1) is the driver's own N = 1 run (2.66-2.70 ms in round 4 -> 0.443-0.450 ms)",
            }
        if not sharded and not args.no_cpu_baseline and not check_failed:
            line["cpu_baseline"] = cpu_baseline(rowptr, col, val, n, k, H, args.graph)
        print(json.dumps(line), flush=True)

    if world > 1:
        shard.close_exchange()
        dist.destroy_process_group()
    if check_failed:
        sys.exit(f"bench.py: output check failed: rel err {rel_all} > {TOL}")


if __name__ == "__main__":
    main()
