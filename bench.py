#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric: SpMM GFLOP/s (+ achieved GB/s against the HBM
roofline) on the Reddit-shaped graph (n = 232 965, nnz ≈ 114.85 M incl. self-loops),
feature width 128, fp32, no reorder, on 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one aggregation layer over the whole graph: C = Â·H (N = 1), or, for N > 1,
every rank's row-block SpMM followed by the RCCL all-gather of the layer output (the next
layer's input) — strong scaling, the graph is fixed.  Inputs are resident in HBM when the
timed region starts.  Prints ONE JSON line on rank 0.
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
from gcn_amd.dist import PipelinedAggregation, RowShardedAdjacency   # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md); measured copy peak 6.29e12
K_FEAT = 128


def algorithmic_bytes(m, nnz, k):
    # one gathered feature row + col + val per non-zero, rowptr, C store
    return nnz * (4 + 4 + 4 * k) + (m + 1) * 4 + m * k * 4


def pmc_traffic(graph, k, passes):
    """HBM-side bytes per main-kernel launch from the committed PMC summary (collected in separate
    `rocprofv3 --pmc` passes of this same bench command, tools/pmc_summary.py), or None when the
    summary was taken for a different configuration."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        if d.get("graph") == graph and d.get("k") == k and d.get("launches_per_spmm") == passes:
            return int(d["traffic_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(rowptr, col, val, n, k, seed):
    """pygcn's CPU path: torch.spmm(adj_sparse_coo_fp32, dense) exactly as gcn1.py:53 issues it,
    on this box's host cores (baseline only; bounded to ~30 s)."""
    rp = rowptr.cpu().long()
    rows = torch.repeat_interleave(torch.arange(n, dtype=torch.int64), rp[1:] - rp[:-1])
    idx = torch.stack([rows, col.cpu().long()])
    adj = torch.sparse_coo_tensor(idx, val.cpu(), (n, n))      # as utils.py:243-250 builds it
    B = graphgen.random_features(n, k, seed=seed, device="cpu")
    nnz = int(val.numel())
    times = []
    t_begin = time.perf_counter()
    torch.spmm(adj, B)                                         # warm-up
    for _ in range(3):
        t0 = time.perf_counter()
        torch.spmm(adj, B)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_begin > 30:
            break
    times.sort()
    med = times[len(times) // 2]
    return {
        "value": round(2.0 * nnz * k / med / 1e9, 3), "unit": "GFLOP/s",
        "cores": torch.get_num_threads(), "kind": "reference",
        "sample": f"torch.spmm(sparse_coo fp32, dense) as pygcn/gcn1.py:53, full Reddit-shaped graph "
                  f"(nnz={nnz}, k={k}), median of {len(times)} runs after 1 warm-up, "
                  f"os.cpu_count()={os.cpu_count()}, torch {torch.__version__}",
        "seconds_per_spmm": round(med, 4),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--graph", default="reddit")
    ap.add_argument("--k", type=int, default=K_FEAT)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-shard", action="store_true",
                    help="debug: run the row-sharded pipelined path even with one rank (no collective)")
    ap.add_argument("--chunk", type=int, default=0, help="debug: chunk size in non-zeros (0 = automatic)")
    ap.add_argument("--gather-width", type=int, default=-1,
                    help="debug: non-zeros per gather instruction of the 64-column kernel (0 auto, 1, 4)")
    ap.add_argument("--blocks-per-cu", type=int, default=0, help="debug: persistent-grid blocks per CU (1..8)")
    ap.add_argument("--no-plane-streams", action="store_true",
                    help="debug: N > 1 / --sim-world: all column planes on one stream (default: one stream per plane)")
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

    # ---- inputs (same seeds on every rank → identical graph everywhere) -------------------
    rowptr, col, val, n = graphgen.make_graph(args.graph, device=dev, seed=1, scale=args.scale)
    nnz, k = int(col.numel()), args.k
    H = graphgen.random_features(n, k, seed=2, device=dev)

    if world > 1:
        # every rank generated the graph itself (same seeds): make sure they really agree before the
        # partition is derived from it — a mismatch would desynchronise the all-gather shapes
        sig = torch.stack([torch.tensor(float(nnz), device=dev, dtype=torch.float64),
                           rowptr.double().sum(), col.double().sum()])     # integer sums: exact in fp64
        lo, hi = sig.clone(), sig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise RuntimeError(f"rank {rank}: synthetic graph differs between ranks: {sig.tolist()}")

    shard_check = None
    sharded = world > 1 or args.force_shard or args.sim_world > 1
    if not sharded:
        adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True, chunk_nnz=args.chunk)
        out = torch.empty((n, k), dtype=torch.float32, device=dev)

        def step():
            adj.matmul_raw(H, out=out)
        local_adj, local_nnz, local_m = adj, nnz, n
        launches_per_step = 1
    else:
        sim = args.sim_world if (world == 1 and args.sim_world > 1) else 0
        # Â = D^-1/2 (A+I) D^-1/2: its values factor as u[r]·u[c] with u = sqrt(diag Â); a row block with
        # renumbered columns cannot see that by itself, so the factor travels with the partition
        rows_of = torch.repeat_interleave(torch.arange(n, device=dev), (rowptr[1:] - rowptr[:-1]).long())
        on_diag = rows_of == col.long()
        u = torch.zeros(n, device=dev)
        u[rows_of[on_diag]] = val[on_diag].sqrt()
        del rows_of, on_diag
        shard = RowShardedAdjacency(rowptr, col, val, n, rank, sim or world,
                                    lambda rp, ci, va, shape: gcn_amd.CsrAdjacency(rp, ci, va, shape, chunk_nnz=args.chunk),
                                    value_factor=u)
        if sim:
            shard.collective = False
        # column planes of 64: the RCCL all-gather of one plane overlaps the SpMM of the next
        pipe = PipelinedAggregation(shard, k, dev, plane_cols=64, streams=False if args.no_plane_streams else None)
        pipe.load(H)
        if world > 1 or sim:
            # One stream + one operator per plane (PipelinedAggregation): the tail kernels and launch gaps of
            # one plane hide under the main kernel of the other.  Grid of 8 blocks per CU and plane: the two
            # concurrent main kernels keep every CU full (4 resident blocks of the 108-VGPR kernel) and are
            # 4x oversubscribed together, so blocks retire every few tens of microseconds and the RCCL
            # all-gather (high-priority stream) gets its workgroups in as they do.  Rank-0 share of an 8-way
            # partition, compute only (profiles/r01f_sim8_streams.log): 0.501 ms/step, against 0.546 ms on one
            # stream with 3 blocks per CU held free for RCCL, 0.527 ms on one stream with 8.
            pipe.set_local_option("set_blocks_per_cu", 8)
        launches_per_step = len(pipe.widths)

        def step():                       # layer l+1 consumes the all-gathered output of layer l
            pipe.step()
        local_adj, local_nnz, local_m = shard.local, shard.local_nnz, shard.rows
        if world > 1:
            # one untimed layer through the sharded path (row blocks + all-gathers), checked on rank 0
            # against the same layer on the unpartitioned matrix: the N > 1 result must be the 1-GPU result
            pipe.step()
            got = pipe.result()
            if rank == 0:
                full = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True).matmul_raw(H)
                shard_check = float((got - full).abs().max() / full.abs().max())
                del full
                if not shard_check <= 1e-5:
                    raise RuntimeError(f"sharded layer differs from the single-GPU layer: rel err {shard_check}")
            del got
            pipe.load(H)
        del rowptr, col, val
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
    local_adj.profile_begin(args.steps * launches_per_step)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if sharded:
        pipe.finish()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = local_adj.profile_end()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (the plan's main kernel), this rank's launches ---------
    # (N = 1: one timed interval per SpMM = all column passes of the main kernel; N > 1: one per
    #  64-column plane SpMM of this rank's row block)
    kp = k if not sharded else min(k, 64)                 # columns per timed SpMM
    passes = local_adj.num_passes(kp)                     # main-kernel launches per timed SpMM
    spmm_avg = sum(kernel_ms) / max(len(kernel_ms), 1) * 1e-3
    kavg = spmm_avg / passes                              # per LAUNCH, what rocprofv3 --stats averages
    balg = algorithmic_bytes(local_m, local_nnz, kp) / passes
    achieved = balg / kavg if kavg > 0 else 0.0
    traffic = pmc_traffic(args.graph, k, passes) if not sharded and args.scale == 1.0 else None

    if rank == 0:
        flops = 2.0 * nnz * k
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
                "workload": f"{args.graph}-shaped R-MAT graph, n={n}, nnz={nnz} (incl. self-loops), "
                            f"feat={k}, fp32, no reorder; step = C = Â·H"
                            + ("" if world == 1 else " per row block + RCCL all-gather of the layer output"),
                "n": n, "nnz": nnz, "k": k,
                "parallelism": "single GPU" if world == 1 else f"1-D row partition x{world} (nnz-balanced), all-gather per layer",
                "chunks": f"{local_adj.num_chunks} x {local_adj.chunk_size} nnz",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": local_adj.main_kernel(kp),
                "slices": local_adj.num_slices,
                "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK, 4),
                "frac_of_measured_copy_peak_6.29TBps": round(achieved / 6.29e12, 4),
                "algorithmic_bytes_per_launch": int(balg),
                # compulsory traffic (each of A, B, C touched once) and the rate it implies
                "compulsory_bytes_per_spmm": int(local_nnz * 8 + (local_m + 1) * 4 + n * kp * 4 + local_m * kp * 4),
                "compulsory_GBps": round((local_nnz * 8 + (local_m + 1) * 4 + n * kp * 4 + local_m * kp * 4)
                                         / spmm_avg / 1e9, 1) if spmm_avg > 0 else None,
                "launches_per_spmm": passes, "columns_per_launch": kp // passes,
                "kernel_ms_avg": round(kavg * 1e3, 4),
                "spmm_ms_min": round(min(kernel_ms), 4) if kernel_ms else None,
                "spmms_timed": len(kernel_ms),
                "timing": "HIP events recorded by libgcnspmm on the launch stream around the main-kernel "
                          "passes of every timed SpMM (gcn_spmm_profile_begin/_end)",
                "traffic": traffic,
                # what the kernel really moves across the XCD <-> memory-side fabric (L2 misses, Infinity-Cache
                # hits included), as a rate and against the 6.29 TB/s this GPU reaches in a plain copy
                "traffic_GBps": None if traffic is None or kavg <= 0 else round(traffic / kavg / 1e9, 1),
                "traffic_frac_of_measured_copy_peak_6.29TBps": None if traffic is None or kavg <= 0
                else round(traffic / kavg / 6.29e12, 4),
                "traffic_source": None if traffic is None else
                "profiles/pmc_latest.json (separate rocprofv3 --pmc passes; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; "
                "L2-miss bytes incl. Infinity-Cache hits)",
            },
            "sharded_vs_single_gpu_rel_err": shard_check,
            "gflops_kernel_only": round(2.0 * local_nnz * kp / spmm_avg / 1e9, 1) if spmm_avg > 0 else None,
        }
        if not sharded and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(rowptr, col, val, n, k, seed=2)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
