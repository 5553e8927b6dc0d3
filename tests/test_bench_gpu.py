"""bench.py end to end on a shrunken graph: the default invocation (with the CPU baseline leg), the
row-sharded path on one rank and the simulated rank share must all print the one JSON line the driver
parses.  (--scale invalidates the metric; this only guards the plumbing.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "0.02", "--steps", "2",
                          "--warmup", "1", *flags], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_default_invocation_prints_the_contract_line_with_roofline_and_cpu_baseline():
    d = _bench()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "l2" and r["achieved"] > 0 and r["peak"] == 8000.0     # (a shrunken graph: the table sits in the caches)
    assert abs(r["algorithmic_over_hbm_peak"] - r["achieved"] / r["peak"]) < 1e-3      # the contract figure, labelled as such
    for key in ("frac", "frac_l2", "frac_hbm_compulsory"):                           # every fraction is a physical one
        assert 0 < r[key] <= 1, (key, r[key])
    assert r["traffic"] is None and r["frac_fabric"] is None and r["traffic_source"] == "not collected in this run"
    assert r["kernel"].startswith("gcn::spmm_")
    chk = d["check"]
    assert chk["passed"] and 0 <= chk["rel_err"] <= 1e-5 and chk["rows_per_rank"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["value"] > 0 and c["cores"] >= 1 and "torch.spmm" in c["sample"]
    assert 0 < d["config"]["plan_build_seconds"] < 60           # device CSR -> plan + first SpMM, outside the timed steps


@pytest.mark.parametrize("flags", [("--no-cpu-baseline", "--force-shard"), ("--no-cpu-baseline", "--sim-world", "4"),
                                   ("--no-cpu-baseline", "--sim-world", "4", "--no-plane-streams"),
                                   ("--no-cpu-baseline", "--gather-width", "1", "--blocks-per-cu", "8")])
def test_debug_paths_of_the_bench_run(flags):
    d = _bench(*flags)
    assert d["value"] > 0 and "cpu_baseline" not in d and d["roofline"]["kernel_ms_avg"] > 0
    assert d["check"]["passed"] and d["check"]["rel_err"] <= 1e-5


@pytest.mark.parametrize("flags,order", [(("--graph", "products", "--scale", "0.01"), "rcm"),
                                         (("--graph", "rmat24", "--rmat-scale", "14", "--order", "deg"), "deg"),
                                         (("--graph", "rmat24", "--rmat-scale", "14", "--order", "rcm"), "rcm"),
                                         (("--graph", "rmat24", "--rmat-scale", "13", "--order", "gorder"), "gorder"),
                                         (("--graph", "rmat24", "--rmat-scale", "14"), "none")])
def test_config3_and_config5_modes_renumber_then_multiply_and_check(flags, order):
    """BASELINE configs 3 and 5 through bench.py at a reduced size: the graph is renumbered by the library's own
    reorderers (device RCM / degree, host Gorder), the SpMM runs on the renumbered matrix and the timed output is
    checked against fp64 on sampled rows; the line says which config it is and that it is not the headline one"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags, "--steps", "2", "--warmup", "1"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["check"]["passed"] and d["check"]["rel_err"] <= 1e-5
    assert d["config"]["order"] == order and "NOT the headline config" in d["config"]["workload"]
    assert d["config"]["k"] == (256 if "products" in flags else 512)
    assert d["roofline"]["bound"] in ("l2", "hbm") and 0 < d["roofline"]["frac"]
    if order != "none":
        assert d["config"]["ordering_seconds"] > 0
    assert "cpu_baseline" in d and "leading" in d["cpu_baseline"]["sample"] or "full" in d["cpu_baseline"]["sample"]


@pytest.mark.parametrize("graph,scale,k", [("reddit-dcsbm", "0.05", 128), ("products-dcsbm", "0.02", 256)])
def test_structured_stand_ins_renumbered_by_rabbit_on_the_device_and_autotuned(graph, scale, k):
    """round 4's degree-corrected planted partitions through bench.py at a reduced size: generated, renumbered by the parallel
    Rabbit on the device, plan shape chosen by CsrAdjacency.autotune() (slices x column tile), EVERY output row checked against
    fp64 (n <= 3 M), the generator's recipe in the line"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--graph", graph, "--scale", scale, "--order", "rabbit",
                          "--autotune", "--mixing", "0.3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["check"]["passed"] and d["check"]["full_matrix"] and d["check"]["rows_per_rank"] == d["config"]["n"]
    assert d["config"]["k"] == k and d["config"]["order"] == "rabbit" and d["config"]["ordering_seconds"] > 0
    assert "planted-partition" in d["config"]["workload"] and "mixing 0.3" in d["config"]["workload"]
    assert "NOT the headline config" in d["config"]["workload"]
    tuned = d["config"]["autotune"]
    assert len(tuned) >= 3 and "slices=0,tile=0" in tuned and "slices=0,tile=64" in tuned and min(tuned.values()) > 0


def test_papers100m_mode_runs_a_rank_share_built_from_its_own_block():
    """BASELINE config 4 through the product path at a reduced scale: the rank is built by from_row_block from the
    block generator (the whole graph never exists), one GPU computes rank 0's share of the 8-way partition"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--graph", "papers100m", "--scale", "0.002",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["check"]["passed"] and "8-way partition" in d["config"]["workload"] and d["config"]["n"] == int(111059956 * 0.002)


def test_two_rank_rehearsal_on_one_gpu_exchanges_and_checks_on_every_rank():
    """the N = 2 bench path end to end on ONE GPU (both ranks on cuda:0, gloo instead of RCCL): rank-local graph
    blocks, all three exchange forms (push: the peers' buffers mapped through IPC handles, shards written by
    hipMemcpyAsync, one flag per layer — both ranks live on the same GPU here, the mechanism is the same), the
    all-reduced output check"""
    for exchange in ("all_gather", "direct", "push"):
        env = dict(os.environ, GCN_AMD_BENCH_REHEARSAL="1")
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"),
                              "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1", "--exchange", exchange],
                             cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert d["n_gpus"] == 2 and d["check"]["passed"] and d["config"]["ranks_seen"] == 2
        assert ("isend" in d["config"]["collective"]) == (exchange == "direct")
        assert ("IPC" in d["config"]["collective"]) == (exchange == "push")


@pytest.mark.parametrize("exchange", ["push", "direct"])
def test_four_rank_rehearsal_on_one_gpu(exchange):
    """four ranks on ONE GPU (the box allows six processes on its card): with more than one peer the staggered peer order,
    the per-peer flags of the push form and the slot arithmetic are no longer degenerate; pre-laid chain on"""
    env = dict(os.environ, GCN_AMD_BENCH_REHEARSAL="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                          "--master-addr", "127.0.0.1", "--master-port", "29751", os.path.join(ROOT, "bench.py"),
                          "--gpus", "4", "--scale", "0.12", "--steps", "2", "--warmup", "1", "--exchange", exchange],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 4 and d["check"]["passed"] and d["check"]["rel_err"] <= 1e-5 and d["config"]["ranks_seen"] == 4
    assert d["config"]["prelaid"] is True and d["check"]["full_matrix"]
    assert ("IPC" in d["config"]["collective"]) == (exchange == "push")


def test_config5_gorder_leg_loads_an_offline_rank_for_the_cpu_generated_graph(tmp_path):
    """BASELINE config 5's Gorder leg at full size uses a rank computed once, off-line (tools/gorder_rmat24.py: the host
    Gorder needs two hours at scale 24).  The same mechanism at scale 12: the tool writes rank + hashes, bench.py
    regenerates the CPU-generator graph, checks the hashes, loads the rank and says so; a tampered rank is refused."""
    import numpy as np
    rank_f = os.path.join(ROOT, "artifacts", "gorder_rmat12_rank.npy")
    meta_f = os.path.join(ROOT, "profiles", "r03_gorder_rmat12.json")
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gorder_rmat24.py"), "--scale", "12"], cwd=ROOT,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and os.path.exists(rank_f) and os.path.exists(meta_f), out.stderr[-1500:]
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--graph", "rmat24", "--rmat-scale", "12", "--graph-device", "cpu",
               "--order", "gorder", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
        out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-1500:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert d["check"]["passed"] and d["config"]["order"] == "gorder" and "CPU generator" in d["config"]["workload"]
        assert "off-line" in d["config"]["ordering_ran_on"] and d["config"]["ordering_seconds"] == json.load(open(meta_f))["gorder_host_seconds"]
        r = np.load(rank_f)
        r[[0, 1]] = r[[1, 0]]
        np.save(rank_f, r)                                           # not the rank the hashes were taken of
        out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert out.returncode != 0 and "hash mismatch" in out.stderr
    finally:
        for f in (rank_f, meta_f):
            if os.path.exists(f):
                os.remove(f)


@pytest.mark.parametrize("exchange", ["all_gather", "push"])
def test_two_rank_rehearsal_with_both_planes_on_one_stream(exchange):
    """what PipelinedAggregation chooses by itself when a plane's main kernel is long (a rank of 2 or 4 of the headline
    graph): both planes on ONE stream, the exchange of plane 0 (asynchronous collective, or pushes + flag wait on the
    stream) under the SpMM of plane 1 — forced here on a small graph"""
    env = dict(os.environ, GCN_AMD_BENCH_REHEARSAL="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29761", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--scale", "0.12", "--steps", "3", "--warmup", "1", "--exchange", exchange,
                          "--no-plane-streams"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["check"]["passed"] and d["check"]["rel_err"] <= 1e-5
    assert d["config"]["prelaid"] is True and d["roofline"]["concurrent_planes"] == 1


@pytest.mark.parametrize("exchange", ["all_gather", "direct", "push"])
def test_two_rank_rehearsal_with_the_prelaid_exchange_buffers(exchange):
    """the N = 2 bench path on ONE GPU at a size where the pre-laid chain switches on (slots of whole column slices, the
    exchange buffer is the next layer's scaled input): both ranks agree on the layout (one MIN all-reduce), exchange
    their slots over gloo, and the all-reduced output check passes"""
    env = dict(os.environ, GCN_AMD_BENCH_REHEARSAL="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29741", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--scale", "0.12", "--steps", "3", "--warmup", "1", "--exchange", exchange],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["check"]["passed"] and d["check"]["rel_err"] <= 1e-5
    assert d["config"]["prelaid"] is True and d["roofline"]["slices"] >= 2
    assert "scaling_budget" in d and d["roofline"]["concurrent_planes"] == 2
