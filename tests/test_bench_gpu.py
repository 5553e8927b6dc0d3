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
    assert r["bound"] == "hbm" and r["achieved"] > 0 and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["kernel"].startswith("gcn::spmm_")
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["value"] > 0 and c["cores"] >= 1 and "torch.spmm" in c["sample"]


@pytest.mark.parametrize("flags", [("--no-cpu-baseline", "--force-shard"), ("--no-cpu-baseline", "--sim-world", "4"),
                                   ("--no-cpu-baseline", "--sim-world", "4", "--no-plane-streams"),
                                   ("--no-cpu-baseline", "--gather-width", "1", "--blocks-per-cu", "8")])
def test_debug_paths_of_the_bench_run(flags):
    d = _bench(*flags)
    assert d["value"] > 0 and "cpu_baseline" not in d and d["roofline"]["kernel_ms_avg"] > 0
