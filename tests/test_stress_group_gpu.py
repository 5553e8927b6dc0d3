"""Randomised stress of the group kernels (spmm_group.hip) — the value-free pass with its LDS ring and the weighted
pass — over graph shapes that hit their corners: hub rows spanning many chunks, runs of one-entry rows (ring
overflow: more than four rows end inside one block), empty virtual rows, slices that end up empty, partial last
column tiles, row-padded and odd-width copies, every epilogue.  Each case against the fp64 oracle; seeds are fixed."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import gcn_amd
from util import oracle_spmm, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _graph(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(300, 6000))
    style = seed % 4
    if style == 0:                       # near-regular, moderate degree
        e = n * int(rng.integers(8, 40))
        u, v = rng.integers(0, n, e), rng.integers(0, n, e)
    elif style == 1:                     # a few hubs that see everyone + a sparse rest (rows of 1-3 entries)
        hubs = rng.choice(n, int(rng.integers(1, 6)), replace=False)
        u = np.concatenate([np.repeat(hubs, n // 2), rng.integers(0, n, n)])
        v = np.concatenate([rng.integers(0, n, len(hubs) * (n // 2)), rng.integers(0, n, n)])
    elif style == 2:                     # power-law-ish
        w = 1.0 / np.arange(1, n + 1) ** 0.9
        w /= w.sum()
        e = n * int(rng.integers(4, 25))
        u, v = rng.choice(n, e, p=w), rng.integers(0, n, e)
    else:                                # two blocks: one dense, one almost empty (slices that end up empty)
        half = n // 2
        e = half * int(rng.integers(20, 60))
        u, v = rng.integers(0, half, e), rng.integers(0, half, e)
    A = sp.coo_matrix((np.ones(len(u)), (u, v)), shape=(n, n))
    A = (A + A.T).tocsr(); A.setdiag(0); A.eliminate_zeros(); A.data[:] = 1.0
    A = (A + sp.eye(n)).tocsr()
    d = np.asarray(A.sum(1)).ravel() ** -0.5
    A = (sp.diags(d) @ A @ sp.diags(d)).tocsr(); A.sort_indices()
    return n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32), rng


@pytest.mark.parametrize("seed", range(48))
def test_group_kernels_random(seed):
    n, rowptr, col, val, rng = _graph(seed)
    d = torch.device("cuda:0")
    S = int(rng.choice([2, 3, 5, 8, 13, 16]))
    k = int(rng.choice([36, 64, 100, 128, 41, 192]))
    weighted = seed % 3 == 2
    if weighted:
        val = (val * (1.0 + 0.5 * rng.random(len(val)))).astype(np.float32)      # no longer u[r]*u[c]
    adj = gcn_amd.CsrAdjacency(torch.from_numpy(rowptr).to(d), torch.from_numpy(col).to(d), torch.from_numpy(val).to(d),
                               (n, n), slices=S)
    assert adj.num_slices == S
    name = adj.main_kernel(k)
    assert name.startswith("gcn::spmm_group") and (not weighted or "weighted" in name), name
    assert adj.has_value_factors == (not weighted)
    B = rng.standard_normal((n, k)).astype(np.float32)
    ref = oracle_spmm(rowptr, col, val, B)
    Bd = torch.from_numpy(B).to(d)
    C = adj.matmul_raw(Bd)
    assert rel_err(C.cpu().numpy(), ref) <= TOL, (seed, n, S, k, weighted)
    assert torch.equal(C, adj.matmul_raw(Bd))                                    # bitwise reproducible
    bias = rng.standard_normal(k).astype(np.float32)
    Ce = adj.matmul_raw(Bd, bias=torch.from_numpy(bias).to(d), relu=True).cpu().numpy()
    assert rel_err(Ce, np.maximum(ref + bias, 0)) <= TOL, (seed, n, S, k, weighted)


@pytest.mark.parametrize("seed", range(32))
def test_group8_kernel_random(seed):
    """k <= 32 on a value-free sliced plan: eight 8-lane row engines per wave (spmm_group8_kernel), LDS ring of eight
    rows per group; widths that are not a multiple of 4 ride the k' detour into the same kernel; k = 8 stays off it."""
    n, rowptr, col, val, rng = _graph(seed + 100)
    d = torch.device("cuda:0")
    S = int(rng.choice([2, 4, 8, 15]))
    k = int(rng.choice([12, 16, 20, 24, 28, 32, 17, 30]))
    adj = gcn_amd.CsrAdjacency(torch.from_numpy(rowptr).to(d), torch.from_numpy(col).to(d), torch.from_numpy(val).to(d),
                               (n, n), slices=S)
    assert adj.num_slices == S and adj.has_value_factors
    name = adj.main_kernel(k)
    knobs_off = os.environ.get("GCN_AMD_GROUP8", "1") == "0"
    if len(col) // n >= 48 and not knobs_off:                    # (below: the weighted pass, see valless_pays)
        assert name.startswith("gcn::spmm_group8_kernel<"), (name, k)
    assert not adj.main_kernel(8).startswith("gcn::spmm_group")
    B = rng.standard_normal((n, k)).astype(np.float32)
    ref = oracle_spmm(rowptr, col, val, B)
    Bd = torch.from_numpy(B).to(d)
    C = adj.matmul_raw(Bd)
    assert rel_err(C.cpu().numpy(), ref) <= TOL, (seed, n, S, k, name)
    assert torch.equal(C, adj.matmul_raw(Bd))
    bias = rng.standard_normal(k).astype(np.float32)
    Ce = adj.matmul_raw(Bd, bias=torch.from_numpy(bias).to(d), relu=True).cpu().numpy()
    assert rel_err(Ce, np.maximum(ref + bias, 0)) <= TOL, (seed, n, S, k, name)


@pytest.mark.parametrize("seed", range(24))
def test_group12_kernel_random(seed):
    """33 <= k <= 48 on a value-free sliced plan: five 12-lane row engines per wave (spmm_group12_kernel) on the SAME stream
    as the 16-lane kernel — entries through ds_bpermute, two row-end ballots, a ring of five rows per group; odd widths
    ride the k' detour into it; ragged graphs (empty rows, rows longer than a chunk), several slice counts"""
    n, rowptr, col, val, rng = _graph(seed + 200)
    d = torch.device("cuda:0")
    S = int(rng.choice([2, 3, 4, 8, 15]))
    k = int(rng.choice([33, 36, 37, 40, 41, 44, 45, 47, 48]))
    adj = gcn_amd.CsrAdjacency(torch.from_numpy(rowptr).to(d), torch.from_numpy(col).to(d), torch.from_numpy(val).to(d),
                               (n, n), slices=S)
    assert adj.num_slices == S and adj.has_value_factors
    name = adj.main_kernel(k)
    knobs_off = os.environ.get("GCN_AMD_GROUP12", "1") == "0" or os.environ.get("GCN_AMD_GROUP_BIG", "0") == "1"   # (64-bit slice bases: the 64-column pass)
    if len(col) // n >= 48 and not knobs_off:                    # (below: the weighted pass, see valless_pays)
        assert name == "gcn::spmm_group12_kernel", (name, k)
    assert name.startswith("gcn::spmm_group"), (name, k)
    B = rng.standard_normal((n, k)).astype(np.float32)
    ref = oracle_spmm(rowptr, col, val, B)
    Bd = torch.from_numpy(B).to(d)
    C = adj.matmul_raw(Bd)
    assert rel_err(C.cpu().numpy(), ref) <= TOL, (seed, n, S, k, name)
    assert torch.equal(C, adj.matmul_raw(Bd))
    bias = rng.standard_normal(k).astype(np.float32)
    Ce = adj.matmul_raw(Bd, bias=torch.from_numpy(bias).to(d), relu=True).cpu().numpy()
    assert rel_err(Ce, np.maximum(ref + bias, 0)) <= TOL, (seed, n, S, k, name)


def _big_child():
    """runs in a child process with GCN_AMD_GROUP_BIG=1 (the knob is read once per process): every group kernel in
    its 64-bit slice-base variant — what tables of 4 GiB and more get — against the fp64 oracle"""
    d = torch.device("cuda:0")
    seen = set()
    for seed in range(12):
        n, rowptr, col, val, rng = _graph(seed)
        S = int(rng.choice([2, 3, 5, 8, 13, 16]))
        weighted = seed % 3 == 2
        if weighted:
            val = (val * (1.0 + 0.5 * rng.random(len(val)))).astype(np.float32)
        adj = gcn_amd.CsrAdjacency(torch.from_numpy(rowptr).to(d), torch.from_numpy(col).to(d), torch.from_numpy(val).to(d),
                                   (n, n), slices=S)
        for k in (int(rng.choice([36, 64, 100, 128, 41, 192])), int(rng.choice([12, 16, 20, 24, 32]))):
            name = adj.main_kernel(k)
            if name.startswith("gcn::spmm_group"):
                assert name.endswith("true>"), name                          # the BIG instantiation
                seen.add(name.split("<")[0])
            B = rng.standard_normal((n, k)).astype(np.float32)
            C = adj.matmul_raw(torch.from_numpy(B).to(d))
            err = rel_err(C.cpu().numpy(), oracle_spmm(rowptr, col, val, B))
            assert err <= TOL, (seed, n, S, k, name, err)
    want = {"gcn::spmm_group_ring_kernel", "gcn::spmm_group_weighted_kernel", "gcn::spmm_group8_kernel"}
    if os.environ.get("GCN_AMD_GROUP8", "1") == "0":
        want.discard("gcn::spmm_group8_kernel")                 # (the switch that keeps narrow widths off the eight-engine kernel)
    assert want <= seen, seen
    print("big ok", sorted(seen))


def test_group_kernels_with_64_bit_slice_bases():
    """the BIG variants of the group kernels (slice base added in 64 bits; tables past 4 GiB or 2^24 rows use them,
    gcn_spmm_group_addressing) forced on small graphs through the development knob, in ONE child process"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GCN_AMD_GROUP_BIG="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--big-child"], env=env, capture_output=True, text=True,
                         timeout=600, cwd=os.path.dirname(os.path.abspath(__file__)))
    assert out.returncode == 0 and "big ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def _fixup_child():
    """runs in a child process with GCN_AMD_GROUP_FUSED_FIXUP=0: the cut rows' pieces added by group_fixup_kernel in a
    pass of its own (what the drop-in flexspmm runs) instead of inside the slice reduction — against the fp64 oracle"""
    d = torch.device("cuda:0")
    for seed in range(12):
        n, rowptr, col, val, rng = _graph(seed + 40)
        S = int(rng.choice([2, 3, 5, 8, 13, 16]))
        if seed % 3 == 2:
            val = (val * (1.0 + 0.5 * rng.random(len(val)))).astype(np.float32)
        adj = gcn_amd.CsrAdjacency(torch.from_numpy(rowptr).to(d), torch.from_numpy(col).to(d), torch.from_numpy(val).to(d),
                                   (n, n), slices=S)
        for k in (int(rng.choice([36, 64, 100, 128, 41, 192])), int(rng.choice([12, 16, 20, 24, 32]))):
            B = rng.standard_normal((n, k)).astype(np.float32)
            C = adj.matmul_raw(torch.from_numpy(B).to(d))
            err = rel_err(C.cpu().numpy(), oracle_spmm(rowptr, col, val, B))
            assert err <= TOL, (seed, n, S, k, adj.main_kernel(k), err)
    print("fixup ok")


def test_group_kernels_with_the_fix_up_pass_of_its_own():
    """GCN_AMD_GROUP_FUSED_FIXUP=0: the same plans with group_fixup_kernel in front of the slice reduction (the default adds
    the pieces inside the reduction), in ONE child process"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GCN_AMD_GROUP_FUSED_FIXUP="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--fixup-child"], env=env, capture_output=True, text=True,
                         timeout=600, cwd=os.path.dirname(os.path.abspath(__file__)))
    assert out.returncode == 0 and "fixup ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def _segments_child():
    """runs in a child process with GCN_AMD_GROUP_SEGMENTS=<runs per XCD>: the (column tile, block) order inside the merged
    launch is placement only — prints a digest of the results of several plans and widths (incl. widths of 3-8 tiles, runs
    that do not divide an XCD's blocks, a weighted plan); the parent compares the digests of different orders"""
    import hashlib
    d = torch.device("cuda:0")
    h = hashlib.sha256()
    for seed in (1, 4, 7, 10):
        n, rowptr, col, val, rng = _graph(seed + 80)
        S = int(rng.choice([2, 3, 5, 8]))
        if seed == 7:
            val = (val * (1.0 + 0.5 * rng.random(len(val)))).astype(np.float32)
        adj = gcn_amd.CsrAdjacency(torch.from_numpy(rowptr).to(d), torch.from_numpy(col).to(d), torch.from_numpy(val).to(d),
                                   (n, n), slices=S)
        for k in (128, 192, 320, 100, 512):
            assert adj.main_kernel(k).startswith("gcn::spmm_group"), adj.main_kernel(k)
            B = rng.standard_normal((n, k)).astype(np.float32)
            C = adj.matmul_raw(torch.from_numpy(B).to(d)).cpu().numpy()
            assert rel_err(C, oracle_spmm(rowptr, col, val, B)) <= TOL, (seed, k)
            h.update(C.tobytes())
    from util import sym_norm_graph
    n = 17000                                              # 37 blocks per XCD: runs of 37, 19, 13 and 6 blocks, the last one shorter
    rowptr, col, val = sym_norm_graph(n, 1200000, seed=12)
    adj = gcn_amd.CsrAdjacency(torch.from_numpy(rowptr).to(d), torch.from_numpy(col).to(d), torch.from_numpy(val).to(d), (n, n))
    for k in (128, 256):
        B = np.random.default_rng(k).standard_normal((n, k)).astype(np.float32)
        C = adj.matmul_raw(torch.from_numpy(B).to(d)).cpu().numpy()
        assert adj.main_kernel(k).startswith("gcn::spmm_group_ring") and rel_err(C, oracle_spmm(rowptr, col, val, B)) <= TOL
        h.update(C.tobytes())
    print("segments digest", h.hexdigest())


def test_tile_order_inside_the_merged_launch_is_placement_only():
    """GCN_AMD_GROUP_SEGMENTS = 1 (tile-major), 2, 3, 7 runs per XCD: bit-identical results (and each within 1e-5 of the
    oracle) — the order in which a launch walks its (tile, block) pairs (spmm_group.hip, launch_group_t) never shows"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for runs in ("1", "2", "3", "7"):
        env = dict(os.environ, GCN_AMD_GROUP_SEGMENTS=runs, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--segments-child"], env=env, capture_output=True,
                             text=True, timeout=600, cwd=os.path.dirname(os.path.abspath(__file__)))
        assert out.returncode == 0 and "segments digest" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
        digests.append(out.stdout.split("segments digest")[1].split()[0])
    assert len(set(digests)) == 1, digests


if __name__ == "__main__":
    import sys
    if "--big-child" in sys.argv:
        _big_child()
    if "--fixup-child" in sys.argv:
        _fixup_child()
    if "--segments-child" in sys.argv:
        _segments_child()
