"""GPU parity tests: the HIP SpMM path, called through the C ABI (libgcnspmm.so), against
the CPU oracle (oracle/spmm_oracle.c, fp64 accumulation) on the same seeded inputs and
against the golden outputs of the Python reference (tests/golden/gcn1_*.npz).

Tolerance (BASELINE.json north_star): 1e-5 relative fp32 —
``max|C - C*| / max|C*| <= 1e-5``.  Integer outputs are bit-exact.
"""
import ctypes
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import gcn_amd
from gcn_amd import _lib, dropin, graphgen
from util import GOLDEN, oracle_spmm, random_csr, rel_err, sym_norm_graph

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _adj(rowptr, col, val, m, n, **kw):
    d = _dev()
    return gcn_amd.CsrAdjacency(torch.from_numpy(rowptr).to(d), torch.from_numpy(col).to(d),
                                torch.from_numpy(val).to(d), (m, n), **kw)


def _run(rowptr, col, val, m, n, k, seed=0, gather_width=0, **kw):
    rng = np.random.default_rng(seed + 99)
    B = rng.standard_normal((n, k)).astype(np.float32)
    adj = _adj(rowptr, col, val, m, n, **kw)
    adj.set_gather_width(gather_width)
    C = adj.matmul_raw(torch.from_numpy(B).to(_dev()))
    torch.cuda.synchronize()
    return C.cpu().numpy(), oracle_spmm(rowptr, col, val, B), adj


# every feature width the reference's launcher distinguishes (flexspmm.cu:510-541: 8, 16, 32,
# <32, >32) plus the BASELINE widths 128/256/512 and awkward ones (odd, non-multiple of 64)
# (gather width 0 = automatic: the one-per-gather / narrow kernels on this 20-per-row matrix; 4 = the
#  four-per-gather kernel forced, which the automatic rule only picks from ~48 non-zeros per row up)
@pytest.mark.parametrize("width", [0, 4])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 24, 32, 33, 64, 100, 128, 130, 192, 256, 300, 512])
def test_parity_feature_widths(k, width):
    m = n = 3000
    rowptr, col, val = random_csr(m, n, 60000, seed=k)
    C, Cref, _ = _run(rowptr, col, val, m, n, k, seed=k, gather_width=width)
    assert rel_err(C, Cref) <= TOL


@pytest.mark.parametrize("k,tile", [(128, 64), (128, 128), (256, 64), (256, 128), (256, 256), (512, 64),
                                    (512, 256), (300, 64), (300, 128), (300, 256), (130, 128), (33, 64)])
def test_parity_column_tile_passes(k, tile):
    """a k-wide SpMM as ceil(k/tile) sequential column-slice passes (gcn_spmm_plan_set_tile_cols)"""
    m = n = 2000
    rowptr, col, val = random_csr(m, n, 40000, seed=k + tile, empty_rows=0.05, long_rows=[(3, 1800)])
    rng = np.random.default_rng(k)
    B = rng.standard_normal((n, k)).astype(np.float32)
    adj = _adj(rowptr, col, val, m, n, chunk_nnz=128)
    adj.set_tile_cols(tile)
    C = adj.matmul_raw(torch.from_numpy(B).to(_dev())).cpu().numpy()
    assert rel_err(C, oracle_spmm(rowptr, col, val, B)) <= TOL


@pytest.mark.parametrize("chunk", [64, 128, 512, 0])
def test_parity_ragged_rows_and_chunk_sizes(chunk):
    """empty rows (leading, trailing, in runs), rows longer than several chunks, rows
    that end exactly on a chunk boundary"""
    m, n, k = 2500, 4000, 128
    rowptr, col, val = random_csr(m, n, 40000, seed=3, empty_rows=0.3,
                                  long_rows=[(0, 0), (1, 0), (7, 3000), (8, 64), (9, 128), (1200, 2111),
                                             (m - 1, 0), (m - 2, 0), (m - 3, 777)])
    C, Cref, adj = _run(rowptr, col, val, m, n, k, chunk_nnz=chunk)
    assert rel_err(C, Cref) <= TOL
    # empty rows are written (as zeros) even though C is torch.empty
    lens = np.diff(rowptr)
    assert np.all(C[lens == 0] == 0.0)
    if chunk:
        assert adj.chunk_size == chunk


def test_edge_cases_empty_and_tiny():
    d = _dev()
    # nnz == 0: all-zero output
    rowptr = np.zeros(11, np.int32)
    adj = _adj(rowptr, np.zeros(0, np.int32), np.zeros(0, np.float32), 10, 10)
    C = adj.matmul_raw(torch.ones((10, 16), device=d))
    assert C.shape == (10, 16) and float(C.abs().max()) == 0.0
    # single non-zero
    rowptr = np.array([0, 0, 1, 1], np.int32)
    adj = _adj(rowptr, np.array([2], np.int32), np.array([2.5], np.float32), 3, 3)
    B = torch.arange(9, dtype=torch.float32, device=d).reshape(3, 3)
    C = adj.matmul_raw(B).cpu().numpy()
    assert np.array_equal(C, np.array([[0, 0, 0], [15, 17.5, 20], [0, 0, 0]], np.float32))
    # one row holding everything (max row length), non-square
    n = 5000
    rowptr = np.array([0, n], np.int32)
    col = np.arange(n, dtype=np.int32); val = np.full(n, 1.0 / n, np.float32)
    C, Cref, _ = _run(rowptr, col, val, 1, n, 128)
    assert rel_err(C, Cref) <= TOL


@pytest.mark.parametrize("width", [0, 4])
@pytest.mark.parametrize("k", [4, 8, 16, 20, 33, 47, 64, 100, 128, 256])       # 20, 47, 100: padded feature rows
def test_nan_inf_do_not_leak_between_rows(k, width):
    """a row of B holding Inf/NaN only poisons the output rows that reference it — for every kernel
    family (narrow k <= 16, one-non-zero-per-gather, four-per-gather).  Row 0 is the row the kernels'
    padding lanes gather (ragged last block), row 317 an ordinary one."""
    m = n = 600
    rowptr, col, val = random_csr(m, n, 6003, seed=11)          # nnz % 64 != 0: ragged last block
    rng = np.random.default_rng(5)
    B = rng.standard_normal((n, k)).astype(np.float32)
    B[0, :] = np.inf
    B[317, :] = np.nan
    adj = _adj(rowptr, col, val, m, n)
    adj.set_gather_width(width)
    C = adj.matmul_raw(torch.from_numpy(B).to(_dev())).cpu().numpy()
    touched = np.array([bool(np.isin([0, 317], col[rowptr[r]:rowptr[r + 1]]).any()) for r in range(m)])
    assert touched.any() and (~touched).any()
    assert np.all(np.isfinite(C[~touched]))
    assert not np.any(np.isfinite(C[touched]).all(axis=1))      # and the rows that do reference them show it
    assert rel_err(C[~touched], oracle_spmm(rowptr, col, val, np.where(np.isfinite(B), B, 0))[~touched]) <= TOL


@pytest.mark.parametrize("width", [1, 4])
@pytest.mark.parametrize("k", [12, 16, 20, 32, 64, 128, 200])
def test_parity_gather_widths(width, k):
    """the 64-column tile with one and with four non-zeros per gather instruction
    (gcn_spmm_plan_set_gather_width), on ragged input: empty rows, hub rows, rows of 1..5 non-zeros,
    rows ending on block and chunk boundaries; plain and with the fused epilogue"""
    m, n = 3000, 5000
    rowptr, col, val = random_csr(m, n, 30000, seed=k + width, empty_rows=0.2,
                                  long_rows=[(0, 0), (5, 4097), (6, 64), (7, 128), (8, 1), (9, 2), (10, 3),
                                             (11, 63), (12, 65), (2000, 2500), (m - 1, 0), (m - 2, 5)])
    rng = np.random.default_rng(k)
    B = rng.standard_normal((n, k)).astype(np.float32)
    bias = rng.standard_normal(k).astype(np.float32)
    for chunk in (0, 64, 256):
        adj = _adj(rowptr, col, val, m, n, chunk_nnz=chunk)
        adj.set_gather_width(width)
        assert ("quad" in adj.main_kernel(k)) == (width == 4)
        Bd = torch.from_numpy(B).to(_dev())
        C = adj.matmul_raw(Bd).cpu().numpy()
        Cref = oracle_spmm(rowptr, col, val, B)
        assert rel_err(C, Cref) <= TOL
        assert np.all(C[np.diff(rowptr) == 0] == 0.0)
        Ce = adj.matmul_raw(Bd, bias=torch.from_numpy(bias).to(_dev()), relu=True).cpu().numpy()
        assert rel_err(Ce, np.maximum(Cref + bias, 0)) <= TOL


def test_main_kernel_families_are_selected_as_documented():
    """which kernel runs for which feature width (gcn_spmm_plan_main_kernel), so that the parity
    tests above are known to cover every family"""
    m = n = 1000
    rowptr, col, val = random_csr(m, n, 80000, seed=1)                       # 80 non-zeros per row: long rows
    adj = _adj(rowptr, col, val, m, n)
    assert adj.main_kernel(4).startswith("gcn::spmm_narrow_kernel<4,")
    assert adj.main_kernel(8).startswith("gcn::spmm_narrow_kernel<8,")
    assert adj.main_kernel(15) == "gcn::spmm_narrow16_dpp_kernel<false>"
    assert adj.main_kernel(16) == "gcn::spmm_quad_kernel<4, false, false, false>"          # 16 non-zeros per gather
    assert adj.main_kernel(17) == "gcn::spmm_quad_kernel<16, false, false, false>"         # odd widths run at k rounded up to 4
    assert adj.main_kernel(32) == "gcn::spmm_quad_kernel<16, false, false, false>"         # 4 per gather, half the lanes idle
    assert adj.main_kernel(128) == "gcn::spmm_quad_kernel<16, false, false, false>"        # 4 per gather
    assert adj.main_kernel(128, epilogue=True) == "gcn::spmm_quad_kernel<16, true, false, false>"
    adj.set_gather_width(1)
    assert adj.main_kernel(16) == "gcn::spmm_narrow16_dpp_kernel<false>"
    adj.set_gather_width(0)
    assert adj.main_kernel(130) == "gcn::spmm_quad_kernel<16, false, false, false>"        # k' = 132
    adj.set_gather_width(1)
    assert adj.main_kernel(130).startswith("gcn::spmm_chunk_kernel<1,")      # one non-zero per gather, caller's layout
    assert adj.main_kernel(128).startswith("gcn::spmm_chunk_kernel<1,")
    adj.set_gather_width(0)
    # short rows (20 per row): the per-row reduction of the four-per-gather layout does not pay
    rp2, ci2, va2 = random_csr(m, n, 20000, seed=2)
    short = _adj(rp2, ci2, va2, m, n)
    assert short.main_kernel(128).startswith("gcn::spmm_chunk_kernel<1,")
    assert short.main_kernel(16) == "gcn::spmm_narrow16_dpp_kernel<false>"
    assert short.main_kernel(130).startswith("gcn::spmm_chunk_kernel<1,")    # and no detour over k' = 132
    short.set_gather_width(4)
    assert short.main_kernel(128) == "gcn::spmm_quad_kernel<16, false, false, false>"
    adj.set_tile_cols(256)
    assert adj.main_kernel(256).startswith("gcn::spmm_chunk_kernel<4,")
    adj.set_tile_cols(128)
    assert adj.main_kernel(256).startswith("gcn::spmm_chunk_kernel<2,")


def test_deterministic_and_geometry_independent():
    """bitwise identical across runs and across chunk sizes is NOT promised (different
    partial splits), but the same plan must reproduce bit for bit (no atomics)"""
    m = n = 4000
    rowptr, col, val = sym_norm_graph(n, 120000, seed=2)
    B = torch.from_numpy(np.random.default_rng(0).standard_normal((n, 128)).astype(np.float32)).to(_dev())
    adj = _adj(rowptr, col, val, m, n)
    C1 = adj.matmul_raw(B).clone()
    C2 = adj.matmul_raw(B).clone()
    assert torch.equal(C1, C2)
    adj2 = _adj(rowptr, col, val, m, n)          # fresh plan
    assert torch.equal(C1, adj2.matmul_raw(B))


def test_bias_relu_epilogue():
    m = n = 2000
    rowptr, col, val = random_csr(m, n, 50000, seed=21, empty_rows=0.1, long_rows=[(5, 1500)])
    rng = np.random.default_rng(1)
    B = rng.standard_normal((n, 128)).astype(np.float32)
    bias = rng.standard_normal(128).astype(np.float32)
    adj = _adj(rowptr, col, val, m, n, chunk_nnz=128)
    d = _dev()
    C = adj.matmul_raw(torch.from_numpy(B).to(d), bias=torch.from_numpy(bias).to(d), relu=True).cpu().numpy()
    Cref = np.maximum(oracle_spmm(rowptr, col, val, B) + bias, 0)
    assert rel_err(C, Cref) <= TOL
    C = adj.matmul_raw(torch.from_numpy(B).to(d), bias=torch.from_numpy(bias).to(d)).cpu().numpy()
    assert rel_err(C, oracle_spmm(rowptr, col, val, B) + bias) <= TOL


# --- golden vectors from the Python reference (gcn1.py:53 = torch.spmm(adj, support)) ---------
@pytest.mark.parametrize("name", ["gcn1_tiny", "gcn1_cora_shaped"])
def test_golden_gcn1_aggregation(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = int(g["n"])
    import scipy.sparse as sp
    A = sp.coo_matrix((g["adj_val"], (g["adj_row"], g["adj_col"])), shape=(n, n)).tocsr()
    A.sort_indices()
    adj = gcn_amd.CsrAdjacency.from_scipy(A, device=_dev(), symmetric=True)
    for sup, agg in (("support1", "agg1"), ("support2", "agg2")):
        C = adj.matmul_raw(torch.from_numpy(g[sup]).to(_dev())).cpu().numpy()
        assert rel_err(C, g[agg]) <= TOL, (name, agg)


def test_golden_gcn1_forward_through_routed_torch_spmm():
    """2-layer GCN forward exactly as gcn1.py:40-58,102-126 writes it (torch.spmm on a sparse
    COO adj) with gcn_amd.install() active; compared with the reference's recorded output."""
    g = np.load(os.path.join(GOLDEN, "gcn1_cora_shaped.npz"))
    n, d = int(g["n"]), _dev()
    adj = torch.sparse_coo_tensor(torch.from_numpy(np.vstack([g["adj_row"], g["adj_col"]]).astype(np.int64)),
                                  torch.from_numpy(g["adj_val"]), (n, n)).to(d)
    X = torch.sparse_coo_tensor(torch.from_numpy(np.vstack([g["x_row"], g["x_col"]]).astype(np.int64)),
                                torch.from_numpy(g["x_val"]), (n, int(g["nfeat"]))).to(d).to_dense()
    w1, b1, w2, b2 = (torch.from_numpy(g[k]).to(d) for k in ("w1", "b1", "w2", "b2"))
    gcn_amd.install()
    try:
        h = torch.spmm(adj, torch.mm(X, w1)) + b1
        h = torch.relu(h)
        out = torch.log_softmax(torch.sparse.mm(adj, torch.mm(h, w2)) + b2, dim=1)
        assert id(adj) in gcn_amd.spmm.__globals__["_csr_cache"], "torch.spmm was not routed to the HIP path"
    finally:
        gcn_amd.uninstall()
    assert rel_err(out.cpu().numpy(), g["out"]) <= TOL


def test_autograd_backward_matches_transpose():
    m, n, k = 700, 900, 64
    rowptr, col, val = random_csr(m, n, 9000, seed=8)
    adj = _adj(rowptr, col, val, m, n, symmetric=False)
    rng = np.random.default_rng(2)
    B = torch.from_numpy(rng.standard_normal((n, k)).astype(np.float32)).to(_dev()).requires_grad_(True)
    G = rng.standard_normal((m, k)).astype(np.float32)
    C = gcn_amd.spmm(adj, B)
    C.backward(torch.from_numpy(G).to(_dev()))
    import scipy.sparse as sp
    At = sp.csr_matrix((val, col, rowptr), shape=(m, n)).T.tocsr(); At.sort_indices()
    gref = oracle_spmm(At.indptr.astype(np.int32), At.indices.astype(np.int32), At.data.astype(np.float32), G)
    assert rel_err(B.grad.cpu().numpy(), gref) <= TOL


# --- the gcn6 drop-in symbols (interface B1) ------------------------------------------------
def test_dropin_csr2tile_flexspmm_permutate_cuspmm():
    n = 3000
    rowptr, col, val = sym_norm_graph(n, 40000, seed=4)
    nnz = len(col)
    d = _dev()
    t_rp, t_ci, t_va = torch.from_numpy(rowptr.copy()), torch.from_numpy(col.copy()), torch.from_numpy(val.copy())
    vo_mp = torch.arange(n, dtype=torch.int32)
    seg_rowPtr, segNzCV, segVoMap, tail, nxt, n_segs = dropin.csr2tile(t_rp, t_ci, t_va, n, n, nnz, vo_mp)
    assert seg_rowPtr.numel() == 9 * int(n_segs[0]) and segVoMap.numel() == 8 * int(n_segs[0])
    assert tail.numel() == 256 and nxt.numel() == 256               # defect D2: never 257 entries
    dev = [t.to(d) for t in (seg_rowPtr, segNzCV, segVoMap, tail, nxt)]
    rng = np.random.default_rng(3)
    for k in (16, 41, 47, 100, 128):
        X = rng.standard_normal((n, k)).astype(np.float32)
        Xd = torch.from_numpy(X).to(d).requires_grad_(True)
        out = dropin.flexspmm.apply(dev[0], dev[1], dev[2], n, n, int(n_segs[0]), dev[3], dev[4], Xd)
        Cref = oracle_spmm(rowptr, col, val, X)
        assert rel_err(out.detach().cpu().numpy(), Cref) <= TOL
        G = rng.standard_normal((n, k)).astype(np.float32)
        out.backward(torch.from_numpy(G).to(d))
        assert rel_err(Xd.grad.cpu().numpy(), oracle_spmm(rowptr, col, val, G)) <= TOL   # Â symmetric
    # cuspmm symbol (cuspmm.cu:23-24)
    X = rng.standard_normal((n, 64)).astype(np.float32)
    C = torch.empty((n, 64), device=d)
    dropin.cuspmm(t_rp.to(d), t_ci.to(d), t_va.to(d), torch.from_numpy(X).to(d), C, n, n, nnz, 64)
    torch.cuda.synchronize()
    assert rel_err(C.cpu().numpy(), oracle_spmm(rowptr, col, val, X)) <= TOL
    # permutate symbol: B[r,:] <- B[vomp[r],:] in place, labels untouched (permutate.cu:17,35)
    perm = rng.permutation(n).astype(np.int32)
    Xd = torch.from_numpy(X).to(d).clone()
    labels = torch.arange(n, dtype=torch.int32, device=d)
    dropin.permutate(Xd, torch.from_numpy(perm).to(d), labels, n, n, 64)
    assert np.array_equal(Xd.cpu().numpy(), X[perm])                 # bit-exact copy
    assert torch.equal(labels, torch.arange(n, dtype=torch.int32, device=d))


def test_gather_rows_bit_exact():
    rng = np.random.default_rng(0)
    for k in (7, 128, 130):
        X = rng.standard_normal((1000, k)).astype(np.float32)
        idx = rng.integers(0, 1000, 2500).astype(np.int32)
        out = gcn_amd.gather_rows(torch.from_numpy(X).to(_dev()), torch.from_numpy(idx).to(_dev()))
        assert np.array_equal(out.cpu().numpy(), X[idx])


def test_reorder_then_spmm_is_a_permutation_of_the_result():
    """P·Â·Pᵀ · (P·B) = P·(Â·B): renumbering + feature permutation (gcn6.py steps 1 and 4)
    reproduce the un-reordered result up to row order."""
    n, k = 2000, 128
    rowptr, col, val = sym_norm_graph(n, 30000, seed=6)
    X = np.random.default_rng(1).standard_normal((n, k)).astype(np.float32)
    base = oracle_spmm(rowptr, col, val, X)
    for fn in (gcn_amd.reorder.gorder, gcn_amd.reorder.dfs, gcn_amd.reorder.rabbit):
        rp2, ci2, va2, vomp = fn(rowptr, col, val)
        adj = _adj(rp2, ci2, va2, n, n, symmetric=True)
        Xp = gcn_amd.gather_rows(torch.from_numpy(X).to(_dev()), torch.from_numpy(vomp).to(_dev()))
        C = adj.matmul_raw(Xp).cpu().numpy()
        assert rel_err(C, base[vomp]) <= TOL


# --- BASELINE-size checks through size-independent properties ---------------------------------
def test_full_size_reddit_shape_properties():
    """Reddit-shaped graph at full size (n = 232 965, nnz ≈ 114.85 M, k = 128):
    (i) sampled rows against the fp64 C oracle and ALL rows against fp64 on the device, (ii) Â·1 = row sums, (iii) linearity."""
    d = _dev()
    rowptr, col, val, n = graphgen.make_graph("reddit", device=d, seed=1)
    nnz = int(col.numel())
    assert n == 232965 and abs(nnz - 114.85e6) < 0.05e6
    adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True)
    k = 128
    B = graphgen.random_features(n, k, seed=2, device=d)
    C = adj.matmul_raw(B)
    # (i) 4096 sampled rows vs fp64
    rows = np.random.default_rng(0).choice(n, 4096, replace=False); rows.sort()
    rp = rowptr.cpu().numpy()
    seg = [np.arange(rp[r], rp[r + 1]) for r in rows]
    idx = torch.from_numpy(np.concatenate(seg)).to(d)
    sub_rp = np.zeros(len(rows) + 1, np.int32); sub_rp[1:] = np.cumsum([len(s) for s in seg])
    Cref = oracle_spmm(sub_rp, col[idx].cpu().numpy(), val[idx].cpu().numpy(), B.cpu().numpy())
    assert rel_err(C[torch.from_numpy(rows).to(d)].cpu().numpy(), Cref) <= TOL
    # (i') the FULL matrix, every one of the 232 965 rows, against an fp64 evaluation on the device in blocks of rows
    # ("full matrix for n <= 3 M"; gcn_amd/check.py: plain torch arithmetic, no library code)
    from gcn_amd.check import sampled_rows_rel_err
    rel, checked = sampled_rows_rel_err(rowptr, col, val, B, C, torch.arange(n), batch_nnz=1 << 22)
    assert checked == n and rel <= TOL, rel
    # (ii) Â·1: every output column equals the row sum of Â
    ones = torch.ones((n, 64), device=d)
    rs = adj.matmul_raw(ones)
    rowsum = torch.zeros(n, dtype=torch.float64, device=d).index_add_(
        0, torch.repeat_interleave(torch.arange(n, device=d), (rowptr[1:] - rowptr[:-1]).long()), val.double())
    assert float((rs[:, 0].double() - rowsum).abs().max() / rowsum.abs().max()) <= TOL
    assert torch.equal(rs[:, 0], rs[:, 63])
    # (iii) linearity: Â(2B + B') = 2ÂB + ÂB'
    B2 = graphgen.random_features(n, k, seed=3, device=d)
    lhs = adj.matmul_raw(2 * B + B2)
    rhs = 2 * C + adj.matmul_raw(B2)
    assert float((lhs - rhs).abs().max() / rhs.abs().max()) <= TOL


# --- the N>1 building blocks on the real kernel (the collective itself is covered by the gloo
# --- tests in test_dist.py; one box has one GPU) ------------------------------------------------
def test_row_sharded_blocks_reproduce_the_single_gpu_result():
    from gcn_amd.dist import PipelinedAggregation, RowShardedAdjacency
    n, k, d = 6000, 128, _dev()
    rowptr, col, val = sym_norm_graph(n, 150000, seed=9)
    t = [torch.from_numpy(x).to(d) for x in (rowptr, col, val)]
    H = torch.from_numpy(np.random.default_rng(4).standard_normal((n, k)).astype(np.float32)).to(d)
    full = gcn_amd.CsrAdjacency(t[0], t[1], t[2], (n, n), symmetric=True, chunk_nnz=64)
    ref = full.matmul_raw(H)
    for world in (2, 3, 8):
        shards = [RowShardedAdjacency(t[0], t[1], t[2], n, r, world,
                                      lambda rp, ci, va, shape: gcn_amd.CsrAdjacency(rp, ci, va, shape, chunk_nnz=64))
                  for r in range(world)]
        Hp = shards[0].to_padded(H)
        out = shards[0].new_buffer(k, d)
        for r, sh in enumerate(shards):                      # what each rank writes into its slot
            sh.local.matmul_raw(Hp, out=out[r * sh.max_rows: r * sh.max_rows + sh.rows])
        got = shards[0].from_padded(out)
        assert rel_err(got.cpu().numpy(), oracle_spmm(rowptr, col, val, H.cpu().numpy())) <= TOL
        assert sum(sh.local_nnz for sh in shards) == len(col)
    # world = 1 pipeline (planes of 64 columns) over 3 layers == 3 chained single-GPU SpMMs per plane
    sh = RowShardedAdjacency(t[0], t[1], t[2], n, 0, 1, lambda rp, ci, va, shape: gcn_amd.CsrAdjacency(rp, ci, va, shape))
    pipe = PipelinedAggregation(sh, k, d, plane_cols=64)
    pipe.load(H)
    for _ in range(3):
        pipe.step()
    chain = H
    for _ in range(3):
        chain = full.matmul_raw(chain)
    assert rel_err(pipe.result().cpu().numpy(), chain.cpu().numpy()) <= TOL
    assert rel_err(ref.cpu().numpy(), oracle_spmm(rowptr, col, val, H.cpu().numpy())) <= TOL


@pytest.mark.parametrize("S,k", [(2, 64), (8, 128), (16, 128), (16, 33), (5, 256), (32, 64)])
def test_parity_xcd_column_slicing(S, k):
    """slice-major virtual CSR + reduction over slices (gcn_spmm_plan_enable_slicing), incl. the
    fused epilogue, empty rows, hub rows and slices that end up empty"""
    m, n = 3000, 3500
    rowptr, col, val = random_csr(m, n, 70000, seed=S * 100 + k, empty_rows=0.1, long_rows=[(4, 2500), (9, 1)])
    rng = np.random.default_rng(S + k)
    B = rng.standard_normal((n, k)).astype(np.float32)
    bias = rng.standard_normal(k).astype(np.float32)
    adj = _adj(rowptr, col, val, m, n, chunk_nnz=128)
    adj.enable_slicing(S)
    assert adj.num_slices == S
    d = _dev()
    C = adj.matmul_raw(torch.from_numpy(B).to(d)).cpu().numpy()
    Cref = oracle_spmm(rowptr, col, val, B)
    assert rel_err(C, Cref) <= TOL
    C2 = adj.matmul_raw(torch.from_numpy(B).to(d), bias=torch.from_numpy(bias).to(d), relu=True).cpu().numpy()
    assert rel_err(C2, np.maximum(Cref + bias, 0)) <= TOL
    assert torch.equal(adj.matmul_raw(torch.from_numpy(B).to(d)), adj.matmul_raw(torch.from_numpy(B).to(d)))
    adj.enable_slicing(0)                                       # off again → plain path
    assert adj.num_slices == 0
    assert rel_err(adj.matmul_raw(torch.from_numpy(B).to(d)).cpu().numpy(), Cref) <= TOL


def test_slicing_refuses_unsorted_rows():
    m = n = 500
    rowptr, col, val = random_csr(m, n, 5000, seed=1, sorted_cols=False)
    adj = _adj(rowptr, col, val, m, n)
    with pytest.raises(gcn_amd.GcnAmdError):
        adj.enable_slicing(8)
    B = np.random.default_rng(0).standard_normal((n, 64)).astype(np.float32)   # plain path still fine
    assert rel_err(adj.matmul_raw(torch.from_numpy(B).to(_dev())).cpu().numpy(), oracle_spmm(rowptr, col, val, B)) <= TOL


def test_dropin_pair_on_a_dense_graph_runs_the_group_kernels():
    """mean degree >= 128 and a feature table larger than an L2: csr2tile packs the group kernels' format (stream,
    chunk metadata, cut rows; value-free for a normalised adjacency, weighted otherwise) into the caller's buffers and
    flexspmm — which only sees device pointers, m, n, k, n_segs — runs the plan API's kernels on it, at every width"""
    n = 17000                                # 64-column table 4.35 MB > one 4 MiB L2 -> 2 slices (auto_slices)
    rowptr, col, val = sym_norm_graph(n, 1200000, seed=12)
    nnz = len(col)
    assert nnz // n >= 128
    d = _dev()
    rng = np.random.default_rng(5)
    for weighted in (False, True):
        v = val.copy()
        if weighted:
            v = (v * (1.0 + 0.5 * rng.random(nnz))).astype(np.float32)
        t_rp, t_ci, t_va = torch.from_numpy(rowptr.copy()), torch.from_numpy(col.copy()), torch.from_numpy(v.copy())
        out = dropin.csr2tile(t_rp, t_ci, t_va, n, n, nnz, torch.arange(n, dtype=torch.int32))
        seg_rowPtr, segNzCV, segVoMap, tail, nxt, n_segs = out
        hdr = seg_rowPtr.numpy()[:9]
        assert hdr[0] == 0x47434E47 and hdr[1] == 2 and hdr[6] == (0 if weighted else 1) and hdr[8] == nnz
        dev = [t.to(d) for t in (seg_rowPtr, segNzCV, segVoMap, tail, nxt)]
        for k in (1, 3, 7, 16, 32, 41, 100, 128, 200):
            X = rng.standard_normal((n, k)).astype(np.float32)
            Xd = torch.from_numpy(X).to(d).requires_grad_(True)
            C = dropin.flexspmm.apply(dev[0], dev[1], dev[2], n, n, int(n_segs[0]), dev[3], dev[4], Xd)
            assert rel_err(C.detach().cpu().numpy(), oracle_spmm(rowptr, col, v, X)) <= TOL, (weighted, k)
            if k == 128 and not weighted:                                   # backward = the same op (Â symmetric)
                G = rng.standard_normal((n, k)).astype(np.float32)
                C.backward(torch.from_numpy(G).to(d))
                assert rel_err(Xd.grad.cpu().numpy(), oracle_spmm(rowptr, col, v, G)) <= TOL


def test_dropin_flexspmm_refuses_buffers_it_did_not_pack():
    """for a graph shape that takes the group-kernel format flexspmm checks csr2tile's header: buffers without it
    (here: zeros) are refused with one line on stderr and the call RETURNS — C keeps what the caller put there, the
    process lives on (the reference's convention is void / print, cuspmm.cu:3-21).  In a child process, to read stderr."""
    import subprocess
    import sys
    code = (
        "import torch, ctypes, gcn_amd\n"
        "n, nnz = 17000, 2400000\n"
        "d = torch.device('cuda:0')\n"
        "vp = lambda t: ctypes.c_void_p(t.data_ptr())\n"
        "z = lambda k, dt: torch.zeros(k, dtype=dt, device=d)\n"
        "a, b, c = z(9 * (nnz // 9), torch.int32), z(2 * nnz, torch.float32), z(8 * (nnz // 9), torch.int32)\n"
        "t, x = z(256, torch.int32), z(256, torch.int32)\n"
        "B, C = torch.ones((n, 64), device=d), torch.full((n, 64), 7.0, device=d)\n"
        "gcn_amd.load_library().flexspmm(vp(a), vp(b), vp(c), vp(t), vp(x), n, n, 64, nnz // 9, vp(B), vp(C))\n"
        "torch.cuda.synchronize(); print('returned', bool((C == 7.0).all()))\n"
        "gcn_amd.load_library().flexspmm(vp(a), vp(b), vp(c), vp(t), vp(x), n, n, 64, 0, vp(B), vp(C))\n"
        "torch.cuda.synchronize(); print('returned again', bool((C == 7.0).all()))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-1500:]
    assert "returned True" in r.stdout and "returned again True" in r.stdout
    assert "not packed by this library's csr2tile" in r.stderr and "C left untouched" in r.stderr
    assert "csr2tile packed nothing" in r.stderr                                # n_segs = 0


def test_dropin_flexspmm_reads_its_launch_parameters_on_the_device(capfd):
    """after the first call on a set of buffers flexspmm only enqueues kernels: the chunk and cut-row counts come
    from the header ON THE DEVICE (dropin_guard_kernel), so (i) a call can be captured into a HIP graph and replayed,
    and (ii) buffers whose header is overwritten later are not walked — C stays as handed over, nothing faults"""
    n = 17000
    rowptr, col, val = sym_norm_graph(n, 1200000, seed=12)
    nnz = len(col)
    d = _dev()
    out = dropin.csr2tile(torch.from_numpy(rowptr.copy()), torch.from_numpy(col.copy()), torch.from_numpy(val.copy()),
                          n, n, nnz, torch.arange(n, dtype=torch.int32))
    seg_rowPtr, segNzCV, segVoMap, tail, nxt, n_segs = out
    assert int(n_segs[0]) % 2 == 1 and seg_rowPtr[0] == 0x47434E47              # value-free, group format
    dev = [t.to(d) for t in (seg_rowPtr, segNzCV, segVoMap, tail, nxt)]
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    lib = gcn_amd.load_library()
    k = 64
    X = torch.from_numpy(np.random.default_rng(3).standard_normal((n, k)).astype(np.float32)).to(d)
    ref = oracle_spmm(rowptr, col, val, X.cpu().numpy())
    C = torch.zeros((n, k), device=d)
    call = lambda: lib.flexspmm(vp(dev[0]), vp(dev[1]), vp(dev[2]), vp(dev[3]), vp(dev[4]), n, n, k, int(n_segs[0]), vp(X), vp(C))
    call()                                                                    # first sight: header read once on the host
    assert rel_err(C.cpu().numpy(), ref) <= TOL
    # (i) the steady-state call has no host round trip: behind ~100 ms of queued work on the same (legacy) stream
    # three calls return at once — a synchronous header read would wait for the queue to drain
    import time
    big = torch.randn((8192, 8192), device=d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        big = (big @ big) * 1e-4
    queued = time.perf_counter() - t0
    for _ in range(3):
        call()
    returned = time.perf_counter() - t0
    torch.cuda.synchronize()
    drained = time.perf_counter() - t0
    assert drained > 0.05, drained                                            # (the queue really was that long)
    assert returned - queued < 0.3 * drained, (queued, returned, drained)
    assert rel_err(C.cpu().numpy(), ref) <= TOL
    # (ii) the header disappears under a cached set of buffers: the device-side guard skips every kernel
    dev[0][:16] = 0
    C.fill_(3.0)
    call()
    torch.cuda.synchronize()
    assert bool((C == 3.0).all())
    # ... and is not silent for good: the guard counts its refusals in host-mapped memory, the NEXT call
    # reports them and forgets every remembered set of buffers — so this one has its header read again, on the host
    capfd.readouterr()
    call()
    torch.cuda.synchronize()
    err = capfd.readouterr().err
    assert "refused on the device" in err and "not packed by this library's csr2tile" in err, err
    assert bool((C == 3.0).all())


def test_row_constant_and_column_constant_values_take_the_value_free_pass_too():
    """values that depend on the row only or on the column only factor as u_row[r]·1 / 1·u_col[c]: an unweighted
    adjacency (all ones — sum aggregation), Kipf's row-normalised D^-1 (A+I) and its transpose (the backward pass of
    that model) run the sliced pass without their value stream, found automatically; one entry off and they do not"""
    n = 17000
    rowptr, col, _ = sym_norm_graph(n, 1200000, seed=31)
    nnz = len(col)
    deg = np.diff(rowptr).astype(np.float32)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    rng = np.random.default_rng(6)
    d = _dev()
    B = rng.standard_normal((n, 64)).astype(np.float32)
    Bd = torch.from_numpy(B).to(d)
    cases = {"ones": np.ones(nnz, np.float32), "row-normalised": (1.0 / deg)[rows].astype(np.float32),
             "column-normalised": (1.0 / deg)[col].astype(np.float32)}
    for name, v in cases.items():
        adj = _adj(rowptr, col, v, n, n)
        assert adj.num_slices >= 2 and adj.has_value_factors, name
        k_name = adj.main_kernel(64)
        assert k_name.startswith("gcn::spmm_group") and "weighted" not in k_name, (name, k_name)
        assert rel_err(adj.matmul_raw(Bd).cpu().numpy(), oracle_spmm(rowptr, col, v, B)) <= TOL, name
        v2 = v.copy(); v2[nnz // 3] *= 1.001                                  # one entry off: the values travel again
        adj2 = _adj(rowptr, col, v2, n, n)
        assert not adj2.has_value_factors and "weighted" in adj2.main_kernel(64), name
        assert rel_err(adj2.matmul_raw(Bd).cpu().numpy(), oracle_spmm(rowptr, col, v2, B)) <= TOL, name
    # a rectangular block (rows 2000..12000) of the row-normalised matrix: detection does not need a square matrix
    lo, hi = 2000, 12000
    e0, e1 = int(rowptr[lo]), int(rowptr[hi])
    v = cases["row-normalised"][e0:e1]
    blk = _adj((rowptr[lo:hi + 1] - e0).astype(np.int32), col[e0:e1], v, hi - lo, n, slices=4)
    assert blk.has_value_factors and "weighted" not in blk.main_kernel(64)
    assert rel_err(blk.matmul_raw(Bd).cpu().numpy(), oracle_spmm((rowptr[lo:hi + 1] - e0).astype(np.int32), col[e0:e1], v, B)) <= TOL


def test_prelaid_chain_needs_no_feature_copy_and_writes_the_next_layers_input():
    """gcn_spmm_csr_f32_prelaid: B handed over in the plan's slice-by-slice, column-scaled layout B' (no per-call copy),
    the result written straight into a consumer's B' (gapped rows, times the consumer's column factor).  A square
    normalised adjacency chained with itself: two layers equal Â²H; a row block (rectangular, renumbered columns)
    writes its rows into its slot of a larger B'; the zero rows behind the slices are never touched."""
    n = 17000                                           # 64-column table > one L2 -> sliced, value-free
    rowptr, col, val = sym_norm_graph(n, 1200000, seed=21)
    d = _dev()
    rng = np.random.default_rng(4)
    deg = np.diff(rowptr).astype(np.float64)
    u = torch.from_numpy(np.sqrt((deg ** -0.5 * deg ** -0.5).astype(np.float32))).to(d)
    for k, S in ((64, "auto"), (128, 4), (32, 3)):
        adj = _adj(rowptr, col, val, n, n) if S == "auto" else _adj(rowptr, col, val, n, n, slices=S)
        lay = adj.prelaid_layout(k)
        assert lay is not None and lay["slices"] == adj.num_slices and lay["ld"] == k
        w = lay["slice_cols"]
        assert w == -(-n // lay["slices"]) and lay["table_rows"] == lay["slices"] * (w + 1)
        H = rng.standard_normal((n, k)).astype(np.float32)
        Bp = adj.to_prelaid(torch.from_numpy(H).to(d), u)
        ref1 = oracle_spmm(rowptr, col, val, H)
        # plain result from a pre-laid input
        C = torch.full((n, k), 5.0, device=d)
        adj.matmul_prelaid(Bp, C)
        assert rel_err(C.cpu().numpy(), ref1) <= TOL, (k, S)
        # chained: layer 1 writes layer 2's B' (scaled by u, gapped), layer 2 reads it
        Bp2 = torch.zeros_like(Bp)
        adj.matmul_prelaid(Bp, Bp2, out_scale=u, out_gap=w)
        assert float(Bp2[w::w + 1].abs().max()) == 0.0                      # the zero row behind every slice
        r = torch.arange(n, device=d)
        got1 = (Bp2[r + r // w] / u[:, None]).cpu().numpy()
        assert rel_err(got1, ref1) <= TOL, (k, S)
        C2 = torch.empty((n, k), device=d)
        adj.matmul_prelaid(Bp2, C2)
        assert rel_err(C2.cpu().numpy(), oracle_spmm(rowptr, col, val, ref1)) <= TOL, (k, S)
        assert torch.equal(C2, adj.matmul_prelaid(Bp2, torch.empty_like(C2)))     # reproducible
    # widths and plans without the layout say so instead of computing something else
    assert _adj(rowptr, col, val, n, n).prelaid_layout(30) is None              # k % 4 != 0
    assert _adj(rowptr, col, val, n, n, slices=0).prelaid_layout(64) is None    # unsliced
    val2 = val.copy(); val2[7] *= 1.01
    assert _adj(rowptr, col, val2, n, n).prelaid_layout(64) is None             # values do not factor
    with pytest.raises(gcn_amd.GcnAmdError):
        _adj(rowptr, col, val, n, n).matmul_prelaid(torch.zeros((10, 64), device=d), torch.zeros((n, 64), device=d))
    # a row block (the multi-GPU shards): rows [lo, n) of the matrix, all columns, lo on a slice boundary as dist.py
    # cuts its slots — factors handed over, the output goes into ITS rows of a B' of the whole column space
    S = 4
    w = -(-n // S)
    lo, hi = w, n
    e0, e1 = int(rowptr[lo]), int(rowptr[hi])
    blk = _adj((rowptr[lo:hi + 1] - e0).astype(np.int32), col[e0:e1], val[e0:e1], hi - lo, n, slices=S)
    blk.set_value_factors(u[lo:hi], u)
    lay = blk.prelaid_layout(64)
    assert lay is not None and lay["slices"] == S and lay["slice_cols"] == w
    H = rng.standard_normal((n, 64)).astype(np.float32)
    Bp = blk.to_prelaid(torch.from_numpy(H).to(d), u)
    nxt = torch.zeros_like(Bp)
    blk.matmul_prelaid(Bp, nxt[lo + lo // w:], out_scale=u[lo:hi], out_gap=w)
    rr = torch.arange(lo, hi, device=d)
    got = (nxt[rr + rr // w] / u[lo:hi, None]).cpu().numpy()
    assert rel_err(got, oracle_spmm(rowptr, col, val, H)[lo:hi]) <= TOL
    assert float(nxt[w::w + 1].abs().max()) == 0.0 and float(nxt[: lo + lo // w].abs().max()) == 0.0


def test_full_size_products_shape_rcm_reordered():
    """BASELINE config 3: products-shaped graph (n = 2 449 029, nnz ≈ 126.2 M), k = 256,
    RCM-reordered by the build's order_rcm.  Checked through size-independent properties:
    sampled rows vs the fp64 oracle, and P·Â·Pᵀ·(P·B) = P·(Â·B) against the un-reordered run."""
    d = _dev()
    rowptr, col, val, n = graphgen.make_graph("products", device=d, seed=3)
    nnz, k = int(col.numel()), 256
    assert n == 2449029 and abs(nnz - 126.2e6) < 0.1e6
    B = graphgen.random_features(n, k, seed=2, device=d)
    base = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True).matmul_raw(B)
    rp, ci, va = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    rank = gcn_amd.reorder.order_rcm(rp, ci)
    assert np.array_equal(np.sort(rank), np.arange(n))                      # a permutation
    rp2, ci2, va2, vomp = gcn_amd.reorder.apply_rank(rp, ci, va, rank)
    adj = gcn_amd.CsrAdjacency(torch.from_numpy(rp2).to(d), torch.from_numpy(ci2).to(d),
                               torch.from_numpy(va2).to(d), (n, n), symmetric=True)
    vomp_d = torch.from_numpy(vomp).to(d)
    C = adj.matmul_raw(gcn_amd.gather_rows(B, vomp_d))
    assert float((C - base[vomp_d.long()]).abs().max() / base.abs().max()) <= TOL
    rows = np.random.default_rng(1).choice(n, 2048, replace=False); rows.sort()
    seg = [np.arange(rp[r], rp[r + 1]) for r in rows]
    sub_rp = np.zeros(len(rows) + 1, np.int32); sub_rp[1:] = np.cumsum([len(s) for s in seg])
    idx = np.concatenate(seg)
    Cref = oracle_spmm(sub_rp, ci[idx], va[idx], B.cpu().numpy())
    assert rel_err(base[torch.from_numpy(rows).to(d)].cpu().numpy(), Cref) <= TOL


def test_rccl_collectives_used_by_the_multi_gpu_path_single_rank():
    """One box has one GPU, so the N>1 exchange cannot run here; this at least drives the exact
    RCCL calls of gcn_amd/dist.py (in-place all_gather_into_tensor with async_op, barrier,
    all_reduce MAX) through backend 'nccl' with world_size 1, around a real sharded SpMM."""
    import socket
    import torch.distributed as dist
    from gcn_amd.dist import PipelinedAggregation, RowShardedAdjacency
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    d = _dev()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=d)
    try:
        n, k = 4000, 128
        rowptr, col, val = sym_norm_graph(n, 60000, seed=14)
        t = [torch.from_numpy(x).to(d) for x in (rowptr, col, val)]
        H = torch.from_numpy(np.random.default_rng(2).standard_normal((n, k)).astype(np.float32)).to(d)
        sh = RowShardedAdjacency(t[0], t[1], t[2], n, 0, 1, lambda rp, ci, va, shape: gcn_amd.CsrAdjacency(rp, ci, va, shape))
        sh.world = 1
        buf_in, buf_out = sh.to_padded(H), sh.new_buffer(k, d)
        slot = buf_out[0: sh.max_rows]
        sh.local.matmul_raw(buf_in, out=slot[: sh.rows])
        assert sh.collective_form() == "none"                     # (one rank: layer() would not exchange at all)
        work = sh._exchange(buf_out, slot, None, True)            # the in-place async all-gather, as RCCL runs it
        work.wait()
        dist.barrier()
        tmax = torch.tensor([1.5], dtype=torch.float64, device=d)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        assert float(tmax) == 1.5
        assert rel_err(sh.from_padded(buf_out).cpu().numpy(), oracle_spmm(rowptr, col, val, H.cpu().numpy())) <= TOL
    finally:
        dist.destroy_process_group()


def test_wide_column_space_beyond_2_pow_24_and_4GiB_of_features():
    """papers100M-like shard shape in miniature: a row block with a column space of 40 M vertices
    (column ids above 2^24, where the reference's float-encoded columns lose exactness, tile.cu:67)
    and a feature matrix of 10 GB (> 4 GiB → 64-bit flat addressing path, 64-bit row offsets).
    Checked on sampled rows against an fp64 gather-sum."""
    d = _dev()
    m, n, k, deg = 1_000_000, 40_000_000, 64, 24
    g = torch.Generator(device=d); g.manual_seed(7)
    col = torch.randint(0, n, (m * deg,), generator=g, device=d, dtype=torch.int64)
    col[:deg] = torch.arange(n - deg, n, device=d)                  # make sure the last columns are hit
    key = torch.sort(torch.arange(m, device=d).repeat_interleave(deg) * n + col).values
    col = (key % n).to(torch.int32)
    rowptr = (torch.arange(m + 1, device=d) * deg).to(torch.int32)
    val = torch.rand(m * deg, generator=g, device=d) - 0.5
    B = torch.randn((n, k), generator=g, device=d)
    assert B.numel() * 4 > (1 << 32) and int(col.max()) > (1 << 24)
    adj = gcn_amd.CsrAdjacency(rowptr, col, val, (m, n))
    C = adj.matmul_raw(B)
    rows = torch.from_numpy(np.random.default_rng(3).choice(m, 4096, replace=False)).to(d)
    idx = (rows[:, None] * deg + torch.arange(deg, device=d)[None, :]).reshape(-1)
    ref = (val[idx].double()[:, None] * B[col[idx].long()].double()).reshape(len(rows), deg, k).sum(1)
    err = float((C[rows].double() - ref).abs().max() / ref.abs().max())
    assert err <= TOL


def _banded_csr(n, half_band, extra, seed, hub=None):
    """near-diagonal matrix (what a renumbered community graph looks like) + a few far entries"""
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for r in range(n):
        lo, hi = max(0, r - half_band), min(n, r + half_band + 1)
        c = rng.choice(np.arange(lo, hi), size=min(hi - lo, int(rng.integers(0, 2 * half_band // 3 + 2))), replace=False)
        far = rng.integers(0, n, extra)
        cc = np.unique(np.concatenate([c, far]))
        if hub is not None and r == hub[0]:
            cc = np.unique(rng.choice(n, hub[1], replace=False))
        rows.append(np.full(len(cc), r)); cols.append(cc)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    import scipy.sparse as sp
    A = sp.csr_matrix((rng.standard_normal(len(rows)).astype(np.float32), (rows, cols)), shape=(n, n))
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32)


@pytest.mark.parametrize("k", [33, 64, 100, 128, 200])
def test_parity_lds_staged_row_panels(k):
    """spmm_panel_kernel: window hits from LDS, misses from global, hub row by the whole workgroup,
    empty rows, last partial panel, epilogue; forced on and automatic"""
    n = 3001
    rowptr, col, val = _banded_csr(n, 150, 3, seed=k, hub=(777, 2600))
    rng = np.random.default_rng(k)
    B = rng.standard_normal((n, k)).astype(np.float32)
    bias = rng.standard_normal(k).astype(np.float32)
    d = _dev()
    adj = _adj(rowptr, col, val, n, n, panels=1)
    assert adj.panel_rows > 0 and adj.panel_coverage > 0.8
    Cref = oracle_spmm(rowptr, col, val, B)
    assert rel_err(adj.matmul_raw(torch.from_numpy(B).to(d)).cpu().numpy(), Cref) <= TOL
    C2 = adj.matmul_raw(torch.from_numpy(B).to(d), bias=torch.from_numpy(bias).to(d), relu=True).cpu().numpy()
    assert rel_err(C2, np.maximum(Cref + bias, 0)) <= TOL
    auto = _adj(rowptr, col, val, n, n, panels="auto")       # by coverage: >= 0.5 → panels on
    assert auto.panel_rows > 0 and auto.num_slices == 0
    assert torch.equal(auto.matmul_raw(torch.from_numpy(B).to(d)), adj.matmul_raw(torch.from_numpy(B).to(d)))


def test_panels_stay_off_for_unstructured_graphs_and_work_when_forced():
    m, n, k = 2500, 3000, 128
    rowptr, col, val = random_csr(m, n, 60000, seed=5, empty_rows=0.1, long_rows=[(9, 2500)])
    auto = _adj(rowptr, col, val, m, n, panels="auto")
    assert auto.panel_rows == 0 and auto.panel_coverage < 0.5
    forced = _adj(rowptr, col, val, m, n, panels=1)           # non-square, coverage ~ 17 %: still exact
    B = np.random.default_rng(0).standard_normal((n, k)).astype(np.float32)
    assert rel_err(forced.matmul_raw(torch.from_numpy(B).to(_dev())).cpu().numpy(), oracle_spmm(rowptr, col, val, B)) <= TOL


def test_panels_with_everything_staged_and_with_nothing_staged():
    """both degenerate splits: A_out empty (epilogue-only second phase) and A_in (almost) empty"""
    d = _dev()
    n, k = 2000, 96
    rowptr, col, val = _banded_csr(n, 100, 0, seed=3)             # every entry inside its panel's window
    rng = np.random.default_rng(4)
    B = rng.standard_normal((n, k)).astype(np.float32)
    bias = rng.standard_normal(k).astype(np.float32)
    adj = _adj(rowptr, col, val, n, n, panels=1)
    assert adj.panel_coverage == 1.0
    Cref = oracle_spmm(rowptr, col, val, B)
    assert rel_err(adj.matmul_raw(torch.from_numpy(B).to(d)).cpu().numpy(), Cref) <= TOL
    C2 = adj.matmul_raw(torch.from_numpy(B).to(d), bias=torch.from_numpy(bias).to(d), relu=True).cpu().numpy()
    assert rel_err(C2, np.maximum(Cref + bias, 0)) <= TOL
    # columns far from the rows: (almost) nothing staged, everything through the accumulate path
    m2, n2 = 1500, 200000
    rp2, ci2, va2 = random_csr(m2, n2, 30000, seed=6, empty_rows=0.2)
    B2 = rng.standard_normal((n2, k)).astype(np.float32)
    adj2 = _adj(rp2, ci2, va2, m2, n2, panels=1)
    assert adj2.panel_coverage < 0.05
    assert rel_err(adj2.matmul_raw(torch.from_numpy(B2).to(d)).cpu().numpy(), oracle_spmm(rp2, ci2, va2, B2)) <= TOL


def test_autotune_keeps_the_faster_of_sliced_and_unsliced_and_the_result():
    """CsrAdjacency.autotune measures both configurations on the matrix itself; whatever it keeps,
    the product is the same"""
    rowptr, col, val, n = graphgen.make_sbm(30000, device="cuda:0", seed=7)
    adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True)
    B = graphgen.random_features(n, 128, seed=3, device="cuda:0")
    before = adj.matmul_raw(B).clone()
    timings = adj.autotune(k=128)
    assert len(timings) >= 3 and (0, 0) in timings and (0, 64) in timings and all(t > 0 for t in timings.values())
    assert adj.num_slices == next(iter(timings))[0]           # the fastest shape (slices, tile) stays configured
    after = adj.matmul_raw(B)
    assert float((after - before).abs().max() / before.abs().max()) <= 1e-6


def test_value_free_sliced_pass_for_normalised_adjacencies():
    """Â = D^-1/2 (A+I) D^-1/2 has values u[r]*u[c]: the sliced main pass then runs without its value
    stream on B scaled by u (gcn_spmm_plan_enable_slicing detects it).  Same result to 1e-5; any other
    value pattern — even one perturbed entry — keeps the ordinary kernel."""
    n = 17000                                           # 64-column table > one L2 -> sliced
    rowptr, col, val = sym_norm_graph(n, 1200000, seed=21)
    rng = np.random.default_rng(8)
    for k in (64, 128, 100, 41, 47):                    # 100: the scaling rides on the row-padding copy; 41, 47: on the odd-width copy
        B = rng.standard_normal((n, k)).astype(np.float32)
        bias = rng.standard_normal(k).astype(np.float32)
        adj = _adj(rowptr, col, val, n, n)
        assert adj.num_slices >= 2 and adj.main_kernel(k).startswith("gcn::spmm_group")
        Bd = torch.from_numpy(B).to(_dev())
        Cref = oracle_spmm(rowptr, col, val, B)
        assert rel_err(adj.matmul_raw(Bd).cpu().numpy(), Cref) <= TOL
        Ce = adj.matmul_raw(Bd, bias=torch.from_numpy(bias).to(_dev()), relu=True).cpu().numpy()
        assert rel_err(Ce, np.maximum(Cref + bias, 0)) <= TOL
        assert torch.equal(adj.matmul_raw(Bd), adj.matmul_raw(Bd))          # reproducible
    # several slice counts, on a graph whose size is not a multiple of any of them
    B = rng.standard_normal((n, 128)).astype(np.float32)
    Cref = oracle_spmm(rowptr, col, val, B)
    for S in (2, 3, 4, 7, 8):
        adj = _adj(rowptr, col, val, n, n, slices=S)
        assert adj.num_slices == S and adj.main_kernel(128).startswith("gcn::spmm_group")
        assert rel_err(adj.matmul_raw(torch.from_numpy(B).to(_dev())).cpu().numpy(), Cref) <= TOL
    # one entry off by 1e-4 relative: no longer rank-1 -> the same walk WITH its values, and the right answer
    val2 = val.copy(); val2[len(val2) // 2] *= 1.0001
    adj2 = _adj(rowptr, col, val2, n, n)
    assert adj2.num_slices >= 2 and adj2.main_kernel(128).startswith("gcn::spmm_group_weighted_kernel<")
    B = rng.standard_normal((n, 128)).astype(np.float32)
    assert rel_err(adj2.matmul_raw(torch.from_numpy(B).to(_dev())).cpu().numpy(), oracle_spmm(rowptr, col, val2, B)) <= TOL
    # arbitrary weights
    val3 = (rng.random(len(val)) + 0.1).astype(np.float32)
    adj3 = _adj(rowptr, col, val3, n, n)
    assert adj3.main_kernel(128).startswith("gcn::spmm_group_weighted_kernel<")
    if os.environ.get("GCN_AMD_GROUP8", "1") != "0":
        assert adj3.main_kernel(16).startswith("gcn::spmm_group8_weighted_kernel<")   # k <= 32: eight engines per wave
    for k in (64, 100, 41, 16, 20, 30):                  # 100, 41, 30: on the row-padded / odd-width copies
        Bk = rng.standard_normal((n, k)).astype(np.float32)
        assert rel_err(adj3.matmul_raw(torch.from_numpy(Bk).to(_dev())).cpu().numpy(), oracle_spmm(rowptr, col, val3, Bk)) <= TOL
    assert rel_err(adj3.matmul_raw(torch.from_numpy(B).to(_dev())).cpu().numpy(), oracle_spmm(rowptr, col, val3, B)) <= TOL


def test_explicit_value_factors_on_a_row_block_with_renumbered_columns():
    """a row block of Â with permuted column numbering still has values u_row[r]*u_col[c]: handed over
    explicitly (gcn_spmm_plan_set_value_factors) the sliced pass drops the value stream; factors that do
    not match are refused"""
    n = 20000
    rowptr, col, val = sym_norm_graph(n, 1500000, seed=5)
    A = sp.csr_matrix((val, col, rowptr), shape=(n, n))
    u = np.sqrt(A.diagonal()).astype(np.float32)
    perm = np.random.default_rng(2).permutation(n)              # new column id of old column c
    lo, hi = 3000, 18000                                        # 15 000 rows x ~150: > 96 non-zeros per column
    Ablk = A[lo:hi][:, np.argsort(perm)].tocsr(); Ablk.sort_indices()   # column j of the block = old column argsort(perm)[j]
    u_col = u[np.argsort(perm)]
    rp, ci, va = Ablk.indptr.astype(np.int32), Ablk.indices.astype(np.int32), Ablk.data.astype(np.float32)
    adj = _adj(rp, ci, va, hi - lo, n, slices=2)
    assert not adj.has_value_factors                            # rectangular: nothing to detect
    assert adj.main_kernel(128).startswith("gcn::spmm_group_weighted_kernel<")
    B = np.random.default_rng(3).standard_normal((n, 128)).astype(np.float32)
    Bd = torch.from_numpy(B).to(_dev())
    plain = adj.matmul_raw(Bd).cpu().numpy()
    adj.set_value_factors(torch.from_numpy(u[lo:hi]), torch.from_numpy(u_col))
    assert adj.has_value_factors and adj.main_kernel(128).startswith("gcn::spmm_group")
    fast = adj.matmul_raw(Bd).cpu().numpy()
    Cref = oracle_spmm(rp, ci, va, B)
    assert rel_err(plain, Cref) <= TOL and rel_err(fast, Cref) <= TOL
    with pytest.raises(gcn_amd.GcnAmdError) as refused:
        adj.set_value_factors(torch.from_numpy(u[lo:hi] * 1.001), torch.from_numpy(u_col))
    assert refused.value.status == _lib.ERR_NOT_FACTORED        # its own status: dist.py hides this one and nothing else
    assert not adj.has_value_factors                            # a refused hand-over leaves none behind
    assert rel_err(adj.matmul_raw(Bd).cpu().numpy(), Cref) <= TOL
    adj.set_value_factors(torch.from_numpy(u[lo:hi]), torch.from_numpy(u_col))
    adj.set_value_factors(None, None)                           # forgotten on request: back to the value stream
    assert not adj.has_value_factors and adj.main_kernel(128).startswith("gcn::spmm_group_weighted_kernel<")
    assert rel_err(adj.matmul_raw(Bd).cpu().numpy(), Cref) <= TOL


def test_automatic_slice_count_follows_the_value_factors():
    """auto_slices on the group-kernel path: one slice per 4 MiB of the 64-column table, with the values beside the
    stream until they are known to factor, value-free afterwards — also when the factors arrive AFTER the automatic
    slicing (row blocks of the multi-GPU path: gcn_spmm_plan_set_value_factors rebuilds the streams).  Sampled rows
    against the fp64 oracle on both plans; an explicit slice count is kept."""
    from util import sampled_rows_oracle_err
    from gcn_amd import graphgen
    d = _dev()
    n, world = 170000, 8                                        # table 43.5 MB -> 11 slices
    rowptr, col, val, n, lo, hi, deg = graphgen.make_rmat_row_block(n, 60000000, world, 0, device=d, seed=9)
    m = hi - lo
    assert int(col.numel()) // m >= 128 and int(col.numel()) // n >= 48
    adj = gcn_amd.CsrAdjacency(rowptr, col, val, (m, n), symmetric=False)
    assert adj.num_slices == 11 and not adj.has_value_factors
    assert adj.main_kernel(128).startswith("gcn::spmm_group_weighted_kernel<")
    B = torch.randn((n, 128), device=d, generator=torch.Generator(device=d).manual_seed(1))
    rows = np.arange(0, m, max(1, m // 300), dtype=np.int64)
    with_values = adj.matmul_raw(B)
    assert sampled_rows_oracle_err(rowptr, col, val, B, with_values, rows)[0] <= TOL
    u = graphgen.value_factor_from_degrees(deg)
    adj.set_value_factors(u[lo:hi], u)
    assert adj.has_value_factors and adj.num_slices == 11 and adj.main_kernel(128).startswith("gcn::spmm_group_ring_kernel<")
    value_free = adj.matmul_raw(B)
    assert sampled_rows_oracle_err(rowptr, col, val, B, value_free, rows)[0] <= TOL
    adj.enable_slicing(4)                                       # an explicit count is kept
    adj.set_value_factors(u[lo:hi], u)
    assert adj.num_slices == 4


def _hub_graph(n, e, hubs, seed):
    """normalised adjacency with a few vertices adjacent to (nearly) everything: virtual rows of n/S entries
    (they cross many chunks of the group kernel) next to rows with a handful of entries and empty slices"""
    rng = np.random.default_rng(seed)
    u, v = rng.integers(0, n, e), rng.integers(0, n, e)
    for h in hubs:
        hv = rng.choice(n, size=int(0.9 * n), replace=False)
        u, v = np.concatenate([u, np.full(len(hv), h)]), np.concatenate([v, hv])
    A = sp.coo_matrix((np.ones(len(u)), (u, v)), shape=(n, n)); A = (A + A.T).tocsr()
    A.setdiag(0); A.eliminate_zeros(); A.data[:] = 1.0
    A = (A + sp.eye(n)).tocsr()
    d = np.asarray(A.sum(1)).ravel() ** -0.5
    A = (sp.diags(d) @ A @ sp.diags(d)).tocsr(); A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32)


@pytest.mark.parametrize("S", [2, 3, 5, 8, 16, 32])
def test_group_kernel_value_free_sliced_pass(S):
    """the sliced pass of a normalised adjacency on the 15-bit stream (spmm_group.hip): four independent row
    engines per wave, row ends as stream bits, padding entries for empty virtual rows; hub rows cross chunks"""
    n = 9000 + S                                          # not a multiple of the slice count
    rowptr, col, val = _hub_graph(n, 900000, hubs=(0, n // 2, n - 1), seed=30 + S)
    rng = np.random.default_rng(S)
    adj = _adj(rowptr, col, val, n, n, slices=S)
    assert adj.num_slices == S and adj.has_value_factors
    for k in (64, 128, 100, 36, 41):
        assert adj.main_kernel(k).startswith("gcn::spmm_group"), adj.main_kernel(k)
        B = rng.standard_normal((n, k)).astype(np.float32)
        Bd = torch.from_numpy(B).to(_dev())
        Cref = oracle_spmm(rowptr, col, val, B)
        C = adj.matmul_raw(Bd)
        assert rel_err(C.cpu().numpy(), Cref) <= TOL
        assert torch.equal(C, adj.matmul_raw(Bd))                             # reproducible
        bias = rng.standard_normal(k).astype(np.float32)
        Ce = adj.matmul_raw(Bd, bias=torch.from_numpy(bias).to(_dev()), relu=True).cpu().numpy()
        assert rel_err(Ce, np.maximum(Cref + bias, 0)) <= TOL
    # a NaN / Inf feature row reaches exactly the rows of A that reference it
    B = rng.standard_normal((n, 64)).astype(np.float32)
    bad = [7, n // 3]
    B[bad[0], 5] = np.nan; B[bad[1], 60] = np.inf
    C = adj.matmul_raw(torch.from_numpy(B).to(_dev())).cpu().numpy()
    A = sp.csr_matrix((val, col, rowptr), shape=(n, n))
    touched = np.asarray((A[:, bad] != 0).sum(1)).ravel() > 0
    assert np.all(np.isfinite(C[~touched])) and np.all(~np.isfinite(C[touched]).all(1) | True)
    assert np.all(np.isnan(C[np.asarray((A[:, [bad[0]]] != 0).sum(1)).ravel() > 0, 5]))


@pytest.mark.parametrize("S", [2, 8, 15])
def test_narrow_widths_on_the_eight_engine_kernel(S):
    """k <= 32 on a value-free sliced plan runs spmm_group8_kernel (eight 8-lane row engines per wave, 8-slot LDS
    ring); hub rows cross many chunks, bias/ReLU ride in the slice reduction, results are bitwise reproducible"""
    n = 9000
    rowptr, col, val = _hub_graph(n, 700000, hubs=(5, 4000, 8999), seed=31 + S)
    rng = np.random.default_rng(S)
    adj = _adj(rowptr, col, val, n, n, slices=S)
    assert adj.num_slices == S and adj.has_value_factors
    g8_on = os.environ.get("GCN_AMD_GROUP8", "1") != "0"            # (tools/knob_matrix.sh runs the suite with the knob off)
    for k in (12, 16, 20, 24, 28, 32, 17, 30, 31):
        assert not g8_on or adj.main_kernel(k).startswith("gcn::spmm_group8_kernel<"), (k, adj.main_kernel(k))
        B = rng.standard_normal((n, k)).astype(np.float32)
        bias = rng.standard_normal(k).astype(np.float32)
        Bd = torch.from_numpy(B).to(_dev())
        Cref = oracle_spmm(rowptr, col, val, B)
        C = adj.matmul_raw(Bd)
        assert rel_err(C.cpu().numpy(), Cref) <= TOL, k
        assert torch.equal(C, adj.matmul_raw(Bd))
        Ce = adj.matmul_raw(Bd, bias=torch.from_numpy(bias).to(_dev()), relu=True).cpu().numpy()
        assert rel_err(Ce, np.maximum(Cref + bias, 0)) <= TOL, k
    assert adj.main_kernel(36) in ("gcn::spmm_group12_kernel", "gcn::spmm_group_ring_kernel<2, false>")   # (GCN_AMD_GROUP12=0: the latter)
    assert adj.main_kernel(52).startswith("gcn::spmm_group_ring_kernel<")
    assert not adj.main_kernel(8).startswith("gcn::spmm_group")


def test_value_free_pass_with_slices_wider_than_the_15_bit_stream():
    """slices of more than 32 767 columns cannot use the group kernel's 15-bit entries: the value-free pass then
    runs the four-per-gather kernel on its 16-bit column stream (<= 65 535 columns per slice, <= 8 slices)"""
    n = 70000
    rowptr, col, val = sym_norm_graph(n, 3600000, seed=77)          # ~100 non-zeros per row and column
    adj = _adj(rowptr, col, val, n, n, slices=2)                    # 35 000 columns per slice
    adj.set_gather_width(4)                                         # (52 entries per virtual row: force the quad layout)
    assert adj.num_slices == 2 and adj.has_value_factors
    assert adj.main_kernel(128) == "gcn::spmm_quad_kernel<16, false, true, true>"
    B = np.random.default_rng(5).standard_normal((n, 128)).astype(np.float32)
    C = adj.matmul_raw(torch.from_numpy(B).to(_dev())).cpu().numpy()
    assert rel_err(C, oracle_spmm(rowptr, col, val, B)) <= TOL
    # ... and values that do not factor keep the four-per-gather kernel with its 32-bit columns and value stream
    val2 = val.copy(); val2[7] *= 1.01
    adj2 = _adj(rowptr, col, val2, n, n, slices=2)
    adj2.set_gather_width(4)
    assert not adj2.has_value_factors and adj2.main_kernel(128) == "gcn::spmm_quad_kernel<16, false, false, false>"
    assert rel_err(adj2.matmul_raw(torch.from_numpy(B).to(_dev())).cpu().numpy(), oracle_spmm(rowptr, col, val2, B)) <= TOL


def _dense_band_csr(n, half_band, density, seed, sparse_from=None):
    """rows hold `density` of the columns within +-half_band of the diagonal (rows >= sparse_from: 3 % instead),
    plus a few far entries: the 128 x 512 windows of the panels are 25-60 % dense"""
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for r in range(n):
        lo, hi = max(0, r - half_band), min(n, r + half_band + 1)
        dens = density if sparse_from is None or r < sparse_from else 0.03
        c = np.flatnonzero(rng.random(hi - lo) < dens) + lo
        far = rng.integers(0, n, 2)
        c = np.unique(np.concatenate([c, far]))
        rows.append(np.full(len(c), r)); cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    A = sp.csr_matrix(((rng.standard_normal(len(rows)) * 0.5).astype(np.float32), (rows, cols)), shape=(n, n))
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32)


@pytest.mark.parametrize("k", [64, 128, 100, 36])
def test_dense_panels_run_on_the_matrix_cores(k):
    """panels whose 128 x 512 window is >= 25 % dense are stored as dense fp32 tiles and contracted with
    v_mfma_f32_32x32x2_f32 (exact fp32: the 1e-5 contract holds); sparse panels of the same matrix keep the LDS
    kernel, entries outside the windows the accumulate pass; last partial panel, window clipped at the matrix
    edge, bias + ReLU"""
    n = 2500                                              # 19.5 panels; the last window ends at the matrix edge
    rowptr, col, val = _dense_band_csr(n, 200, 0.6, seed=k, sparse_from=1700)
    d = _dev()
    adj = _adj(rowptr, col, val, n, n, panels=1)
    assert adj.panel_rows == 128 and 10 <= adj.dense_panels <= 14        # rows < 1700 dense, the rest sparse
    rng = np.random.default_rng(k)
    B = rng.standard_normal((n, k)).astype(np.float32)
    bias = rng.standard_normal(k).astype(np.float32)
    Bd = torch.from_numpy(B).to(d)
    Cref = oracle_spmm(rowptr, col, val, B)
    C = adj.matmul_raw(Bd)
    assert rel_err(C.cpu().numpy(), Cref) <= TOL
    assert torch.equal(C, adj.matmul_raw(Bd))                             # reproducible
    C2 = adj.matmul_raw(Bd, bias=torch.from_numpy(bias).to(d), relu=True).cpu().numpy()
    assert rel_err(C2, np.maximum(Cref + bias, 0)) <= TOL
    auto = _adj(rowptr, col, val, n, n, panels="auto")                    # coverage >= 0.5: on by itself, same tiles
    assert auto.dense_panels == adj.dense_panels
    assert torch.equal(auto.matmul_raw(Bd), C)


def test_dense_panels_keep_non_finite_features_out_of_rows_that_do_not_reference_them():
    """a dense contraction would multiply the zeros of A with an Inf feature value (0 * Inf = NaN) and poison the
    whole panel; a window that holds a non-finite value is summed entry by entry instead"""
    n, k = 1536, 64
    rowptr, col, val = _dense_band_csr(n, 200, 0.5, seed=3)
    adj = _adj(rowptr, col, val, n, n, panels=1)
    assert adj.dense_panels == 12
    B = np.random.default_rng(3).standard_normal((n, k)).astype(np.float32)
    B[300, 7] = np.inf; B[900, 50] = np.nan
    C = adj.matmul_raw(torch.from_numpy(B).to(_dev())).cpu().numpy()
    A = sp.csr_matrix((val, col, rowptr), shape=(n, n))
    touched = np.asarray((A[:, [300, 900]] != 0).sum(1)).ravel() > 0
    assert np.all(np.isfinite(C[~touched]))                                # nothing leaked
    assert np.all(~np.isfinite(C[np.asarray((A[:, [300]] != 0).sum(1)).ravel() > 0, 7]))
    Bf = B.copy(); Bf[300, 7] = 0.0; Bf[900, 50] = 0.0
    Cref = oracle_spmm(rowptr, col, val, Bf)
    assert rel_err(C[~touched], Cref[~touched]) <= TOL                     # and the clean rows are exact


@pytest.mark.parametrize("shape", ["small", "sliced", "sliced_epilogue", "weighted"])
def test_spmm_can_be_captured_in_a_hip_graph_and_replayed(shape):
    """after one eager call (workspaces allocated, streams built) gcn_spmm_csr_f32 only enqueues kernels on the stream it
    is given: the call is captured in a HIP graph (torch.cuda.CUDAGraph) and replayed on NEW operand contents — the path
    a launch-bound training loop on a small graph takes; the sliced group kernels (partial slab, fused fix-up) too"""
    d = _dev()
    n, e, k = (3000, 15000, 16) if shape == "small" else (20000, 1500000, 128)
    rowptr, col, val = sym_norm_graph(n, e, seed=11)
    if shape == "weighted":
        val = (val * (1.0 + 0.5 * np.random.default_rng(5).random(len(val)))).astype(np.float32)
    adj = _adj(rowptr, col, val, n, n, **({} if shape == "small" else {"slices": 4}))
    name = adj.main_kernel(k)
    if shape != "small":
        assert name.startswith("gcn::spmm_group") and (shape != "weighted" or "weighted" in name), name
    rng = np.random.default_rng(12)
    B0 = rng.standard_normal((n, k)).astype(np.float32)
    B1 = rng.standard_normal((n, k)).astype(np.float32)
    bias = rng.standard_normal(k).astype(np.float32) if shape == "sliced_epilogue" else None
    bias_d = torch.from_numpy(bias).to(d) if bias is not None else None
    Bd = torch.from_numpy(B0).to(d)
    out = torch.empty((n, k), dtype=torch.float32, device=d)
    side = torch.cuda.Stream(d)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        adj.matmul_raw(Bd, out=out, bias=bias_d, relu=bias is not None)           # eager warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        adj.matmul_raw(Bd, out=out, bias=bias_d, relu=bias is not None)
    for B in (B1, B0):
        Bd.copy_(torch.from_numpy(B))
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ref = oracle_spmm(rowptr, col, val, B)
        if bias is not None:
            ref = np.maximum(ref + bias, 0)
        assert rel_err(out.cpu().numpy(), ref) <= TOL, (shape, name)


@pytest.mark.parametrize("scale", [0.5, 1.0])
def test_second_slice_set_for_narrow_widths(scale):
    """a value-free plan with an automatic slice count cuts the matrix again, into fewer slices, at its first call of a
    narrow width — k <= 32 (a table row is 128 bytes): Reddit-shaped at half / full size 8 -> 4 / 15 -> 8 slices;
    33 <= k <= 48 stay on the plan's own slices (rows of 48 floats for the five-engine kernel).  Wide calls stay on the plan's own slices, narrow ones (incl. odd widths and the epilogue) run on the second
    set; every result against the fp64 oracle on sampled rows, bitwise repeatable; an explicit slice count builds none."""
    from util import sampled_rows_oracle_err
    d = _dev()
    rowptr, col, val, n = graphgen.make_graph("reddit", device=d, seed=1, scale=scale)
    adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True)
    S = adj.num_slices
    assert S == (15 if scale == 1.0 else 8) and adj.has_value_factors and adj.narrow_slices == 0
    rows = np.random.default_rng(3).choice(n, 1024, replace=False); rows.sort()
    B128 = graphgen.random_features(n, 128, seed=2, device=d)
    C128 = adj.matmul_raw(B128)
    assert adj.narrow_slices == 0                                            # (no narrow call so far)
    env = os.environ.get
    g8_on = env("GCN_AMD_GROUP8", "1") != "0"          # (tools/knob_matrix.sh runs the suite with these two off: the
    g12_on = env("GCN_AMD_GROUP12", "1") != "0"        #  numbers below are checked either way)
    expect = 8 if scale == 1.0 else 4
    for k in (16, 32, 12, 30, 7):
        B = graphgen.random_features(n, k, seed=10 + k, device=d)
        C = adj.matmul_raw(B)
        if k >= 12:
            assert adj.narrow_slices == expect, (k, adj.narrow_slices)
            if g8_on:
                assert adj.main_kernel(k).startswith("gcn::spmm_group8_kernel<"), adj.main_kernel(k)
        assert sampled_rows_oracle_err(rowptr, col, val, B, C, rows)[0] <= TOL, k
        assert torch.equal(C, adj.matmul_raw(B))
    for k in (44, 48, 36, 41, 47):                                           # 33..48: the plan's own slices, rows of 48 floats
        B = graphgen.random_features(n, k, seed=10 + k, device=d)
        C = adj.matmul_raw(B)
        assert adj.narrow_slices_for(k) == 0 and (not g12_on or adj.main_kernel(k) == "gcn::spmm_group12_kernel"), (k, adj.main_kernel(k))
        assert sampled_rows_oracle_err(rowptr, col, val, B, C, rows)[0] <= TOL, k
        assert torch.equal(C, adj.matmul_raw(B))
    assert adj.narrow_slices_for(52) == 0 and adj.narrow_slices_for(16) == expect
    bias = torch.randn(16, device=d)
    B = graphgen.random_features(n, 16, seed=77, device=d)
    Ce = adj.matmul_raw(B, bias=bias, relu=True)
    assert float((Ce - torch.relu(adj.matmul_raw(B) + bias)).abs().max()) <= 1e-5 * float(Ce.abs().max())
    assert torch.equal(C128, adj.matmul_raw(B128)) and adj.num_slices == S    # the wide path did not move
    assert sampled_rows_oracle_err(rowptr, col, val, B128, C128, rows)[0] <= TOL
    if scale == 0.5:
        fixed = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True, slices=8)
        fixed.matmul_raw(graphgen.random_features(n, 16, seed=5, device=d))
        assert fixed.num_slices == 8 and fixed.narrow_slices == 0
