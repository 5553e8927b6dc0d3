"""The C-ABI library loads and exports every symbol include/gcn_spmm.h declares; host-only
entry points validate their arguments.  (No compute calls here: those need a GPU.)"""
import ctypes
import os
import re

import numpy as np
import pytest

import gcn_amd
from gcn_amd import _lib
from util import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gcn_spmm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text)
    skip = {"defined", "C"}
    return sorted({n for n in names if n not in skip and not n.isupper()})


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(gcn_amd.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/gcn_spmm.h but not exported"
    # and the Python binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == declared


@pytest.mark.parametrize("name", ["flexspmm.so", "cuspmm.so", "tile.so", "permutate.so", "renumber.so"])
def test_dropin_objects_carry_the_reference_symbols(name):
    """gcn6.py:21-25 loads these five file names; each must resolve the symbols gcn6 calls."""
    lib = ctypes.CDLL(os.path.join(gcn_amd.DROPIN_DIR, name))
    wanted = {"flexspmm.so": ["flexspmm"], "cuspmm.so": ["cuspmm"], "tile.so": ["csr2tile"],
              "permutate.so": ["permutate"], "renumber.so": ["dfs", "gorder", "rabbit", "perm_apply"]}[name]
    for sym in wanted:
        assert hasattr(lib, sym)


def test_version_and_status_strings():
    lib = gcn_amd.load_library()
    assert lib.gcn_version().decode() == gcn_amd.__version__
    assert lib.gcn_status_string(0) == b"ok"
    assert b"invalid" in lib.gcn_status_string(1)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(gcn_amd.GcnAmdError):
        _lib.load(str(tmp_path / "nope.so"))


def test_cpu_tensors_are_rejected_not_silently_computed():
    """there is no CPU fallback in the product path"""
    import torch
    rp = torch.tensor([0, 1, 2], dtype=torch.int32)
    with pytest.raises(gcn_amd.GcnAmdError):
        gcn_amd.CsrAdjacency(rp, torch.tensor([0, 1], dtype=torch.int32), torch.ones(2), (2, 2))
    with pytest.raises(gcn_amd.GcnAmdError):
        gcn_amd.gather_rows(torch.ones(2, 2), torch.tensor([0, 1]))


def test_reorder_entry_points_validate_input():
    lib = gcn_amd.load_library()
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    rp = np.array([0, 2, 1], np.int32)        # not monotone
    ci = np.array([0, 1], np.int32)
    out = np.zeros(2, np.int64)
    assert lib.gcn_order_deg(p(rp), p(ci), 2, 2, 0, 1, p(out)) == 1
    rp = np.array([0, 1, 2], np.int32)
    ci = np.array([0, 5], np.int32)           # column out of range
    assert lib.gcn_order_rcm(p(rp), p(ci), 2, 2, 1, p(out)) == 1
    ci = np.array([0, 1], np.int32)
    assert lib.gcn_order_gorder(p(rp), p(ci), 2, 2, 0, p(out)) == 1     # window < 1
    va = np.ones(2, np.float32)
    bad_rank = np.array([0, 0], np.int64)     # not a bijection
    assert lib.gcn_csr_apply_rank(p(rp), p(ci), p(va), 2, 2, p(bad_rank), None) == 1


def test_csr2tile_packing_respects_caller_capacities():
    """host-only drop-in: seg_rowPtr nnz ints, segNzCV 2*nnz floats, segVoMap nnz ints,
    grouped_tailSeg / next_seg exactly 256 ints (gcn6.py:334-339; reference defect D2 wrote 257)"""
    import torch
    from gcn_amd import dropin
    from util import sym_norm_graph
    n = 500
    rowptr, col, val = sym_norm_graph(n, 3000, seed=1)
    nnz = len(col)
    guard = 7
    tail = torch.full((256 + guard,), -77, dtype=torch.int32)
    nxt = torch.full((256 + guard,), -77, dtype=torch.int32)
    seg_rowPtr = torch.full((nnz + guard,), -77, dtype=torch.int32)
    segVoMap = torch.full((nnz + guard,), -77, dtype=torch.int32)
    segNzCV = torch.full((2 * nnz + guard,), -77.0)
    n_segs = torch.zeros(1, dtype=torch.int32)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    gcn_amd.load_library().csr2tile(vp(torch.from_numpy(rowptr)), vp(torch.from_numpy(col)), vp(torch.from_numpy(val)),
                                    n, n, nnz, vp(torch.arange(n, dtype=torch.int32)), vp(segVoMap), vp(seg_rowPtr),
                                    vp(segNzCV), vp(tail), vp(nxt), 8, vp(n_segs))
    ns = int(n_segs[0])
    assert ns == nnz // 9 and 9 * ns >= n + 1
    for t, cap in ((tail, 256), (nxt, 256), (seg_rowPtr, nnz), (segVoMap, nnz), (segNzCV, 2 * nnz)):
        assert bool((t[cap:] == -77).all()), "wrote past the caller's buffer"
    assert np.array_equal(seg_rowPtr[: n + 1].numpy(), rowptr)
    assert np.array_equal(segNzCV[:nnz].numpy().view(np.int32), col)        # exact int32 columns
    assert np.array_equal(segNzCV[nnz:2 * nnz].numpy(), val)


def test_csr2tile_slice_major_packing_is_a_permutation_of_the_matrix():
    """host-only: for a dense graph csr2tile packs S*m virtual rows (row r's entries split by
    column slice, slice-major); unpacking them gives back exactly the input matrix"""
    import scipy.sparse as sp
    import torch
    from gcn_amd import dropin
    from util import sym_norm_graph
    n = 17000                                # 64-column table 4.35 MB > one 4 MiB L2 -> 2 slices (auto_slices)
    rowptr, col, val = sym_norm_graph(n, 1200000, seed=2)
    nnz = len(col)
    assert nnz // n >= 128
    rng = np.random.default_rng(0)           # hand the rows over UNSORTED: csr2tile sorts them itself
    col_u, val_u = col.copy(), val.copy()
    for r in range(n):
        p = rng.permutation(rowptr[r + 1] - rowptr[r]) + rowptr[r]
        col_u[rowptr[r]:rowptr[r + 1]], val_u[rowptr[r]:rowptr[r + 1]] = col[p], val[p]
    seg_rowPtr, segNzCV, segVoMap, tail, nxt, n_segs = dropin.csr2tile(
        torch.from_numpy(rowptr.copy()), torch.from_numpy(col_u), torch.from_numpy(val_u), n, n, nnz,
        torch.arange(n, dtype=torch.int32))
    S, w = 2, (n + 1) // 2
    vrp = seg_rowPtr.numpy()[: S * n + 1]
    assert vrp[0] == 0 and vrp[-1] == nnz and np.all(np.diff(vrp) >= 0)
    vcol = segNzCV[:nnz].numpy().view(np.int32)
    vval = segNzCV[nnz:2 * nnz].numpy()
    rows = np.repeat(np.arange(S * n), np.diff(vrp))
    assert np.array_equal(vcol // w, rows // n)                       # every entry sits in its slice
    A = sp.csr_matrix((val, col, rowptr), shape=(n, n))
    Bm = sp.coo_matrix((vval, (rows % n, vcol)), shape=(n, n)).tocsr()
    Bm.sort_indices()
    assert np.array_equal(Bm.indptr, A.indptr) and np.array_equal(Bm.indices, A.indices)
    assert np.array_equal(Bm.data, A.data)
    # chunk_row of the virtual CSR: monotone over the used prefix, within range
    cr = segVoMap.numpy()
    lead = np.trim_zeros(cr, "b")
    assert len(lead) >= 1 and np.all(np.diff(lead) >= 0) and cr.max() < S * n
