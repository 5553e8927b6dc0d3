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
    """gcn6.py:21-25 loads these five file names; each must resolve the symbols gcn6 calls (and the one other `extern "C"`
    symbol the reference's objects export: csr2seg_Cmajor, tile.cu:11)."""
    lib = ctypes.CDLL(os.path.join(gcn_amd.DROPIN_DIR, name))
    wanted = {"flexspmm.so": ["flexspmm"], "cuspmm.so": ["cuspmm"], "tile.so": ["csr2tile", "csr2seg_Cmajor"],
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


def test_group_kernel_addressing_rule_at_the_4gib_boundary():
    """host-only: the group kernels keep (entry + slice base) * row_bytes in 32 bits only while the sliced copy of B
    stays below 4 GiB and 2^24 rows; past either bound the slice base is added in 64 bits (the BIG variants), and a
    table that needs 64 bits with rows of 128 KiB or more is not served by them at all (n = 8 M, k = 128
    and n = 2.4 M, k = 512 used to wrap silently)."""
    f = gcn_amd.load_library().gcn_spmm_group_addressing
    assert f(232965 + 15, 128) == 0                              # Reddit-shaped, k = 128: 119 MB
    rows_4g_k128 = (1 << 32) // 512                              # 8 388 608 rows of 512 bytes = exactly 4 GiB
    assert f(rows_4g_k128 - 1, 128) == 0 and f(rows_4g_k128, 128) == 1
    rows_4g_k512 = (1 << 32) // 2048                             # 2 097 152 rows of 2 KiB
    assert f(rows_4g_k512 - 1, 512) == 0 and f(rows_4g_k512, 512) == 1
    assert f(2449029 + 75, 512) == 1                             # products-shaped n at k = 512: 5 GB
    assert f((1 << 24) - 1, 16) == 0 and f(1 << 24, 16) == 1     # the 24-bit multiplier's row bound
    assert f(1 << 24, 32768) == -1 and f(1000, 32768) == 0       # 128 KiB rows: only while 32 bits reach the table
    assert f(1000, 126) == -1 and f(0, 128) == -1                # stride not a multiple of 4 floats; no table


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
    # n_segs is nnz/9, or one less so that its lowest bit says "values are u[r]*u[c]" (true for this matrix) — the
    # only scalar that travels from csr2tile to flexspmm (gcn6.py:353-366 shrinks the buffers by it)
    assert ns in (nnz // 9, nnz // 9 - 1) and ns % 2 == 1 and 9 * ns >= n + 1
    for t, cap in ((tail, 256), (nxt, 256), (seg_rowPtr, nnz), (segVoMap, nnz), (segNzCV, 2 * nnz)):
        assert bool((t[cap:] == -77).all()), "wrote past the caller's buffer"
    assert np.array_equal(seg_rowPtr[: n + 1].numpy(), rowptr)
    assert np.array_equal(segNzCV[:nnz].numpy().view(np.int32), col)        # exact int32 columns
    assert np.array_equal(segNzCV[nnz:2 * nnz].numpy(), val)


def _group_format_offsets(n, n_segs, S):
    """where the fix list and the values live (api_dropin.cpp dropin_group: pure functions of (m, n, n_segs))"""
    T = 512
    total_ub = (9 * n_segs + 17 + S * n + (S + 64) * T + 63) // (64 * T) * (64 * T) + 64 * T
    nchunks_ub = total_ub // T
    return (16 + 2 * nchunks_ub + 3) // 4 * 4, (total_ub * 2 + 15) // 16 * 4


def _unpack_group_format(seg_rowPtr, segNzCV, segVoMap, n):
    """decode what csr2tile packs for graphs that qualify for slicing (api_dropin.cpp: the group-kernel format)
    → (header dict, rows, cols, vals or None, u or None, logical stream)"""
    h = seg_rowPtr.numpy()[:16]
    hd = dict(magic=int(h[0]), S=int(h[1]), T=int(h[2]), w=int(h[3]), nchunks=int(h[4]), nfix=int(h[5]),
              value_free=int(h[6]), total=int(h[7]), nnz=int(h[8]))
    total, S, w = hd["total"], hd["S"], hd["w"]
    p = np.arange(total, dtype=np.int64)
    r = p & 63
    phys = (p & ~np.int64(63)) + (r & 15) * 4 + (r >> 4)              # lane-major runs of 64 entries
    stream = segNzCV.numpy().view(np.uint16)[:total][phys]
    ends, off = (stream >> 15).astype(np.int64), (stream & 0x7FFF).astype(np.int64)
    vrow = np.concatenate([[0], np.cumsum(ends)[:-1]])
    assert int(ends.sum()) == S * n                                    # every virtual row ends exactly once
    real = off < w                                                     # (off == w: padding entry, the all-zero row)
    rows, cols = (vrow % n)[real], ((vrow // n) * w + off)[real]
    vals = u = None
    if hd["value_free"]:
        u = segVoMap.numpy()[:n].view(np.float32)
    else:
        voff = _group_format_offsets(n, seg_rowPtr.numel() // 9, S)[1]
        vals = segNzCV.numpy()[voff: voff + total][phys][real]
        assert np.all(segNzCV.numpy()[voff: voff + total][phys][~real] == 0)      # padding weighs nothing
    return hd, rows, cols, vals, u, (vrow, ends, real)


def test_csr2tile_group_format_is_the_matrix():
    """host-only: for a graph that qualifies for slicing csr2tile packs the group kernels' format — slice-major 15-bit
    stream in lane-major runs, chunk metadata, list of cut rows; value-free when the values are u[r]*u[c], with the
    values beside the stream otherwise.  Unpacking gives back exactly the input matrix."""
    import scipy.sparse as sp
    import torch
    from gcn_amd import dropin
    from util import sym_norm_graph
    n = 17000                                # 64-column table 4.35 MB > one 4 MiB L2 -> 2 slices (auto_slices)
    rowptr, col, val = sym_norm_graph(n, 1200000, seed=2)
    nnz = len(col)
    assert nnz // n >= 128
    A = sp.csr_matrix((val, col, rowptr), shape=(n, n))
    rng = np.random.default_rng(0)           # hand the rows over UNSORTED: csr2tile sorts them itself
    col_u, val_u = col.copy(), val.copy()
    for r in range(n):
        p = rng.permutation(rowptr[r + 1] - rowptr[r]) + rowptr[r]
        col_u[rowptr[r]:rowptr[r + 1]], val_u[rowptr[r]:rowptr[r + 1]] = col[p], val[p]
    for weighted in (False, True):
        vin = val_u.copy()
        if weighted:
            vin[5] *= 1.5                    # no longer u[r]*u[c]: the values travel beside the stream
        seg_rowPtr, segNzCV, segVoMap, tail, nxt, n_segs = dropin.csr2tile(
            torch.from_numpy(rowptr.copy()), torch.from_numpy(col_u.copy()), torch.from_numpy(vin), n, n, nnz,
            torch.arange(n, dtype=torch.int32))
        hd, rows, cols, vals, u, (vrow, ends, real) = _unpack_group_format(seg_rowPtr, segNzCV, segVoMap, n)
        assert hd["magic"] == 0x47434E47 and hd["S"] == 2 and hd["T"] == 512 and hd["w"] == (n + 1) // 2
        assert hd["nnz"] == nnz and hd["total"] == hd["nchunks"] * 512 and hd["nchunks"] % 64 == 0
        assert hd["value_free"] == (0 if weighted else 1)
        Bm = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n)).tocsr()
        Bm.sort_indices()
        assert np.array_equal(Bm.indptr, A.indptr) and np.array_equal(Bm.indices, A.indices)   # the same pattern
        if weighted:
            got = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr(); got.sort_indices()
            ref = sp.csr_matrix((vin, col_u, rowptr), shape=(n, n)); ref.sort_indices()
            assert np.array_equal(got.data, ref.data)                                          # the values, exactly
        else:
            assert np.allclose(u[rows] * u[cols], np.asarray(A[rows, cols]).ravel(), rtol=1e-6, atol=0)
        # chunk metadata: chunk c starts inside virtual row vrow[c*T]; flag = the row began in an earlier chunk
        meta = seg_rowPtr.numpy()[16: 16 + 2 * hd["nchunks"]].reshape(-1, 2)
        starts = np.arange(hd["nchunks"], dtype=np.int64) * 512
        assert np.array_equal(meta[:, 0] >> 1, vrow[starts])
        began = np.concatenate([[0], (ends[starts[1:] - 1] == 0).astype(np.int64)])
        assert np.array_equal(meta[:, 0] & 1, began)
        assert np.array_equal(meta[:, 1], (vrow[starts] // n) * (hd["w"] + 1))
        # fix list: exactly the rows that begin in chunk c-1 and run on into chunk c
        foff = _group_format_offsets(n, int(n_segs[0]), hd["S"])[0]
        assert int(n_segs[0]) % 2 == hd["value_free"]                         # the flag flexspmm reads on the host
        fix = seg_rowPtr.numpy()[foff: foff + 4 * hd["nfix"]].reshape(-1, 4)
        cut = np.flatnonzero(began[1:] == 1) + 1
        first_of_row = np.concatenate([[0], np.flatnonzero(ends[:-1] == 1) + 1])              # start entry of every vrow
        cut = cut[first_of_row[vrow[starts[cut]]] // 512 == cut - 1]
        assert hd["nfix"] == len(cut) and np.array_equal(np.sort(fix[:, 1]), cut)
        assert np.array_equal(fix[np.argsort(fix[:, 1]), 0], vrow[starts[cut]])


def test_csr2seg_cmajor_resolves_prints_and_touches_nothing(capfd):
    """tile.cu:11-12 exports it beside csr2tile; no call site binds it: ours says so on stderr and returns"""
    lib = gcn_amd.load_library()
    rp = np.array([0, 2, 3], np.int32); ci = np.array([0, 1, 1], np.int32); va = np.ones(3, np.float32)
    bufs = [np.full(16, 7, np.int32) for _ in range(3)] + [np.full(16, 7.0, np.float32)]
    nseg = np.array([-5], np.int32)
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)
    lib.csr2seg_Cmajor(0, vp(rp), vp(ci), vp(va), 2, 2, 3, vp(bufs[0]), vp(bufs[1]), vp(bufs[2]), vp(bufs[3]), 8, vp(nseg))
    assert "csr2seg_Cmajor" in capfd.readouterr().err
    assert nseg[0] == -5 and all(np.all(b == 7) for b in bufs)
