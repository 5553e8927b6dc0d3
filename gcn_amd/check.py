"""Sampled-row fp64 verification of an SpMM result, on the device, with plain torch arithmetic (index_select,
double-precision multiply-add) — independent of libgcnspmm.  bench.py runs it on the very output it timed
("verify output vs fp64 golden on a sampled row set"); the tests use the C oracle instead."""
import torch


def sampled_rows_rel_err(rowptr, col, val, B, C, rows, batch_nnz=1 << 20):
    """max |C[r, :] − Σ_e val[e]·B[col[e], :]| over the sampled rows r, divided by the largest reference
    magnitude among them; everything in fp64.  rowptr/col/val: CSR on the device of B and C (col indexes
    the rows of B), rows: 1-D int64 tensor of row ids.  → (rel_err, rows_checked)"""
    dev = C.device
    rows = rows.to(device=dev, dtype=torch.int64)
    rp = rowptr.to(torch.int64)
    start, end = rp[rows], rp[rows + 1]
    lens = end - start
    err = torch.zeros((), dtype=torch.float64, device=dev)
    ref_max = torch.zeros((), dtype=torch.float64, device=dev)
    i, nrows = 0, int(rows.numel())
    csum = torch.cumsum(lens, 0)
    while i < nrows:
        # as many rows as hold ~batch_nnz entries
        base = int(csum[i - 1]) if i > 0 else 0
        j = int(torch.searchsorted(csum, torch.tensor(base + batch_nnz, device=dev))) + 1
        j = min(max(j, i + 1), nrows)
        l = lens[i:j]
        seg = torch.repeat_interleave(torch.arange(j - i, device=dev), l)
        first = torch.cumsum(l, 0) - l
        e = start[i:j][seg] + (torch.arange(int(l.sum()), device=dev) - first[seg])
        prod = val[e].double()[:, None] * B.index_select(0, col[e].long()).double()
        ref = torch.zeros((j - i, C.shape[1]), dtype=torch.float64, device=dev).index_add_(0, seg, prod)
        got = C.index_select(0, rows[i:j]).double()
        err = torch.maximum(err, (got - ref).abs().max())
        ref_max = torch.maximum(ref_max, ref.abs().max())
        i = j
    return float(err / torch.clamp(ref_max, min=1e-300)), nrows
