"""Randomised parity sweep on the GPU: many small CSR shapes (empty rows in runs, rows ending on
and around chunk / step boundaries, hub rows, non-square, k from 1 to 300) x chunk sizes x column
slices x epilogue, each against the fp64 oracle.  Seeds are fixed, so a failure is reproducible."""
import numpy as np
import pytest
import torch

import gcn_amd
from util import oracle_spmm, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _case(seed):
    rng = np.random.default_rng(seed)
    m = int(rng.integers(1, 1500))
    n = int(rng.integers(1, 1500))
    style = seed % 4
    if style == 0:      # short rows, many empty
        lens = rng.integers(0, 4, m)
    elif style == 1:    # lengths clustered around the step / chunk sizes
        lens = rng.choice([0, 1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129], m)
    elif style == 2:    # a few hub rows among short ones
        lens = rng.integers(0, 6, m)
        lens[rng.integers(0, m, 3)] = rng.integers(200, 1400, 3)
    else:               # dense-ish
        lens = rng.integers(20, 90, m)
    lens = np.minimum(lens, n)
    lens[rng.random(m) < 0.15] = 0
    rowptr = np.zeros(m + 1, np.int32); rowptr[1:] = np.cumsum(lens)
    col = np.concatenate([np.sort(rng.choice(n, L, replace=False)) for L in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    val = rng.standard_normal(len(col)).astype(np.float32)
    k = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 13, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 200, 300]))
    chunk = int(rng.choice([0, 64, 128, 256]))
    slices = int(rng.choice([0, 0, 2, 3, 8]))
    epi = bool(rng.integers(0, 2))
    panels = int(rng.choice([0, 0, 0, 1]))
    return m, n, rowptr, col, val, k, chunk, slices, epi, panels, rng


@pytest.mark.parametrize("seed", range(80))
def test_random_shape(seed):
    m, n, rowptr, col, val, k, chunk, slices, epi, panels, rng = _case(seed)
    d = torch.device("cuda:0")
    B = rng.standard_normal((n, k)).astype(np.float32)
    adj = gcn_amd.CsrAdjacency(torch.from_numpy(rowptr).to(d), torch.from_numpy(col).to(d),
                               torch.from_numpy(val).to(d), (m, n), chunk_nnz=chunk, slices=slices, panels=panels)
    ref = oracle_spmm(rowptr, col, val, B)
    if epi:
        bias = rng.standard_normal(k).astype(np.float32)
        C = adj.matmul_raw(torch.from_numpy(B).to(d), bias=torch.from_numpy(bias).to(d), relu=True).cpu().numpy()
        ref = np.maximum(ref + bias, 0)
    else:
        C = adj.matmul_raw(torch.from_numpy(B).to(d)).cpu().numpy()
    assert C.shape == ref.shape
    assert rel_err(C, ref) <= TOL, (seed, m, n, k, chunk, slices, epi, panels)
