"""BASELINE configs 4 and 5 at FULL SIZE through the product path, checked by size-independent properties:
sampled rows against the fp64 C oracle, and for reorderings P·Â·Pᵀ·(P·B) = P·(Â·B) against the un-reordered run.

  config 4: ogbn-papers100M-shaped graph (n = 111 059 956, 1.616 G directed R-MAT samples, ≈ 3.3 G non-zeros
            in the whole graph — past int32), feat = 128, row-partitioned 8 ways: ONE GPU holds rank 0's block, built
            by RowShardedAdjacency.from_row_block from that block alone (global int64-safe ids remapped on the
            device); the exchange itself is covered by the gloo tests and the 2-rank rehearsal.
  config 5: R-MAT scale 24 (n = 16 777 216, ≈ 538 M non-zeros), feat = 512: no reorder / degree / RCM at full
            size with the device reorderers; the Gorder leg (inherently serial host algorithm, `gorder` of
            renumber.cu:157-230) at the largest scale whose host time fits a test: scale 20, stated below.
"""
import time

import numpy as np
import pytest
import torch

import gcn_amd
from gcn_amd import graphgen
from gcn_amd.dist import RowShardedAdjacency
from util import sampled_rows_oracle_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev():
    return torch.device("cuda:0")


def test_config4_papers100m_rank_share_built_from_its_own_block_full_size():
    d = _dev()
    n, samples, world, rank, k = 111059956, 1615685872, 8, 0, 128
    lrp, lcol, lval, n, lo, hi, deg = graphgen.make_rmat_row_block(n, samples, world, rank, device=d, seed=4)
    total_nnz = int(deg.sum())
    assert total_nnz > 2 ** 31 and lo == 0 and hi == (n + world - 1) // world       # the whole graph is past int32
    rows_per = (n + world - 1) // world
    bounds = [min(n, p * rows_per) for p in range(world + 1)]
    u = graphgen.value_factor_from_degrees(deg)
    del deg
    shard = RowShardedAdjacency.from_row_block(
        lrp, lcol, lval, bounds, rank, world, lambda rp, ci, va, shape: gcn_amd.CsrAdjacency(rp, ci, va, shape),
        value_factor=u, total_nnz=total_nnz)
    del lcol, u
    shard.collective = False                                     # one GPU: this rank's block only
    assert shard.rows == hi - lo and shard.local_nnz == int(lval.numel()) and 3.5e8 < shard.local_nnz < 2 ** 31
    g = torch.Generator(device=d)
    g.manual_seed(11)
    H = torch.randn((world * shard.max_rows, k), generator=g, device=d)        # the gathered layer input, padded layout (57 GB)
    out = torch.zeros((shard.max_rows, k), device=d)
    shard.local.matmul_raw(H, out=out[: shard.rows])             # what layer() runs for this rank
    rows = np.sort(np.random.default_rng(4).choice(shard.rows, 2048, replace=False))
    la = shard._local_args                                       # (local rowptr, padded int32 columns, values)
    err, entries = sampled_rows_oracle_err(la[0], la[1], la[2], H, out, rows)
    assert entries > 20000 and err <= TOL, err
    # the padded numbering is the global one shifted block by block: spot-check the remap against the ids it came from
    pc = la[1][:1000].long()
    owner = pc // shard.max_rows
    assert torch.all(pc - owner * shard.max_rows + torch.tensor(bounds, device=d)[owner] < n)


def test_config5_rmat24_k512_orderings_full_size_and_gorder_at_scale_20():
    d = _dev()
    rowptr, col, val, n = graphgen.make_rmat(24, device=d, seed=5)
    nnz, k = int(col.numel()), 512
    assert n == 1 << 24 and 5.0e8 < nnz < 5.6e8
    B = graphgen.random_features(n, k, seed=2, device=d)
    base_adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True)
    base = base_adj.matmul_raw(B)
    rows = np.sort(np.random.default_rng(5).choice(n, 1024, replace=False))
    err, entries = sampled_rows_oracle_err(rowptr, col, val, B, base, rows)
    assert entries > 10000 and err <= TOL, err
    scale = float(base.abs().max())
    for name in ("deg", "rcm"):
        rank = (gcn_amd.reorder.order_deg_device(rowptr, col) if name == "deg"
                else gcn_amd.reorder.order_rcm_device(rowptr, col))
        rp2, ci2, va2, vomp = gcn_amd.reorder.apply_rank_device(rowptr, col, val, rank)
        assert int(rp2[-1]) == nnz and torch.equal(torch.sort(vomp.long()).values, torch.arange(n, device=d))
        adj = gcn_amd.CsrAdjacency(rp2, ci2, va2, (n, n), symmetric=True)
        C = adj.matmul_raw(gcn_amd.gather_rows(B, vomp))         # (P Â Pᵀ)(P B)
        # = P (Â B): compared on a large sample of rows (the full difference would need two more 34 GB buffers)
        pick = torch.from_numpy(np.random.default_rng(6).choice(n, 200000, replace=False)).to(d)
        assert float((C[pick] - base[vomp.long()[pick]]).abs().max()) <= TOL * scale, name
        err, _ = sampled_rows_oracle_err(rp2, ci2, va2, gcn_amd.gather_rows(B, vomp), C, rows)
        assert err <= TOL, (name, err)
        del adj, C, rp2, ci2, va2, vomp
    del base, B, base_adj
    torch.cuda.empty_cache()
    # Gorder (window 3, through RCM, as the reference's C ABI runs it): serial on the host, so at scale 20
    rowptr, col, val, n = graphgen.make_rmat(20, device=d, seed=5)
    k = 512
    B = graphgen.random_features(n, k, seed=2, device=d)
    base = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True).matmul_raw(B)
    rp, ci, va = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    t0 = time.time()
    rank = gcn_amd.reorder.order_gorder(rp, ci, window=3)
    host_s = time.time() - t0
    print(f"gorder host time at scale 20 (n={n}, nnz={len(ci)}): {host_s:.1f} s")
    rp2, ci2, va2, vomp = gcn_amd.reorder.apply_rank(rp, ci, va, rank)
    adj = gcn_amd.CsrAdjacency(torch.from_numpy(rp2).to(d), torch.from_numpy(ci2).to(d), torch.from_numpy(va2).to(d),
                               (n, n), symmetric=True)
    vomp_d = torch.from_numpy(vomp).to(d)
    C = adj.matmul_raw(gcn_amd.gather_rows(B, vomp_d))
    assert float((C - base[vomp_d.long()]).abs().max() / base.abs().max()) <= TOL
