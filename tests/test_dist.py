"""The N>1 path (1-D row partition + all-gather per layer) on CPU: world_size-2/3 gloo
processes.  The product has no CPU compute, so the local row-block SpMM is injected from the
oracle here (tests may use the oracle); what is under test is the partition, the padded
column remap and the in-place all-gather of gcn_amd/dist.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gcn_amd.dist import PipelinedAggregation, RowShardedAdjacency, partition_rows
from util import oracle_spmm, sym_norm_graph


class _OracleLocal:
    """CPU stand-in for gcn_amd.CsrAdjacency with the same matmul_raw(out=) contract."""

    def __init__(self, rowptr, col, val, shape):
        self.rp, self.ci, self.va, self.shape = rowptr.numpy(), col.numpy(), val.numpy(), shape

    def matmul_raw(self, dense, out=None):
        assert dense.shape[0] == self.shape[1]
        C = torch.from_numpy(oracle_spmm(self.rp, self.ci, self.va, dense.numpy(), fp64=False))
        if out is None:
            return C
        out.copy_(C)
        return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, balance, q, exchange="all_gather", from_block=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, k = 700, 24
        rp, ci, va = sym_norm_graph(n, 6000, seed=3)
        H = torch.from_numpy(np.random.default_rng(1).standard_normal((n, k)).astype(np.float32))
        if from_block:
            # the rank is handed ITS rows only: local row pointer, GLOBAL int64 column ids, values, all ranks' bounds
            bounds = partition_rows(rp, world, balance)
            lo, hi = int(bounds[rank]), int(bounds[rank + 1])
            e0, e1 = int(rp[lo]), int(rp[hi])
            shard = RowShardedAdjacency.from_row_block(
                torch.from_numpy((rp[lo:hi + 1] - e0).astype(np.int32)), torch.from_numpy(ci[e0:e1].astype(np.int64)),
                torch.from_numpy(va[e0:e1]), bounds, rank, world, _OracleLocal, total_nnz=len(ci), exchange=exchange)
            whole = RowShardedAdjacency(torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(va), n,
                                        rank, world, _OracleLocal, balance=balance)
            for x, y in zip(shard._local_args[:3], whole._local_args[:3]):      # the same block, bit for bit
                assert torch.equal(x, y)
            assert shard._local_args[3] == whole._local_args[3] and shard.max_rows == whole.max_rows
        else:
            shard = RowShardedAdjacency(torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(va), n,
                                        rank, world, _OracleLocal, balance=balance, exchange=exchange)
        assert ("isend" in shard.collective_form()) == (exchange == "direct")
        a, b = shard.to_padded(H), shard.new_buffer(k, "cpu")
        shard.layer(a, b)            # layer 1
        shard.layer(b, a)            # layer 2 consumes the all-gathered output of layer 1
        got1, got2 = shard.from_padded(b).numpy(), shard.from_padded(a).numpy()
        ref1 = oracle_spmm(rp, ci, va, H.numpy(), fp64=False)
        ref2 = oracle_spmm(rp, ci, va, ref1, fp64=False)
        # a row partition keeps every row's summation order → bit-identical to the unsharded run
        ok = np.array_equal(got1, ref1) and np.array_equal(got2, ref2)
        # the pipelined form (column planes, all-gathers left in flight) over three layers
        pipe = PipelinedAggregation(shard, k, "cpu", plane_cols=16)
        pipe.load(H)
        for _ in range(3):
            pipe.step()
        ref3 = oracle_spmm(rp, ci, va, ref2, fp64=False)
        ok = ok and len(pipe.widths) == 2 and np.array_equal(pipe.result().numpy(), ref3)
        q.put((rank, bool(ok), int(shard.local_nnz), int(shard.rows)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,balance,exchange,from_block",
                         [(2, "nnz", "all_gather", False), (3, "nnz", "all_gather", False), (2, "rows", "all_gather", False),
                          (2, "nnz", "direct", False), (3, "nnz", "direct", True), (2, "nnz", "all_gather", True)])
def test_row_sharded_layers_match_single_process(world, balance, exchange, from_block):
    """whole-graph and own-block constructors, all-gather and direct (grouped send/recv) exchange: every
    combination reproduces the unsharded layers bit for bit"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, balance, q, exchange, from_block)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert sum(nz for _, _, nz, _ in res) == len(sym_norm_graph(700, 6000, seed=3)[1])
    assert sum(rows for _, _, _, rows in res) == 700


def test_partition_rows_balances_nnz_and_covers_all_rows():
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 50, 1000); lens[10] = 5000; lens[500:520] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    for world in (1, 2, 4, 8):
        b = partition_rows(rowptr, world)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0) and len(b) == world + 1
        per = np.diff(rowptr[b])
        assert per.sum() == rowptr[-1]
        assert per.max() <= rowptr[-1] / world + 5000          # within one (hub) row of the ideal
    assert np.array_equal(partition_rows(rowptr, 4, "rows"), [0, 250, 500, 750, 1000])


def test_from_row_block_remaps_global_int64_ids_and_refuses_what_cannot_fit():
    """the rank gets GLOBAL int64 column ids and remaps them itself to the padded int32 numbering
    (owner * max_rows + index inside the owner's block) — nothing whole-graph is needed; a padded space or a
    block that cannot fit int32 is refused up front"""
    world, rank = 4, 2
    bounds = np.array([0, 5, 12, 15, 30], dtype=np.int64)        # unequal blocks: max_rows = 15
    lrp = torch.tensor([0, 2, 2, 5], dtype=torch.int32)          # this rank's 3 rows
    gcol = torch.tensor([4, 29, 5, 11, 14], dtype=torch.int64)
    seen = {}

    def make_local(rp, ci, va, shape):
        seen["args"] = (rp, ci, va, shape)
        return object()

    shard = RowShardedAdjacency.from_row_block(lrp, gcol, torch.ones(5), bounds, rank, world, make_local)
    rp, ci, va, shape = seen["args"]
    assert shape == (3, 4 * 15) and ci.dtype == torch.int32 and shard.n == 30 and shard.rows == 3
    assert ci.tolist() == [0 * 15 + 4, 3 * 15 + 14, 1 * 15 + 0, 1 * 15 + 6, 2 * 15 + 2]
    H = torch.arange(30, dtype=torch.float32)[:, None]
    assert torch.equal(shard.from_padded(shard.to_padded(H)), H)
    assert shard.to_padded(H)[3 * 15 + 14, 0] == 29 and shard.to_padded(H)[5, 0] == 0   # (padding rows stay zero)
    with pytest.raises(ValueError):                               # world * max_rows must stay below 2^31
        RowShardedAdjacency.from_row_block(lrp, gcol, torch.ones(5), np.array([0, 3, 2 ** 30, 2 ** 30 + 1, 2 ** 30 + 2]),
                                           0, world, make_local)
    with pytest.raises(ValueError):
        RowShardedAdjacency.from_row_block(lrp, gcol, torch.ones(5), bounds, rank, world, make_local, exchange="ring")
