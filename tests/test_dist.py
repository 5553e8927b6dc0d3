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


def _worker(rank, world, port, balance, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, k = 700, 24
        rp, ci, va = sym_norm_graph(n, 6000, seed=3)
        H = torch.from_numpy(np.random.default_rng(1).standard_normal((n, k)).astype(np.float32))
        shard = RowShardedAdjacency(torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(va), n,
                                    rank, world, _OracleLocal, balance=balance)
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


@pytest.mark.parametrize("world,balance", [(2, "nnz"), (3, "nnz"), (2, "rows")])
def test_row_sharded_layers_match_single_process(world, balance):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, balance, q)) for r in range(world)]
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
