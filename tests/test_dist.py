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


def test_only_the_values_do_not_factor_status_is_hidden_when_a_shard_builds_its_operator():
    """RowShardedAdjacency hands its value factors to the local operator; "they do not factor" (GCN_ERR_NOT_FACTORED)
    means keep the value stream — every other failure of that call must surface (dist.py swallowed all)"""
    from gcn_amd import _lib

    def local_raising(status, exc=None):
        class _Local(_OracleLocal):
            def set_value_factors(self, u_row, u_col):
                if exc is not None:
                    raise exc
                err = _lib.GcnAmdError(f"status {status}")
                err.status = status
                raise err
        return _Local

    rp, ci, va = sym_norm_graph(300, 2000, seed=5)
    u = torch.ones(300)
    args = (torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(va), 300, 0, 1)
    shard = RowShardedAdjacency(*args, local_raising(_lib.ERR_NOT_FACTORED), value_factor=u)
    assert shard._factors is None                                   # hidden: the shard works on its value stream
    with pytest.raises(_lib.GcnAmdError):
        RowShardedAdjacency(*args, local_raising(2), value_factor=u)             # a HIP error is not "does not factor"
    with pytest.raises(ZeroDivisionError):
        RowShardedAdjacency(*args, local_raising(0, ZeroDivisionError()), value_factor=u)


class _OraclePrelaidLocal(_OracleLocal):
    """CPU stand-in for a CsrAdjacency that offers the pre-laid feature layout (prelaid_layout / matmul_prelaid /
    set_value_factors with the semantics of include/gcn_spmm.h), the arithmetic done by the oracle."""

    def __init__(self, rowptr, col, val, shape, slices="auto"):
        super().__init__(rowptr, col, val, shape)
        self.S = None if slices == "auto" else int(slices)
        self.u_row = self.u_col = None

    def set_value_factors(self, u_row, u_col):
        rows = np.repeat(np.arange(self.shape[0]), np.diff(self.rp))
        assert np.allclose(u_row.numpy()[rows] * u_col.numpy()[self.ci], self.va, rtol=1e-6)
        self.u_row, self.u_col = u_row.numpy(), u_col.numpy()

    def prelaid_layout(self, k):
        if self.S is None or self.u_row is None:
            return None
        w = -(-self.shape[1] // self.S)
        return dict(slices=self.S, slice_cols=w, table_rows=self.S * (w + 1), ld=k)

    def matmul_prelaid(self, Bp, out, out_scale=None, out_gap=0):
        lay = self.prelaid_layout(out.shape[1])
        w = lay["slice_cols"]
        assert tuple(Bp.shape) == (lay["table_rows"], out.shape[1])
        assert float(Bp[w::w + 1].abs().max()) == 0.0                           # the zero row behind every slice
        c = np.arange(self.shape[1])
        ones = np.ones_like(self.va)
        res = self.u_row[:, None] * oracle_spmm(self.rp, self.ci, ones, Bp.numpy()[c + c // w], fp64=False)
        if out_scale is not None:
            res = res * out_scale.numpy()[:, None]
        r = np.arange(self.shape[0])
        out[torch.from_numpy(r + r // out_gap if out_gap else r)] = torch.from_numpy(res.astype(np.float32))
        return out


def _prelaid_worker(rank, world, port, q, exchange):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, k = 17000, 16                                     # 64-column table 4.35 MB > one L2: the plan would slice
        rp, ci, va = sym_norm_graph(n, 1200000, seed=12)
        deg = np.diff(rp).astype(np.float64)
        u = torch.from_numpy(np.sqrt((deg ** -0.5 * deg ** -0.5).astype(np.float32)))   # as graphgen.value_factor_from_degrees
        H = torch.from_numpy(np.random.default_rng(1).standard_normal((n, k)).astype(np.float32))
        shard = RowShardedAdjacency(torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(va), n, rank, world,
                                    _OraclePrelaidLocal, value_factor=u, exchange=exchange, plane_cols=8)
        qq, w = shard.slices_per_rank, shard.slice_cols
        ok = shard.prelaid and qq >= 1 and shard.max_rows == qq * w and shard.slot_rows == qq * (w + 1)
        ok = ok and shard.local.S == world * qq                               # the plan is told exactly the slots' fractions
        pipe = PipelinedAggregation(shard, k, "cpu", plane_cols=8)
        pipe.load(H)
        ok = ok and np.allclose(pipe.result().numpy(), H.numpy(), rtol=1e-6, atol=1e-7)    # B' -> H round trip
        for _ in range(2):
            pipe.step()
        ref = oracle_spmm(rp, ci, va, oracle_spmm(rp, ci, va, H.numpy()))
        got = pipe.result().numpy()
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        lo, hi = shard.row_lo, shard.row_hi
        ok = ok and err <= 1e-5 and np.array_equal(pipe.local_rows().numpy(), got[lo:hi])
        for b in pipe.src:                                                    # zero rows survived two layers + exchanges
            ok = ok and float(b[w::w + 1].abs().max()) == 0.0
        q.put((rank, bool(ok), err))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["all_gather", "direct"])
def test_prelaid_chain_exchanges_the_next_layers_input_directly(exchange):
    """world-2 gloo: slots of whole column slices, the exchange buffer IS the next layer's pre-laid input
    (diag(u)·H with a zero row behind every slice, written by the local SpMM's epilogue): two layers equal Â²H"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prelaid_worker, args=(r, 2, port, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


class _RefusingPrelaidLocal(_OraclePrelaidLocal):
    """an operator that does NOT offer the pre-laid layout (a rank whose block falls on the other side of a rule)"""

    def prelaid_layout(self, k):
        return None


def _subgroup_worker(rank, nproc, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=nproc)
    try:
        members = [0, 2]
        sub = dist.new_group(members)                        # (every process of the job calls new_group)
        if rank not in members:
            q.put((rank, True, "bystander"))
            return
        world, me = len(members), members.index(rank)
        n, k = 17000, 8
        rp, ci, va = sym_norm_graph(n, 1200000, seed=12)
        deg = np.diff(rp).astype(np.float64)
        u = torch.from_numpy(np.sqrt((deg ** -0.5 * deg ** -0.5).astype(np.float32)))
        H = torch.from_numpy(np.random.default_rng(1).standard_normal((n, k)).astype(np.float32))
        args = (torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(va), n, me, world)
        # without the group the shard cannot find its peers: a 2-rank shard in a 3-rank job is refused, not mis-agreed
        try:
            RowShardedAdjacency(*args, _OraclePrelaidLocal, value_factor=u, plane_cols=8)
            refused = False
        except ValueError:
            refused = True
        # rank `members[1]`'s operator refuses the layout: BOTH ranks must fall back to the copying path together
        local_cls = _RefusingPrelaidLocal if me == 1 else _OraclePrelaidLocal
        shard = RowShardedAdjacency(*args, local_cls, value_factor=u, plane_cols=8, group=sub)
        ok = refused and not shard.prelaid and shard.local.S is None      # downgraded, operator rebuilt without forced slices
        a, b = shard.to_padded(H), shard.new_buffer(k, "cpu")
        shard.layer(a, b)                                    # runs on the shard's own group
        ref = oracle_spmm(rp, ci, va, H.numpy(), fp64=False)
        ok = ok and np.array_equal(shard.from_padded(b).numpy(), ref)
        q.put((rank, bool(ok), "member"))
    finally:
        dist.destroy_process_group()


def test_shard_on_a_subgroup_agrees_on_that_group_and_downgrades_together():
    """the pre-laid agreement runs on the group the shard is given (not the default group), a shard whose
    world differs from its group's size is refused, and a downgrade rebuilds the local operator"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
