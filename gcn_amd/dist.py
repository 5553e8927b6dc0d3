"""1-D row partition of Â across the GPUs of one node + all-gather of layer outputs.

The reference is single-GPU (device 0 hard-coded, flexspmm.cu:507; 32-bit `row*k` offsets, flexspmm.cu:67;
column indices stored as floats, tile.cu:67); this module is the MI355X-native extension BASELINE.json asks
for: rank p owns a contiguous row block Â[rows_p, :], computes H'_p = Â[rows_p, :] · H with the same HIP
kernel, and the next layer's input H' is assembled with ONE exchange per layer over xGMI.

A rank is built from ITS OWN block only (`from_row_block`: local row pointer, GLOBAL int64 column ids,
values, the row boundaries of all ranks) — the whole graph never has to exist on any rank, and its global
non-zero count may exceed 2³¹ (papers100M: 3.3 G); only a rank's block must fit int32.  The whole-graph
constructor is a convenience for graphs that do fit.

Zero-copy layout: shards have unequal row counts, so every shard is padded to `max_rows` and the gathered
buffer is [world * max_rows, k].  Instead of compacting that buffer after every layer, the column indices
of the local block are remapped ONCE (on the device) to the padded numbering
(col' = owner(col) * max_rows + local_index(col), int32); the local SpMM writes straight into this rank's
slot of the buffer and the exchange is done in place.  Summation order inside a row is unchanged by the
partition, so every row equals the single-GPU result to rounding.

Pre-laid chains (`prelaid`, on by default where the local operator offers it): when the values factor as
u[r]·u[c] (a normalised adjacency) the sliced kernels gather from B' = diag(u)·H laid out slice by slice with a
zero row behind every slice (gcn_spmm_plan_prelaid_layout).  A rank's slot is then sized to a whole number q of
slices (slot = q·w columns, w = ceil(max_rows / q)), the local plan is told exactly world·q slices, and the
exchange buffer IS B': the local SpMM writes its rows — already multiplied by u for the NEXT layer — into its
slot of it (gcn_spmm_csr_f32_prelaid), peers' slots arrive by the exchange, and the O(n·k) scaled copy that
every rank otherwise makes per layer and plane disappears.

Exchange forms (`exchange=`): "all_gather" — one RCCL all-gather per layer (`all_gather_into_tensor` on
nccl, the list form on gloo; chosen once from the backend, never by catching an error: a rank-local
failure must not make one rank issue a different collective than its peers); "direct" — every rank sends
its shard to each peer and receives theirs in ONE grouped batch of point-to-point operations
(`batch_isend_irecv`): on MI355X every GPU pair has its own xGMI link, so each shard crosses exactly one
link once, where a ring all-gather forwards every shard over world-1 hops; "push" (r04) — no collective
library at all on the data path: every rank maps its peers' exchange buffers ONCE (IPC handles, exchanged through the
process group's object all-gather), then per layer writes its shard straight into every peer's buffer with the runtime's
copy path (hipMemcpyAsync between devices: the SDMA engines, no compute units), raises one flag per peer behind the data,
and one wave waits for its own world-1 flags before the next layer reads the buffer (csrc/exchange.hip) — so nothing
competes for the CUs with the two oversubscribed SpMM main kernels of a layer.  All forms fill the same buffer with the
same bytes.  None has been measured on multi-GPU hardware by this build.
"""
import ctypes
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def partition_rows(rowptr, world, balance="nnz"):
    """Row boundaries [world+1] of contiguous blocks with (nearly) equal nnz (or rows)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    m = len(rowptr) - 1
    if balance == "rows":
        return np.array([(m * p) // world for p in range(world + 1)], dtype=np.int64)
    nnz = int(rowptr[-1])
    targets = np.array([(nnz * p) // world for p in range(1, world)], dtype=np.int64)
    cuts = np.searchsorted(rowptr, targets, side="left")
    bounds = np.concatenate([[0], cuts, [m]]).astype(np.int64)
    return np.maximum.accumulate(np.minimum(bounds, m))


class _Works:
    """what batch_isend_irecv returns, behind the one-handle interface of an async collective"""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


class RowShardedAdjacency:
    """This rank's row block of Â in the padded-gather numbering.

    make_local(rowptr_i32, col_i32, val_f32, shape) must return an object with
    ``matmul_raw(dense, out=...)`` — gcn_amd.CsrAdjacency on the GPU; the gloo/CPU tests
    inject an oracle-backed stand-in (the product path has no CPU compute).
    """

    def __init__(self, rowptr, col, val, n, rank, world, make_local, balance="nnz", value_factor=None,
                 exchange="all_gather", prelaid="auto", plane_cols=64, group=None):
        """Whole-graph convenience constructor: every rank passes the same CSR (int32 rowptr/col on the
        device); the partition is derived from it and this rank keeps its block."""
        rowptr_h = rowptr.detach().cpu().numpy().astype(np.int64)
        bounds = partition_rows(rowptr_h, world, balance)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        e_lo, e_hi = int(rowptr_h[lo]), int(rowptr_h[hi])
        local_rowptr = torch.from_numpy(rowptr_h[lo:hi + 1] - e_lo).to(col.device)
        self._setup(local_rowptr, col[e_lo:e_hi], val[e_lo:e_hi], bounds, rank, world, make_local,
                    value_factor, int(rowptr_h[-1]), exchange, prelaid, plane_cols, group)
        if int(n) != self.n:
            raise ValueError("n does not match the row pointer")

    @classmethod
    def from_row_block(cls, local_rowptr, local_col, local_val, bounds, rank, world, make_local,
                       value_factor=None, total_nnz=None, exchange="all_gather", prelaid="auto", plane_cols=64, group=None):
        """A rank from its own row block: `local_rowptr` [rows+1] (starts at 0), `local_col` GLOBAL column
        ids (any integer dtype; int64 for graphs past 2³¹ columns·entries), `local_val` fp32, `bounds`
        [world+1] the first global row of every rank's block (the same on all ranks).  `group`: the process group
        the shard's layers will run on (None = the default group); its size must be `world`."""
        self = cls.__new__(cls)
        self._setup(local_rowptr, local_col, local_val, np.asarray(bounds, dtype=np.int64), rank, world,
                    make_local, value_factor, total_nnz, exchange, prelaid, plane_cols, group)
        return self

    def _setup(self, local_rowptr, gcol, val, bounds, rank, world, make_local, value_factor, total_nnz, exchange,
               prelaid="auto", plane_cols=64, group=None):
        if exchange not in ("all_gather", "direct", "push"):
            raise ValueError("exchange must be 'all_gather', 'direct' or 'push'")
        self._push = {}            # exchange="push": data_ptr of a registered buffer -> _PushPeers
        self.rank, self.world, self.exchange, self.group = int(rank), int(world), exchange, group
        self.bounds = np.asarray(bounds, dtype=np.int64)
        if len(self.bounds) != self.world + 1 or self.bounds[0] != 0 or np.any(np.diff(self.bounds) < 0):
            raise ValueError("bounds must be world+1 non-decreasing row boundaries starting at 0")
        # the ranks this shard exchanges with: its process group must have exactly `world` members (size 1: one process
        # standing in for one rank of `world`, collective = False — nobody to exchange or agree with)
        peers = False
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            gsize = dist.get_world_size(group)
            if gsize not in (1, self.world):
                raise ValueError(f"the shard is one of {self.world} but its process group has {gsize} ranks: pass the group "
                                 "the layers run on (group=...)")
            peers = gsize == self.world
        self.n = int(self.bounds[-1])
        self.collective = True     # False = compute this rank's block only (single-GPU rehearsal of rank r of W)
        self.max_rows = int(np.diff(self.bounds).max())
        self.row_lo, self.row_hi = int(self.bounds[rank]), int(self.bounds[rank + 1])
        self.rows = self.row_hi - self.row_lo
        device = gcol.device
        if int(local_rowptr.numel()) != self.rows + 1:
            raise ValueError("local_rowptr must have rows+1 entries")
        self.local_nnz = int(gcol.numel())
        if self.local_nnz >= 2 ** 31:
            raise ValueError("a rank's block must hold fewer than 2^31 non-zeros (use more ranks)")
        # Pre-laid chains: slots of q whole column slices.  Decided HERE, from numbers every rank has (bounds, world,
        # the global nnz) — every rank must size its slot alike — and only kept if the local operator then really
        # offers the layout (checked below, again from numbers that are the same on all ranks).
        self.prelaid, self.slices_per_rank, self.slice_cols = False, 0, 0
        if prelaid not in ("auto", True, False):
            raise ValueError("prelaid must be 'auto', True or False")
        want = prelaid is True or (prelaid == "auto" and value_factor is not None)
        if want and value_factor is not None and total_nnz:
            mean_rows = max(1, self.n // self.world)
            S_auto = int(_lib.load().gcn_spmm_auto_slices(mean_rows, self.world * self.max_rows,
                                                          max(1, int(total_nnz) // self.world), 1))
            if S_auto > 1:
                q = -(-S_auto // self.world)
                w = -(-self.max_rows // q)
                if w <= 32767:
                    self.prelaid, self.slices_per_rank, self.slice_cols = True, q, w
                    self.max_rows = q * w                 # the slot: q slices of w columns (>= the longest block)
        if prelaid is True and not self.prelaid:
            raise _lib.GcnAmdError("prelaid=True: this partition does not take the pre-laid layout (needs value factors, "
                                   "a graph the sliced kernels pay on, slices of at most 32 767 columns)")
        if self.world * self.max_rows >= 2 ** 31:
            raise ValueError("padded column space does not fit int32")
        self.total_nnz = int(total_nnz) if total_nnz is not None else None
        bounds_t = torch.from_numpy(self.bounds).to(device)
        self._bounds_t = bounds_t
        # global column id -> padded numbering, on the device, in pieces (the temporaries are int64)
        pcol = torch.empty(self.local_nnz, dtype=torch.int32, device=device)
        step = 1 << 26
        for s in range(0, self.local_nnz, step):
            g = gcol[s:s + step].long()
            owner = torch.bucketize(g, bounds_t[1:], right=True)
            pcol[s:s + step] = (owner * self.max_rows + (g - bounds_t[owner])).to(torch.int32)
            del g, owner
        self._local_args = (local_rowptr.to(device=device, dtype=torch.int32).contiguous(), pcol,
                            val.to(torch.float32).contiguous(), (self.rows, self.world * self.max_rows))
        self._make_local = make_local
        # value_factor: u [n] with Â[r, c] = u[r]·u[c] (u = D^-1/2 of a normalised adjacency).  The row block
        # with renumbered columns still factors — as u[rows] x u in the padded numbering — and telling the
        # operator so lets its sliced main pass drop the value stream (CsrAdjacency.set_value_factors)
        self._factors = None
        if value_factor is not None:
            u = value_factor.to(device=device, dtype=torch.float32)
            self._factors = (u[self.row_lo:self.row_hi].contiguous(), self.to_padded(u[:, None])[:, 0].contiguous())
        self.local = self._new_local()
        if self.prelaid:                                   # does the operator serve the layout this slot was cut for?
            lay = self.local.prelaid_layout(plane_cols) if (self._factors is not None and hasattr(self.local, "prelaid_layout")) else None
            ok = (lay is not None and lay["slices"] == self.world * self.slices_per_rank and lay["slice_cols"] == self.slice_cols
                  and lay["ld"] == plane_cols)
            # the ranks must AGREE: a rank whose block falls on the other side of a rule (non-zeros per column, table
            # sizes) would size its exchange buffers differently and the collective would hang.  One MIN over the ranks
            # of the group the layers will run on (the default group is not necessarily that group).
            if peers:
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok = bool(int(flag.item()))
            if not ok:
                if prelaid is True:
                    raise _lib.GcnAmdError(f"prelaid=True: the local operator does not offer the layout (got {lay})")
                self.prelaid = False                       # (the wider slot stays: harmless for the copying path)
                self.local = self._new_local()             # ... and the operator goes back to its own slice count

    # -- global <-> padded numbering (device side) ---------------------------------------------
    def _pad_index(self, device):
        idx = torch.arange(self.n, dtype=torch.int64, device=device)
        b = self._bounds_t.to(device)
        owner = torch.bucketize(idx, b[1:], right=True)
        return owner * self.max_rows + (idx - b[owner])

    def to_padded(self, H):
        """global [n, k] -> padded [world*max_rows, k] (rows of rank p at p*max_rows ...; the rest zero)"""
        out = torch.zeros((self.world * self.max_rows, H.shape[1]), dtype=H.dtype, device=H.device)
        out.index_copy_(0, self._pad_index(H.device), H)
        return out

    def from_padded(self, Hp):
        """padded [world*max_rows, k] -> global [n, k]"""
        return Hp.index_select(0, self._pad_index(Hp.device))

    def new_buffer(self, k, device, dtype=torch.float32):
        return torch.zeros((self.world * self.max_rows, k), dtype=dtype, device=device)

    # -- the pre-laid layout: B' = diag(u)·H, slice by slice, one zero row behind every slice ---------------------
    @property
    def slot_rows(self):
        """rows of one rank's slot in an exchange buffer (pre-laid: its q slices with their zero rows)"""
        return self.slices_per_rank * (self.slice_cols + 1) if self.prelaid else self.max_rows

    def new_prelaid_buffer(self, k, device):
        return torch.zeros((self.world * self.slot_rows, k), dtype=torch.float32, device=device)

    def _prelaid_index(self, device):
        c = self._pad_index(device)                        # global row -> padded column id
        return c + c // self.slice_cols                    # ... -> row of B'

    def to_prelaid(self, H):
        """global [n, k] -> B' [world*q*(w+1), k]: rows scaled by u, in the slices' layout (zero rows stay zero)"""
        out = self.new_prelaid_buffer(H.shape[1], H.device)
        u = self._u_global(H.device)
        out.index_copy_(0, self._prelaid_index(H.device), H * u[:, None])
        return out

    def from_prelaid(self, Bp):
        """B' -> global [n, k] (the scaling undone)"""
        u = self._u_global(Bp.device)
        return Bp.index_select(0, self._prelaid_index(Bp.device)) / u[:, None]

    def _u_global(self, device):
        # u of every global vertex, recovered from the padded column factors this shard keeps
        return self._factors[1].to(device).index_select(0, self._pad_index(device))

    # -- the exchange ---------------------------------------------------------------------------
    def collective_form(self, group=None):
        """which operation a layer's exchange issues (decided from the backend, once)"""
        if self.world == 1 or not self.collective:
            return "none"
        if self.exchange == "direct":
            return "batch_isend_irecv (direct exchange, one shard per peer link)"
        if self.exchange == "push":
            return "push (peer buffers mapped through IPC, hipMemcpyAsync per peer + one flag per layer, no collective kernel)"
        return "all_gather_into_tensor" if dist.get_backend(group) == "nccl" else "all_gather (list form)"

    def register_exchange_buffers(self, buffers, group=None):
        """exchange="push": map every peer's counterpart of these exchange buffers into this process, once.  Every rank
        calls it with the same number of buffers of the same shapes, in the same order (a collective: IPC handles travel
        through all_gather_object).  The buffers must stay alive and keep their storage while the shard is used."""
        if self.exchange != "push" or self.world == 1 or not self.collective:
            return
        group = group if group is not None else self.group
        for buf in buffers:
            self._push[buf.data_ptr()] = _PushPeers(self, buf, group)

    def close_exchange(self, group=None):
        """exchange="push": release the IPC mappings and the flag memory (a collective: every rank calls it)"""
        group = group if group is not None else self.group
        for reg in self._push.values():
            reg.close(group)
        self._push = {}

    def check_exchange(self):
        """exchange="push": did a wait give up (a peer that never signalled)?  Synchronises; raises GcnAmdError."""
        for reg in self._push.values():
            torch.cuda.synchronize(reg.status.device)
            bad = int(reg.status.item())
            if bad:
                raise _lib.GcnAmdError(f"push exchange: rank {self.rank} gave up waiting for the flag of rank {bad - 1}")

    def _exchange(self, out_padded, slot, group, async_op):
        if self.exchange == "push":
            reg = self._push.get(out_padded.data_ptr())
            if reg is None:
                raise _lib.GcnAmdError("exchange='push': the buffer was not registered (register_exchange_buffers; "
                                       "PipelinedAggregation does it for its planes)")
            reg.push_and_wait(slot)
            return None                                    # (stream-ordered: nothing to wait for on the host)
        if self.exchange == "direct":
            ops = []
            for off in range(1, self.world):               # staggered peer order: rank r starts with r+1
                peer = (self.rank + off) % self.world
                src = (self.rank - off) % self.world
                dst_view = out_padded[src * self.slot_rows: (src + 1) * self.slot_rows]
                ops.append(dist.P2POp(dist.isend, slot, dist.get_global_rank(group, peer) if group is not None else peer,
                                      group=group))
                ops.append(dist.P2POp(dist.irecv, dst_view, dist.get_global_rank(group, src) if group is not None else src,
                                      group=group))
            works = _Works(dist.batch_isend_irecv(ops))
            if async_op:
                return works
            works.wait()
            return None
        if dist.get_backend(group) == "nccl":
            return dist.all_gather_into_tensor(out_padded, slot, group=group, async_op=async_op)
        views = [out_padded[p * self.slot_rows: (p + 1) * self.slot_rows] for p in range(self.world)]
        return dist.all_gather(views, slot.clone(), group=group, async_op=async_op)

    def another_local(self):
        """A second operator over the same row block (own plan, own workspaces) — what lets two
        SpMMs of this block run on different streams at the same time (PipelinedAggregation)."""
        return self._new_local()

    def _new_local(self):
        if self.prelaid:                                   # exactly world*q slices: the slices ARE the slots' fractions
            local = self._make_local(*self._local_args, slices=self.world * self.slices_per_rank)
        else:
            local = self._make_local(*self._local_args)
        if self._factors is not None and hasattr(local, "set_value_factors"):
            try:
                local.set_value_factors(*self._factors)
            except _lib.GcnAmdError as e:
                if e.status != _lib.ERR_NOT_FACTORED:
                    raise                           # a real failure (HIP error, allocation, bad arguments): not ours to hide
                self._factors = None                # the values do not factor that way: keep reading them
        return local

    def layer_async(self, H_padded, out_padded, group=None, local=None):
        """Like layer(), but the exchange is only ENQUEUED (on the communicator's stream, behind
        the SpMM that fills the slot); returns the Work handle (None for world == 1).  The caller
        waits on it before the next read of out_padded — this is what lets the exchange of one
        column plane overlap the SpMM of the next (PipelinedAggregation)."""
        slot = self._slot(out_padded)
        if self.rows:
            self._local_spmm(local or self.local, H_padded, slot)
        if self.world > 1 and self.collective:
            return self._exchange(out_padded, slot, group if group is not None else self.group, True)
        return None

    def buffer_rows_of_local_rows(self, device):
        """where this rank's rows live in an exchange buffer (either layout)"""
        r = torch.arange(self.rows, device=device)
        if self.prelaid:
            return self.rank * self.slot_rows + r + r // self.slice_cols
        return self.rank * self.max_rows + r

    def buffer_rows_of_rank(self, q, device):
        """... and the rows of rank q's block (the same arithmetic on every rank)"""
        r = torch.arange(int(self.bounds[q + 1] - self.bounds[q]), device=device)
        if self.prelaid:
            return q * self.slot_rows + r + r // self.slice_cols
        return q * self.max_rows + r

    def as_buffer_operator(self):
        """(rowptr, col, val) of the map  exchange buffer -> this rank's rows of the next exchange buffer  — what a
        layer computes, in buffer coordinates, for checking a result against plain arithmetic: the block itself, or for
        the pre-laid layout the pattern with B' row ids and the values u[r]² (B' in, B' out)."""
        rp, pcol, val, _shape = self._local_args
        if not self.prelaid:
            return rp, pcol, val
        u = self._factors[0]
        rows = torch.repeat_interleave(torch.arange(self.rows, device=pcol.device), (rp[1:] - rp[:-1]).long())
        return rp, pcol + pcol // self.slice_cols, (u * u)[rows]

    def _slot(self, buf):
        return buf[self.rank * self.slot_rows: (self.rank + 1) * self.slot_rows]

    def _local_spmm(self, local, src, slot):
        if self.prelaid:        # src IS B'; the result lands in this rank's slot of the next B', scaled for the next layer
            local.matmul_prelaid(src, slot, out_scale=self._factors[0], out_gap=self.slice_cols)
        else:
            local.matmul_raw(src, out=slot[: self.rows])

    def layer(self, H_padded, out_padded, group=None):
        """out = Â · H for the whole graph, in the padded layout, on every rank:
        local row-block SpMM into this rank's slot, then one in-place exchange."""
        slot = self._slot(out_padded)
        if self.rows:
            self._local_spmm(self.local, H_padded, slot)
        if self.world > 1 and self.collective:
            self._exchange(out_padded, slot, group if group is not None else self.group, False)
        return out_padded


class _PushPeers:
    """One exchange buffer of this rank and its counterparts on the peers (exchange="push")."""
    RING = 1 << 12                                         # layer counters travel as values of a constant table

    def __init__(self, shard, buf, group):
        from torch.multiprocessing.reductions import reduce_tensor
        self.shard, self.buf = shard, buf
        dev = buf.device
        lib = _lib.load()
        # flags[q]: the last layer rank q's shard landed for — fine-grained device memory (gcn_exchange_flags_create)
        fl, handle = ctypes.c_void_p(), ctypes.create_string_buffer(64)
        with torch.cuda.device(dev):
            _lib.check(lib.gcn_exchange_flags_create(shard.world, ctypes.cast(ctypes.byref(fl), ctypes.c_void_p),
                                                     ctypes.cast(handle, ctypes.c_void_p)), "gcn_exchange_flags_create")
        self.flags_ptr = fl.value
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.consts = torch.arange(self.RING, dtype=torch.int32, device=dev)
        self.count = 0
        torch.cuda.synchronize(dev)                        # (the buffer's zero fill is done before any peer may push into it)
        handles = [None] * shard.world
        dist.all_gather_object(handles, (reduce_tensor(buf), handle.raw), group=group)
        self.peer_buf, self.peer_flags = [None] * shard.world, [0] * shard.world
        for q, ((fb, ab), fh) in enumerate(handles):
            if q == shard.rank:
                continue
            self.peer_buf[q] = fb(*ab)                     # the peer's buffer, mapped into this process (torch's IPC)
            pf = ctypes.c_void_p()
            with torch.cuda.device(dev):
                _lib.check(lib.gcn_exchange_flags_open(ctypes.cast(ctypes.create_string_buffer(fh, 64), ctypes.c_void_p),
                                                       ctypes.cast(ctypes.byref(pf), ctypes.c_void_p)), "gcn_exchange_flags_open")
            self.peer_flags[q] = pf.value
        dist.barrier(group=group)                          # every mapping exists before anyone pushes

    def close(self, group):
        """unmap the peers' flags, then (after a barrier: nobody writes any more) free this rank's"""
        lib = _lib.load()
        torch.cuda.synchronize(self.buf.device)
        for q, pf in enumerate(self.peer_flags):
            if pf:
                lib.gcn_exchange_flags_close(ctypes.c_void_p(pf))
                self.peer_flags[q] = 0
        self.peer_buf = [None] * len(self.peer_buf)
        dist.barrier(group=group)
        if self.flags_ptr:
            lib.gcn_exchange_flags_destroy(ctypes.c_void_p(self.flags_ptr))
            self.flags_ptr = 0

    def push_and_wait(self, slot):
        sh, lib = self.shard, _lib.load()
        self.count += 1
        value = self.count % self.RING
        st = ctypes.c_void_p(torch.cuda.current_stream(self.buf.device).cuda_stream)
        lo = sh.rank * sh.slot_rows
        nbytes = slot.numel() * slot.element_size()
        vptr = ctypes.c_void_p(self.consts.data_ptr() + 4 * value)
        for off in range(1, sh.world):                     # staggered peer order: rank r starts with r+1
            q = (sh.rank + off) % sh.world
            dst = self.peer_buf[q][lo: lo + sh.slot_rows]
            _lib.check(lib.gcn_exchange_push(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(slot.data_ptr()), nbytes, st),
                       "gcn_exchange_push")
            _lib.check(lib.gcn_exchange_signal(ctypes.c_void_p(self.peer_flags[q] + 4 * sh.rank), vptr, st),
                       "gcn_exchange_signal")
        _lib.check(lib.gcn_exchange_wait(ctypes.c_void_p(self.flags_ptr), sh.world, sh.rank, value,
                                         ctypes.c_void_p(self.status.data_ptr()), 20.0, st), "gcn_exchange_wait")


class PipelinedAggregation:
    """Repeated aggregation layers H ← Â·H on a row-sharded Â with the exchange hidden.

    The k feature columns are kept as independent PLANES of ≤ `plane_cols` columns (each plane a
    padded [world·max_rows, cols] buffer pair).  Â·H acts on every column independently, so
    plane p of layer l+1 depends only on the gathered plane p of layer l: while RCCL moves
    plane p over xGMI on its own stream, the compute stream already runs the SpMM of plane p+1
    (and, at the layer seam, plane 0 of the next layer).  Per layer every rank still does all of
    its row-block SpMM work and one exchange per plane; only the waiting is gone.  Plane width 64
    is also the kernel's preferred column tile when n·256 B fits the Infinity Cache.
    """

    def __init__(self, shard, k, device, plane_cols=64, group=None, streams=None):
        self.shard, self.k, self.group = shard, int(k), group
        self.widths = [min(plane_cols, k - c) for c in range(0, k, plane_cols)]
        if shard.prelaid and any(w != plane_cols for w in self.widths):
            raise _lib.GcnAmdError("a pre-laid shard was cut for planes of %d columns; k must be a multiple of it" % plane_cols)
        make = shard.new_prelaid_buffer if shard.prelaid else shard.new_buffer
        self.src = [make(w, device) for w in self.widths]
        self.dst = [make(w, device) for w in self.widths]
        shard.register_exchange_buffers(self.src + self.dst, group)       # (exchange="push": peers' buffers mapped once)
        self.pending = [None] * len(self.widths)
        # One HIP stream and one operator (plan + workspaces) per plane: the planes' chains
        # (SpMM passes -> fix-up -> slice reduction -> exchange) are independent, so on separate
        # streams the short tail kernels and launch gaps of one plane hide under the main kernel of the
        # other (rank-0 share of an 8-way partition of the Reddit-shaped graph, compute only: 0.546 ->
        # 0.501 ms per layer, profiles/r01f_sim8_streams.log).  Off on the CPU (gloo tests) and for a single plane.
        dev = torch.device(device)
        if streams is None:
            # ... when a plane's main kernel is SHORT.  A kernel of many rounds of blocks fills the chip by itself and only
            # loses by sharing it; one of fewer than ~2.5 rounds leaves its last round half empty, which a second plane's
            # kernel fills (r04, rank shares of the Reddit-shaped graph, one stream / two streams: N = 2 (6.8 rounds) 1.371 /
            # 1.464 ms, N = 4 (3.4) 0.713 / 0.760, N = 8 (1.7) 0.402 / 0.380; profiles/r04z_rank_share_plane_streams.log).
            # On one stream the planes still pipeline: plane 0's exchange runs under plane 1's SpMM.
            rounds = 0.0
            if dev.type == "cuda":
                cu = int(_lib.load().gcn_device_cu_count())
                rounds = shard.local_nnz / 8192.0 / (4.0 * max(cu, 1))      # blocks of 16 chunks x 512 entries, 4 resident per CU
            # (Only for the sliced group kernels that rule was measured on — the pre-laid chain: the unsliced, HBM-bound
            #  kernel of a papers100M-shaped block does gain from a second plane beside it, 29.4-30.0 ms per layer on two
            #  streams against 31.3 on one, profiles/r04z_*.)
            long_kernels = bool(getattr(shard, "prelaid", False)) and rounds >= 2.5
            streams = dev.type == "cuda" and len(self.widths) > 1 and not long_kernels
        # Staggered priorities: two chains of equal length started together stay in phase — both main kernels share
        # the chip and both tails (fix-up, slice reduction) end up exposed behind them, once per layer (rank share of an
        # 8-way partition: 2 x 166 us of main kernels, 390 us per layer).  With plane 0 ahead in priority its main
        # kernel takes the chip first and its tail then runs beside plane 1's main kernel (0.387 -> 0.384 ms at N = 8).
        # (r04: the planes' tails on ONE high-priority stream of their own, and the planes' main kernels taking turns, were built
        #  and measured — 0.417 / 0.450 ms per layer against 0.377, profiles/r04h_sim8_tail_stream_and_turns.log: with equal
        #  priorities the two main kernels fall into phase and both tails end up exposed behind them; with turns the cross-stream
        #  event waits leave 31-38 us between alternate main kernels and a main kernel alone is slower than its share of two.
        #  Removed again)
        self.streams = ([torch.cuda.Stream(dev, priority=(-1 if p % 2 == 0 else 0)) for p in range(len(self.widths))]
                        if streams else None)
        self.locals = [shard.local] + [shard.another_local() if streams else shard.local for _ in self.widths[1:]]

    def set_local_option(self, name, *args):
        """apply a CsrAdjacency setter (set_blocks_per_cu, set_gather_width, ...) to every plane's operator"""
        for loc in {id(l): l for l in self.locals}.values():
            getattr(loc, name)(*args)

    def load(self, H):
        """global [n, k] features → the planes' source buffers"""
        c = 0
        conv = self.shard.to_prelaid if self.shard.prelaid else self.shard.to_padded
        for p, w in enumerate(self.widths):
            self.src[p].copy_(conv(H[:, c:c + w].contiguous()))
            c += w

    def load_padded_block(self, fill):
        """fill(plane_index, buffer[world*max_rows, width]) writes the planes' source buffers in place — for
        features that only ever exist in the padded layout (graphs whose global [n, k] matrix is never built)"""
        for p, buf in enumerate(self.src):
            fill(p, buf)

    def _one_plane(self, p):
        if self.pending[p] is not None:
            self.pending[p].wait()                  # the exchange that produced src[p]
            self.pending[p] = None
        self.pending[p] = self.shard.layer_async(self.src[p], self.dst[p], self.group, local=self.locals[p])
        self.src[p], self.dst[p] = self.dst[p], self.src[p]

    def step(self):
        """one aggregation layer over all planes (exchanges left in flight)"""
        if self.streams is None:
            for p in range(len(self.widths)):
                self._one_plane(p)
            return
        cur = torch.cuda.current_stream()
        for p, s in enumerate(self.streams):
            s.wait_stream(cur)                      # whatever filled src[p] on the caller's stream
            with torch.cuda.stream(s):
                self._one_plane(p)

    def finish(self):
        if self.streams is not None:
            cur = torch.cuda.current_stream()
            for p, s in enumerate(self.streams):
                with torch.cuda.stream(s):
                    if self.pending[p] is not None:
                        self.pending[p].wait()
                        self.pending[p] = None
                cur.wait_stream(s)
            return
        for p, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[p] = None

    def result(self):
        """global [n, k] view of the current layer output (waits for outstanding exchanges)"""
        self.finish()
        conv = self.shard.from_prelaid if self.shard.prelaid else self.shard.from_padded
        return torch.cat([conv(b) for b in self.src], 1)

    def local_rows(self):
        """this rank's own rows of the current layer output, [rows, k] (no global matrix involved)"""
        self.finish()
        sh = self.shard
        if sh.prelaid:
            r = torch.arange(sh.rows, device=self.src[0].device)
            idx = sh.rank * sh.slot_rows + r + r // sh.slice_cols
            u = sh._factors[0].to(self.src[0].device)
            return torch.cat([b.index_select(0, idx) / u[:, None] for b in self.src], 1)
        lo = sh.rank * sh.max_rows
        return torch.cat([b[lo: lo + sh.rows] for b in self.src], 1)
