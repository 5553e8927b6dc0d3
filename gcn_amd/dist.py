"""1-D row partition of Â across the GPUs of one node + all-gather of layer outputs.

The reference is single-GPU (device 0 hard-coded, flexspmm.cu:507); this module is the
MI355X-native extension BASELINE.json asks for: rank p owns a contiguous,
nnz-balanced row block Â[rows_p, :], computes H'_p = Â[rows_p, :] · H with the same HIP
kernel, and the next layer's input H' is assembled with ONE RCCL all-gather over xGMI.

Zero-copy layout: shards have unequal row counts, so every shard is padded to
`max_rows` and the gathered buffer is [world * max_rows, k].  Instead of compacting
that buffer after every layer, the column indices of the local block are remapped ONCE
to the padded numbering (col' = owner(col) * max_rows + local_index(col)); the local
SpMM writes straight into this rank's slot of the buffer and the all-gather is done
in place.  Summation order inside a row is unchanged by the partition, so every row
equals the single-GPU result bit for bit.
"""
import numpy as np
import torch
import torch.distributed as dist


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


class RowShardedAdjacency:
    """This rank's row block of Â in the padded-gather numbering.

    make_local(rowptr_i32, col_i32, val_f32, shape) must return an object with
    ``matmul_raw(dense, out=...)`` — gcn_amd.CsrAdjacency on the GPU; the gloo/CPU tests
    inject an oracle-backed stand-in (the product path has no CPU compute).
    """

    def __init__(self, rowptr, col, val, n, rank, world, make_local, balance="nnz", value_factor=None):
        rowptr_h = rowptr.detach().cpu().numpy().astype(np.int64)
        self.n, self.rank, self.world = int(n), int(rank), int(world)
        self.collective = True     # False = compute this rank's block only (single-GPU rehearsal of rank r of W)
        self.bounds = partition_rows(rowptr_h, world, balance)
        sizes = np.diff(self.bounds)
        self.max_rows = int(sizes.max())
        self.row_lo, self.row_hi = int(self.bounds[rank]), int(self.bounds[rank + 1])
        self.rows = self.row_hi - self.row_lo
        device = col.device
        e_lo, e_hi = int(rowptr_h[self.row_lo]), int(rowptr_h[self.row_hi])
        local_rowptr = torch.from_numpy((rowptr_h[self.row_lo:self.row_hi + 1] - e_lo).astype(np.int32)).to(device)
        gcol = col[e_lo:e_hi].long()
        bounds_t = torch.from_numpy(self.bounds).to(device)
        owner = torch.bucketize(gcol, bounds_t[1:], right=True)
        pcol = owner * self.max_rows + (gcol - bounds_t[owner])
        self.local_nnz = e_hi - e_lo
        self.total_nnz = int(rowptr_h[-1])
        self._local_args = (local_rowptr, pcol.to(torch.int32), val[e_lo:e_hi].contiguous(),
                            (self.rows, self.world * self.max_rows))
        self._make_local = make_local
        # value_factor: u [n] with Â[r, c] = u[r]·u[c] (u = D^-1/2 of a normalised adjacency).  The row block
        # with renumbered columns still factors — as u[rows] x u in the padded numbering — and telling the
        # operator so lets its sliced main pass drop the value stream (CsrAdjacency.set_value_factors)
        self._factors = None
        if value_factor is not None:
            u = value_factor.to(device=device, dtype=torch.float32)
            self._factors = (u[self.row_lo:self.row_hi].contiguous(), self.to_padded(u[:, None])[:, 0].contiguous())
        self.local = self._new_local()
        self._bounds_t = bounds_t

    # global [n, k] -> padded [world*max_rows, k]
    def to_padded(self, H):
        out = torch.zeros((self.world * self.max_rows, H.shape[1]), dtype=H.dtype, device=H.device)
        for p in range(self.world):
            lo, hi = int(self.bounds[p]), int(self.bounds[p + 1])
            out[p * self.max_rows: p * self.max_rows + (hi - lo)] = H[lo:hi]
        return out

    # padded [world*max_rows, k] -> global [n, k]
    def from_padded(self, Hp):
        parts = []
        for p in range(self.world):
            lo, hi = int(self.bounds[p]), int(self.bounds[p + 1])
            parts.append(Hp[p * self.max_rows: p * self.max_rows + (hi - lo)])
        return torch.cat(parts, 0)

    def new_buffer(self, k, device, dtype=torch.float32):
        return torch.zeros((self.world * self.max_rows, k), dtype=dtype, device=device)

    def _all_gather(self, out_padded, slot, group, async_op):
        try:
            return dist.all_gather_into_tensor(out_padded, slot, group=group, async_op=async_op)
        except (RuntimeError, NotImplementedError):         # backends without the flat form
            views = [out_padded[p * self.max_rows: (p + 1) * self.max_rows] for p in range(self.world)]
            return dist.all_gather(views, slot.clone(), group=group, async_op=async_op)

    def another_local(self):
        """A second operator over the same row block (own plan, own workspaces) — what lets two
        SpMMs of this block run on different streams at the same time (PipelinedAggregation)."""
        return self._new_local()

    def _new_local(self):
        local = self._make_local(*self._local_args)
        if self._factors is not None and hasattr(local, "set_value_factors"):
            try:
                local.set_value_factors(*self._factors)
            except Exception:                       # the values do not factor that way: keep reading them
                self._factors = None
        return local

    def layer_async(self, H_padded, out_padded, group=None, local=None):
        """Like layer(), but the all-gather is only ENQUEUED (on the communicator's stream, behind
        the SpMM that fills the slot); returns the Work handle (None for world == 1).  The caller
        waits on it before the next read of out_padded — this is what lets the all-gather of one
        column plane overlap the SpMM of the next (PipelinedAggregation)."""
        slot = out_padded[self.rank * self.max_rows: (self.rank + 1) * self.max_rows]
        if self.rows:
            (local or self.local).matmul_raw(H_padded, out=slot[: self.rows])
        if self.world > 1 and self.collective:
            return self._all_gather(out_padded, slot, group, True)
        return None

    def layer(self, H_padded, out_padded, group=None):
        """out = Â · H for the whole graph, in the padded layout, on every rank:
        local row-block SpMM into this rank's slot, then one in-place all-gather."""
        k = H_padded.shape[1]
        slot = out_padded[self.rank * self.max_rows: (self.rank + 1) * self.max_rows]
        if self.rows:
            self.local.matmul_raw(H_padded, out=slot[: self.rows])
        if self.world > 1 and self.collective:
            try:
                dist.all_gather_into_tensor(out_padded, slot, group=group)
            except (RuntimeError, NotImplementedError):     # backends without the flat form
                views = [out_padded[p * self.max_rows: (p + 1) * self.max_rows] for p in range(self.world)]
                dist.all_gather(views, slot.clone(), group=group)
        return out_padded


class PipelinedAggregation:
    """Repeated aggregation layers H ← Â·H on a row-sharded Â with the exchange hidden.

    The k feature columns are kept as independent PLANES of ≤ `plane_cols` columns (each plane a
    padded [world·max_rows, cols] buffer pair).  Â·H acts on every column independently, so
    plane p of layer l+1 depends only on the gathered plane p of layer l: while RCCL all-gathers
    plane p over xGMI on its own stream, the compute stream already runs the SpMM of plane p+1
    (and, at the layer seam, plane 0 of the next layer).  Per layer every rank still does all of
    its row-block SpMM work and one all-gather per plane; only the waiting is gone.  Plane width 64
    is also the kernel's preferred column tile when n·256 B fits the Infinity Cache.
    """

    def __init__(self, shard, k, device, plane_cols=64, group=None, streams=None):
        self.shard, self.k, self.group = shard, int(k), group
        self.widths = [min(plane_cols, k - c) for c in range(0, k, plane_cols)]
        self.src = [shard.new_buffer(w, device) for w in self.widths]
        self.dst = [shard.new_buffer(w, device) for w in self.widths]
        self.pending = [None] * len(self.widths)
        # One HIP stream and one operator (plan + workspaces) per plane: the planes' chains
        # (SpMM passes -> fix-up -> slice reduction -> all-gather) are independent, so on separate
        # streams the short tail kernels and launch gaps of one plane hide under the main kernel of the
        # other (rank-0 share of an 8-way partition of the Reddit-shaped graph, compute only: 0.546 ->
        # 0.501 ms per layer, profiles/r01f_sim8_streams.log).  Off on the CPU (gloo tests) and for a single plane.
        dev = torch.device(device)
        if streams is None:
            streams = dev.type == "cuda" and len(self.widths) > 1
        self.streams = [torch.cuda.Stream(dev) for _ in self.widths] if streams else None
        self.locals = [shard.local] + [shard.another_local() if streams else shard.local for _ in self.widths[1:]]

    def set_local_option(self, name, *args):
        """apply a CsrAdjacency setter (set_blocks_per_cu, set_gather_width, ...) to every plane's operator"""
        for loc in {id(l): l for l in self.locals}.values():
            getattr(loc, name)(*args)

    def load(self, H):
        """global [n, k] features → the planes' source buffers"""
        c = 0
        for p, w in enumerate(self.widths):
            self.src[p].copy_(self.shard.to_padded(H[:, c:c + w].contiguous()))
            c += w

    def _one_plane(self, p):
        if self.pending[p] is not None:
            self.pending[p].wait()                  # the gather that produced src[p]
            self.pending[p] = None
        self.pending[p] = self.shard.layer_async(self.src[p], self.dst[p], self.group, local=self.locals[p])
        self.src[p], self.dst[p] = self.dst[p], self.src[p]

    def step(self):
        """one aggregation layer over all planes (all-gathers left in flight)"""
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
        """global [n, k] view of the current layer output (waits for outstanding gathers)"""
        self.finish()
        return torch.cat([self.shard.from_padded(b) for b in self.src], 1)
