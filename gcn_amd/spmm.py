"""Host-side op surface for the aggregation SpMM  C = Â · B  on MI355X.

Mirrors the two interfaces the reference reaches its SpMM through:

* B2 — ``torch.spmm(adj, dense)`` / ``torch.sparse.mm(adj, dense)`` with a
  ``torch.sparse_coo`` fp32 ``adj`` (pygcn/gcn1.py:53, gcn2.py:92,147, gcn3.py:87,146,
  gcn4.py:56,106, gcn5.py:90,135).  ``install()`` routes exactly that case (GPU
  sparse fp32 × GPU dense fp32) to the HIP kernel and leaves everything else to
  stock PyTorch, so gcn1–5 run unchanged.
* B1 — the ``flexspmm`` autograd Function of pygcn/gcn6.py:34-62 (see dropin.py).

PyTorch is plumbing here (device memory, streams, autograd); the arithmetic is
libgcnspmm.so.  There is no CPU fallback: CPU tensors raise.
"""
import ctypes
import weakref

import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class CsrAdjacency:
    """Device-resident CSR matrix (int32 rowptr / int32 col / fp32 val) + its cached
    SpMM plan.  The layout is the reference's own CSR hand-off (gcn6.py:302-311:
    ``to_sparse_csr()``, crow/col cast to int32)."""

    def __init__(self, rowptr, col, val, shape, symmetric=None, chunk_nnz=0, slices="auto", panels=0):
        if not (rowptr.is_cuda and col.is_cuda and val.is_cuda):
            raise _lib.GcnAmdError("CsrAdjacency needs CUDA/HIP tensors (no CPU path in gcn_amd)")
        self.rowptr = rowptr.to(torch.int32).contiguous()
        self.col = col.to(torch.int32).contiguous()
        self.val = val.to(torch.float32).contiguous()
        self.m, self.n = int(shape[0]), int(shape[1])
        self.nnz = int(self.col.numel())
        if self.rowptr.numel() != self.m + 1:
            raise ValueError("rowptr must have m+1 entries")
        if self.nnz >= 2 ** 31:
            raise ValueError("nnz must fit int32 (row-partition the matrix first)")
        self.symmetric = symmetric
        self.chunk_nnz = int(chunk_nnz)
        self.slices = -1 if slices == "auto" else int(slices)    # XCD-aware column slicing
        # LDS-staged row panels: off unless asked for ("auto" = by measured window coverage) — on MI355X
        # the chunk kernel is still faster even on community-ordered graphs
        self.panels = -1 if panels == "auto" else int(bool(panels))
        self._plan = None
        self._transpose = None
        self.device = self.val.device

    # -- constructors ---------------------------------------------------------
    @classmethod
    def from_scipy(cls, mat, device="cuda", symmetric=None, chunk_nnz=0):
        mat = mat.tocsr()
        mat.sort_indices()
        dev = torch.device(device)
        return cls(torch.from_numpy(mat.indptr.astype("int32")).to(dev),
                   torch.from_numpy(mat.indices.astype("int32")).to(dev),
                   torch.from_numpy(mat.data.astype("float32")).to(dev),
                   mat.shape, symmetric=symmetric, chunk_nnz=chunk_nnz)

    @classmethod
    def from_torch_sparse(cls, adj, symmetric=None, chunk_nnz=0):
        """From a torch sparse COO/CSR tensor (the object pygcn builds in
        utils.py:243-250).  COO is coalesced and converted once."""
        if adj.layout == torch.sparse_coo:
            adj = adj.coalesce().to_sparse_csr()
        elif adj.layout != torch.sparse_csr:
            raise TypeError(f"unsupported layout {adj.layout}")
        return cls(adj.crow_indices(), adj.col_indices(), adj.values(), adj.shape,
                   symmetric=symmetric, chunk_nnz=chunk_nnz)

    # -- plan -------------------------------------------------------------------
    @property
    def plan(self):
        if self._plan is None:
            lib = _lib.load()
            handle = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                st = lib.gcn_spmm_plan_create(ctypes.byref(handle), _ptr(self.rowptr), self.m, self.n,
                                              self.nnz, self.chunk_nnz, _stream_ptr(self.device))
            _lib.check(st, "gcn_spmm_plan_create")
            self._plan = handle
            weakref.finalize(self, _destroy_plan, handle)
            if self.panels != 0:
                self.enable_panels(self.panels)
            if self.slices not in (0, 1) and self.panel_rows == 0:
                self.enable_slicing(self.slices)
        return self._plan

    @property
    def num_chunks(self):
        return int(_lib.load().gcn_spmm_plan_num_chunks(self.plan))

    @property
    def chunk_size(self):
        return int(_lib.load().gcn_spmm_plan_chunk_nnz(self.plan))

    def set_tile_cols(self, cols):
        """Feature-column tile per kernel pass (0 auto, 64, 128, 256)."""
        _lib.check(_lib.load().gcn_spmm_plan_set_tile_cols(self.plan, int(cols)), "gcn_spmm_plan_set_tile_cols")

    def set_gather_width(self, nz_per_gather):
        """Non-zeros per gather instruction of the 64-column kernel: 0 auto, 1 or 4."""
        _lib.check(_lib.load().gcn_spmm_plan_set_gather_width(self.plan, int(nz_per_gather)),
                   "gcn_spmm_plan_set_gather_width")

    def set_blocks_per_cu(self, blocks):
        """Main-kernel grid in blocks of 4 waves per CU (1..64, default 32 = oversubscribed); below the
        resident count (8, or 4 for the four-per-gather kernel) it leaves room for a concurrent kernel."""
        _lib.check(_lib.load().gcn_spmm_plan_set_blocks_per_cu(self.plan, int(blocks)),
                   "gcn_spmm_plan_set_blocks_per_cu")

    def prepare_width(self, k):
        """Build now whatever a k-wide call would build at first use (gcn_spmm_plan_prepare_width): e.g. before a capture."""
        with torch.cuda.device(self.device):
            st = _lib.load().gcn_spmm_plan_prepare_width(self.plan, _ptr(self.rowptr), _ptr(self.col), _ptr(self.val), int(k),
                                                         _stream_ptr(self.device))
        _lib.check(st, "gcn_spmm_plan_prepare_width")

    def enable_panels(self, mode):
        """LDS-staged row panels (gcn_spmm_plan_enable_panels): 0 off, 1 on, -1 automatic."""
        with torch.cuda.device(self.device):
            st = _lib.load().gcn_spmm_plan_enable_panels(self.plan, _ptr(self.rowptr), _ptr(self.col),
                                                         _ptr(self.val), int(mode), _stream_ptr(self.device))
        _lib.check(st, "gcn_spmm_plan_enable_panels")

    @property
    def panel_rows(self):
        return int(_lib.load().gcn_spmm_plan_panel_rows(self.plan))

    @property
    def dense_panels(self):
        """panels of the plan that run as dense tiles on the matrix cores (gcn_spmm_plan_dense_panels)"""
        return int(_lib.load().gcn_spmm_plan_dense_panels(self.plan))

    @property
    def panel_coverage(self):
        return float(_lib.load().gcn_spmm_plan_panel_coverage(self.plan))

    def enable_slicing(self, slices):
        """XCD-aware column slicing (gcn_spmm_plan_enable_slicing): 0/1 off, -1 automatic."""
        with torch.cuda.device(self.device):
            st = _lib.load().gcn_spmm_plan_enable_slicing(self.plan, _ptr(self.rowptr), _ptr(self.col),
                                                          _ptr(self.val), int(slices), _stream_ptr(self.device))
        _lib.check(st, "gcn_spmm_plan_enable_slicing")

    def set_value_factors(self, u_row, u_col):
        """Declare val[r, c] == u_row[r] * u_col[c] (checked on the device; GcnAmdError if an entry does not
        factor): lets the sliced main pass run without its value stream.  Square normalised adjacencies
        are detected automatically; this is for row blocks / renumbered columns.  (None, None) forgets."""
        if u_row is None and u_col is None:
            _lib.check(_lib.load().gcn_spmm_plan_set_value_factors(self.plan, None, None, None, None, None,
                                                                   _stream_ptr(self.device)), "gcn_spmm_plan_set_value_factors")
            return
        ur = u_row.to(device=self.device, dtype=torch.float32).contiguous()
        uc = u_col.to(device=self.device, dtype=torch.float32).contiguous()
        if ur.numel() != self.m or uc.numel() != self.n:
            raise ValueError("u_row needs m entries and u_col n entries")
        with torch.cuda.device(self.device):
            st = _lib.load().gcn_spmm_plan_set_value_factors(self.plan, _ptr(self.rowptr), _ptr(self.col), _ptr(self.val),
                                                             _ptr(ur), _ptr(uc), _stream_ptr(self.device))
        _lib.check(st, "gcn_spmm_plan_set_value_factors")

    @property
    def has_value_factors(self):
        return bool(_lib.load().gcn_spmm_plan_has_value_factors(self.plan))

    @property
    def num_slices(self):
        return int(_lib.load().gcn_spmm_plan_num_slices(self.plan))

    def narrow_slices_for(self, k):
        """slices of the slice set a k-wide call runs on when that is not the plan's own (k <= 32, 33..48: built at the first
        such call of a value-free, auto-sliced plan); 0: the plan's own"""
        return int(_lib.load().gcn_spmm_plan_narrow_slices(self.plan, int(k)))

    @property
    def narrow_slices(self):
        return self.narrow_slices_for(16)

    def num_passes(self, k):
        """main-kernel launches (column passes) one k-wide SpMM issues"""
        return int(_lib.load().gcn_spmm_plan_num_passes(self.plan, int(k)))

    def prelaid_layout(self, k):
        """The slice-by-slice, column-scaled feature layout B' a k-wide SpMM of this plan gathers from
        (gcn_spmm_plan_prelaid_layout) → dict(slices, slice_cols, table_rows, ld), or None when this plan does not take
        the value-free sliced pass for that width (then matmul_raw is the only form)."""
        S, w, ld = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        rows = ctypes.c_int64(0)
        vp = lambda x: ctypes.cast(ctypes.byref(x), ctypes.c_void_p)
        st = _lib.load().gcn_spmm_plan_prelaid_layout(self.plan, int(k), vp(S), vp(w), vp(rows), vp(ld))
        if st != 0:
            return None
        return dict(slices=S.value, slice_cols=w.value, table_rows=rows.value, ld=ld.value)

    def to_prelaid(self, dense, u_col):
        """B' for `dense` [n x k] (torch arithmetic; the first layer of a chain — later layers get theirs from
        matmul_prelaid): rows scaled by u_col, slice s at rows [s*(w+1), (s+1)*(w+1)), row w of every slice zero."""
        lay = self.prelaid_layout(int(dense.shape[1]))
        if lay is None:
            raise _lib.GcnAmdError("this plan has no pre-laid feature layout for that width")
        w, k = lay["slice_cols"], int(dense.shape[1])
        out = torch.zeros((lay["table_rows"], lay["ld"]), dtype=torch.float32, device=dense.device)
        c = torch.arange(self.n, device=dense.device)
        out[c + c // w, :k] = dense * u_col.to(dense.device)[:, None]
        return out

    def matmul_prelaid(self, Bp, out, out_scale=None, out_gap=0):
        """out[r + r // out_gap] = out_scale[r] * (Â·B)[r] with B handed over as B' (prelaid_layout): no per-call copy
        of the features, and the result lands in the B' layout of a consumer whose slices hold `out_gap` of these rows
        each (out_gap = 0: a plain [m x k] result).  gcn_spmm_csr_f32_prelaid."""
        lay = self.prelaid_layout(int(out.shape[1]))
        if lay is None or not (Bp.is_cuda and Bp.dtype == torch.float32 and Bp.is_contiguous()
                               and tuple(Bp.shape) == (lay["table_rows"], lay["ld"])):
            raise _lib.GcnAmdError("matmul_prelaid: B' must be the contiguous fp32 [table_rows x ld] array of prelaid_layout(k)")
        k = int(out.shape[1])
        need = self.m + (self.m - 1) // out_gap if (out_gap and self.m) else self.m
        if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] >= need):
            raise ValueError("out must be a contiguous fp32 array with room for every (gapped) row")
        sp = ctypes.c_void_p()
        if out_scale is not None:
            if not (out_scale.is_cuda and out_scale.dtype == torch.float32 and out_scale.is_contiguous()
                    and out_scale.numel() == self.m):
                raise ValueError("out_scale must be a contiguous fp32 device vector with m entries")
            sp = _ptr(out_scale)
        with torch.cuda.device(self.device):
            st = _lib.load().gcn_spmm_csr_f32_prelaid(self.plan, _ptr(self.rowptr), _ptr(self.col), _ptr(self.val),
                                                      _ptr(Bp), _ptr(out), sp, int(out_gap), k, _stream_ptr(self.device))
        _lib.check(st, "gcn_spmm_csr_f32_prelaid")
        return out

    def autotune(self, k=128, reps=3, verbose=False):
        """Measure, don't guess: time a k-wide SpMM on this matrix in the plan shapes that can win and keep the fastest —
        column slices {automatic, 8, none} and, unsliced and for k > 64, the column tile per pass {automatic, 64, 128}.
        The automatic rules (auto_slices, auto_tile_cols) are derived from unordered graphs; a matrix whose rows were
        renumbered to sit near their neighbours (Rabbit on a graph with communities) can be faster unsliced,
        and when its table is far larger than the caches, in NARROW tiles: a community's slice of a 64-column tile fits an
        L2 where its 256-column rows do not (products-sized planted partition, Rabbit order, k = 256: 15.8 ms with the
        widest tile, 13.0 ms with 64-column tiles; un-renumbered: 20.1 ms, profiles/r04r_*).
        → dict {(slices, tile_cols): ms} sorted by time; the first entry is what stays configured."""
        k = int(k)
        B = torch.randn((self.n, k), dtype=torch.float32, device=self.device)
        out = torch.empty((self.m, k), dtype=torch.float32, device=self.device)
        tried = {}
        shapes = [(-1, 0), (8, 0), (0, 0)] + ([(0, 64)] if k > 64 else []) + ([(0, 128)] if k > 128 else [])
        for S, tile in shapes:
            try:
                self.enable_slicing(S)
                self.set_tile_cols(tile)
            except _lib.GcnAmdError:
                continue                                       # (unsorted rows, S*m too large, ...)
            key = (self.num_slices, tile)
            if key in tried:
                continue
            for _i in range(2):
                self.matmul_raw(B, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _i in range(reps):
                self.matmul_raw(B, out=out)
            e1.record()
            torch.cuda.synchronize(self.device)
            tried[key] = e0.elapsed_time(e1) / reps
            if verbose:
                print(f"autotune: slices={key[0]} tile={key[1]}: {tried[key]:.4f} ms")
        best = min(tried, key=tried.get)
        self.enable_slicing(best[0])
        self.set_tile_cols(best[1])
        self.slices, self.tile_cols = best
        return dict(sorted(tried.items(), key=lambda kv: kv[1]))

    def main_kernel(self, k, epilogue=False):
        """name of the main kernel a k-wide SpMM on this plan launches (as rocprofv3 prints it)"""
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().gcn_spmm_plan_main_kernel(self.plan, int(k), int(bool(epilogue)), buf, 128),
                   "gcn_spmm_plan_main_kernel")
        return buf.value.decode()

    def profile_begin(self, capacity):
        """Record HIP-event pairs around the main kernel of the next `capacity` launches."""
        _lib.check(_lib.load().gcn_spmm_profile_begin(self.plan, int(capacity)), "gcn_spmm_profile_begin")
        self._prof_cap = int(capacity)

    def profile_end(self):
        """→ list of per-launch main-kernel durations in ms (synchronises)."""
        buf = (ctypes.c_float * self._prof_cap)()
        cnt = ctypes.c_int32(0)
        _lib.check(_lib.load().gcn_spmm_profile_end(self.plan, ctypes.cast(buf, ctypes.c_void_p),
                                                    ctypes.cast(ctypes.byref(cnt), ctypes.c_void_p)),
                   "gcn_spmm_profile_end")
        return [float(buf[i]) for i in range(cnt.value)]

    def transpose(self):
        """Âᵀ as a CsrAdjacency (cached); Â itself when flagged symmetric — the
        reference's backward reuses Â because Â is symmetric (gcn6.py:50-62)."""
        if self.symmetric:
            return self
        if self._transpose is None:
            csr = torch.sparse_csr_tensor(self.rowptr.long(), self.col.long(), self.val,
                                          size=(self.m, self.n))
            t = csr.to_sparse_coo().t().coalesce().to_sparse_csr()
            self._transpose = CsrAdjacency(t.crow_indices(), t.col_indices(), t.values(),
                                           (self.n, self.m), symmetric=False,
                                           chunk_nnz=self.chunk_nnz)
            self._transpose._transpose = self
        return self._transpose

    # -- the op ------------------------------------------------------------------
    def matmul_raw(self, dense, out=None, bias=None, relu=False, dropout=None):
        """C = dropout(act(Â·dense + bias)) with no autograd; dense is [n x k] fp32 on the same device.
        dropout = (p, seed, offset): the mask of gcn_spmm_csr_f32_epilogue (element i kept iff its Philox word
        passes; kept values scaled by 1/(1-p)); `dropout_rows` applies the same mask to a gradient."""
        if not dense.is_cuda or dense.dtype != torch.float32 or dense.dim() != 2:
            raise _lib.GcnAmdError("dense operand must be a 2-D fp32 CUDA/HIP tensor")
        if dense.shape[0] != self.n:
            raise ValueError(f"shape mismatch: A is {self.m}x{self.n}, B is {tuple(dense.shape)}")
        dense = dense.contiguous()
        k = int(dense.shape[1])
        if out is None:
            out = torch.empty((self.m, k), dtype=torch.float32, device=dense.device)
        elif not (out.is_contiguous() and out.shape == (self.m, k) and out.dtype == torch.float32):
            raise ValueError("out must be a contiguous fp32 [m x k] tensor")
        lib = _lib.load()
        with torch.cuda.device(self.device):
            if dropout is not None and float(dropout[0]) > 0.0:
                bp = _ptr(bias.contiguous()) if bias is not None else ctypes.c_void_p()
                st = lib.gcn_spmm_csr_f32_epilogue(self.plan, _ptr(self.rowptr), _ptr(self.col), _ptr(self.val),
                                                   _ptr(dense), _ptr(out), bp, 1 if relu else 0, float(dropout[0]),
                                                   int(dropout[1]), int(dropout[2]), k, _stream_ptr(self.device))
            elif bias is None and not relu:
                st = lib.gcn_spmm_csr_f32(self.plan, _ptr(self.rowptr), _ptr(self.col), _ptr(self.val),
                                          _ptr(dense), _ptr(out), k, _stream_ptr(self.device))
            else:
                bp = _ptr(bias.contiguous()) if bias is not None else ctypes.c_void_p()
                st = lib.gcn_spmm_csr_f32_bias_relu(self.plan, _ptr(self.rowptr), _ptr(self.col),
                                                    _ptr(self.val), _ptr(dense), _ptr(out), bp,
                                                    1 if relu else 0, k, _stream_ptr(self.device))
        _lib.check(st, "gcn_spmm_csr_f32")
        return out


def _destroy_plan(handle):
    try:
        _lib.load().gcn_spmm_plan_destroy(handle)
    except Exception:  # interpreter shutdown
        pass


class _SpmmFunction(torch.autograd.Function):
    """Autograd wrapper: grad_dense = Âᵀ · grad_out (Â itself when symmetric, as in
    gcn6.py:50-62); no gradient w.r.t. the adjacency (the reference has none either)."""

    @staticmethod
    def forward(ctx, adj, dense):
        ctx.adj = adj
        return adj.matmul_raw(dense)

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.needs_input_grad[1]:                # (a constant operand, e.g. the input features: no Âᵀ·grad to compute)
            return None, None
        return None, ctx.adj.transpose().matmul_raw(grad_out.contiguous())


def spmm(adj, dense):
    """C = adj @ dense on the HIP kernel.  `adj` is a CsrAdjacency or a torch sparse
    (COO/CSR) fp32 tensor on the GPU (converted and cached per tensor object)."""
    if not isinstance(adj, CsrAdjacency):
        adj = _cached_csr(adj)
    return _SpmmFunction.apply(adj, dense)


# --- routing of torch.spmm / torch.sparse.mm (interface B2) -----------------------
_csr_cache = {}          # id(sparse tensor) -> (weakref, CsrAdjacency)
_orig = {}


def _cached_csr(adj):
    key = id(adj)
    hit = _csr_cache.get(key)
    if hit is not None and hit[0]() is adj:
        return hit[1]
    csr = CsrAdjacency.from_torch_sparse(adj)
    _csr_cache[key] = (weakref.ref(adj, lambda _r, k=key: _csr_cache.pop(k, None)), csr)
    return csr


def _routable(a, b):
    return (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor)
            and a.layout in (torch.sparse_coo, torch.sparse_csr) and a.is_cuda
            and a.dtype == torch.float32 and a.dim() == 2
            and b.layout == torch.strided and b.is_cuda and b.dtype == torch.float32 and b.dim() == 2
            and not a.requires_grad)


def install():
    """Route ``torch.spmm`` / ``torch.sparse.mm`` for (GPU sparse fp32) × (GPU dense fp32)
    to the HIP kernel; anything else goes to the original callables."""
    if _orig:
        return
    _lib.load()  # fail loudly now, not at the first layer
    _orig["spmm"] = torch.spmm
    _orig["sparse_mm"] = torch.sparse.mm

    def routed_spmm(a, b, *args, **kw):
        if not args and not kw and _routable(a, b):
            return spmm(a, b)
        return _orig["spmm"](a, b, *args, **kw)

    def routed_sparse_mm(a, b, *args, **kw):
        if not args and not kw and _routable(a, b):
            return spmm(a, b)
        return _orig["sparse_mm"](a, b, *args, **kw)

    torch.spmm = routed_spmm
    torch.sparse.mm = routed_sparse_mm


def uninstall():
    if not _orig:
        return
    torch.spmm = _orig.pop("spmm")
    torch.sparse.mm = _orig.pop("sparse_mm")


def gather_rows(src, idx, out=None):
    """out[r,:] = src[idx[r],:]  (permutate.cu:3-21 counterpart), fp32 on the GPU."""
    if not (src.is_cuda and idx.is_cuda) or src.dtype != torch.float32 or src.dim() != 2:
        raise _lib.GcnAmdError("gather_rows needs a 2-D fp32 CUDA/HIP tensor and a CUDA index")
    src = src.contiguous()
    idx = idx.to(torch.int32).contiguous()
    nrows, k = int(idx.numel()), int(src.shape[1])
    if out is None:
        out = torch.empty((nrows, k), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        st = _lib.load().gcn_gather_rows_f32(_ptr(out), _ptr(src), _ptr(idx), nrows, k,
                                             _stream_ptr(src.device))
    _lib.check(st, "gcn_gather_rows_f32")
    return out


def dropout_rows(x, p, seed, offset, out=None):
    """out = x with the dropout mask of the fused epilogue (gcn_dropout_f32): element i of the contiguous fp32 tensor
    is kept (and scaled by 1/(1-p)) iff its Philox4x32-10 word passes — the same function of (seed, offset, i) the
    SpMM epilogue uses, so applying it to a gradient is the backward of that epilogue."""
    if not x.is_cuda or x.dtype != torch.float32:
        raise _lib.GcnAmdError("dropout_rows needs an fp32 CUDA/HIP tensor")
    x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        st = _lib.load().gcn_dropout_f32(_ptr(out), _ptr(x), int(x.numel()), float(p), int(seed), int(offset),
                                         _stream_ptr(x.device))
    _lib.check(st, "gcn_dropout_f32")
    return out
