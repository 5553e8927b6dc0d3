"""The callers of the aggregation op, mirrored from pygcn/gcn6.py on top of the native API:

* ``GraphConvolution``  — A(XW) layer  (gcn6.py:66-148)
* ``GraphConvolution2`` — (AX)W layer  (gcn6.py:151-199)
* ``GCN``               — the 2-layer model with gcn6's four preprocessing steps in ``fit``
                          (renumber → schedule → to GPU → permute features; gcn6.py:262-410)

Same constructor arguments, parameter initialisation, layer-order rule (A(XW) for layer 2 on
pubmed/flickr, (AX)W otherwise, gcn6.py:214-218), per-layer ``xw / af / bi`` timers and training
loop (Adam, nll_loss on the training rows).  Differences are the ones the path forces: the
adjacency is a ``CsrAdjacency`` handle instead of the nine tile arguments, labels ARE permuted with
the features (the reference forgets to, SURVEY defect D4), and the bias add (+ ReLU for layer 1)
can ride in the SpMM epilogue (``fuse_epilogue=True``).
"""
import math

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from . import preprocess, reorder, timers
from .spmm import CsrAdjacency, _SpmmFunction, dropout_rows, gather_rows


class _FusedSpmmBiasRelu(torch.autograd.Function):
    """out = dropout(relu(Â·X + b)) in one pass over the SpMM output (gcn6.py:141-142, 245-246 as one epilogue);
    backward through the regenerated dropout mask, the ReLU mask, Âᵀ and the bias sum.  dropout = None or
    (p, seed, offset)."""

    @staticmethod
    def forward(ctx, adj, x, bias, relu, dropout=None):
        out = adj.matmul_raw(x, bias=bias, relu=relu, dropout=dropout)
        ctx.adj, ctx.relu, ctx.has_bias, ctx.dropout = adj, relu, bias is not None, dropout
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        if ctx.dropout is not None and ctx.dropout[0] > 0:
            g = dropout_rows(g, *ctx.dropout)          # the same mask, the same 1/(1-p)
        if ctx.relu:
            g = g * (out > 0)                          # (dropped elements are 0 in `out`, and their gradient is 0 already)
        g = g.contiguous()
        gx = ctx.adj.transpose().matmul_raw(g) if ctx.needs_input_grad[1] else None
        return None, gx, (g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None), None, None


class _Layer(nn.Module):
    def __init__(self, in_features, out_features, with_bias=True, name="dataset", layer="layer0"):
        super().__init__()
        self.in_features, self.out_features, self.layer = in_features, out_features, layer
        self.weight = Parameter(torch.empty(in_features, out_features))
        if with_bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()
        self.timers = timers.Timers()

    def reset_parameters(self):                    # gcn6.py:86-94
        stdv = 1.0 / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def reset_timing(self):
        self.timers.reset()

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.in_features} -> {self.out_features})"


class GraphConvolution(_Layer):
    """A(XW): support = X·W, output = Â·support (+ bias)  — gcn6.py:99-143."""

    def forward(self, input, adj, relu=False, fuse_epilogue=False, dropout=None):
        with self.timers.hc.xw:
            support = torch.spmm(input, self.weight) if input.is_sparse else torch.mm(input, self.weight)
        if fuse_epilogue:
            with self.timers.hc.af:
                return _FusedSpmmBiasRelu.apply(adj, support, self.bias, relu, dropout)
        with self.timers.hc.af:
            output = _SpmmFunction.apply(adj, support)
        if self.bias is not None:
            with self.timers.hc.bi:
                output = output + self.bias
        return F.relu(output) if relu else output


class GraphConvolution2(_Layer):
    """(AX)W: support = Â·X, output = support·W (+ bias)  — gcn6.py:184-194."""

    def forward(self, input, adj, relu=False, fuse_epilogue=False):
        with self.timers.hc.af:
            support = _SpmmFunction.apply(adj, input)
        with self.timers.hc.xw:
            output = torch.mm(support, self.weight)
        if self.bias is not None:
            with self.timers.hc.bi:
                output = output + self.bias
        return F.relu(output) if relu else output


class GCN(nn.Module):
    """2-layer GCN with the gcn6 training flow on the native SpMM (gcn6.py:201-441)."""

    def __init__(self, nfeat, nhid, nclass, dataset="dataset", dropout=0.5, lr=0.01, weight_decay=5e-4,
                 with_relu=True, with_bias=True, device=None, order="rabbit", fuse_epilogue=False,
                 layer_order="reference", precompute_ax=False):
        super().__init__()
        assert device is not None, "Please specify 'device'!"
        self.device, self.nfeat, self.hidden_sizes, self.nclass = device, nfeat, [nhid], nclass
        self.dataname = dataset
        self.gc1 = GraphConvolution(nfeat, nhid, with_bias=with_bias, name=dataset, layer="layer1")
        # Â(XW) or (ÂX)W for layer 2: the reference hard-codes it per dataset (gcn6.py:214-218);
        # layer_order="auto" runs the SpMM at the narrower of the two widths instead —
        # Reddit-shaped, hidden 128 -> 41 classes: SpMM at k = 41 (2.0 ms) instead of k = 128 (3.7 ms)
        if layer_order not in ("reference", "auto"):
            raise ValueError("layer_order must be 'reference' or 'auto'")
        a_xw = (nclass <= nhid) if layer_order == "auto" else dataset in ("pubmed", "flickr")
        if a_xw:
            self.gc2 = GraphConvolution(nhid, nclass, with_bias=with_bias, name=dataset, layer="layer2")
        else:
            self.gc2 = GraphConvolution2(nhid, nclass, with_bias=with_bias, name=dataset, layer="layer2")
        self.dropout, self.lr = dropout, lr
        self.weight_decay = weight_decay if with_relu else 0
        self.with_relu, self.with_bias = with_relu, with_bias
        self.order = order                    # None | "dfs" | "gorder" | "rabbit" (gcn6.py:27-30: RBT default) | "rcm" | "deg" | "rabbit_device" (GPU)
        self.fuse_epilogue = fuse_epilogue
        # precompute_ax: layer 1 is Â·(X·W1) with X the CONSTANT input features (the reference applies dropout behind layer 1,
        # gcn6.py:245-246, never to X), and Â(XW) = (ÂX)W: ÂX is aggregated ONCE (one SpMM at the input width) and every
        # epoch's layer 1 is a dense product — no SpMM in its forward pass and none in its backward pass (W1's gradient is
        # (ÂX)ᵀ·g).  Same function, fp32 rounding apart; n x nfeat floats of memory.  Off by default (the reference recomputes).
        self.precompute_ax = precompute_ax
        self._ax = None
        self.output = None
        self.adj = self.features = self.labels = self.vo_mp = None
        self.tuning = None                    # {(slices, column tile): ms} measured by prepare() on a renumbered graph
        # fused-epilogue dropout: Philox keyed on (seed, offset).  The seed is drawn from torch's default generator at
        # the first training forward (so torch.manual_seed governs it like F.dropout's masks, two models or two
        # restarts differ unless seeded alike), the offset counts forward passes; both are part of state_dict-less
        # extra state (get_extra_state) so a checkpoint resumes the same mask sequence
        self.dropout_seed, self._dropout_calls = None, 0
        # ... and OPTIONAL on load: a reference-format checkpoint holds the four weight tensors only
        # (profiling_gcn.py:165-170: torch.save(model.state_dict()) / load_state_dict), as do checkpoints of this class
        # written before the extra state existed; a strict load of those must not fail on a missing "_extra_state"
        self._register_load_state_dict_pre_hook(self._default_extra_state)
        self.dur_fwd = timers.Timer()

    @staticmethod
    def _default_extra_state(state_dict, prefix, *_unused):
        state_dict.setdefault(prefix + "_extra_state", {"dropout_seed": None, "dropout_calls": 0})

    def reset_timing(self):
        self.dur_fwd.reset()
        for gc in (self.gc1, self.gc2):
            gc.reset_timing()

    def _layer1_from_cached_ax(self, x, adj):
        if self._ax is None:
            with torch.no_grad():
                self._ax = adj.matmul_raw(x.contiguous())
        with self.gc1.timers.hc.xw:
            h = torch.mm(self._ax, self.gc1.weight)
        if self.gc1.bias is not None:
            with self.gc1.timers.hc.bi:
                h = h + self.gc1.bias
        return F.relu(h) if self.with_relu else h

    def forward(self, x, adj):
        with self.dur_fwd:
            if self.precompute_ax and x is self.features and adj is self.adj and not x.requires_grad:
                x = self._layer1_from_cached_ax(x, adj)
                x = F.dropout(x, self.dropout, training=self.training)
                x = self.gc2(x, adj)
                return F.log_softmax(x, dim=1)
            # (under HIP-graph capture the Philox offset, a host integer, would be frozen into the graph — every replay the
            #  same mask; torch's own dropout draws from the generator state the graph registers, so it takes over there)
            if self.fuse_epilogue and self.training and self.dropout > 0 and not torch.cuda.is_current_stream_capturing():
                # bias + ReLU + dropout mask in the SpMM epilogue; a fresh Philox offset per forward pass
                if self.dropout_seed is None:
                    self.dropout_seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
                self._dropout_calls += 1
                x = self.gc1(x, adj, relu=self.with_relu, fuse_epilogue=True,
                             dropout=(float(self.dropout), self.dropout_seed, self._dropout_calls))
            else:
                x = self.gc1(x, adj, relu=self.with_relu, fuse_epilogue=self.fuse_epilogue)
                x = F.dropout(x, self.dropout, training=self.training)
            x = self.gc2(x, adj)
            return F.log_softmax(x, dim=1)

    def get_extra_state(self):
        return {"dropout_seed": self.dropout_seed, "dropout_calls": self._dropout_calls}

    def set_extra_state(self, state):
        self.dropout_seed, self._dropout_calls = state.get("dropout_seed"), int(state.get("dropout_calls", 0))

    def initialize(self):
        self.gc1.reset_parameters()
        self.gc2.reset_parameters()

    def prepare(self, features, adj, labels, normalize=True):
        """gcn6.fit steps 1-4 (gcn6.py:271-379): normalise, renumber on the host, build the SpMM
        schedule, move to the GPU, permute features (and labels)."""
        if sp.issparse(features):
            features = np.asarray(features.todense())
        features = torch.as_tensor(np.asarray(features), dtype=torch.float32)
        adj_norm = preprocess.normalize_adj_tensor(adj) if normalize else preprocess.sparse_mx_to_torch_sparse_tensor(adj)
        rp, ci, va, vo_mp = preprocess.to_csr_int32(adj_norm)
        dev = torch.device(self.device)
        if self.order in ("rcm", "deg", "communities", "rabbit_device"):          # step 1 on the GPU
            # the reference library's internal orderings (order_rcm.cu, order_deg.cu), computed by the
            # device kernels — same integers as the host code, ~100x faster (reorder_device.hip)
            rp, ci, va = rp.to(dev), ci.to(dev), va.to(dev)
            rank = (reorder.order_rcm_device(rp, ci) if self.order == "rcm"
                    else reorder.order_communities_device(rp, ci) if self.order == "communities"
                    else reorder.order_rabbit_device(rp, ci) if self.order == "rabbit_device"    # parallel Rabbit (Arai'16)
                    else reorder.order_deg_device(rp, ci, "total", True))
            rp, ci, va, vo_mp = reorder.apply_rank_device(rp, ci, va, rank)
        else:
            rp, ci, va, vo_mp = rp.numpy(), ci.numpy(), va.numpy(), vo_mp.numpy()
            if self.order:                                                       # step 1 on the host
                rp, ci, va, vo_mp = getattr(reorder, self.order)(rp, ci, va)
            rp, ci, va, vo_mp = (torch.from_numpy(x).to(dev) for x in (rp, ci, va, vo_mp))
        n = rp.numel() - 1
        self.adj = CsrAdjacency(rp, ci, va, (n, n), symmetric=True)             # steps 2+3
        if self.order:
            # A renumbered graph may sit near its diagonal (Rabbit / RCM / Gorder on a graph with communities):
            # the XCD-contiguous chunk ranges then already keep neighbours in one L2 and the column slicing the
            # automatic rule picks for UNORDERED graphs of this size can lose to the plain pass (planted
            # partition, n = 60 k, Rabbit: 0.55 ms unsliced, 0.73 ms sliced).  Measure once, keep the faster.
            self.tuning = self.adj.autotune(k=max(self.hidden_sizes[0], 64))
        self.vo_mp = vo_mp
        self._ax = None
        self.features = gather_rows(features.to(dev), self.vo_mp)                # step 4
        self.labels = torch.as_tensor(np.asarray(labels), dtype=torch.int64).to(dev)[self.vo_mp.long()]
        inv = torch.empty(n, dtype=torch.int64)
        inv[vo_mp.cpu().long()] = torch.arange(n)
        self._new_index = inv                 # old vertex id -> row in the renumbered graph (gcn6.py:255-260)
        return self

    def fit(self, features, adj, labels, idx_train, train_iters=200, initialize=True, verbose=False,
            normalize=True, reuse_prepared=False, hip_graph=False):
        """gcn6.fit (gcn6.py:262-410).  hip_graph=True: the training step — forward, loss, backward, Adam — is captured
        once in a HIP graph after three eager iterations and replayed for the rest: for graphs small enough that an epoch
        is a few dozen launch-bound kernels (Cora-, Pubmed-shaped), where the launches, not the kernels, set the pace."""
        if initialize:
            self.initialize()
        if not (reuse_prepared and self.adj is not None):   # (a second fit on the same graph keeps steps 1-4)
            self.prepare(features, adj, labels, normalize)
        idx = self._new_index[torch.as_tensor(np.asarray(idx_train)).long()].to(self.labels.device)
        self.train()
        if hip_graph:
            return self._fit_captured(idx, train_iters, verbose)
        opt = torch.optim.Adam(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        ti = timers.Timers()
        losses = []
        for i in range(train_iters):
            opt.zero_grad()
            with ti.h.fwd:
                output = self.forward(self.features, self.adj)
            loss = F.nll_loss(output[idx], self.labels[idx])
            with ti.h.bwd:
                loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
            if verbose and i % 10 == 0:
                print(f"Epoch {i:3d}, training loss: {losses[-1]:.6f}  Fwd: {ti.h.fwd.avms():.3f} ms/iter"
                      f"  Bwd: {ti.h.bwd.avms():.3f} ms/iter")
                ti.reset()
        self.output = output.detach()       # (detached: a live autograd graph would keep this run's AccumulateGrad nodes —
        return losses                       #  and the stream they were made on — alive into the next fit, see _fit_captured)

    def _fit_captured(self, idx, train_iters, verbose):
        dev = self.labels.device
        # Nothing may keep an autograd graph of an earlier (eager, default-stream) run alive: its AccumulateGrad nodes would
        # be reused, and their work on the default stream inside the capture is what hipStreamEndCapture dies of.
        self.output = None
        for p_ in self.parameters():
            p_.grad = None
        opt = torch.optim.Adam(self.parameters(), lr=self.lr, weight_decay=self.weight_decay, capturable=True)
        target = self.labels[idx]
        losses = torch.zeros(train_iters, dtype=torch.float32, device=dev)

        def step():
            output = self.forward(self.features, self.adj)
            loss = F.nll_loss(output[idx], target)
            loss.backward()
            opt.step()
            return output, loss

        # eager iterations first, on the stream the capture will use: every workspace of the SpMM plans (forward and
        # backward widths) and Adam's state exist before the capture begins
        warm = min(3, train_iters)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        output = None
        with torch.cuda.stream(side):
            for i in range(warm):
                opt.zero_grad(set_to_none=True)
                output, loss = step()
                losses[i] = loss.detach()
            if train_iters > warm:
                opt.zero_grad(set_to_none=True)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    output, loss = step()
                for i in range(warm, train_iters):
                    graph.replay()
                    losses[i].copy_(loss.detach())
        cur.wait_stream(side)
        self.output = output.detach() if output is not None else None
        out = [float(v) for v in losses.tolist()]           # one synchronisation, at the end
        if verbose:
            print(f"captured fit: {train_iters} iterations ({warm} eager), loss {out[0]:.6f} -> {out[-1]:.6f}")
        return out

    def timing_report(self):
        """the per-layer lines gcn6 prints after fit (gcn6.py:401-410)"""
        lines = [f"Forward time: {self.dur_fwd.s():.4f} s for {self.dur_fwd.n_calls} calls."]
        for gc in (self.gc1, self.gc2):
            t = gc.timers
            lines.append(f"{gc.layer} xw: {t.h.xw.avms():6.4f} ms  cu {t.c.xw.avms():6.4f} ms "
                         f" af: {t.h.af.avms():7.4f} ms  cu {t.c.af.avms():7.4f} ms "
                         f" bi: {t.h.bi.avms():7.4f} ms  cu {t.c.bi.avms():7.4f} ms")
        return "\n".join(lines)

    @torch.no_grad()
    def predict(self):
        """log-probabilities in the ORIGINAL vertex order"""
        self.eval()
        out = self.forward(self.features, self.adj)
        return out[self._new_index.to(out.device)]

    def test(self, idx_test, labels):
        out = self.predict()
        idx = torch.as_tensor(np.asarray(idx_test)).long().to(out.device)
        lab = torch.as_tensor(np.asarray(labels), dtype=torch.int64).to(out.device)
        return preprocess.accuracy(out[idx], lab[idx])
