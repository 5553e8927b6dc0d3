"""Python mirror of the gcn6 op surface (interface B1, pygcn/gcn6.py:21-62,334-377).

gcn6.py itself binds the five shared objects under ``gcn_amd/dropin/`` by file name
(``ctypes.cdll.LoadLibrary('./flexspmm.so')`` …) and needs no Python from here; this
module exists so that the parity tests and users can drive the SAME C symbols with
the same argument order, tensor shapes and calling convention as gcn6.py does.
"""
import ctypes

import torch

from . import _lib


def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


def csr2tile(adj_rowPtr, adj_col, adj_values, m, n, nnz, vo_mp, tm=8):
    """Host tiling step, same buffers as gcn6.py:334-354.  Returns
    (seg_rowPtr, segNzCV, segVoMap, grouped_tailSeg, next_seg, n_segs) as CPU tensors,
    already shrunk the way gcn6.py:353-354 does."""
    seg_rowPtr = torch.empty(nnz, dtype=torch.int32)
    segNzCV = torch.empty(2 * nnz, dtype=torch.float32)
    segVoMap = torch.empty(nnz, dtype=torch.int32)
    grouped_tailSeg = torch.empty(256, dtype=torch.int32)
    next_seg = torch.empty(256, dtype=torch.int32)
    n_segs = torch.zeros(1, dtype=torch.int32)
    _lib.load().csr2tile(_vp(adj_rowPtr), _vp(adj_col), _vp(adj_values), m, n, nnz, _vp(vo_mp),
                         _vp(segVoMap), _vp(seg_rowPtr), _vp(segNzCV), _vp(grouped_tailSeg),
                         _vp(next_seg), tm, _vp(n_segs))
    seg_rowPtr.resize_((tm + 1) * int(n_segs[0]))
    segVoMap.resize_(tm * int(n_segs[0]))
    return seg_rowPtr, segNzCV, segVoMap, grouped_tailSeg, next_seg, n_segs


class flexspmm(torch.autograd.Function):
    """Same signature and semantics as the reference's Function (gcn6.py:34-62):
    zero-initialised output, fresh copy of next_seg per call, backward = the same
    op on grad_out (valid because Â is symmetric)."""

    @staticmethod
    def forward(ctx, seg_rowPtr, segNzCV, segVoMap, m, n, n_segs, grouped_tailSeg, next_seg, input):
        output = torch.zeros((m, input.shape[1]), device=input.device)
        next_seg1 = next_seg.clone()
        _lib.load().flexspmm(_vp(seg_rowPtr), _vp(segNzCV), _vp(segVoMap), _vp(grouped_tailSeg),
                             _vp(next_seg1), m, n, input.shape[1], n_segs,
                             _vp(input.contiguous()), _vp(output))
        ctx.backward_flex = seg_rowPtr, segNzCV, segVoMap, m, n, n_segs, grouped_tailSeg, next_seg
        return output

    @staticmethod
    def backward(ctx, grad_out):
        seg_rowPtr, segNzCV, segVoMap, m, n, n_segs, grouped_tailSeg, next_seg = ctx.backward_flex
        grad_x = torch.zeros((m, grad_out.shape[1]), device=grad_out.device)
        next_seg1 = next_seg.clone()
        _lib.load().flexspmm(_vp(seg_rowPtr), _vp(segNzCV), _vp(segVoMap), _vp(grouped_tailSeg),
                             _vp(next_seg1), m, n, grad_out.shape[1], n_segs,
                             _vp(grad_out.contiguous()), _vp(grad_x))
        return None, None, None, None, None, None, None, None, grad_x


def permutate(features_dev, vo_mp_dev, labels_dev, m, n, k):
    """permutate.so:permutate — in-place B[r,:] ← B[vo_mp[r],:] (gcn6.py:374-377)."""
    _lib.load().permutate(_vp(features_dev), _vp(vo_mp_dev), _vp(labels_dev), m, n, k)


def cuspmm(rowptr_dev, col_dev, vals_dev, X_dev, C_dev, m, n, nnz, dim):
    """cuspmm.so:cuspmm (cuspmm.cu:23-24; dead call sites gcn5.py:77-82, gcn6.py:120-125)."""
    _lib.load().cuspmm(_vp(rowptr_dev), _vp(col_dev), _vp(vals_dev), _vp(X_dev), _vp(C_dev),
                       m, n, nnz, dim)
