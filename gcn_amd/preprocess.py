"""Host-side preprocessing the SpMM inputs go through in the reference
(pygcn/gcnio/util/utils.py:78-90, 126-164, 243-250) — restated so that a graph handed to
gcn_amd carries exactly the values pygcn would feed torch.spmm."""
import numpy as np
import scipy.sparse as sp
import torch


def normalize_adj(mx):
    """Â = D^-1/2 (A [+ I]) D^-1/2 in fp64 (utils.py:78-90).  Like the reference, self-loops are
    added only when ``mx[0, 0] == 0`` — the odd test at utils.py:82 is kept on purpose."""
    mx = sp.lil_matrix(mx) if not sp.isspmatrix_lil(mx) else mx
    if mx[0, 0] == 0:
        mx = mx + sp.eye(mx.shape[0])
    rowsum = np.array(mx.sum(1))
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1 / 2).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    r_mat_inv = sp.diags(r_inv)
    return r_mat_inv.dot(mx).dot(r_mat_inv)


def sparse_mx_to_torch_sparse_tensor(sparse_mx):
    """scipy → torch sparse COO fp32 with int64 indices (utils.py:243-250)."""
    sparse_mx = sparse_mx.tocoo().astype(np.float32)
    indices = torch.from_numpy(np.vstack((sparse_mx.row, sparse_mx.col)).astype(np.int64))
    return torch.sparse_coo_tensor(indices, torch.from_numpy(sparse_mx.data), torch.Size(sparse_mx.shape))


def normalize_adj_tensor(adj_scipy, device="cpu"):
    """utils.normalize_adj_tensor2(adj, sparse=True) (utils.py:154-155): the sparse COO fp32 Â
    gcn6.fit builds at gcn6.py:281."""
    return sparse_mx_to_torch_sparse_tensor(normalize_adj(adj_scipy)).to(device)


def to_csr_int32(adj_norm):
    """gcn6.py:302-312: COO → CSR, crow/col cast to int32, values fp32, vo_mp = arange(m)."""
    csr = adj_norm.coalesce().to_sparse_csr()
    m = csr.crow_indices().shape[0] - 1
    return (csr.crow_indices().to(torch.int32), csr.col_indices().to(torch.int32), csr.values(),
            torch.arange(m, dtype=torch.int32))


def accuracy(output, labels):
    """utils.py:214-220."""
    if not isinstance(labels, torch.Tensor):
        labels = torch.LongTensor(labels)
    preds = output.max(1)[1].type_as(labels)
    return preds.eq(labels).double().sum() / len(labels)
