"""gcn_amd — MI355X-native GCN aggregation (CSR SpMM) behind the pygcn op surface.

    import gcn_amd
    gcn_amd.install()                 # torch.spmm / torch.sparse.mm → HIP kernel (gcn1–5)
    adj = gcn_amd.CsrAdjacency.from_scipy(A_hat)   # explicit handle
    C = gcn_amd.spmm(adj, X)

Drop-in shared objects for gcn6.py live in gcn_amd/dropin/ (see INTEGRATION.md).
"""
from ._lib import GcnAmdError, LIB_PATH, DROPIN_DIR, load as load_library  # noqa: F401
from .spmm import CsrAdjacency, spmm, gather_rows, dropout_rows, install, uninstall  # noqa: F401
from . import reorder, dropin  # noqa: F401
from .layers import GCN, GraphConvolution, GraphConvolution2  # noqa: F401

__version__ = "0.3.0"
