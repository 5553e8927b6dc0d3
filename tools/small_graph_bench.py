"""tools/small_graph_bench.py — per-call time of one SpMM on small graphs (Cora- / Pubmed-shaped and up): this library
against torch.sparse.mm on the same GPU (COO and CSR operands).  Development tool."""
import os, time, torch, numpy as np, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd
from gcn_amd import graphgen
d = torch.device("cuda:0")
for (n, deg, k) in [(2485, 5, 16), (2485, 5, 64), (19717, 5, 16), (50000, 20, 64), (232965//8, 100, 128)]:
    g = torch.Generator().manual_seed(0)
    nnz = n * deg
    r = torch.randint(0, n, (nnz,), generator=g); c = torch.randint(0, n, (nnz,), generator=g)
    r = torch.cat([r, c, torch.arange(n)]); c = torch.cat([c, r[:nnz], torch.arange(n)])
    A = torch.sparse_coo_tensor(torch.stack([r, c]), torch.ones(r.numel()), (n, n)).coalesce()
    A = torch.sparse_coo_tensor(A.indices(), torch.ones(A.values().numel()), (n, n)).coalesce()
    dg = torch.sparse.sum(A, 1).to_dense()
    u = dg.pow(-0.5)
    vals = u[A.indices()[0]] * u[A.indices()[1]]
    An = torch.sparse_coo_tensor(A.indices(), vals, (n, n)).coalesce().to(d)
    Acsr = An.to_sparse_csr()
    adj = gcn_amd.CsrAdjacency(Acsr.crow_indices().int(), Acsr.col_indices().int(), Acsr.values(), (n, n), symmetric=True)
    H = torch.randn(n, k, device=d)
    out = torch.empty(n, k, device=d)
    def t(f, reps=300):
        for _ in range(20): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
    t(lambda: adj.matmul_raw(H, out=out))                # (the clocks come up during the first loop)
    ours = t(lambda: adj.matmul_raw(H, out=out))
    ours_fn = t(lambda: gcn_amd.spmm(adj, H))
    tc = t(lambda: torch.sparse.mm(An, H))
    tcsr = t(lambda: torch.sparse.mm(Acsr, H))
    ref = torch.sparse.mm(An, H)
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"n={n} nnz={An.values().numel()} k={k}: gcn_amd matmul_raw {ours:.1f} us, spmm() {ours_fn:.1f} us, torch COO {tc:.1f} us, torch CSR {tcsr:.1f} us, kernel {adj.main_kernel(k)}, err {err:.1e}")
