"""tools/rabbit_stress.py — rabbit_device over and over on graphs of different character (development aid): every run
must return a permutation; prints the spread of modularity and of the guard counters."""
import sys, time, os, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, '.')
from gcn_amd import reorder, graphgen
dev = torch.device('cuda:0')
def run(name, rp, ci, reps):
    n = rp.numel() - 1
    qs, ts = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        rank, comm, stats = reorder.order_rabbit_device(rp, ci, return_communities=True, return_stats=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        assert torch.equal(torch.sort(rank).values, torch.arange(n, device=dev))
        qs.append(reorder.modularity(rp, ci, comm))
    print(f"{name}: n={n} nnz={ci.numel()} reps={reps} Q {min(qs):.4f}..{max(qs):.4f} ms {min(ts)*1e3:.1f}..{max(ts)*1e3:.1f} last {stats}", flush=True)
g = np.load('tests/golden/gcn1_cora_shaped.npz'); n = int(g['n'])
A = sp.coo_matrix((g['adj_val'], (g['adj_row'], g['adj_col'])), shape=(n, n)).tocsr(); A.sort_indices()
run('cora-shaped', torch.from_numpy(A.indptr.astype(np.int32)).to(dev), torch.from_numpy(A.indices.astype(np.int32)).to(dev), 40)
for nn in (2000, 20000):
    rp, ci, va, nn = graphgen.make_sbm(nn, device=dev, seed=7)
    run(f'sbm{nn}', rp, ci, 15)
rp, ci, va, nn = graphgen.make_graph('reddit', device=dev, seed=1, scale=0.05)
run('reddit x0.05', rp, ci, 10)
rp, ci, va, nn = graphgen.make_graph('products', device=dev, seed=1, scale=0.02)
run('products x0.02', rp, ci, 10)
rp, ci, va, nn = graphgen.make_rmat(16, device=dev, seed=5)
run('rmat16', rp, ci, 10)
