import sys, time, os, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, '.')
from gcn_amd import reorder
g = np.load('tests/golden/gcn1_cora_shaped.npz')
n = int(g['n'])
A = sp.coo_matrix((g['adj_val'], (g['adj_row'], g['adj_col'])), shape=(n, n)).tocsr(); A.sort_indices()
dev = torch.device('cuda:0')
rp = torch.from_numpy(A.indptr.astype(np.int32)).to(dev); ci = torch.from_numpy(A.indices.astype(np.int32)).to(dev)
print('graph', n, A.nnz, flush=True)
for rep in range(5):
    t0 = time.perf_counter()
    rank, comm, stats = reorder.order_rabbit_device(rp, ci, return_communities=True, return_stats=True)
    torch.cuda.synchronize()
    print(rep, 'device rabbit', round((time.perf_counter() - t0) * 1e3, 1), 'ms', stats, 'Q', round(reorder.modularity(rp, ci, comm), 4), flush=True)
