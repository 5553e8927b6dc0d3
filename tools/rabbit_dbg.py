import sys, time, torch
sys.path.insert(0, '.')
from gcn_amd import graphgen, reorder
n = int(sys.argv[1])
dev = torch.device('cuda:0')
rowptr, col, val, n = graphgen.make_sbm(n, device=dev, seed=7)
print('graph', n, int(col.numel()), flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
rank, comm, stats = reorder.order_rabbit_device(rowptr, col, return_communities=True, return_stats=True)
torch.cuda.synchronize()
print('device rabbit', round((time.perf_counter() - t0) * 1e3, 1), 'ms', stats, flush=True)
assert torch.equal(torch.sort(rank).values, torch.arange(n, device=dev))
print('Q', reorder.modularity(rowptr, col, comm), 'communities', int(torch.unique(comm).numel()), flush=True)
