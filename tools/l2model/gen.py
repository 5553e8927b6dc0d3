import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from gcn_amd import graphgen
t=time.time()
scale=float(sys.argv[1])
rowptr,col,val,n = graphgen.make_graph('reddit', device='cpu', seed=1, scale=scale)
print(n, col.numel(), time.time()-t)
np.savez(f'/tmp/w/reddit_{scale}.npz', rowptr=rowptr.numpy(), col=col.numpy())
