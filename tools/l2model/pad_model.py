import numpy as np
d = np.load('/tmp/w/reddit_1.0.npz')
rowptr = d['rowptr'].astype(np.int64); col = d['col'].astype(np.int64)
m = len(rowptr)-1; n = m; nnz = len(col)
rows = np.repeat(np.arange(m), np.diff(rowptr))
for S in (8, 12, 16):
    w = (n + S - 1)//S
    g = col // w
    cnt = np.bincount(g*m + rows, minlength=S*m).reshape(S, m)
    for Lmax in (64, 128):
        tot_steps = 0; npieces = 0
        for s in range(S):
            l = cnt[s]
            npc = np.maximum(1, (l + Lmax - 1)//Lmax)
            full = (npc - 1)                     # pieces of Lmax
            last = l - full*Lmax                 # last piece length (0 for empty rows)
            pieces = np.concatenate([np.repeat(Lmax, full.sum()), last])
            pieces = np.sort(pieces)[::-1]
            pad = (-len(pieces)) % 4
            pieces = np.concatenate([pieces, np.zeros(pad, dtype=pieces.dtype)])
            q = pieces.reshape(-1, 4)
            L = np.maximum(4, (q.max(1) + 3)//4*4)
            tot_steps += L.sum(); npieces += len(pieces)
        print(f'S={S} Lmax={Lmax}: pieces={npieces} ({npieces/(S*m):.3f} per vrow) steps*4={4*tot_steps} vs nnz={nnz}: padding {4*tot_steps/nnz-1:.4f}  quads={npieces//4} avg L={tot_steps/(npieces/4):.1f}')
