import sys, time, ctypes, numpy as np
lib = ctypes.CDLL('/tmp/w/l2sim.so')
P = ctypes.c_void_p
lib.simulate.argtypes = [P, ctypes.c_int64, P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
scale = sys.argv[1] if len(sys.argv) > 1 else '0.25'
d = np.load(f'/tmp/w/reddit_{scale}.npz')
rowptr = d['rowptr'].astype(np.int64); col = d['col'].astype(np.int64)
m = len(rowptr) - 1; n = m; nnz = len(col)
deg = np.diff(rowptr)
rows = np.repeat(np.arange(m, dtype=np.int64), deg)
print(f'n={n} nnz={nnz} mean={nnz/n:.1f} max={deg.max()} median={np.median(deg)}', flush=True)
T = 2048
CACHE = int(float(sys.argv[2]) * (1 << 20)) if len(sys.argv) > 2 else (4 << 20)

def build(perm_rank, bounds):
    """perm_rank[c] = new column label; bounds: group boundaries in new-label space [G+1] -> virtual CSR"""
    newc = perm_rank[col]
    g = np.searchsorted(bounds, newc, side='right') - 1
    vrow = g * m + rows
    order = np.argsort(vrow * n + newc, kind='stable')
    vcol = newc[order].astype(np.int32)
    G = len(bounds) - 1
    cnt = np.bincount(vrow, minlength=G * m)
    vrowptr = np.zeros(G * m + 1, dtype=np.int64); vrowptr[1:] = np.cumsum(cnt)
    return vrowptr, vcol, cnt

def sim(vrowptr, vcol, store_alloc=1, col_bytes=2, xcds=range(8), open_=512, ways=16):
    nv = len(vrowptr) - 1
    nchunks = (nnz + T - 1) // T
    H = M = S = 0
    out = np.zeros(3, dtype=np.int64)
    for x in xcds:
        c_lo, c_hi = nchunks * x // 8, nchunks * (x + 1) // 8
        e_lo, e_hi = c_lo * T, min(c_hi * T, nnz)
        lib.simulate(vrowptr.ctypes.data, nv, vcol.ctypes.data, e_lo, e_hi, T, open_, 2, store_alloc, col_bytes, CACHE, ways, out.ctypes.data)
        H += out[0]; M += out[1]; S += out[2]
    f = 8 / len(list(xcds))
    return H * f, M * f, S * f

def report(name, vrowptr, vcol, cnt, **kw):
    t = time.time()
    H, M, S = sim(vrowptr, vcol, **kw)
    nonempty = int((cnt > 0).sum())
    G = (len(vrowptr) - 1) // m
    traffic = M * 128 + (len(vrowptr) - 1) * 256 + nnz * 2     # misses + partial-row stores (all virtual rows) + col16
    reduce_rd = (len(vrowptr) - 1) * 256
    print(f'{name:40s} G={G:3d} hit={H/(H+M):.3f} miss_GB={M*128/1e9:.2f} traffic_GB={traffic/1e9:.2f} vrows={len(vrowptr)-1} nonempty={nonempty} ({nonempty/(len(vrowptr)-1):.2f}) nnz/vrow={nnz/max(nonempty,1):.1f} reduce_read_GB={reduce_rd/1e9:.2f} [{time.time()-t:.0f}s]', flush=True)

ident = np.arange(n, dtype=np.int64)
by_deg = np.empty(n, dtype=np.int64); by_deg[np.argsort(-deg, kind='stable')] = np.arange(n)
cum = np.cumsum(np.sort(deg)[::-1])   # nnz in the top-i columns (sorted labels)

def eq_width(G): return np.array([ (n * i + G - 1) // G if i < G else n for i in range(G + 1)], dtype=np.int64) if False else np.minimum(np.arange(G + 1, dtype=np.int64) * ((n + G - 1) // G), n)
def eq_nnz(G): return np.concatenate([[0], np.searchsorted(cum, nnz * np.arange(1, G) / G), [n]]).astype(np.int64)
def by_size(mb): 
    w = int(mb * (1 << 20) / 256)
    b = list(range(0, n, w)) + [n]
    return np.array(b, dtype=np.int64)

which = sys.argv[3].split(',') if len(sys.argv) > 3 else ['A8']
for wname in which:
    if wname == 'A8': v = build(ident, eq_width(8)); report('identity, 8 equal width', *v); report('identity, 8 equal width, stores no-alloc', *v, store_alloc=0)
    elif wname == 'A16': v = build(ident, eq_width(16)); report('identity, 16 equal width', *v); report('identity, 16 equal width, no-alloc', *v, store_alloc=0)
    elif wname == 'A0': v = build(ident, eq_width(1)); report('identity, unsliced', *v)
    elif wname.startswith('B'): G = int(wname[1:]); v = build(by_deg, eq_nnz(G)); print('  bounds', eq_nnz(G)); report(f'deg-sorted, {G} equal nnz', *v); report(f'deg-sorted, {G} equal nnz, no-alloc', *v, store_alloc=0)
    elif wname.startswith('C'): mb = float(wname[1:]); v = build(by_deg, by_size(mb)); report(f'deg-sorted, groups of {mb} MB', *v); report(f'deg-sorted, groups of {mb} MB, no-alloc', *v, store_alloc=0)
    elif wname.startswith('W'): mb = float(wname[1:]); v = build(ident, by_size(mb)); report(f'identity, groups of {mb} MB', *v); report(f'identity, groups of {mb} MB, no-alloc', *v, store_alloc=0)
for wname in which:
    if wname.startswith('E'):
        G = int(wname[1:]); v = build(ident, eq_width(G)); report(f'identity, {G} equal width', *v); report(f'identity, {G} equal width, no-alloc', *v, store_alloc=0)
