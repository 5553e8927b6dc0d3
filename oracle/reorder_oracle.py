"""oracle/reorder_oracle.py — TEST INFRASTRUCTURE ONLY (never imported by gcn_amd/).

Pure-Python CPU restatement of the reference's vertex reorderers, written from the
reference's sources as an independent second implementation (the product's C++ lives in
gcn_amd/csrc/reorder.cpp).  Small graphs only (pure-Python loops).

Pinned against tests/golden/reorder_*.npz — outputs of the reference's own code compiled
in the build container (oracle/Makefile `ref`, oracle/make_golden.py).

Every function cites the reference file:line it follows.
"""
import math

import numpy as np

INT_MAX = 2 ** 31 - 1


# ---------------------------------------------------------------------------------------
# containers
# ---------------------------------------------------------------------------------------
def edges_from_csr(rowptr, col):
    """edgelist.cuh:16-25 — one directed edge per stored CSR entry, self-loops included."""
    out = []
    for i in range(len(rowptr) - 1):
        for j in range(rowptr[i], rowptr[i + 1]):
            out.append((i, int(col[j])))
    return out


def compute_degrees(n, edges):
    """edgelist.cu:79-102 → (degOut, degIn, deg = in + out)."""
    dout, din = [0] * n, [0] * n
    for u, v in edges:
        dout[u] += 1
        din[v] += 1
    return dout, din, [a + b for a, b in zip(dout, din)]


def rank_from_deg(deg, desc):
    """order_deg.cu:19-39 with the comparators of order_deg.cu:8-13 (ties: node id ascending)."""
    n = len(deg)
    ids = sorted(range(n), key=(lambda u: (-deg[u], u)) if desc else (lambda u: (deg[u], u)))
    rank = [0] * n
    for pos, u in enumerate(ids):
        rank[u] = pos
    return rank


def order_deg(rowptr, col, which="total", desc=True):
    """order_deg.cu:41-56."""
    n = len(rowptr) - 1
    dout, din, deg = compute_degrees(n, edges_from_csr(rowptr, col))
    return rank_from_deg({"total": deg, "out": dout, "in": din}[which], desc)


def build_adjlist(n, edges, rank, kind):
    """adjlist.cu:69-87 (ranked build, neighbours sorted) for
    'D' Dadjlist adjlist.cu:127-150, 'U' Uadjlist :94-121, 'B' Badjlist :157-187.
    Returns a list of neighbour lists; for 'B' slots [0,n) are out-lists, [n,2n) in-lists."""
    slots = 2 * n if kind == "B" else n
    adj = [[] for _ in range(slots)]
    for u, v in edges:
        ru, rv = rank[u], rank[v]
        adj[ru].append(rv)
        if kind == "U":
            adj[rv].append(ru)
        if kind == "B":
            adj[rv + n].append(ru)
    for lst in adj:
        lst.sort()
    return adj


def algo_bfs(n, adj, u0=0):
    """algo_bfs.cu:11-39 — returns the visiting order."""
    placed = [False] * n
    order = []
    i = 0
    for c in range(n):
        u = (c + u0) % n
        if placed[u]:
            continue
        order.append(u)
        placed[u] = True
        while i < len(order):
            w = order[i]
            i += 1
            for v in adj[w]:
                if placed[v]:
                    continue
                placed[v] = True
                order.append(v)
    return order


def rank_from_order(order):
    """tools.cu:31-46."""
    rank = [0] * len(order)
    for pos, u in enumerate(order):
        rank[u] = pos
    return rank


def order_rcm(rowptr, col, directed=True):
    """order_rcm.cu:15-33."""
    n = len(rowptr) - 1
    edges = edges_from_csr(rowptr, col)
    rank_deg = rank_from_deg(compute_degrees(n, edges)[2], desc=False)
    adj = build_adjlist(n, edges, rank_deg, "D" if directed else "U")
    rank_bfs = rank_from_order(algo_bfs(n, adj))
    return [n - 1 - rank_bfs[rank_deg[u]] for u in range(n)]


# ---------------------------------------------------------------------------------------
# Gorder
# ---------------------------------------------------------------------------------------
class UnitHeap:
    """unitheap.cu:16-217 restated with Python lists (same fields, same update rules)."""

    def __init__(self, size):
        self.none = size + 2
        self.infty = INT_MAX // 2
        self.huge = int(math.sqrt(float(size)))                       # unitheap.cu:19
        self.key = [self.infty] * size
        self.prev = [self.none] * size
        self.next = [self.none] * size
        self.update = [self.infty] * size
        self.first, self.second = {}, {}                             # Header[key]
        self.heapsize = 0
        self.top = None

    def _hf(self, k):
        return self.first.get(k, self.none)

    def _hs(self, k):
        return self.second.get(k, self.none)

    def insert(self, index, key):                                    # unitheap.cu:23-28
        self.key[index] = key
        self.update[index] = -key
        self.heapsize += 1

    def reconstruct(self):                                           # unitheap.cu:30-62
        g = sorted(range(self.heapsize), key=lambda a: (-self.key[a], a))
        self.top = g[0]
        cur = self.key[self.top]
        self.first[cur] = self.top
        for i, v in enumerate(g):
            self.prev[v] = g[i - 1] if i > 0 else self.none
            self.next[v] = g[i + 1] if i < len(g) - 1 else self.none
            if self.key[v] != cur:
                self.second[cur] = g[i - 1]
                self.first[self.key[v]] = v
                cur = self.key[v]
        self.second[cur] = g[-1]

    def erase_key_element(self, index, nxt, prv):                    # unitheap.cu:68-76
        k = self.key[index]
        if self._hf(k) == self._hs(k):
            self.first[k] = self.second[k] = self.none
        elif index == self._hf(k):
            self.first[k] = nxt
        elif index == self._hs(k):
            self.second[k] = prv

    def extract_max(self):                                           # unitheap.cu:82-95
        while True:
            tmptop = self.top
            if self.update[self.top] < 0:
                self.decrease_top()
            if self.top == tmptop:
                break
        self.delete(self.top)
        return tmptop

    def decrease_top(self):                                          # unitheap.cu:98-149
        top = self.top
        nxt = self.next[top]
        if nxt == self.none:
            return
        key = self.key[top]
        leftover = int(self.update[top] / 2)                         # C++ int division truncates to 0
        new_key = key + self.update[top] - leftover
        if new_key >= self.key[nxt]:
            return
        self.update[top] = leftover
        level_tail = self._hs(key)
        next_level = self.next[level_tail]
        while next_level != self.none and self.key[next_level] >= new_key:
            level_tail = self._hs(self.key[next_level])
            next_level = self.next[level_tail]
        self.prev[nxt] = self.none
        self.prev[top] = level_tail
        self.next[top] = next_level
        self.next[level_tail] = top
        if next_level != self.none:
            self.prev[next_level] = top
        self.erase_key_element(top, nxt, self.none)
        self.key[top] = new_key
        self.second[new_key] = top
        if self._hf(new_key) == self.none:
            self.first[new_key] = top
        self.top = nxt

    def delete(self, index):                                         # unitheap.cu:152-170
        self.update[index] = self.infty
        prv, nxt = self.prev[index], self.next[index]
        if prv != self.none:
            self.next[prv] = nxt
        if nxt != self.none:
            self.prev[nxt] = prv
        self.erase_key_element(index, nxt, prv)
        if self.top == index:
            self.top = nxt
        self.prev[index] = self.next[index] = self.none
        self.heapsize -= 1

    def lazy_increment(self, index, up):                             # unitheap.cu:177-185
        if self.update[index] == self.infty:
            return
        if self.update[index] == 0 and up > 0:
            self.increment_key(index)
        else:
            self.update[index] += up

    def increment_key(self, index):                                  # unitheap.cu:187-217
        level_head = self._hf(self.key[index])
        prv, nxt = self.prev[index], self.next[index]
        if level_head != index:
            self.next[prv] = nxt
            if nxt != self.none:
                self.prev[nxt] = prv
            prev_level = self.prev[level_head]
            self.prev[index] = prev_level
            self.next[index] = level_head
            self.prev[level_head] = index
            if prev_level != self.none:
                self.next[prev_level] = index
        self.erase_key_element(index, nxt, prv)
        self.key[index] += 1
        k = self.key[index]
        self.second[k] = index
        if self._hf(k) == self.none:
            self.first[k] = index
            if k > self.key[self.top]:
                self.top = index


def move_window(n, adj, heap, new_node, old_node):
    """order_gorder.cu:88-143 (all locality weights are 1, order_gorder.cuh:20-28)."""
    out, inn = (lambda u: adj[u]), (lambda u: adj[u + n])
    old_par, new_par = inn(old_node), inn(new_node)
    oi, ni = 0, 0
    if old_node == new_node:
        oi = len(old_par)
    elif len(out(old_node)) <= heap.huge:
        for child in out(old_node):
            heap.lazy_increment(child, -1)
    tmp_old, tmp_new = [], []
    while True:
        factor = -1
        if oi >= len(old_par):
            if ni >= len(new_par):
                break
            factor = 1
        elif ni < len(new_par):
            if new_par[ni] == old_par[oi]:
                oi += 1
                ni += 1
                continue
            if new_par[ni] < old_par[oi]:
                factor = 1
        if factor == -1:
            if len(out(old_par[oi])) <= heap.huge:
                tmp_old.append(old_par[oi])
            oi += 1
        else:
            if len(out(new_par[ni])) <= heap.huge:
                tmp_new.append(new_par[ni])
            ni += 1
    for parent in tmp_old:
        heap.lazy_increment(parent, -1)
        for sib in out(parent):
            if sib != old_node:
                heap.lazy_increment(sib, -1)
    if len(out(new_node)) <= heap.huge:
        for child in out(new_node):
            heap.lazy_increment(child, +1)
    for parent in tmp_new:
        heap.lazy_increment(parent, +1)
        for sib in out(parent):
            if sib != new_node:
                heap.lazy_increment(sib, +1)


def order_gorder(n, adj, window):
    """order_gorder.cu:35-84."""
    heap = UnitHeap(n)
    order, isolates = [], []
    for u in range(n):
        if len(adj[u]) + len(adj[u + n]) == 0:
            isolates.append(u)
        else:
            heap.insert(u, len(adj[u + n]))                          # keyed by in-degree
    heap.reconstruct()
    hub = heap.top
    order.append(hub)
    heap.delete(hub)
    move_window(n, adj, heap, hub, hub)
    while heap.heapsize > 0:
        new_node = heap.extract_max()
        order.append(new_node)
        old_node = new_node
        if len(order) > window:
            old_node = order[len(order) - window - 1]
        move_window(n, adj, heap, new_node, old_node)
    order.extend(isolates)
    return rank_from_order(order)


def complete_gorder(rowptr, col, window=3):
    """order_gorder.cu:13-31 — RCM, then Gorder on the RCM-relabelled bidirected adjacency."""
    n = len(rowptr) - 1
    edges = edges_from_csr(rowptr, col)
    rank_rcm = order_rcm(rowptr, col, True)
    adj = build_adjlist(n, edges, rank_rcm, "B")
    rank_go = order_gorder(n, adj, window)
    return [rank_go[rank_rcm[u]] for u in range(n)]


# ---------------------------------------------------------------------------------------
# CSR rewrite + C-ABI level functions (renumber.cu)
# ---------------------------------------------------------------------------------------
def apply_rank(rowptr, col, vals, rank):
    """renumber.cu:190-217 — rows to their new index, columns relabelled and sorted ascending,
    values carried; returns (rowptr, col, vals, vomp[new]=old)."""
    n = len(rowptr) - 1
    vomp = [0] * n
    for old in range(n):
        vomp[rank[old]] = old
    nrp = [0]
    for v in vomp:
        nrp.append(nrp[-1] + int(rowptr[v + 1] - rowptr[v]))
    ncol = np.zeros(len(col), np.int32)
    nval = np.zeros(len(col), np.float32)
    for old in range(n):
        pairs = sorted(((rank[int(col[e])], vals[e]) for e in range(rowptr[old], rowptr[old + 1])),
                       key=lambda p: p[0])
        o = nrp[rank[old]]
        for c, v in pairs:
            ncol[o], nval[o] = c, v
            o += 1
    return np.array(nrp, np.int32), ncol, nval, np.array(vomp, np.int32)


def dfs(rowptr, col, vals):
    """renumber.cu:23-155 — pre-order DFS, first root vertex 0, later roots in index order,
    neighbours in stored CSR order."""
    n = len(rowptr) - 1
    rank = [0] * n
    seen = [False] * n
    nxt = 0
    for root in range(n):
        if seen[root]:
            continue
        seen[root] = True
        rank[root] = nxt
        nxt += 1
        stack = [[int(rowptr[root]), int(rowptr[root + 1])]]
        while stack:
            it = stack[-1]
            while it[0] < it[1] and seen[int(col[it[0]])]:
                it[0] += 1
            if it[0] >= it[1]:
                stack.pop()
                continue
            v = int(col[it[0]])
            it[0] += 1
            seen[v] = True
            rank[v] = nxt
            nxt += 1
            stack.append([int(rowptr[v]), int(rowptr[v + 1])])
    return apply_rank(rowptr, col, vals, rank)


def gorder(rowptr, col, vals):
    """renumber.cu:157-230 (window 3, :176)."""
    return apply_rank(rowptr, col, vals, complete_gorder(rowptr, col, 3))


def perm_apply(rowptr, col, vals, vomp):
    """renumber.cu:233-318."""
    n = len(rowptr) - 1
    rank = [0] * n
    for new, old in enumerate(vomp):
        rank[int(old)] = new
    return apply_rank(rowptr, col, vals, rank)[:3]


# ---- libstdc++ std::sort restated (bits/stl_algo.h, GCC 11): introsort with median-of-3,
# ---- heapsort fallback and the final insertion sort.  Needed because rabbit sorts by a key
# ---- that is NOT a total order (renumber.cu:408-409), so the permutation among equal keys is
# ---- whatever this algorithm produces.
def _libstdcxx_sort(a, less):
    n = len(a)
    if n == 0:
        return

    def unguarded_linear_insert(last):
        val = a[last]
        nxt = last - 1
        while less(val, a[nxt]):
            a[last] = a[nxt]
            last = nxt
            nxt -= 1
        a[last] = val

    def insertion_sort(first, last):
        if first == last:
            return
        for i in range(first + 1, last):
            if less(a[i], a[first]):
                val = a[i]
                a[first + 1:i + 1] = a[first:i]
                a[first] = val
            else:
                unguarded_linear_insert(i)

    def push_heap(first, hole, top, value):
        parent = (hole - 1) // 2
        while hole > top and less(a[first + parent], value):
            a[first + hole] = a[first + parent]
            hole = parent
            parent = (hole - 1) // 2
        a[first + hole] = value

    def adjust_heap(first, hole, length, value):
        top = hole
        second = hole
        while second < (length - 1) // 2:
            second = 2 * (second + 1)
            if less(a[first + second], a[first + second - 1]):
                second -= 1
            a[first + hole] = a[first + second]
            hole = second
        if (length & 1) == 0 and second == (length - 2) // 2:
            second = 2 * (second + 1)
            a[first + hole] = a[first + second - 1]
            hole = second - 1
        push_heap(first, hole, top, value)

    def heapsort(first, last):
        length = last - first
        if length >= 2:
            parent = (length - 2) // 2
            while True:
                adjust_heap(first, parent, length, a[first + parent])
                if parent == 0:
                    break
                parent -= 1
        while last - first > 1:
            last -= 1
            value = a[last]
            a[last] = a[first]
            adjust_heap(first, 0, last - first, value)

    def move_median_to_first(result, x, y, z):
        if less(a[x], a[y]):
            if less(a[y], a[z]):
                pick = y
            elif less(a[x], a[z]):
                pick = z
            else:
                pick = x
        elif less(a[x], a[z]):
            pick = x
        elif less(a[y], a[z]):
            pick = z
        else:
            pick = y
        a[result], a[pick] = a[pick], a[result]

    def unguarded_partition(first, last, pivot):
        while True:
            while less(a[first], a[pivot]):
                first += 1
            last -= 1
            while less(a[pivot], a[last]):
                last -= 1
            if not first < last:
                return first
            a[first], a[last] = a[last], a[first]
            first += 1

    def introsort_loop(first, last, depth):
        while last - first > 16:
            if depth == 0:
                heapsort(first, last)
                return
            depth -= 1
            mid = first + (last - first) // 2
            move_median_to_first(first, first + 1, mid, last - 1)
            cut = unguarded_partition(first + 1, last, first)
            introsort_loop(cut, last, depth)
            last = cut

    introsort_loop(0, n, 2 * (n.bit_length() - 1))
    if n > 16:
        insertion_sort(0, 16)
        for i in range(16, n):
            unguarded_linear_insert(i)
    else:
        insertion_sort(0, n)


def rabbit(rowptr, col, vals):
    """renumber.cu:319-522 (opt_iterative = true, hub grouping off, shyness 1).
    std::map is restated as a dict iterated in sorted-key order."""
    n = len(rowptr) - 1
    w = [dict() for _ in range(n)]
    deg = [0] * n
    rnd = [0] * n
    tree = list(range(n))                 # dendrogram node owned by v; None once merged away
    lch, rch = {}, {}                     # cluster node (n+u) -> children
    n_edges = 0
    for v in range(n):                    # renumber.cu:382-398
        for e in range(rowptr[v], rowptr[v + 1]):
            d = int(col[e])
            if d != v:
                w[v][d] = 1
                w[d][v] = 1
        deg[v] = len(w[v])
        n_edges += deg[v]
    two_m_inv = (1.0 / float(2 * n_edges)) if n_edges else float("inf")
    cur = list(range(n))
    nxt = []
    rno = 1
    while cur:
        _libstdcxx_sort(cur, lambda x, y: deg[x] < deg[y])          # renumber.cu:408-409
        for u in cur:
            if rnd[u] == rno:
                continue
            best, v = -1.0, -1
            dv_2m = deg[u] * two_m_inv
            for d in sorted(w[u]):                                  # renumber.cu:419-425
                dq = w[u][d] - deg[d] * dv_2m
                if dq > best:
                    best, v = dq, d
            if best <= 0:
                continue
            deg[v] += deg[u]
            for d in sorted(w[u]):                                  # renumber.cu:433-441
                if d == v:
                    continue
                wt = w[u][d]
                w[v][d] = w[v].get(d, 0) + wt
                if u not in w[d]:
                    continue
                w[d][v] = w[d].get(v, 0) + w[d][u]
                del w[d][u]
            w[v].pop(u, None)
            lch[n + u], rch[n + u] = tree[v], tree[u]               # renumber.cu:445-447
            tree[u] = None
            tree[v] = n + u
            if rnd[v] == rno:
                continue
            rnd[v] = rno
            nxt.append(v)
        cur, nxt = nxt, []
        rno += 1
    vomp = []
    for v in range(n):                                              # renumber.cu:477-489
        if tree[v] is None:
            continue
        st = [tree[v]]
        while st:
            t = st.pop()
            if t in lch:
                st.append(rch[t])
                st.append(lch[t])
            else:
                vomp.append(t)
    out = perm_apply(rowptr, col, vals, vomp)                       # renumber.cu:521
    return out[0], out[1], out[2], np.array(vomp, np.int32)
