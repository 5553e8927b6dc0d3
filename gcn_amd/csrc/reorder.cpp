// Host-side vertex reorderers for the GCN SpMM path — degree, reverse
// Cuthill-McKee, Gorder (RCM∘Gorder, window w), DFS and Rabbit — plus the CSR
// rewrite that applies an ordering.
//
// These are preprocessing steps (run once per graph on the host); the contract
// is that the integer vectors they return are BIT-EXACT with the reference
// (guohaoqiang/gcn @ v1).  The code below is written against CSR directly (the
// reference materialises a 16 B/edge edge list and virtual Adjlist classes —
// edgelist.cuh:8-44, adjlist.cuh:16-117); every place where the reference's
// tie-breaking or traversal order is observable is cited.
#include "reorder.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <map>
#include <ranges>
#include <thread>
#include <utility>

namespace gcn {
namespace reorder {

namespace {

// ---- degrees --------------------------------------------------------------
// One directed edge per stored CSR entry, self-loops included (edgelist.cuh:16-25);
// out = row length, in = column count, total = in + out (edgelist.cu:79-102).
void degrees(const Csr& g, std::vector<u64>& out, std::vector<u64>& in) {
  out.assign(g.n, 0);
  in.assign(g.n, 0);
  for (int64_t u = 0; u < g.n; ++u) out[u] = (u64)(g.rowptr[u + 1] - g.rowptr[u]);
  for (int64_t e = 0; e < g.nnz; ++e) ++in[g.col[e]];
}

// rank[node] = position of node when sorted by (degree asc|desc, node id asc)
// — order_deg.cu:8-13 comparators, :19-39 rank_from_deg.  The key is a strict
// total order, so the sort algorithm is free.
std::vector<u64> rank_by_degree(const std::vector<u64>& deg, bool desc) {
  const size_t n = deg.size();
  std::vector<u64> ids(n);
  for (size_t i = 0; i < n; ++i) ids[i] = i;
  if (desc)
    std::sort(ids.begin(), ids.end(), [&](u64 a, u64 b) {
      return deg[a] > deg[b] || (deg[a] == deg[b] && a < b); });
  else
    std::sort(ids.begin(), ids.end(), [&](u64 a, u64 b) {
      return deg[a] < deg[b] || (deg[a] == deg[b] && a < b); });
  std::vector<u64> rank(n);
  for (size_t i = 0; i < n; ++i) rank[ids[i]] = i;
  return rank;
}

// ---- relabelled adjacency ---------------------------------------------------
// Compact adjacency in the numbering `rank` with every neighbour list sorted
// ascending (adjlist.cu:69-87).  kind: 'D' out-lists (adjlist.cu:127-150),
// 'U' both directions merged into one list per vertex (adjlist.cu:94-121),
// 'B' out-lists in [0,n) and in-lists in [n,2n) (adjlist.cu:157-187).
struct Adj {
  int64_t n = 0;
  std::vector<u64> off;   // cumulative degrees
  std::vector<u32> nb;    // neighbours (32-bit: half the bytes the window slides of Gorder pull through the caches)
  const u32* begin(u64 slot) const { return nb.data() + off[slot]; }
  const u32* end(u64 slot) const { return nb.data() + off[slot + 1]; }
  u64 deg(u64 slot) const { return off[slot + 1] - off[slot]; }
};

Adj build_adj(const Csr& g, const std::vector<u64>& rank, char kind) {
  Adj a;
  a.n = g.n;
  const int64_t slots = (kind == 'B') ? 2 * g.n : g.n;
  std::vector<u64> cnt(slots, 0);
  for (int64_t u = 0; u < g.n; ++u) {
    const u64 ru = rank[u];
    for (int32_t e = g.rowptr[u]; e < g.rowptr[u + 1]; ++e) {
      const u64 rv = rank[g.col[e]];
      ++cnt[ru];
      if (kind == 'U') ++cnt[rv];
      if (kind == 'B') ++cnt[rv + g.n];
    }
  }
  a.off.assign(slots + 1, 0);
  for (int64_t s = 0; s < slots; ++s) a.off[s + 1] = a.off[s] + cnt[s];
  a.nb.assign(a.off[slots], 0);
  std::fill(cnt.begin(), cnt.end(), 0);
  for (int64_t u = 0; u < g.n; ++u) {
    const u64 ru = rank[u];
    for (int32_t e = g.rowptr[u]; e < g.rowptr[u + 1]; ++e) {
      const u64 rv = rank[g.col[e]];
      a.nb[a.off[ru] + cnt[ru]++] = (u32)rv;
      if (kind == 'U') a.nb[a.off[rv] + cnt[rv]++] = (u32)ru;
      if (kind == 'B') a.nb[a.off[rv + g.n] + cnt[rv + g.n]++] = (u32)ru;
    }
  }
  for (int64_t s = 0; s < slots; ++s)
    std::sort(a.nb.begin() + a.off[s], a.nb.begin() + a.off[s + 1]);
  return a;
}

// BFS over all components: start at vertex 0, restart at the next unplaced INDEX,
// neighbours in (sorted) list order — algo_bfs.cu:11-39.  Returns order[i] = vertex.
std::vector<u64> bfs_order(const Adj& a) {
  std::vector<char> placed(a.n, 0);
  std::vector<u64> order;
  order.reserve(a.n);
  size_t head = 0;
  for (int64_t s = 0; s < a.n; ++s) {
    if (placed[s]) continue;
    placed[s] = 1;
    order.push_back(s);
    while (head < order.size()) {
      const u64 w = order[head++];
      for (const u32* v = a.begin(w); v != a.end(w); ++v) {
        if (placed[*v]) continue;
        placed[*v] = 1;
        order.push_back(*v);
      }
    }
  }
  return order;
}

// ---- the Gorder priority structure ------------------------------------------
// Behavioural restatement of the reference's UnitHeap (unitheap.cu:16-217): a
// doubly linked list of vertices kept in key-descending order with per-key
// (first,last) bucket markers and LAZY negative updates that are only applied to
// the current top (DecreaseTop, halving the pending amount each time).  Every
// tie-break below is observable in the final order, so the operations follow the
// reference's semantics one for one.
class LazyBuckets {
 public:
  explicit LazyBuckets(u64 n)
      : huge((u64)std::sqrt((double)n)), none_((u32)(n + 2)), node_(n, Node{kInf, kInf, (u32)(n + 2), (u32)(n + 2)}) {}

  static constexpr int kInf = INT_MAX / 2;
  u64 huge;            // hub cut-off, sqrt(n) truncated (unitheap.cu:19)
  size_t live = 0;
  u64 top = 0;

  // the record a lazy_add of vertex v is about to touch: the updates of a window slide hit vertices all over the
  // graph (one cache miss each at scale 20+), and their order is known many steps ahead
  void prefetch(u64 v) const { __builtin_prefetch(&node_[v], 1, 1); }

  void insert(u64 v, int key) {          // unitheap.cu:23-28
    node_[v].key = key;
    node_[v].pend = -key;
    ++live;
  }

  // unitheap.cu:30-62 — sorts the indices 0..live-1 (the reference's silent
  // assumption: every inserted vertex has an index below `live`).
  void build() {
    std::vector<u32> g(live);
    for (size_t i = 0; i < live; ++i) g[i] = (u32)i;
    std::sort(g.begin(), g.end(), [&](u32 a, u32 b) {
      return node_[a].key > node_[b].key || (node_[a].key == node_[b].key && a < b); });
    top = g[0];
    int cur = node_[top].key;
    bucket(cur).first = (u32)top;
    for (size_t i = 0; i < g.size(); ++i) {
      const u32 v = g[i];
      node_[v].prev = i > 0 ? g[i - 1] : none_;
      node_[v].next = i + 1 < g.size() ? g[i + 1] : none_;
      if (node_[v].key != cur) {
        bucket(cur).last = g[i - 1];
        bucket(node_[v].key).first = v;
        cur = node_[v].key;
      }
    }
    bucket(cur).last = g.back();
  }

  u64 extract_max() {                    // unitheap.cu:82-95
    u64 t;
    do {
      t = top;
      if (node_[top].pend < 0) decrease_top();
    } while (top != t);
    remove(top);
    return t;
  }

  void remove(u64 v) {                   // unitheap.cu:152-170
    Node& x = node_[v];
    x.pend = kInf;
    const u32 p = x.prev, nx = x.next;
    if (p != none_) node_[p].next = nx;
    if (nx != none_) node_[nx].prev = p;
    unbucket((u32)v, nx, p);
    if (top == v) top = nx;
    x.prev = x.next = none_;
    --live;
  }

  // returns false on the reference's "negative" abort condition
  bool lazy_add(u64 v, int up) {         // unitheap.cu:177-185
    Node& x = node_[v];
    if (x.pend == kInf) return true;
    if (x.pend == 0 && up > 0) {
      increment((u32)v);
    } else {
      x.pend += up;
      if (-x.pend > x.key) return false;
    }
    return true;
  }

 private:
  struct Node { int key, pend; u32 prev, next; };   // one 16-byte record per vertex: a lazy update is one cache line
  struct Bucket { u32 first, last; };
  u32 none_;
  std::vector<Node> node_;
  std::vector<Bucket> buckets_;

  Bucket& bucket(int key) {
    if ((size_t)key >= buckets_.size()) buckets_.resize((size_t)key + 64, Bucket{none_, none_});
    return buckets_[key];
  }

  void unbucket(u32 v, u32 nx, u32 p) {  // unitheap.cu:68-76 (note: the first test
    Bucket& b = bucket(node_[v].key);    // does not look at v itself)
    if (b.first == b.last) b.first = b.last = none_;
    else if (v == b.first) b.first = nx;
    else if (v == b.last) b.last = p;
  }

  void decrease_top() {                  // unitheap.cu:98-149
    Node& t = node_[top];
    const u32 nx = t.next;
    if (nx == none_) return;
    const int key = t.key;
    const int leftover = t.pend / 2;
    const int new_key = key + t.pend - leftover;
    if (new_key >= node_[nx].key) return;
    t.pend = leftover;

    u32 tail = bucket(key).last;
    u32 after = node_[tail].next;
    while (after != none_ && node_[after].key >= new_key) {
      tail = bucket(node_[after].key).last;
      after = node_[tail].next;
    }
    node_[nx].prev = none_;
    t.prev = tail;
    t.next = after;
    node_[tail].next = (u32)top;
    if (after != none_) node_[after].prev = (u32)top;

    unbucket((u32)top, nx, none_);
    t.key = new_key;
    Bucket& nb = bucket(new_key);
    nb.last = (u32)top;
    if (nb.first == none_) nb.first = (u32)top;
    top = nx;
  }

  void increment(u32 v) {                // unitheap.cu:187-217
    Node& x = node_[v];
    const u32 head = bucket(x.key).first;
    const u32 p = x.prev, nx = x.next;
    if (head != v) {
      node_[p].next = nx;
      if (nx != none_) node_[nx].prev = p;
      const u32 before = node_[head].prev;
      x.prev = before;
      x.next = head;
      node_[head].prev = v;
      if (before != none_) node_[before].next = v;
    }
    unbucket(v, nx, p);
    const int key = ++x.key;
    Bucket& b = bucket(key);
    b.last = v;
    if (b.first == none_) {
      b.first = v;
      if (key > node_[top].key) top = v;
    }
  }
};

// order_gorder.cu:88-143.  `b` holds out-lists in slots [0,n) and in-lists in [n,2n).
bool slide_window(const Adj& b, LazyBuckets& h, u64 incoming, u64 outgoing) {
  const u64 n = (u64)b.n;
  const u32* op = b.begin(outgoing + n);
  const u32* oe = b.end(outgoing + n);
  const u32* np = b.begin(incoming + n);
  const u32* ne = b.end(incoming + n);
  bool ok = true;
  // The updates of one slide hit heap records all over the graph — one cache miss each from scale 20 on — in an
  // order that is known in advance (the adjacency lists): every list is walked with the record `kAhead` entries on
  // already requested.  Hints only: the sequence of heap operations, and so every tie-break, is unchanged.
  constexpr int kAhead = 12;
  auto add_all = [&](const u32* s, const u32* e, u64 skip, int up) {
    for (const u32* q = s; q != e && q < s + kAhead; ++q) h.prefetch(*q);
    for (const u32* q = s; q != e; ++q) {
      if (q + kAhead < e) h.prefetch(q[kAhead]);
      if (*q != skip) ok &= h.lazy_add(*q, up);
    }
  };

  if (outgoing == incoming) {
    op = oe;                                        // no vertex leaves the window
  } else if (b.deg(outgoing) <= h.huge) {
    add_all(b.begin(outgoing), b.end(outgoing), n + 2, -1);
  }

  // parents of exactly one of the two vertices (sorted-list symmetric difference),
  // hubs (out-degree > sqrt n) skipped
  static thread_local std::vector<u32> leaving, entering;       // scratch, reused across the n calls
  leaving.clear();
  entering.clear();
  while (true) {
    bool take_new;
    if (op >= oe) {
      if (np >= ne) break;
      take_new = true;
    } else if (np < ne) {
      if (*np == *op) { ++op; ++np; continue; }
      take_new = *np < *op;
    } else {
      take_new = false;
    }
    if (take_new) { if (b.deg(*np) <= h.huge) entering.push_back(*np); ++np; }
    else          { if (b.deg(*op) <= h.huge) leaving.push_back(*op);  ++op; }
  }

  // (the next parent's list itself is requested while the current one is walked)
  auto walk = [&](const std::vector<u32>& parents, u64 skip, int up) {
    for (size_t i = 0; i < parents.size(); ++i) {
      const u32 p = parents[i];
      if (i + 1 < parents.size()) { __builtin_prefetch(b.begin(parents[i + 1])); h.prefetch(parents[i + 1]); }
      ok &= h.lazy_add(p, up);
      add_all(b.begin(p), b.end(p), skip, up);
    }
  };
  walk(leaving, outgoing, -1);
  if (b.deg(incoming) <= h.huge) add_all(b.begin(incoming), b.end(incoming), n + 2, +1);
  walk(entering, incoming, +1);
  return ok;
}

// order_gorder.cu:35-84; returns rank (in the numbering of `b`), ok=false if the
// reference would have aborted or run into undefined behaviour.
std::vector<u64> gorder_rank(const Adj& b, u64 window, bool* ok) {
  const u64 n = (u64)b.n;
  *ok = true;
  std::vector<u64> order;
  order.reserve(n);
  LazyBuckets heap(n);
  std::vector<u64> isolated;
  for (u64 u = 0; u < n; ++u) {
    if (b.deg(u) + b.deg(u + n) == 0) isolated.push_back(u);
    else heap.insert(u, (int)b.deg(u + n));          // keyed by IN-degree
  }
  // the reference sorts indices 0..heapsize-1: only defined when no isolated
  // vertex has an index below heapsize (SURVEY §8a "replicate, don't fix")
  for (u64 u : isolated) if (u < heap.live) { *ok = false; return {}; }
  if (heap.live == 0) {
    std::vector<u64> rank(n);
    for (u64 i = 0; i < n; ++i) rank[i] = i;
    return rank;
  }
  heap.build();
  const u64 hub = heap.top;
  order.push_back(hub);
  heap.remove(hub);
  *ok &= slide_window(b, heap, hub, hub);
  while (heap.live > 0 && *ok) {
    const u64 v = heap.extract_max();
    if (v >= n) { *ok = false; break; }
    order.push_back(v);
    u64 old = v;
    if (order.size() > window) old = order[order.size() - window - 1];
    *ok &= slide_window(b, heap, v, old);
  }
  if (!*ok) return {};
  order.insert(order.end(), isolated.begin(), isolated.end());
  std::vector<u64> rank(n);
  for (u64 i = 0; i < n; ++i) rank[order[i]] = i;   // tools.cu:31-46
  return rank;
}

}  // namespace

// ---- public orderings ---------------------------------------------------------

std::vector<u64> order_deg(const Csr& g, DegKind which, bool desc) {
  std::vector<u64> out, in;
  degrees(g, out, in);
  if (which == DEG_OUT) return rank_by_degree(out, desc);       // order_deg.cu:46-50
  if (which == DEG_IN) return rank_by_degree(in, desc);         // order_deg.cu:52-56
  for (int64_t u = 0; u < g.n; ++u) out[u] += in[u];            // order_deg.cu:41-45
  return rank_by_degree(out, desc);
}

// order_rcm.cu:15-33: degree-ASC relabel, BFS, reverse.
std::vector<u64> order_rcm(const Csr& g, bool directed) {
  const std::vector<u64> rdeg = order_deg(g, DEG_TOTAL, false);
  const Adj a = build_adj(g, rdeg, directed ? 'D' : 'U');
  const std::vector<u64> order = bfs_order(a);
  std::vector<u64> pos(g.n);
  for (int64_t i = 0; i < g.n; ++i) pos[order[i]] = i;
  std::vector<u64> rank(g.n);
  for (int64_t u = 0; u < g.n; ++u) rank[u] = (u64)g.n - 1 - pos[rdeg[u]];
  return rank;
}

// order_gorder.cu:13-31
std::vector<u64> order_gorder_complete(const Csr& g, u64 window, bool* ok) {
  const std::vector<u64> rrcm = order_rcm(g, true);
  const Adj b = build_adj(g, rrcm, 'B');
  const std::vector<u64> rgo = gorder_rank(b, window, ok);
  if (!*ok) return {};
  std::vector<u64> rank(g.n);
  for (int64_t u = 0; u < g.n; ++u) rank[u] = rgo[rrcm[u]];
  return rank;
}

// renumber.cu:23-95: iterative pre-order DFS, first tree rooted at vertex 0, next
// roots in index order, neighbours in stored CSR order.
std::vector<u64> order_dfs(const Csr& g) {
  const int64_t n = g.n;
  std::vector<u64> rank(n, 0);
  std::vector<char> seen(n, 0);
  std::vector<std::pair<int32_t, int32_t>> stack;   // (next edge, end edge)
  u64 next_id = 0;
  for (int64_t root = 0; root < n; ++root) {
    if (seen[root]) continue;
    seen[root] = 1;
    rank[root] = next_id++;
    stack.push_back({g.rowptr[root], g.rowptr[root + 1]});
    while (!stack.empty()) {
      auto& top = stack.back();
      while (top.first < top.second && seen[g.col[top.first]]) ++top.first;
      if (top.first >= top.second) { stack.pop_back(); continue; }
      const int32_t v = g.col[top.first++];
      seen[v] = 1;
      rank[v] = next_id++;
      stack.push_back({g.rowptr[v], g.rowptr[v + 1]});
    }
  }
  return rank;
}

// Neighbour -> merged edge count, one per vertex.  The reference keeps a std::map<int,int> here
// (renumber.cu:331-336) and its observable behaviour depends on the key ORDER in exactly one place:
// the merge target is the first maximum of dQ when the map is scanned in key order with a strict '>'
// (renumber.cu:14-20, :419-425) — i.e. the maximum, ties to the smallest key.  Everything else
// (weight accumulation, erasing the merged vertex from its neighbours' maps) is order-independent, so
// an open-addressing table with that tie rule gives identical results without the per-node
// allocations and pointer chasing that dominate the reference's 29.5 s per 3.4 M non-zeros.
class NbrTable {
 public:
  int size() const { return live_; }
  template <typename F> void for_each(F&& f) const {
    for (size_t i = 0; i < key_.size(); ++i)
      if (key_[i] >= 0) f(key_[i], val_[i]);
  }
  // w[k] += dv (inserting k with 0 first, like std::map::operator[])
  void add(int k, int dv) {
    if ((used_ + 1) * 10 > (int)key_.size() * 7) grow();
    const size_t mask = key_.size() - 1;
    size_t i = hash(k) & mask, first_free = (size_t)-1;
    while (key_[i] != kEmpty) {
      if (key_[i] == k) { val_[i] += dv; return; }
      if (key_[i] == kDead && first_free == (size_t)-1) first_free = i;
      i = (i + 1) & mask;
    }
    if (first_free != (size_t)-1) i = first_free; else ++used_;
    key_[i] = k; val_[i] = dv; ++live_;
  }
  // value of k, or -1 when absent (weights are >= 1)
  int find(int k) const {
    if (key_.empty()) return -1;
    const size_t mask = key_.size() - 1;
    size_t i = hash(k) & mask;
    while (key_[i] != kEmpty) {
      if (key_[i] == k) return val_[i];
      i = (i + 1) & mask;
    }
    return -1;
  }
  void erase(int k) {
    if (key_.empty()) return;
    const size_t mask = key_.size() - 1;
    size_t i = hash(k) & mask;
    while (key_[i] != kEmpty) {
      if (key_[i] == k) { key_[i] = kDead; --live_; return; }
      i = (i + 1) & mask;
    }
  }
  void release() { std::vector<int>().swap(key_); std::vector<int>().swap(val_); live_ = used_ = 0; }

 private:
  static constexpr int kEmpty = -1, kDead = -2;
  static size_t hash(int k) { return (size_t)((unsigned)k * 2654435761u) >> 7; }
  void grow() {
    size_t cap = 8;
    while (cap * 7 < (size_t)(live_ + 1) * 20) cap *= 2;      // <= 35 % full after a rebuild
    std::vector<int> ok(cap, kEmpty), ov(cap, 0);
    ok.swap(key_); ov.swap(val_);
    live_ = used_ = 0;
    for (size_t i = 0; i < ok.size(); ++i)
      if (ok[i] >= 0) {
        const size_t mask = key_.size() - 1;
        size_t j = hash(ok[i]) & mask;
        while (key_[j] != kEmpty) j = (j + 1) & mask;
        key_[j] = ok[i]; val_[j] = ov[i]; ++live_; ++used_;
      }
  }
  std::vector<int> key_, val_;
  int live_ = 0, used_ = 0;     // live entries; slots ever occupied (live + tombstones)
};

// renumber.cu:319-520 (opt_iterative = true, hub grouping off, shyness 1).
std::vector<int32_t> order_rabbit_vomp(const Csr& g, bool verbose, std::vector<int32_t>* community_out) {
  const int n = (int)g.n;
  struct Vtx {
    NbrTable w;             // neighbour -> merged edge count (unit weights)
    int deg = 0;
    int round = 0;
    int tree = -1;          // dendrogram node id owned by this vertex, -1 = merged away
  };
  std::vector<Vtx> V(n);
  // dendrogram: node v in [0,n) is leaf v; node n+u is the cluster created when u was merged
  std::vector<int> lch(2 * (size_t)n, -1), rch(2 * (size_t)n, -1);
  std::vector<unsigned int> cur(n), nxt;
  long long n_edges = 0;

  // undirected simple graph, self-loops dropped; a vertex's degree is taken right
  // after ITS row has been scanned (renumber.cu:382-398) — reverse edges inserted
  // later by higher-numbered rows are not counted, exactly as in the reference.
  for (int v = 0; v < n; ++v) {
    for (int32_t e = g.rowptr[v]; e < g.rowptr[v + 1]; ++e) {
      const int d = g.col[e];
      if (d == v) continue;
      if (V[v].w.find(d) < 0) V[v].w.add(d, 1);          // map[d] = 1: insert-or-assign, weights stay 1
      if (V[d].w.find(v) < 0) V[d].w.add(v, 1);
    }
    V[v].deg = V[v].w.size();
    n_edges += V[v].deg;
    V[v].tree = v;
    cur[v] = v;
  }
  const double two_m_inv = 1.0 / double(2 * (int)n_edges);   // int arithmetic as renumber.cu:402

  for (int round = 1; !cur.empty(); ++round) {
    // sort by CURRENT degree only: not a total order, so the permutation of equal
    // keys is whatever libstdc++'s introsort produces — same call as renumber.cu:408
    std::ranges::sort(cur, std::ranges::less(), [&](auto i) { return V[i].deg; });
    if (verbose) std::printf("Rabbit round %2d, n elts %zd\n", round, cur.size());
    for (auto u : cur) {
      Vtx& uo = V[u];
      if (uo.round == round) continue;
      double best = -1;
      int v = -1;
      const double dv_2m = uo.deg * two_m_inv;
      uo.w.for_each([&](int d, int w) {                  // first maximum in key order = max, ties to the smallest key
        const double dq = w - V[d].deg * dv_2m;
        if (dq > best || (dq == best && v >= 0 && d < v)) { best = dq; v = d; }
      });
      if (best <= 0) continue;
      Vtx& vo = V[v];
      vo.deg += uo.deg;
      uo.w.for_each([&](int d, int w) {                  // (no self keys: neither vo.w nor V[d].w is uo.w)
        if (d == v) return;
        vo.w.add(d, w);
        auto& dm = V[d].w;
        const int back = dm.find((int)u);
        if (back < 0) return;
        dm.add(v, back);
        dm.erase((int)u);
      });
      vo.w.erase((int)u);
      uo.w.release();            // u is out of every neighbour's table (the relation stays symmetric): never read again
      lch[(size_t)n + u] = vo.tree;                    // (target's tree, merged tree)
      rch[(size_t)n + u] = uo.tree;
      uo.tree = -1;
      vo.tree = n + (int)u;
      if (vo.round == round) continue;
      vo.round = round;
      nxt.push_back(v);
    }
    std::swap(cur, nxt);
    nxt.clear();
  }

  // leaves of every surviving dendrogram, roots in vertex-index order, left before right
  std::vector<int32_t> vomp;
  vomp.reserve(n);
  std::vector<int> st;
  int n_comm = 0;
  if (community_out) community_out->assign((size_t)n, -1);
  for (int v = 0; v < n; ++v) {
    if (V[v].tree < 0) continue;
    ++n_comm;
    st.push_back(V[v].tree);
    while (!st.empty()) {
      const int t = st.back();
      st.pop_back();
      if (lch[t] >= 0) { st.push_back(rch[t]); st.push_back(lch[t]); }
      else { vomp.push_back(t); if (community_out) (*community_out)[(size_t)t] = v; }
    }
  }
  if (verbose) std::printf("Rabbit found %d communities, edges %lld\n", n_comm, n_edges);
  return vomp;
}

// renumber.cu:190-217 / :244-278: rows moved to their new index, columns relabelled,
// each row sorted by new column with the values carried along (same sort call and
// element type as the reference so that duplicate columns, if any, permute alike).
void csr_apply_rank(int32_t* rowptr, int32_t* col, float* vals, int64_t n, int64_t nnz,
                    const u64* rank) {
  std::vector<int32_t> nrow(n + 1, 0), ncol(nnz);
  std::vector<float> nval(nnz);
  for (int64_t u = 0; u < n; ++u) nrow[rank[u] + 1] = rowptr[u + 1] - rowptr[u];
  for (int64_t i = 0; i < n; ++i) nrow[i + 1] += nrow[i];
  // rows are independent (each writes its own range of the output), so the per-row sorts run on all
  // host cores; the result does not depend on the thread count
  auto rows = [&](int64_t lo, int64_t hi) {
    std::vector<std::pair<float, unsigned int>> row;
    for (int64_t u = lo; u < hi; ++u) {
      row.clear();
      for (int32_t e = rowptr[u]; e < rowptr[u + 1]; ++e)
        row.emplace_back(vals[e], (unsigned int)rank[col[e]]);
      std::ranges::sort(row, std::ranges::less(), [](auto& p) { return p.second; });
      int32_t o = nrow[rank[u]];
      for (auto& [val, c] : row) { ncol[o] = (int32_t)c; nval[o++] = val; }
    }
  };
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > 64) nt = 64;
  if (nt < 2 || nnz < (1 << 18)) {
    rows(0, n);
  } else {
    // equal shares of the non-zeros, not of the rows
    std::vector<std::thread> pool;
    int64_t lo = 0;
    for (unsigned t = 0; t < nt; ++t) {
      const int64_t target = nnz / nt * (t + 1);
      int64_t hi = (t + 1 == nt) ? n : (int64_t)(std::upper_bound(rowptr, rowptr + n + 1, (int32_t)target) - rowptr) - 1;
      if (hi < lo) hi = lo;
      if (hi > n) hi = n;
      pool.emplace_back(rows, lo, hi);
      lo = hi;
    }
    for (auto& th : pool) th.join();
  }
  std::copy(nrow.begin(), nrow.end(), rowptr);
  std::copy(ncol.begin(), ncol.end(), col);
  std::copy(nval.begin(), nval.end(), vals);
}

}  // namespace reorder
}  // namespace gcn
