// Host reorderers under AddressSanitizer + UBSan (CPU only; built and run by tests/test_sanitizers.py).
// Exercises every algorithm of gcn_amd/csrc/reorder.cpp on graphs with the shapes that stress its index
// arithmetic: isolated vertices, a star, a path, duplicate-degree ties, many components, a dense block.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <utility>
#include <vector>

#include "../gcn_amd/csrc/reorder.h"

using gcn::reorder::Csr;
using gcn::reorder::u64;

struct Graph { std::vector<int32_t> rowptr, col; std::vector<float> val; int64_t n; };

static Graph from_edges(int64_t n, const std::set<std::pair<int, int>>& und, bool self_loops) {
  std::vector<std::vector<int>> adj((size_t)n);
  for (auto [u, v] : und) { adj[u].push_back(v); if (u != v) adj[v].push_back(u); }
  if (self_loops) for (int i = 0; i < n; ++i) adj[i].push_back(i);
  Graph g; g.n = n; g.rowptr.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) {
    std::set<int> s(adj[i].begin(), adj[i].end());
    for (int c : s) { g.col.push_back(c); g.val.push_back(1.0f / (1 + (c + i) % 7)); }
    g.rowptr[i + 1] = (int32_t)g.col.size();
  }
  return g;
}

static void check_perm(const std::vector<u64>& r, int64_t n, const char* what) {
  std::vector<char> hit((size_t)n, 0);
  if ((int64_t)r.size() != n) { std::fprintf(stderr, "%s: wrong size\n", what); std::exit(2); }
  for (u64 x : r) { if (x >= (u64)n || hit[x]) { std::fprintf(stderr, "%s: not a permutation\n", what); std::exit(2); } hit[x] = 1; }
}

static void run_all(Graph g, const char* name, bool gorder_ok) {
  Csr c{g.rowptr.data(), g.col.data(), g.n, (int64_t)g.col.size()};
  check_perm(gcn::reorder::order_deg(c, gcn::reorder::DEG_TOTAL, true), g.n, "order_deg");
  check_perm(gcn::reorder::order_deg(c, gcn::reorder::DEG_IN, false), g.n, "order_deg in");
  check_perm(gcn::reorder::order_rcm(c, true), g.n, "order_rcm directed");
  check_perm(gcn::reorder::order_rcm(c, false), g.n, "order_rcm");
  check_perm(gcn::reorder::order_dfs(c), g.n, "order_dfs");
  if (gorder_ok) {
    bool ok = true;
    auto r = gcn::reorder::order_gorder_complete(c, 3, &ok);
    if (ok) check_perm(r, g.n, "gorder");
  }
  auto vo = gcn::reorder::order_rabbit_vomp(c, false);
  std::vector<u64> rank((size_t)g.n);
  for (int64_t i = 0; i < g.n; ++i) rank[(size_t)vo[(size_t)i]] = (u64)i;
  check_perm(rank, g.n, "rabbit");
  gcn::reorder::csr_apply_rank(g.rowptr.data(), g.col.data(), g.val.data(), g.n, (int64_t)g.col.size(), rank.data());
  std::printf("%s ok (n=%lld nnz=%zu)\n", name, (long long)g.n, g.col.size());
}

int main() {
  std::mt19937 rng(7);
  { std::set<std::pair<int, int>> e; for (int i = 0; i < 3000; ++i) { int u = rng() % 500, v = rng() % 500; if (u != v) e.insert({std::min(u, v), std::max(u, v)}); }
    run_all(from_edges(500, e, true), "random+loops", true); }
  { std::set<std::pair<int, int>> e; for (int i = 1; i < 200; ++i) e.insert({0, i});
    run_all(from_edges(200, e, true), "star", true); }
  { std::set<std::pair<int, int>> e; for (int i = 0; i + 1 < 300; ++i) e.insert({i, i + 1});
    run_all(from_edges(300, e, false), "path, no loops", true); }
  { std::set<std::pair<int, int>> e; for (int c = 0; c < 20; ++c) for (int i = 0; i < 6; ++i) for (int j = i + 1; j < 6; ++j) e.insert({c * 10 + i, c * 10 + j});
    run_all(from_edges(200, e, true), "20 cliques + isolated tails", true); }
  { std::set<std::pair<int, int>> e; for (int i = 0; i < 64; ++i) for (int j = i + 1; j < 64; ++j) e.insert({i, j});
    run_all(from_edges(100, e, false), "dense block + isolated vertices, no loops", false); }
  { std::set<std::pair<int, int>> e;
    run_all(from_edges(17, e, true), "only self-loops", true); }
  return 0;
}
