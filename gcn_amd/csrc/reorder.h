// Host-side vertex reorderers (preprocessing for the SpMM path).  Internal header;
// the public contract is include/gcn_spmm.h.  All functions reproduce the integer
// vectors of the reference bit for bit (see reorder.cpp for file:line citations).
#pragma once
#include <cstdint>
#include <vector>

namespace gcn {
namespace reorder {

typedef uint64_t u64;   // the reference's `ul` (tools.cuh:80)
typedef uint32_t u32;   // vertex ids inside the adjacency copies and the Gorder heap (n < 2^32 - 2)

struct Csr {
  const int32_t* rowptr;   // [n+1]
  const int32_t* col;      // [nnz]
  int64_t n;
  int64_t nnz;
};

enum DegKind { DEG_TOTAL = 0, DEG_OUT = 1, DEG_IN = 2 };

// all return rank[old] = new
std::vector<u64> order_deg(const Csr& g, DegKind which, bool desc);
std::vector<u64> order_rcm(const Csr& g, bool directed);
// RCM then Gorder, composed.  ok=false when the graph hits a case in which the
// reference's behaviour is undefined (isolated vertices inside the heap range).
std::vector<u64> order_gorder_complete(const Csr& g, u64 window, bool* ok);
// DFS pre-order over all components, roots in index order, neighbours in stored order
std::vector<u64> order_dfs(const Csr& g);
// Rabbit (serial modularity merging); returns vomp[new] = old
// (community_out, optional: the surviving top-level vertex every vertex ended up under)
std::vector<int32_t> order_rabbit_vomp(const Csr& g, bool verbose, std::vector<int32_t>* community_out = nullptr);

// CSR rewrite in the new numbering (rows moved, columns relabelled and sorted
// ascending, values carried along).  rank[old] = new.
void csr_apply_rank(int32_t* rowptr, int32_t* col, float* vals, int64_t n, int64_t nnz,
                    const u64* rank);

}  // namespace reorder
}  // namespace gcn
