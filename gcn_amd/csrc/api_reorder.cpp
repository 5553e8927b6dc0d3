// api_reorder.cpp — C ABI of the vertex reorderers: the host versions (bit-exact with the reference's integer
// vectors, reorder.cpp) and the device versions (reorder_device.hip).  Contract: include/gcn_spmm.h (1c), (1d).
#include "plan.h"

#include <vector>

#include "reorder.h"

namespace gcn {

bool csr_ok(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t nnz) {
  if (n < 0 || nnz < 0 || !rowptr || (nnz > 0 && !col)) return false;
  if (rowptr[0] != 0 || rowptr[n] != nnz) return false;
  for (int32_t i = 0; i < n; ++i) if (rowptr[i + 1] < rowptr[i]) return false;
  for (int32_t e = 0; e < nnz; ++e) if (col[e] < 0 || col[e] >= n) return false;
  return true;
}

}  // namespace gcn

using gcn::csr_ok;

extern "C" {

int gcn_order_deg(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t nnz, int32_t which,
                  int32_t desc, int64_t* rank_out) {
  if (!rank_out || which < 0 || which > 2 || !csr_ok(rowptr, col, n, nnz)) return GCN_ERR_INVALID_ARG;
  gcn::reorder::Csr g{rowptr, col, n, nnz};
  auto r = gcn::reorder::order_deg(g, (gcn::reorder::DegKind)which, desc != 0);
  for (int32_t i = 0; i < n; ++i) rank_out[i] = (int64_t)r[i];
  return GCN_OK;
}

int gcn_order_rcm(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t nnz,
                  int32_t directed, int64_t* rank_out) {
  if (!rank_out || !csr_ok(rowptr, col, n, nnz)) return GCN_ERR_INVALID_ARG;
  gcn::reorder::Csr g{rowptr, col, n, nnz};
  auto r = gcn::reorder::order_rcm(g, directed != 0);
  for (int32_t i = 0; i < n; ++i) rank_out[i] = (int64_t)r[i];
  return GCN_OK;
}

int gcn_order_deg_device(const int32_t* rowptr_dev, const int32_t* col_dev, int32_t n, int32_t nnz,
                         int32_t which, int32_t desc, int32_t* rank_out_dev, void* stream) {
  if (n < 0 || nnz < 0 || which < 0 || which > 2 || (n > 0 && (!rowptr_dev || !rank_out_dev)) || (nnz > 0 && !col_dev))
    return GCN_ERR_INVALID_ARG;
  return gcn::device_order_deg(rowptr_dev, col_dev, n, nnz, which, desc ? 1 : 0, rank_out_dev,
                               (hipStream_t)stream) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

int gcn_order_rcm_device(const int32_t* rowptr_dev, const int32_t* col_dev, int32_t n, int32_t nnz,
                         int32_t* rank_out_dev, int32_t* bfs_levels_out, void* stream) {
  if (n < 0 || nnz < 0 || (n > 0 && (!rowptr_dev || !rank_out_dev)) || (nnz > 0 && !col_dev))
    return GCN_ERR_INVALID_ARG;
  int levels = 0;
  const hipError_t e = gcn::device_order_rcm(rowptr_dev, col_dev, n, nnz, rank_out_dev, &levels, (hipStream_t)stream);
  if (bfs_levels_out) *bfs_levels_out = levels;
  return e == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

int gcn_csr_apply_rank_device(const int32_t* rowptr_dev, const int32_t* col_dev, const float* val_dev,
                              const int32_t* rank_dev, int32_t n, int32_t nnz, int32_t* out_rowptr_dev,
                              int32_t* out_col_dev, float* out_val_dev, int32_t* vomp_out_dev, void* stream) {
  if (n < 0 || nnz < 0) return GCN_ERR_INVALID_ARG;
  if (n > 0 && (!rowptr_dev || !rank_dev || !out_rowptr_dev || !vomp_out_dev)) return GCN_ERR_INVALID_ARG;
  if (nnz > 0 && (!col_dev || !val_dev || !out_col_dev || !out_val_dev)) return GCN_ERR_INVALID_ARG;
  if (out_col_dev == col_dev || out_val_dev == val_dev || out_rowptr_dev == rowptr_dev) return GCN_ERR_INVALID_ARG;
  int bad = 0;
  const hipError_t e = gcn::device_csr_apply_rank(rowptr_dev, col_dev, val_dev, rank_dev, n, nnz, out_rowptr_dev,
                                                  out_col_dev, out_val_dev, vomp_out_dev, &bad, (hipStream_t)stream);
  if (e != hipSuccess) return GCN_ERR_HIP;
  return bad ? GCN_ERR_INVALID_ARG : GCN_OK;
}

int gcn_order_rabbit_device(const int32_t* rowptr_dev, const int32_t* col_dev, int32_t n, int32_t nnz,
                            int32_t* rank_out_dev, int32_t* community_out_dev, int64_t* stats_host, void* stream) {
  if (n < 0 || nnz < 0 || (n > 0 && (!rowptr_dev || !rank_out_dev)) || (nnz > 0 && !col_dev)) return GCN_ERR_INVALID_ARG;
  long long st8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const hipError_t e = gcn::device_order_rabbit(rowptr_dev, col_dev, n, nnz, rank_out_dev, community_out_dev, st8, (hipStream_t)stream);
  if (stats_host) for (int i = 0; i < 8; ++i) stats_host[i] = (int64_t)st8[i];
  if (e == hipErrorAssert) return GCN_ERR_INTERNAL;      // a loop guard tripped (stats_host[4..7]): no ordering was written
  return e == hipSuccess ? GCN_OK : (e == hipErrorOutOfMemory ? GCN_ERR_ALLOC : GCN_ERR_HIP);
}

int gcn_order_rabbit(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t nnz, int32_t* vomp_out,
                     int32_t* community_out) {
  if (!vomp_out || !csr_ok(rowptr, col, n, nnz)) return GCN_ERR_INVALID_ARG;
  gcn::reorder::Csr g{rowptr, col, n, nnz};
  std::vector<int32_t> comm;
  const auto vo = gcn::reorder::order_rabbit_vomp(g, false, community_out ? &comm : nullptr);
  for (int32_t i = 0; i < n; ++i) vomp_out[i] = vo[i];
  if (community_out) for (int32_t i = 0; i < n; ++i) community_out[i] = comm[i];
  return GCN_OK;
}

int gcn_order_gorder(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t nnz,
                     int32_t window, int64_t* rank_out) {
  if (!rank_out || window < 1 || !csr_ok(rowptr, col, n, nnz)) return GCN_ERR_INVALID_ARG;
  gcn::reorder::Csr g{rowptr, col, n, nnz};
  bool ok = true;
  auto r = gcn::reorder::order_gorder_complete(g, (gcn::reorder::u64)window, &ok);
  if (!ok) return GCN_ERR_INVALID_ARG;
  for (int32_t i = 0; i < n; ++i) rank_out[i] = (int64_t)r[i];
  return GCN_OK;
}

int gcn_csr_apply_rank(int32_t* rowptr, int32_t* col, float* vals, int32_t n, int32_t nnz,
                       const int64_t* rank, int32_t* vomp_out) {
  if (!rank || !vals || !csr_ok(rowptr, col, n, nnz)) return GCN_ERR_INVALID_ARG;
  std::vector<gcn::reorder::u64> r(n);
  std::vector<char> hit(n, 0);
  for (int32_t i = 0; i < n; ++i) {
    if (rank[i] < 0 || rank[i] >= n || hit[rank[i]]) return GCN_ERR_INVALID_ARG;   // bijection
    hit[rank[i]] = 1;
    r[i] = (gcn::reorder::u64)rank[i];
  }
  gcn::reorder::csr_apply_rank(rowptr, col, vals, n, nnz, r.data());
  if (vomp_out) for (int32_t i = 0; i < n; ++i) vomp_out[rank[i]] = i;
  return GCN_OK;
}

}  // extern "C"
