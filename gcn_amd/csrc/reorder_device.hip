// reorder_device.hip — the degree / RCM orderings and the CSR rewrite ON THE GPU, with the same
// integer results as the host versions (reorder.cpp, i.e. the reference's order_deg.cu:19-45,
// order_rcm.cu:15-33 + algo_bfs.cu:11-39 and renumber.cu:190-217) — SURVEY.md §8(f).4.
//
// The reference's RCM is a SERIAL queue BFS over the degree-ascending relabelled graph: start at
// vertex 0, neighbours in ascending (relabelled) order, restart at the next unplaced index, reverse.
// That order is reproduced exactly by a level-synchronous formulation:
//   * the restarts visit the connected components in the order of their smallest vertex id, and a
//     component's BFS starts at that vertex  ->  label the components by their minimum id
//     (label propagation + pointer jumping) and start ALL of them at once (multi-source BFS);
//   * inside a component the serial queue holds level L+1 in the order (position of the FIRST
//     level-L vertex adjacent to y, then y ascending)  ->  per level: atomicMin of the parent's
//     frontier position into every unvisited neighbour, then a radix sort of the newly reached
//     vertices by (parent position, id).  Positions are global over all components, which preserves
//     the order inside each component (induction over levels);
//   * the serial output is component-major, level-major inside  ->  one final sort by
//     (component label, discovery sequence number).
// Works on the symmetrised pattern A ∪ Aᵀ (the reference's `Uadjlist`, directed = 0); for the
// symmetric patterns of GCN adjacencies that is also the `Dadjlist` (directed = 1) result.
// Everything is int32 / uint64 keys, hipCUB radix sorts and scans; no floating point.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include "spmm_kernels.h"

namespace gcn {

namespace {

typedef unsigned long long u64;

struct DevBuf {                                    // hipMalloc'ed scratch, freed on scope exit
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

#define GCN_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

inline int nblk(long long items, int per = 256) { return (int)((items + per - 1) / per); }

// total degree = stored entries in the row + stored entries in the column (edgelist.cu:97-99)
__global__ void rd_degree_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int n, int nnz,
                                 unsigned* __restrict__ deg) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) atomicAdd(&deg[t], (unsigned)(rowptr[t + 1] - rowptr[t]));
  if (t < nnz) atomicAdd(&deg[col[t]], 1u);
}

__global__ void rd_degree_keys_kernel(const unsigned* __restrict__ deg, int n, int desc, unsigned maxdeg,
                                      u64* __restrict__ keys) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const unsigned d = desc ? maxdeg - deg[v] : deg[v];          // (degree asc|desc, id asc): a strict total order
  keys[v] = ((u64)d << 32) | (unsigned)v;
}

// rank[low32(sorted[i])] = i
__global__ void rd_rank_from_sorted_kernel(const u64* __restrict__ sorted, int n, int* __restrict__ rank) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rank[(unsigned)sorted[i]] = i;
}

// both directions of every stored entry, in the relabelled numbering; one wave per row
__global__ void rd_edge_keys_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                    const int* __restrict__ rel, int n, u64* __restrict__ keys) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nw = gridDim.x * (blockDim.x >> 6);
  for (int r = wave; r < n; r += nw) {
    const u64 ru = (unsigned)rel[r];
    for (int e = rowptr[r] + lane; e < rowptr[r + 1]; e += 64) {
      const u64 rv = (unsigned)rel[col[e]];
      keys[2 * (size_t)e] = (ru << 32) | rv;
      keys[2 * (size_t)e + 1] = (rv << 32) | ru;
    }
  }
}

// off[v] = first index i with (keys[i] >> 32) >= v, v in [0, n]
__global__ void rd_offsets_kernel(const u64* __restrict__ keys, long long count, int n, long long* __restrict__ off) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v > n) return;
  long long lo = 0, hi = count;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if ((long long)(keys[mid] >> 32) < (long long)v) lo = mid + 1; else hi = mid;
  }
  off[v] = lo;
}

__global__ void rd_iota_kernel(int* __restrict__ a, int n, int base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = base + i;
}

__global__ void rd_fill_kernel(int* __restrict__ a, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

// one round of min-label propagation; one wave per vertex
__global__ void rd_cc_propagate_kernel(const u64* __restrict__ adj, const long long* __restrict__ off, int n,
                                       int* __restrict__ comp, int* __restrict__ changed) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nw = gridDim.x * (blockDim.x >> 6);
  for (int v = wave; v < n; v += nw) {
    int m = comp[v];
    const int mine = m;
    for (long long e = off[v] + lane; e < off[v + 1]; e += 64) m = min(m, comp[(unsigned)adj[e]]);
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
    if (lane == 0 && m < mine) {
      atomicMin(&comp[v], m);
      atomicMin(&comp[mine], m);             // hook the old representative as well: faster convergence
      *changed = 1;
    }
  }
}

__global__ void rd_cc_jump_kernel(int* __restrict__ comp, int n) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  int c = comp[v];
  while (comp[c] != c) c = comp[c];              // labels only ever decrease: the chain ends at a root
  comp[v] = c;
}

__global__ void rd_flag_roots_kernel(const int* __restrict__ comp, int n, char* __restrict__ flag) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) flag[v] = comp[v] == v;
}

// the frontier (positions 0..fcount-1) claims its unvisited neighbours: par[y] = min parent position;
// the first claim of a vertex appends it to the candidate list
__global__ void rd_bfs_expand_kernel(const u64* __restrict__ adj, const long long* __restrict__ off,
                                     const int* __restrict__ frontier, int fcount,
                                     const int* __restrict__ level, int* __restrict__ par,
                                     int* __restrict__ cand, int* __restrict__ ccount) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nw = gridDim.x * (blockDim.x >> 6);
  for (int p = wave; p < fcount; p += nw) {
    const int x = frontier[p];
    for (long long e = off[x] + lane; e < off[x + 1]; e += 64) {
      const int y = (int)(unsigned)adj[e];
      if (level[y] >= 0) continue;
      const int old = atomicMin(&par[y], p);
      if (old == 0x7fffffff) cand[atomicAdd(ccount, 1)] = y;
    }
  }
}

__global__ void rd_bfs_keys_kernel(const int* __restrict__ cand, int count, const int* __restrict__ par,
                                   u64* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) keys[i] = ((u64)(unsigned)par[cand[i]] << 32) | (unsigned)cand[i];
}

// the sorted new frontier: record level and discovery sequence number
__global__ void rd_bfs_commit_kernel(const u64* __restrict__ sorted, int count, int lvl, int base,
                                     int* __restrict__ frontier, int* __restrict__ level, int* __restrict__ seq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int y = (int)(unsigned)sorted[i];
  frontier[i] = y;
  level[y] = lvl;
  seq[y] = base + i;
}

__global__ void rd_final_keys_kernel(const int* __restrict__ comp, const int* __restrict__ seq, int n,
                                     u64* __restrict__ keys, int* __restrict__ ids) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  keys[v] = ((u64)(unsigned)comp[v] << 32) | (unsigned)seq[v];
  ids[v] = v;
}

// order[i] = relabelled vertex at BFS position i  ->  pos[order[i]] = n-1-i (reversed)
__global__ void rd_reverse_pos_kernel(const int* __restrict__ order, int n, int* __restrict__ rpos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rpos[order[i]] = n - 1 - i;
}

__global__ void rd_compose_kernel(const int* __restrict__ rel, const int* __restrict__ rpos, int n,
                                  int* __restrict__ rank) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n) rank[u] = rpos[rel[u]];
}

// ---- CSR rewrite ----
__global__ void rd_apply_keys_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                     const int* __restrict__ rank, int n, u64* __restrict__ keys,
                                     int* __restrict__ newlen, int* __restrict__ vomp) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nw = gridDim.x * (blockDim.x >> 6);
  for (int r = wave; r < n; r += nw) {
    const u64 ru = (unsigned)rank[r];
    if (lane == 0) { newlen[ru] = rowptr[r + 1] - rowptr[r]; vomp[ru] = r; }
    for (int e = rowptr[r] + lane; e < rowptr[r + 1]; e += 64) keys[e] = (ru << 32) | (unsigned)rank[col[e]];
  }
}

__global__ void rd_low32_kernel(const u64* __restrict__ keys, long long count, int* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = (int)(unsigned)keys[i];
}

__global__ void rd_check_perm_kernel(const int* __restrict__ rank, int n, int* __restrict__ hits, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = rank[i];
  if (r < 0 || r >= n) { *bad = 1; return; }
  if (atomicAdd(&hits[r], 1) != 0) *bad = 1;
}

int bits_for(unsigned long long maxval) {          // number of low bits that can be non-zero
  int b = 0;
  while (b < 64 && (maxval >> b) != 0) ++b;
  return b < 1 ? 1 : b;
}

hipError_t sort_keys(u64* in, u64* out, long long count, int end_bit, DevBuf& tmp, size_t& tmp_bytes, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  size_t need = 0;
  GCN_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, need, in, out, count, 0, end_bit, st));
  if (need > tmp_bytes) {
    if (tmp.p) { (void)hipFree(tmp.p); tmp.p = nullptr; }
    GCN_TRY(tmp.alloc(need));
    tmp_bytes = need;
  }
  return hipcub::DeviceRadixSort::SortKeys(tmp.p, need, in, out, count, 0, end_bit, st);
}

}  // namespace

// rank_out[old] = new (device int32 [n]);  which: 0 total (in+out), 1 out, 2 in;  desc != 0: largest first
hipError_t device_order_deg(const int* rowptr, const int* col, int n, int nnz, int which, int desc,
                            int* rank_out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  DevBuf deg, keys, sorted, tmp;
  size_t tmp_bytes = 0;
  GCN_TRY(deg.alloc(sizeof(unsigned) * (size_t)n));
  GCN_TRY(keys.alloc(sizeof(u64) * (size_t)n));
  GCN_TRY(sorted.alloc(sizeof(u64) * (size_t)n));
  GCN_TRY(hipMemsetAsync(deg.p, 0, sizeof(unsigned) * (size_t)n, st));
  const long long items = n > nnz ? n : nnz;
  // which = 1: rows only (pass nnz = 0 to the column part), which = 2: columns only (n = 0 to the row part)
  rd_degree_kernel<<<nblk(items), 256, 0, st>>>(rowptr, col, which == 2 ? 0 : n, which == 1 ? 0 : nnz, deg.as<unsigned>());
  rd_degree_keys_kernel<<<nblk(n), 256, 0, st>>>(deg.as<unsigned>(), n, desc, 0xffffffffu, keys.as<u64>());
  GCN_TRY(sort_keys(keys.as<u64>(), sorted.as<u64>(), n, 64, tmp, tmp_bytes, st));
  rd_rank_from_sorted_kernel<<<nblk(n), 256, 0, st>>>(sorted.as<u64>(), n, rank_out);
  GCN_TRY(hipGetLastError());
  return hipStreamSynchronize(st);
}

// rank_out[old] = new (device int32 [n]); levels_out (host, optional) = BFS levels of the deepest component
hipError_t device_order_rcm(const int* rowptr, const int* col, int n, int nnz, int* rank_out,
                            int* levels_out, hipStream_t st) {
  if (levels_out) *levels_out = 0;
  if (n <= 0) return hipSuccess;
  DevBuf rel, ekeys, esorted, tmp, off, comp, flag, level, par, seq, frontier, cand, counters, fkeys, fsorted, ids, order;
  size_t tmp_bytes = 0;
  GCN_TRY(rel.alloc(sizeof(int) * (size_t)n));
  GCN_TRY(device_order_deg(rowptr, col, n, nnz, 0, 0, rel.as<int>(), st));     // degree-ascending relabel

  // relabelled, symmetrised, de-duplicated adjacency with ascending neighbour lists
  const long long e2 = 2LL * nnz;
  GCN_TRY(ekeys.alloc(sizeof(u64) * (size_t)(e2 + 1)));
  GCN_TRY(esorted.alloc(sizeof(u64) * (size_t)(e2 + 1)));
  GCN_TRY(counters.alloc(sizeof(long long) * 4));
  rd_edge_keys_kernel<<<nblk((long long)n * 64 > (1LL << 24) ? (1LL << 24) : (long long)n * 64), 256, 0, st>>>(
      rowptr, col, rel.as<int>(), n, ekeys.as<u64>());
  const int idbits = bits_for((unsigned long long)(n > 1 ? n - 1 : 1));
  GCN_TRY(sort_keys(ekeys.as<u64>(), esorted.as<u64>(), e2, 32 + idbits, tmp, tmp_bytes, st));
  long long ucount = 0;
  if (e2 > 0) {
    size_t need = 0;
    GCN_TRY(hipcub::DeviceSelect::Unique(nullptr, need, esorted.as<u64>(), ekeys.as<u64>(), counters.as<long long>(), e2, st));
    if (need > tmp_bytes) { if (tmp.p) { (void)hipFree(tmp.p); tmp.p = nullptr; } GCN_TRY(tmp.alloc(need)); tmp_bytes = need; }
    GCN_TRY(hipcub::DeviceSelect::Unique(tmp.p, need, esorted.as<u64>(), ekeys.as<u64>(), counters.as<long long>(), e2, st));
    GCN_TRY(hipMemcpyAsync(&ucount, counters.p, sizeof(long long), hipMemcpyDeviceToHost, st));
    GCN_TRY(hipStreamSynchronize(st));
  }
  const u64* adj = ekeys.as<u64>();               // unique keys: low 32 bits = neighbour
  if (esorted.p) { (void)hipFree(esorted.p); esorted.p = nullptr; }
  GCN_TRY(off.alloc(sizeof(long long) * (size_t)(n + 1)));
  rd_offsets_kernel<<<nblk(n + 1), 256, 0, st>>>(adj, ucount, n, off.as<long long>());

  // connected components labelled by their smallest (relabelled) vertex
  GCN_TRY(comp.alloc(sizeof(int) * (size_t)n));
  rd_iota_kernel<<<nblk(n), 256, 0, st>>>(comp.as<int>(), n, 0);
  int* changed = reinterpret_cast<int*>(counters.as<long long>() + 1);
  const int wave_blocks = nblk((long long)n * 64 > (1LL << 24) ? (1LL << 24) : (long long)n * 64);
  for (int it = 0; it < n + 1; ++it) {
    GCN_TRY(hipMemsetAsync(changed, 0, sizeof(int), st));
    rd_cc_propagate_kernel<<<wave_blocks, 256, 0, st>>>(adj, off.as<long long>(), n, comp.as<int>(), changed);
    rd_cc_jump_kernel<<<nblk(n), 256, 0, st>>>(comp.as<int>(), n);
    int h = 0;
    GCN_TRY(hipMemcpyAsync(&h, changed, sizeof(int), hipMemcpyDeviceToHost, st));
    GCN_TRY(hipStreamSynchronize(st));
    if (!h) break;
  }

  // multi-source BFS from every component's smallest vertex
  GCN_TRY(level.alloc(sizeof(int) * (size_t)n));
  GCN_TRY(par.alloc(sizeof(int) * (size_t)n));
  GCN_TRY(seq.alloc(sizeof(int) * (size_t)n));
  GCN_TRY(frontier.alloc(sizeof(int) * (size_t)n));
  GCN_TRY(cand.alloc(sizeof(int) * (size_t)n));
  GCN_TRY(fkeys.alloc(sizeof(u64) * (size_t)n));
  GCN_TRY(fsorted.alloc(sizeof(u64) * (size_t)n));
  GCN_TRY(flag.alloc((size_t)n));
  GCN_TRY(ids.alloc(sizeof(int) * (size_t)n));
  rd_fill_kernel<<<nblk(n), 256, 0, st>>>(level.as<int>(), n, -1);
  rd_fill_kernel<<<nblk(n), 256, 0, st>>>(par.as<int>(), n, 0x7fffffff);
  rd_flag_roots_kernel<<<nblk(n), 256, 0, st>>>(comp.as<int>(), n, flag.as<char>());
  rd_iota_kernel<<<nblk(n), 256, 0, st>>>(ids.as<int>(), n, 0);
  int* ccount = reinterpret_cast<int*>(counters.as<long long>() + 2);
  int fcount = 0;
  {
    size_t need = 0;
    GCN_TRY(hipcub::DeviceSelect::Flagged(nullptr, need, ids.as<int>(), flag.as<char>(), cand.as<int>(), ccount, n, st));
    if (need > tmp_bytes) { if (tmp.p) { (void)hipFree(tmp.p); tmp.p = nullptr; } GCN_TRY(tmp.alloc(need)); tmp_bytes = need; }
    GCN_TRY(hipcub::DeviceSelect::Flagged(tmp.p, need, ids.as<int>(), flag.as<char>(), cand.as<int>(), ccount, n, st));
    GCN_TRY(hipMemcpyAsync(&fcount, ccount, sizeof(int), hipMemcpyDeviceToHost, st));
    GCN_TRY(hipStreamSynchronize(st));
  }
  // level 0: the roots in ascending order (Flagged keeps the input order); key = (0, id) is already sorted
  // (the commit kernel only reads the low 32 bits of a key = the vertex; the high half written here is unused)
  rd_bfs_keys_kernel<<<nblk(fcount), 256, 0, st>>>(cand.as<int>(), fcount, level.as<int>(), fsorted.as<u64>());
  rd_bfs_commit_kernel<<<nblk(fcount), 256, 0, st>>>(fsorted.as<u64>(), fcount, 0, 0, frontier.as<int>(), level.as<int>(), seq.as<int>());
  int base = fcount, lvl = 0;
  const int posbits = 32;
  while (fcount > 0 && base < n) {
    GCN_TRY(hipMemsetAsync(ccount, 0, sizeof(int), st));
    const long long waves = (long long)fcount * 64;
    rd_bfs_expand_kernel<<<nblk(waves > (1LL << 24) ? (1LL << 24) : waves), 256, 0, st>>>(
        adj, off.as<long long>(), frontier.as<int>(), fcount, level.as<int>(), par.as<int>(), cand.as<int>(), ccount);
    int cnt = 0;
    GCN_TRY(hipMemcpyAsync(&cnt, ccount, sizeof(int), hipMemcpyDeviceToHost, st));
    GCN_TRY(hipStreamSynchronize(st));
    if (cnt == 0) break;
    ++lvl;
    rd_bfs_keys_kernel<<<nblk(cnt), 256, 0, st>>>(cand.as<int>(), cnt, par.as<int>(), fkeys.as<u64>());
    GCN_TRY(sort_keys(fkeys.as<u64>(), fsorted.as<u64>(), cnt, posbits + 32, tmp, tmp_bytes, st));
    rd_bfs_commit_kernel<<<nblk(cnt), 256, 0, st>>>(fsorted.as<u64>(), cnt, lvl, base, frontier.as<int>(), level.as<int>(), seq.as<int>());
    base += cnt;
    fcount = cnt;
  }
  if (levels_out) *levels_out = lvl + 1;
  if (base != n) return hipErrorUnknown;          // every vertex belongs to some root's component

  // component-major, discovery order inside; reverse; compose with the degree relabel
  GCN_TRY(order.alloc(sizeof(int) * (size_t)n));
  rd_final_keys_kernel<<<nblk(n), 256, 0, st>>>(comp.as<int>(), seq.as<int>(), n, fkeys.as<u64>(), ids.as<int>());
  {
    size_t need = 0;
    GCN_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, fkeys.as<u64>(), fsorted.as<u64>(), ids.as<int>(), order.as<int>(), n, 0, 64, st));
    if (need > tmp_bytes) { if (tmp.p) { (void)hipFree(tmp.p); tmp.p = nullptr; } GCN_TRY(tmp.alloc(need)); tmp_bytes = need; }
    GCN_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, need, fkeys.as<u64>(), fsorted.as<u64>(), ids.as<int>(), order.as<int>(), n, 0, 64, st));
  }
  rd_reverse_pos_kernel<<<nblk(n), 256, 0, st>>>(order.as<int>(), n, par.as<int>());       // par reused as rpos
  rd_compose_kernel<<<nblk(n), 256, 0, st>>>(rel.as<int>(), par.as<int>(), n, rank_out);
  GCN_TRY(hipGetLastError());
  return hipStreamSynchronize(st);
}

// CSR rewrite under rank[old] = new: rows and columns relabelled, every row's columns ascending, values
// carried along (renumber.cu:190-217); vomp_out[new] = old.  -1 in *bad_rank_host: rank is not a permutation.
hipError_t device_csr_apply_rank(const int* rowptr, const int* col, const float* val, const int* rank, int n,
                                 int nnz, int* out_rowptr, int* out_col, float* out_val, int* vomp_out,
                                 int* bad_rank_host, hipStream_t st) {
  if (bad_rank_host) *bad_rank_host = 0;
  if (n <= 0) return hipSuccess;
  DevBuf keys, sorted, newlen, tmp, chk;
  size_t tmp_bytes = 0;
  GCN_TRY(chk.alloc(sizeof(int) * ((size_t)n + 1)));
  GCN_TRY(hipMemsetAsync(chk.p, 0, sizeof(int) * ((size_t)n + 1), st));
  rd_check_perm_kernel<<<nblk(n), 256, 0, st>>>(rank, n, chk.as<int>(), chk.as<int>() + n);
  int bad = 0;
  GCN_TRY(hipMemcpyAsync(&bad, chk.as<int>() + n, sizeof(int), hipMemcpyDeviceToHost, st));
  GCN_TRY(hipStreamSynchronize(st));
  if (bad) { if (bad_rank_host) *bad_rank_host = 1; return hipSuccess; }
  GCN_TRY(keys.alloc(sizeof(u64) * (size_t)(nnz + 1)));
  GCN_TRY(sorted.alloc(sizeof(u64) * (size_t)(nnz + 1)));
  GCN_TRY(newlen.alloc(sizeof(int) * ((size_t)n + 1)));
  GCN_TRY(hipMemsetAsync(newlen.p, 0, sizeof(int) * ((size_t)n + 1), st));
  rd_apply_keys_kernel<<<nblk((long long)n * 64 > (1LL << 24) ? (1LL << 24) : (long long)n * 64), 256, 0, st>>>(
      rowptr, col, rank, n, keys.as<u64>(), newlen.as<int>(), vomp_out);
  if (nnz > 0) {
    const int idbits = bits_for((unsigned long long)(n > 1 ? n - 1 : 1));
    size_t need = 0;
    GCN_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys.as<u64>(), sorted.as<u64>(), val, out_val, nnz, 0, 32 + idbits, st));
    GCN_TRY(tmp.alloc(need));
    tmp_bytes = need;
    GCN_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, need, keys.as<u64>(), sorted.as<u64>(), val, out_val, nnz, 0, 32 + idbits, st));
    rd_low32_kernel<<<nblk(nnz), 256, 0, st>>>(sorted.as<u64>(), nnz, out_col);
  }
  {
    size_t need = 0;
    GCN_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, newlen.as<int>(), out_rowptr, n + 1, st));
    if (need > tmp_bytes) { if (tmp.p) { (void)hipFree(tmp.p); tmp.p = nullptr; } GCN_TRY(tmp.alloc(need)); tmp_bytes = need; }
    GCN_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, need, newlen.as<int>(), out_rowptr, n + 1, st));
  }
  GCN_TRY(hipGetLastError());
  return hipStreamSynchronize(st);
}

}  // namespace gcn
