// rabbit_device.hip — Rabbit ordering on the GPU: parallel incremental aggregation.
//
// The reference's `rabbit` (renumber.cu:319-522) is the SERIAL modularity merging of Shiokawa'13 and names, at
// renumber.cu:328-330, the algorithm "Rabbit properly refers to": Arai et al., "Rabbit Order: Just-in-time Parallel
// Reordering for Fast Graph Analysis", IPDPS 2016.  This file is that parallel algorithm written for MI355X (SURVEY
// §8f.4); it does NOT reproduce the serial code's integers (gcn_amd.reorder.rabbit does, on the host) — its contract
// is the quality of the communities and of the ordering, which the tests bound against the host version.
//
//   * every vertex is processed exactly once, in ascending degree order, by ONE WAVE (a work counter hands the
//     order out; 1 024 waves are in flight, so ~1 000 consecutive vertices are merged concurrently);
//   * the wave LOCKS its vertex u (a bit in u's atom: from then on nobody can merge into u), gathers u's edges
//     LAZILY — u's own row plus the already aggregated (community, weight) lists of the vertices merged into u
//     so far — mapping every endpoint to its current community by following `dest` pointers, and adds them up per
//     community in the wave's hash table;
//   * the neighbouring community v with the largest modularity gain  w(u,v) - d(u)·d(v)/2m  (> 0; ties to the
//     smaller id, as the reference's key-ordered scan) takes u: one compare-and-swap on v's 64-bit atom
//     {lock, degree, newest child} adds d(u) and pushes u on v's child list; u's aggregated list is kept in a pool
//     for the moment v aggregates;  a target that is locked at that moment sends u to the retry list of the next pass;
//   * vertices that gain nothing stay top-level.  The order is the reference's dendrogram order (renumber.cu:477-489):
//     top-level vertices in index order, below each its merged vertices in merge order, depth first.
//
// Integer weights (unit edges), so the sums in the tables are exact and independent of the lanes' arrival order; the
// RESULT still depends on which merges race (as in the paper) — runs differ in detail, not in quality.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <vector>
#include "spmm_kernels.h"

namespace gcn {

namespace {

constexpr unsigned long long kLock = 1ull << 63;
constexpr unsigned kNone = 0xFFFFFFFFu;
constexpr int kHashCap = 1 << 16;                     // slots per wave (global memory); lists up to half of it are aggregated
constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ unsigned atom_deg(unsigned long long a) { return (unsigned)((a & ~kLock) >> 32); }
__device__ __forceinline__ unsigned atom_child(unsigned long long a) { return (unsigned)a; }

struct RabbitArgs {
  const int* rowptr; const int* col; int n;
  unsigned long long* atom;            // [n] {lock:1, degree:31, newest child:32}
  unsigned* dest;                      // [n] community a vertex was merged into (itself: top-level so far)
  unsigned* sibling;                   // [n] next (older) child of the same parent
  unsigned long long* agg_ptr;         // [n] where the aggregated list of a merged vertex starts in the pool
  unsigned* agg_len;                   // [n]
  uint2* pool; unsigned long long* pool_head; unsigned long long pool_cap;
  const unsigned* list; unsigned count; unsigned* counter;      // this pass's vertices, in processing order
  unsigned* retry; unsigned* nretry;                             // ... and the next pass's
  unsigned* hkey; unsigned* hval;                                // [waves * kHashCap]
  unsigned* skipped;                                             // vertices whose lists did not fit (left top-level)
  double two_m_inv;
};

// current community of x: follow the merge pointers (they only ever move up, so a stale read still lands on an ancestor)
__device__ __forceinline__ unsigned find_root(unsigned* dest, unsigned x) {
  unsigned p = dest[x];
  if (p == x) return x;
  const unsigned x0 = x;
  do { x = p; p = dest[x]; } while (p != x);
  dest[x0] = x;                                       // one-step shortcut (benign race: any ancestor is a valid value)
  return x;
}

__device__ __forceinline__ void table_add(unsigned* hkey, unsigned* hval, unsigned key, unsigned w) {
  unsigned i = (key * 2654435761u) >> 16;             // kHashCap = 2^16
  for (;;) {
    const unsigned k = hkey[i];
    if (k == key) break;
    if (k == kNone) {
      const unsigned old = atomicCAS(&hkey[i], kNone, key);
      if (old == kNone || old == key) break;
    }
    i = (i + 1) & (kHashCap - 1);
  }
  atomicAdd(&hval[i], w);
}

__global__ void __launch_bounds__(64 * kWavesPerBlock)
rabbit_pass_kernel(RabbitArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  unsigned* hkey = a.hkey + (size_t)wave * kHashCap;
  unsigned* hval = a.hval + (size_t)wave * kHashCap;
  for (;;) {
    unsigned idx = 0;
    if (lane == 0) idx = atomicAdd(a.counter, 1u);
    idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
    if (idx >= a.count) break;
    const unsigned u = a.list[idx];
    // 1. lock u: nobody merges into it from here on; its degree and child list are final for this step
    unsigned long long au = 0;
    if (lane == 0) au = atomicOr(&a.atom[u], kLock);
    au = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(au >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)au);
    const unsigned du = atom_deg(au);
    // 2. how long are the lists to aggregate?  (own row + the lists of the merged vertices)
    const int rbeg = a.rowptr[u], rend = a.rowptr[u + 1];
    unsigned long long total = (unsigned long long)(rend - rbeg);
    for (unsigned c = atom_child(au); c != kNone; c = a.sibling[c]) total += a.agg_len[c];
    if (total > (unsigned long long)kHashCap / 2 || du == 0) {   // too long for the table (hubs of hubs), or isolated: stays top-level
      if (lane == 0) {
        atomicAnd(&a.atom[u], ~kLock);
        if (du != 0) atomicAdd(a.skipped, 1u);
      }
      continue;
    }
    // 3. lazy aggregation: weights per current community
    for (int e = rbeg + lane; e < rend; e += 64) {
      const unsigned x = (unsigned)a.col[e];
      if (x == u) continue;
      const unsigned r = find_root(a.dest, x);
      if (r != u) table_add(hkey, hval, r, 1u);
    }
    for (unsigned c = atom_child(au); c != kNone; c = a.sibling[c]) {
      const uint2* lst = a.pool + a.agg_ptr[c];
      const unsigned len = a.agg_len[c];
      for (unsigned i = lane; i < len; i += 64) {
        const uint2 kw = lst[i];
        const unsigned r = find_root(a.dest, kw.x);
        if (r != u) table_add(hkey, hval, r, kw.y);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // 4. scan the table: best gain, number of entries
    double best = 0.0;
    unsigned bestv = kNone, mine = 0;
    const double du_2m = (double)du * a.two_m_inv;
    for (int i = lane; i < kHashCap; i += 64) {
      const unsigned k = hkey[i];
      if (k == kNone) continue;
      ++mine;
      const double dq = (double)hval[i] - (double)atom_deg(a.atom[k]) * du_2m;
      if (dq > best || (dq == best && dq > 0.0 && k < bestv)) { best = dq; bestv = k; }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off);
      const unsigned ov = __shfl_xor(bestv, off);
      if (ob > best || (ob == best && ob > 0.0 && ov < bestv)) { best = ob; bestv = ov; }
    }
    // 5. the entries leave the table (into the pool when u is going to be merged), the table is empty again
    unsigned incl = mine;                               // inclusive prefix of the lanes' counts
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    const unsigned cnt = __shfl(incl, 63);
    const bool merging = bestv != kNone && best > 0.0;
    unsigned long long base = 0;
    bool stored = false;
    if (merging) {
      if (lane == 0) base = atomicAdd(a.pool_head, (unsigned long long)cnt);
      base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
      stored = base + cnt <= a.pool_cap;
    }
    unsigned pos = incl - mine;
    for (int i = lane; i < kHashCap; i += 64) {
      const unsigned k = hkey[i];
      if (k == kNone) continue;
      if (stored) a.pool[base + pos] = make_uint2(k, hval[i]);
      ++pos;
      hkey[i] = kNone;
      hval[i] = 0;
    }
    if (lane == 0) {
      bool done = false;
      if (merging && stored) {
        a.agg_ptr[u] = base;
        a.agg_len[u] = cnt;
        // 6. push u on bestv's child list and add its degree: one CAS on bestv's atom
        for (int tries = 0; tries < 8 && !done; ++tries) {
          const unsigned long long av = __hip_atomic_load(&a.atom[bestv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (av & kLock) break;                         // being processed (or merged away): try again next pass
          a.sibling[u] = atom_child(av);
          __threadfence();                               // list, length and sibling are visible before u becomes a child
          const unsigned long long want = ((unsigned long long)(atom_deg(av) + du) << 32) | u;
          if (atomicCAS(&a.atom[bestv], av, want) == av) {
            __threadfence();
            a.dest[u] = bestv;                           // (u keeps its lock for good: nothing merges into a merged vertex)
            done = true;
          }
        }
        if (!done) {                                     // u stays a candidate
          atomicAnd(&a.atom[u], ~kLock);
          a.retry[atomicAdd(a.nretry, 1u)] = u;
        }
      } else {
        if (merging && !stored) atomicAdd(a.skipped, 1u);   // pool exhausted: u stays top-level
        atomicAnd(&a.atom[u], ~kLock);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void rabbit_init_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                   unsigned long long* __restrict__ atom, unsigned* __restrict__ dest,
                                   unsigned* __restrict__ sibling, unsigned* __restrict__ agg_len,
                                   unsigned long long* __restrict__ two_m) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  unsigned d = 0;
  for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) d += col[e] != v;     // self-loops carry no modularity
  atom[v] = ((unsigned long long)d << 32) | kNone;
  dest[v] = (unsigned)v;
  sibling[v] = kNone;
  agg_len[v] = 0;
  if (d) atomicAdd(two_m, (unsigned long long)d);
}

__global__ void rabbit_order_list_kernel(const int* __restrict__ rank, int n, unsigned* __restrict__ list) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) list[rank[v]] = (unsigned)v;
}

__global__ void rabbit_fill_kernel(unsigned* p, size_t count, unsigned v) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) p[i] = v;
}

template <class T>
struct Dev {                                             // scoped device buffer
  T* p = nullptr;
  ~Dev() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t count) { return hipMalloc((void**)&p, sizeof(T) * (count ? count : 1)); }
};

}  // namespace

// rank_out_dev [n] (rank[old] = new), community_out_dev [n] or null (top-level vertex of every vertex),
// stats_host[4] or null: {communities, passes, retried vertices, vertices left top-level for lack of table / pool room}
hipError_t device_order_rabbit(const int* rowptr, const int* col, int n, int nnz, int* rank_out_dev, int* community_out_dev,
                               long long* stats_host, hipStream_t st) {
  if (stats_host) stats_host[0] = stats_host[1] = stats_host[2] = stats_host[3] = 0;
  if (n <= 0) return hipSuccess;
#define GCN_R(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)
  int cu = 256;
  { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cu = prop.multiProcessorCount; }
  const int nblocks = cu;                                // 4 waves per CU: 1 024 vertices in flight on MI355X
  const int nwaves = nblocks * kWavesPerBlock;
  Dev<unsigned long long> atom, agg_ptr, scal;           // scal: {two_m, pool_head}
  Dev<unsigned> dest, sibling, agg_len, listA, listB, hkey, hval, cnt;   // cnt: {counter, nretry, skipped}
  Dev<uint2> pool;
  Dev<int> degrank;
  const unsigned long long pool_cap = 4ull * (unsigned long long)(nnz > 0 ? nnz : 1) + (unsigned long long)n;
  GCN_R(atom.alloc(n)); GCN_R(agg_ptr.alloc(n)); GCN_R(scal.alloc(2));
  GCN_R(dest.alloc(n)); GCN_R(sibling.alloc(n)); GCN_R(agg_len.alloc(n)); GCN_R(listA.alloc(n)); GCN_R(listB.alloc(n));
  GCN_R(hkey.alloc((size_t)nwaves * kHashCap)); GCN_R(hval.alloc((size_t)nwaves * kHashCap)); GCN_R(cnt.alloc(4));
  GCN_R(pool.alloc(pool_cap)); GCN_R(degrank.alloc(n));
  GCN_R(hipMemsetAsync(scal.p, 0, sizeof(unsigned long long) * 2, st));
  GCN_R(hipMemsetAsync(cnt.p, 0, sizeof(unsigned) * 4, st));
  GCN_R(hipMemsetAsync(hval.p, 0, sizeof(unsigned) * (size_t)nwaves * kHashCap, st));
  rabbit_fill_kernel<<<1024, 256, 0, st>>>(hkey.p, (size_t)nwaves * kHashCap, kNone);
  rabbit_init_kernel<<<(n + 255) / 256, 256, 0, st>>>(rowptr, col, n, atom.p, dest.p, sibling.p, agg_len.p, scal.p);
  GCN_R(hipGetLastError());
  // processing order: ascending degree, ties by vertex id (the strict total order of order_deg, device version)
  GCN_R(device_order_deg(rowptr, col, n, nnz, /*which = out*/ 1, /*desc*/ 0, degrank.p, st));
  rabbit_order_list_kernel<<<(n + 255) / 256, 256, 0, st>>>(degrank.p, n, listA.p);
  GCN_R(hipGetLastError());
  unsigned long long two_m = 0;
  GCN_R(hipMemcpyAsync(&two_m, scal.p, sizeof(two_m), hipMemcpyDeviceToHost, st));
  GCN_R(hipStreamSynchronize(st));
  long long passes = 0, retried = 0;
  if (two_m > 0) {
    RabbitArgs a;
    a.rowptr = rowptr; a.col = col; a.n = n; a.atom = atom.p; a.dest = dest.p; a.sibling = sibling.p;
    a.agg_ptr = agg_ptr.p; a.agg_len = agg_len.p; a.pool = pool.p; a.pool_head = scal.p + 1; a.pool_cap = pool_cap;
    a.counter = cnt.p; a.nretry = cnt.p + 1; a.skipped = cnt.p + 2; a.hkey = hkey.p; a.hval = hval.p;
    a.two_m_inv = 1.0 / (double)two_m;
    unsigned count = (unsigned)n;
    unsigned* cur = listA.p;
    unsigned* nxt = listB.p;
    // a vertex whose target was locked at the moment of the merge comes back in the next pass; a handful of passes
    // empties the list (what is left after 16 stays top-level)
    for (; count > 0 && passes < 16; ++passes) {
      a.list = cur; a.count = count; a.retry = nxt;
      GCN_R(hipMemsetAsync(cnt.p, 0, sizeof(unsigned) * 2, st));
      rabbit_pass_kernel<<<nblocks, 64 * kWavesPerBlock, 0, st>>>(a);
      GCN_R(hipGetLastError());
      unsigned h[2] = {0, 0};
      GCN_R(hipMemcpyAsync(h, cnt.p, sizeof(h), hipMemcpyDeviceToHost, st));
      GCN_R(hipStreamSynchronize(st));
      count = h[1];
      retried += count;
      unsigned* t = cur; cur = nxt; nxt = t;
    }
  }
  // the dendrogram -> the order (renumber.cu:477-489): top-level vertices in index order, each followed depth-first by
  // the vertices merged into it, in merge order.  O(n) pointer chasing: on the host.
  std::vector<unsigned long long> h_atom((size_t)n);
  std::vector<unsigned> h_dest((size_t)n), h_sib((size_t)n);
  unsigned h_skipped = 0;
  GCN_R(hipMemcpyAsync(h_atom.data(), atom.p, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost, st));
  GCN_R(hipMemcpyAsync(h_dest.data(), dest.p, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToHost, st));
  GCN_R(hipMemcpyAsync(h_sib.data(), sibling.p, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToHost, st));
  GCN_R(hipMemcpyAsync(&h_skipped, cnt.p + 2, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  GCN_R(hipStreamSynchronize(st));
  std::vector<int> rank((size_t)n), comm((size_t)n);
  std::vector<unsigned> stack, kids;
  long long ncomm = 0;
  int next = 0;
  for (int r = 0; r < n; ++r) {
    if (h_dest[(size_t)r] != (unsigned)r) continue;
    ++ncomm;
    stack.push_back((unsigned)r);
    while (!stack.empty()) {
      const unsigned v = stack.back();
      stack.pop_back();
      rank[v] = next++;
      comm[v] = r;
      // children newest first on the chain; pushed in that order they are popped oldest first (merge order)
      for (unsigned c = (unsigned)h_atom[v]; c != kNone; c = h_sib[c]) stack.push_back(c);
    }
  }
  if (next != n) return hipErrorUnknown;                 // (a broken dendrogram: never observed; refuse rather than return a non-permutation)
  GCN_R(hipMemcpyAsync(rank_out_dev, rank.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
  if (community_out_dev) GCN_R(hipMemcpyAsync(community_out_dev, comm.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
  GCN_R(hipStreamSynchronize(st));
  if (stats_host) { stats_host[0] = ncomm; stats_host[1] = passes; stats_host[2] = retried; stats_host[3] = (long long)h_skipped; }
#undef GCN_R
  return hipSuccess;
}

}  // namespace gcn
