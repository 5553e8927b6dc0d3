// rabbit_device.hip — Rabbit ordering on the GPU: parallel incremental aggregation.
//
// The reference's `rabbit` (renumber.cu:319-522) is the SERIAL modularity merging of Shiokawa'13 and names, at
// renumber.cu:328-330, the algorithm "Rabbit properly refers to": Arai et al., "Rabbit Order: Just-in-time Parallel
// Reordering for Fast Graph Analysis", IPDPS 2016.  This file is that parallel algorithm written for MI355X (SURVEY
// §8f.4); it does NOT reproduce the serial code's integers (gcn_amd.reorder.rabbit does, on the host) — its contract
// is the quality of the communities and of the ordering, which the tests bound against the host version.
//
//   * every vertex is processed exactly once, in ascending degree order, by ONE WAVE (a work counter hands the
//     order out; 512 waves are in flight — one per 64 KiB of LDS — so ~500 consecutive vertices are merged concurrently);
//   * the wave LOCKS its vertex u (a bit in u's atom: from then on nobody can merge into u), gathers u's edges
//     LAZILY — u's own row plus the already aggregated (community, weight) lists of the vertices merged into u
//     so far — mapping every endpoint to its current community by following `dest` pointers, and adds them up per
//     community in the wave's hash table (LDS for lists of up to 4 096 entries, memory beyond);
//   * the neighbouring community v with the largest modularity gain  w(u,v) - d(u)·d(v)/2m  (> 0; ties to the
//     smaller id, as the reference's key-ordered scan) takes u: one compare-and-swap on v's 64-bit atom
//     {lock, degree, newest child} adds d(u) and pushes u on v's child list; u's aggregated list is kept in a pool
//     for the moment v aggregates;  a target that is locked at that moment sends u to the retry list of the next pass;
//   * the vertices that come back are processed in ROUNDS, smallest community first (the reference sorts every round by
//     current degree too, renumber.cu:408-409), each starting from the list it aggregated last time plus the children
//     merged since (every list is folded once); at most 1 in 64 vertices of a round is in flight, so merges are seen
//     promptly; a community whose lists outgrow the wave's table (64 k entries) stays top-level;
//   * vertices that gain nothing stay top-level.  The order is the reference's dendrogram order (renumber.cu:477-489):
//     top-level vertices in index order, below each its merged vertices in merge order, depth first.
//
// Integer weights (unit edges), so the sums in the tables are exact and independent of the lanes' arrival order; the
// RESULT still depends on which merges race (as in the paper) — runs differ in detail, not in quality.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
#include "spmm_kernels.h"

namespace gcn {

namespace {

constexpr unsigned long long kLock = 1ull << 63;
constexpr unsigned kNone = 0xFFFFFFFFu;
constexpr int kLdsCap = 1 << 13;                      // slots of the wave's LDS table (64 KiB: keys + weights); lists up to half of it
constexpr int kBigCap = 1 << 17;                      // slots of the wave's table in memory for longer lists (up to half of it)
constexpr int kMaxPasses = 64;
constexpr int kCntWords = 8 + 2 * kMaxPasses;       // {-, -, skipped, -, guard trips x4}, then {work counter, retry count} of every pass
constexpr int kBlocksPerCu = 2;                       // one wave per block, 64 KiB of LDS each

__device__ __forceinline__ unsigned atom_deg(unsigned long long a) { return (unsigned)((a & ~kLock) >> 32); }
__device__ __forceinline__ unsigned atom_child(unsigned long long a) { return (unsigned)a; }

// Everything another wave may have written a moment ago is read past the CU's L1 (which is not coherent with other CUs'
// stores and never sees an atomic's result): device-scope relaxed atomic loads.
__device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct RabbitArgs {
  const int* rowptr; const int* col; int n;
  unsigned long long* atom;            // [n] {lock:1, degree:31, newest child:32}
  unsigned* dest;                      // [n] community a vertex was merged into (itself: top-level so far)
  unsigned* sibling;                   // [n] next (older) child of the same parent
  unsigned long long* agg_ptr;         // [n] where the aggregated list of a vertex starts in the pool (written whenever it is aggregated)
  unsigned* agg_len;                   // [n] its length; 0: never aggregated (the list is the vertex's own row)
  unsigned* agg_child;                 // [n] newest child that list already covers (children merged later are newer on the chain)
  unsigned long long* pool;            // {community, weight} pairs, 8 bytes each
  unsigned long long* pool_head; unsigned long long pool_cap;
  const unsigned* list; unsigned count; unsigned* counter;      // this pass's vertices, in processing order
  unsigned* retry; unsigned* nretry;                             // ... and the next pass's
  unsigned* bkey; unsigned* bval; unsigned* btouch;              // per wave: kBigCap slots + the list of slots in use
  unsigned* skipped;                                             // vertices whose lists did not fit (left top-level)
  unsigned* beat;                                                // GCN_AMD_RABBIT_BEAT=1: per wave {vertex, stage, x} watched from the host; null: off
  unsigned* dbg;                                                 // [4] guard trips: pointer chains, child chains, full table (all 0 in a sound run)
  double two_m_inv;
};

#define GCN_BEAT(a_, lane_, stage_, x_) do { if ((a_).beat && (lane_) == 0) { st_u32((a_).beat + 4 * blockIdx.x + 1, (stage_)); \
  st_u32((a_).beat + 4 * blockIdx.x + 2, (unsigned)(x_)); } } while (0)

// current community of x: follow the merge pointers (they only ever move up, so a stale value still names an ancestor)
__device__ __forceinline__ unsigned find_root(unsigned* dest, unsigned x, unsigned* dbg) {
  unsigned p = ld_u32(dest + x);
  if (p == x) return x;
  const unsigned x0 = x;
  int guard = 1 << 16;                                // (chains are short; a bound so that no wave can ever spin for good)
  do { x = p; p = ld_u32(dest + x); } while (p != x && --guard > 0);   // (every value ever stored in dest is a vertex id)
  if (guard <= 0) atomicAdd(dbg + 0, 1u);
  st_u32(dest + x0, x);                               // one-step shortcut (benign race: any ancestor is a valid value)
  return x;
}

// The wave's table of (community, weight): LDS for lists of up to kLdsCap / 2 entries, memory beyond (there every access
// is an atomic or an L1-bypassing load / store, and the slots in use are kept on a touch list so that walking and
// emptying the table costs the entries, not the capacity).
template <bool LDS>
struct Table {
  // touch: the slots in use, in the order they were taken (ntouch of them) — so that walking and emptying the table costs
  // its entries, not its capacity.  LDS table: 16-bit slot numbers in LDS (r04; the walk used to sweep all 8 192 slots twice
  // per vertex, whatever the list length); table in memory: 32-bit slot numbers in memory.
  unsigned* key; unsigned* val; unsigned* touch; unsigned* ntouch; unsigned* dbg; unsigned short* ltouch;
  static constexpr unsigned kMask = (unsigned)(LDS ? kLdsCap : kBigCap) - 1u;
  __device__ __forceinline__ unsigned peek(unsigned i) const { return LDS ? key[i] : ld_u32(key + i); }
  __device__ __forceinline__ unsigned weight(unsigned i) const { return LDS ? val[i] : ld_u32(val + i); }
  __device__ __forceinline__ void clear(unsigned i) {
    if (LDS) { key[i] = kNone; val[i] = 0; } else { st_u32(key + i, kNone); st_u32(val + i, 0u); }
  }
  __device__ __forceinline__ void add(unsigned k, unsigned w) {
    unsigned i = (k * 2654435761u) & kMask;
    for (unsigned probes = 0; probes <= kMask; ++probes) {   // (never more than half full: the bound only rules out a spin)
      const unsigned cur = peek(i);
      if (cur == k) { atomicAdd(&val[i], w); return; }
      if (cur == kNone) {
        const unsigned old = atomicCAS(&key[i], kNone, k);
        if (old == kNone) {
          if (LDS) ltouch[atomicAdd(ntouch, 1u)] = (unsigned short)i;
          else     st_u32(touch + atomicAdd(ntouch, 1u), i);
          atomicAdd(&val[i], w);
          return;
        }
        if (old == k) { atomicAdd(&val[i], w); return; }
      }
      i = (i + 1) & kMask;
    }
    atomicAdd(dbg + 2, 1u);
  }
};

// steps 3-6 for vertex u (locked by this wave; degree du, newest child `child`): aggregate, choose, merge or release
template <bool LDS>
__device__ __noinline__ void rabbit_vertex(const RabbitArgs& a, Table<LDS> t, unsigned u, unsigned du, unsigned child,
                                           unsigned* ntouch_lds, int lane) {
  // 3. lazy aggregation: weights per current community.  A vertex that comes back (its target was locked last time)
  // starts from the list it aggregated then — every list is folded once — plus the children merged since.
  const unsigned own_len = a.agg_len[u];
  const unsigned covered = own_len ? a.agg_child[u] : kNone;
  const unsigned nverts = (unsigned)a.n;
  if (own_len) {
    const unsigned long long op = a.agg_ptr[u];
    const unsigned long long* lst = a.pool + op;
    const unsigned len = op + own_len <= a.pool_cap ? own_len : 0u;
    for (unsigned i = lane; i < len; i += 64) {
      const unsigned long long kw = lst[i];             // (written by this vertex's own earlier step: a kernel boundary ago)
      if ((unsigned)(kw >> 32) >= nverts) { atomicAdd(a.dbg + 3, 1u); continue; }   // (never, in a sound run: an index is checked before it is followed)
      const unsigned r = find_root(a.dest, (unsigned)(kw >> 32), a.dbg);
      if (r != u) t.add(r, (unsigned)kw);
    }
  } else {
    const int rbeg = a.rowptr[u], rend = a.rowptr[u + 1];
    for (int e = rbeg + lane; e < rend; e += 64) {
      const unsigned x = (unsigned)a.col[e];
      if (x == u) continue;
      const unsigned r = find_root(a.dest, x, a.dbg);
      if (r != u) t.add(r, 1u);
    }
  }
  int walked = 0;
  for (unsigned c = child; c != kNone && c != covered; c = ld_u32(a.sibling + c)) {
    if (c >= nverts || ++walked > a.n) { if (lane == 0) atomicAdd(a.dbg + 1, 1u); break; }
    const unsigned long long cp = ld_u64(a.agg_ptr + c);
    const unsigned clen = ld_u32(a.agg_len + c);
    const unsigned long long* lst = a.pool + cp;
    const unsigned len = cp + clen <= a.pool_cap ? clen : 0u;
    if (len != clen && lane == 0) atomicAdd(a.dbg + 3, 1u);
    for (unsigned i = lane; i < len; i += 64) {
      const unsigned long long kw = ld_u64(lst + i);
      if ((unsigned)(kw >> 32) >= nverts) { atomicAdd(a.dbg + 3, 1u); continue; }
      const unsigned r = find_root(a.dest, (unsigned)(kw >> 32), a.dbg);
      if (r != u) t.add(r, (unsigned)kw);
    }
  }
  GCN_BEAT(a, lane, 4, 0);
  __syncthreads();
  // 4. walk the entries — the slots on the touch list: best gain, number of entries
  const unsigned nslots = *ntouch_lds;
  double best = 0.0;
  unsigned bestv = kNone, mine = 0;
  const double du_2m = (double)du * a.two_m_inv;
  for (unsigned q = lane; q < nslots; q += 64) {
    const unsigned i = LDS ? (unsigned)t.ltouch[q] : ld_u32(t.touch + q);
    const unsigned k = t.peek(i);
    if (k == kNone) continue;
    ++mine;
    if (k >= (unsigned)a.n) continue;
    const double dq = (double)t.weight(i) - (double)atom_deg(ld_u64(a.atom + k)) * du_2m;
    if (dq > best || (dq == best && dq > 0.0 && k < bestv)) { best = dq; bestv = k; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_xor(best, off);
    const unsigned ov = __shfl_xor(bestv, off);
    if (ob > best || (ob == best && ob > 0.0 && ov < bestv)) { best = ob; bestv = ov; }
  }
  GCN_BEAT(a, lane, 5, bestv);
  // 5. the entries leave the table (into the pool when u is going to be merged); the table is empty again
  unsigned incl = mine;                               // inclusive prefix of the lanes' counts
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned tt = __shfl_up(incl, off);
    if (lane >= off) incl += tt;
  }
  const unsigned cnt = __shfl(incl, 63);
  const bool merging = bestv < (unsigned)a.n && best > 0.0;
  unsigned long long base = 0;
  bool stored = false;
  if (merging) {
    if (lane == 0) base = atomicAdd(a.pool_head, (unsigned long long)cnt);
    base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
    stored = base + cnt <= a.pool_cap;
  }
  unsigned pos = incl - mine;
  for (unsigned q = lane; q < nslots; q += 64) {
    const unsigned i = LDS ? (unsigned)t.ltouch[q] : ld_u32(t.touch + q);
    const unsigned k = t.peek(i);
    if (k == kNone) continue;
    if (stored) a.pool[base + pos] = ((unsigned long long)k << 32) | t.weight(i);
    ++pos;
    t.clear(i);
  }
  __syncthreads();
  GCN_BEAT(a, lane, 6, cnt);
  if (lane == 0) {
    *ntouch_lds = 0;
    bool done = false;
    if (merging && stored) {
      a.agg_ptr[u] = base;
      a.agg_len[u] = cnt;
      a.agg_child[u] = child;
      // 6. push u on bestv's child list and add its degree: one CAS on bestv's atom
      for (int tries = 0; tries < 8 && !done; ++tries) {
        const unsigned long long av = ld_u64(a.atom + bestv);
        if (av & kLock) break;                         // being processed (or merged away): try again next pass
        a.sibling[u] = atom_child(av);
        __threadfence();                               // list, length and sibling are visible before u becomes a child
        const unsigned long long want = ((unsigned long long)(atom_deg(av) + du) << 32) | u;
        if (atomicCAS(&a.atom[bestv], av, want) == av) {
          __threadfence();
          st_u32(a.dest + u, bestv);                   // (u keeps its lock for good: nothing merges into a merged vertex)
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
  __syncthreads();
}

__global__ void __launch_bounds__(64)
rabbit_pass_kernel(RabbitArgs a) {
  __shared__ unsigned lkey[kLdsCap], lval[kLdsCap];
  __shared__ unsigned short ltouch[kLdsCap / 2];       // (72 KiB per wave in all: two waves per CU, as before)
  __shared__ unsigned ntouch;
  const int lane = threadIdx.x;
  const int wave = blockIdx.x;
  for (int i = lane; i < kLdsCap; i += 64) { lkey[i] = kNone; lval[i] = 0; }
  if (lane == 0) ntouch = 0;
  __syncthreads();
  for (;;) {
    unsigned idx = 0;
    if (lane == 0) idx = atomicAdd(a.counter, 1u);
    idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
    if (idx >= a.count) break;
    const unsigned u = a.list[idx];
    if (a.beat && lane == 0) st_u32(a.beat + 4 * blockIdx.x, u);
    GCN_BEAT(a, lane, 1, idx);
    // 1. lock u: nobody merges into it from here on; its degree and child list are final for this step
    unsigned long long au = 0;
    if (lane == 0) au = atomicOr(&a.atom[u], kLock);
    au = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(au >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)au);
    // ACQUIRE at device scope: everything the vertices merged into u published before their compare-and-swap (their
    // sibling links, list pointers and lengths, the lists themselves — released by the fence in front of that CAS) must be
    // read from memory, not from a copy this XCD's L2 happens to hold from before: the 8 L2s are not coherent with each
    // other, and a relaxed load may hit such a line.  (Found the hard way: a stale list pointer or list entry is an
    // arbitrary integer, and following it faults.)
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (au & kLock) continue;       // (already locked: merged away, or in another wave's hands — never the case for a vertex
                                    //  handed out once; cheap insurance, and the lock is not ours to release)
    const unsigned du = atom_deg(au);
    // 2. how long are the lists to aggregate?  (own row + the lists of the vertices merged into u so far)
    const unsigned own_len = a.agg_len[u];
    const unsigned covered = own_len ? a.agg_child[u] : kNone;
    unsigned long long total = own_len ? (unsigned long long)own_len : (unsigned long long)(a.rowptr[u + 1] - a.rowptr[u]);
    int chain = 0;
    for (unsigned c = atom_child(au); c != kNone && c != covered; c = ld_u32(a.sibling + c)) {
      if (c >= (unsigned)a.n) { if (lane == 0) atomicAdd(a.dbg + 1, 1u); total = ~0ull >> 1; break; }
      total += ld_u32(a.agg_len + c);
      if (++chain > a.n) { if (lane == 0) atomicAdd(a.dbg + 1, 1u); total = ~0ull >> 1; break; }   // (a child list longer than n: broken)
    }
    GCN_BEAT(a, lane, 2, total);
    if (du == 0 || total > (unsigned long long)kBigCap / 2) {
      // isolated, or too long even for the table in memory (hubs of hubs): stays top-level
      if (lane == 0) {
        atomicAnd(&a.atom[u], ~kLock);
        if (du != 0) atomicAdd(a.skipped, 1u);
      }
    } else if (total <= (unsigned long long)kLdsCap / 2) {
      Table<true> t{lkey, lval, nullptr, &ntouch, a.dbg, ltouch};
      rabbit_vertex<true>(a, t, u, du, atom_child(au), &ntouch, lane);
    } else {
      Table<false> t{a.bkey + (size_t)wave * kBigCap, a.bval + (size_t)wave * kBigCap,
                     a.btouch + (size_t)wave * (kBigCap / 2), &ntouch, a.dbg, nullptr};
      rabbit_vertex<false>(a, t, u, du, atom_child(au), &ntouch, lane);
    }
    GCN_BEAT(a, lane, 9, 0);
  }
  GCN_BEAT(a, lane, 10, 0);
}

__global__ void rabbit_init_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                   unsigned long long* __restrict__ atom, unsigned* __restrict__ dest,
                                   unsigned* __restrict__ sibling, unsigned* __restrict__ agg_len,
                                   unsigned* __restrict__ agg_child, unsigned long long* __restrict__ agg_ptr,
                                   unsigned long long* __restrict__ two_m) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  unsigned d = 0;
  for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) d += col[e] != v;     // self-loops carry no modularity
  atom[v] = ((unsigned long long)d << 32) | kNone;
  dest[v] = (unsigned)v;
  sibling[v] = kNone;
  agg_len[v] = 0;
  agg_child[v] = kNone;
  agg_ptr[v] = 0;
  if (d) atomicAdd(two_m, (unsigned long long)d);
}

__global__ void rabbit_order_list_kernel(const int* __restrict__ rank, int n, unsigned* __restrict__ list) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) list[rank[v]] = (unsigned)v;
}

// key of a retry-list vertex: its community's current degree (the reference sorts every round's vertices by it,
// renumber.cu:408-409): the few large communities come last, when the small ones have found their way into them
__global__ void rabbit_retry_keys_kernel(const unsigned* __restrict__ list, unsigned count, const unsigned long long* __restrict__ atom,
                                         unsigned* __restrict__ keys) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) keys[i] = atom_deg(atom[list[i]]);
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
// stats_host[8] or null: {communities, passes, retried vertices, vertices left top-level for lack of table / pool room,
// guard trips: pointer chain, child chain, full table, bad index}.  The four guards bound loops that end by themselves
// in a sound run; a trip means an aggregation was cut short on broken state (a cyclic child list, a stale index): the
// ordering is then NOT returned — hipErrorAssert, one line on stderr, the counts in stats_host.
hipError_t device_order_rabbit(const int* rowptr, const int* col, int n, int nnz, int* rank_out_dev, int* community_out_dev,
                               long long* stats_host, hipStream_t st) {
  if (stats_host) for (int i = 0; i < 8; ++i) stats_host[i] = 0;
  if (n <= 0) return hipSuccess;
#define GCN_R(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)
  int cu = 256;
  { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cu = prop.multiProcessorCount; }
  const int nblocks = cu * kBlocksPerCu;                 // one wave each: 512 vertices in flight on MI355X
  const int nwaves = nblocks;
  Dev<unsigned long long> atom, agg_ptr, scal;           // scal: {two_m, pool_head}
  Dev<unsigned> dest, sibling, agg_len, agg_child, listA, listB, bkey, bval, btouch, cnt, keys, keys2;   // cnt: {counter, nretry, skipped}
  Dev<unsigned long long> pool;
  Dev<int> degrank;
  const unsigned long long pool_cap = 4ull * (unsigned long long)(nnz > 0 ? nnz : 1) + (unsigned long long)n;
  GCN_R(atom.alloc(n)); GCN_R(agg_ptr.alloc(n)); GCN_R(scal.alloc(2));
  GCN_R(dest.alloc(n)); GCN_R(sibling.alloc(n)); GCN_R(agg_len.alloc(n)); GCN_R(agg_child.alloc(n)); GCN_R(listA.alloc(n));
  GCN_R(listB.alloc(n)); GCN_R(keys.alloc(n)); GCN_R(keys2.alloc(n));
  GCN_R(bkey.alloc((size_t)nwaves * kBigCap)); GCN_R(bval.alloc((size_t)nwaves * kBigCap));
  GCN_R(btouch.alloc((size_t)nwaves * (kBigCap / 2))); GCN_R(cnt.alloc(kCntWords));
  GCN_R(pool.alloc(pool_cap)); GCN_R(degrank.alloc(n));
  GCN_R(hipMemsetAsync(scal.p, 0, sizeof(unsigned long long) * 2, st));
  GCN_R(hipMemsetAsync(cnt.p, 0, sizeof(unsigned) * kCntWords, st));     // (once, long before the first pass: see below)
  GCN_R(hipMemsetAsync(bval.p, 0, sizeof(unsigned) * (size_t)nwaves * kBigCap, st));
  rabbit_fill_kernel<<<1024, 256, 0, st>>>(bkey.p, (size_t)nwaves * kBigCap, kNone);
  rabbit_init_kernel<<<(n + 255) / 256, 256, 0, st>>>(rowptr, col, n, atom.p, dest.p, sibling.p, agg_len.p, agg_child.p, agg_ptr.p, scal.p);
  GCN_R(hipGetLastError());
  // processing order: ascending degree, ties by vertex id (the strict total order of order_deg, device version)
  GCN_R(device_order_deg(rowptr, col, n, nnz, /*which = out*/ 1, /*desc*/ 0, degrank.p, st));
  rabbit_order_list_kernel<<<(n + 255) / 256, 256, 0, st>>>(degrank.p, n, listA.p);
  GCN_R(hipGetLastError());
  unsigned long long two_m = 0;
  GCN_R(hipMemcpyAsync(&two_m, scal.p, sizeof(two_m), hipMemcpyDeviceToHost, st));
  GCN_R(hipStreamSynchronize(st));
  long long passes = 0, retried = 0;
  unsigned guards[4] = {0, 0, 0, 0};                     // the counters cnt[4..7] as last read back (they only grow)
  struct SideStream { hipStream_t s = nullptr; ~SideStream() { if (s) (void)hipStreamDestroy(s); } } side_owner;
  const bool verbose = [] { const char* e = getenv("GCN_AMD_VERBOSE"); return e && e[0] && e[0] != '0'; }();
  if (verbose) { std::fprintf(stderr, "rabbit_device: n=%d nnz=%d 2m=%llu waves=%d\n", n, nnz, two_m, nwaves); std::fflush(stderr); }
  if (two_m > 0) {
    RabbitArgs a;
    a.rowptr = rowptr; a.col = col; a.n = n; a.atom = atom.p; a.dest = dest.p; a.sibling = sibling.p;
    a.agg_ptr = agg_ptr.p; a.agg_len = agg_len.p; a.agg_child = agg_child.p; a.pool = pool.p; a.pool_head = scal.p + 1; a.pool_cap = pool_cap;
    a.skipped = cnt.p + 2; a.bkey = bkey.p; a.bval = bval.p; a.btouch = btouch.p; a.dbg = cnt.p + 4;
    a.two_m_inv = 1.0 / (double)two_m;
    unsigned count = (unsigned)n;
    unsigned* cur = listA.p;
    unsigned* nxt = listB.p;
    // a vertex whose target was locked at the moment of the merge comes back in the next pass; a handful of passes
    // empties the list (what is left after 16 stays top-level)
    auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    const int max_passes = 16;
    static_assert(16 <= kMaxPasses, "every pass owns a counter pair");
    const int retry_waves = nblocks;
    // vertices in flight see each other's merges late: keep them a small share of the list (1 in 64 or fewer)
    auto waves_for = [&](unsigned cnt_) { long long w = (long long)cnt_ / 64; if (w < 8) w = 8; if (w > nblocks) w = nblocks; return (int)w; };
    Dev<unsigned> beat;
    hipStream_t side = nullptr;
    a.beat = nullptr;
    if (env_int("GCN_AMD_RABBIT_BEAT", 0)) {             // development: watch the waves of a pass from a side stream
      GCN_R(beat.alloc((size_t)nwaves * 4));
      GCN_R(hipMemsetAsync(beat.p, 0, sizeof(unsigned) * (size_t)nwaves * 4, st));
      GCN_R(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
      side_owner.s = side;                               // (destroyed on every way out)
      a.beat = beat.p;
    }
    void* sort_tmp = nullptr;
    size_t sort_bytes = 0;
    GCN_R(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys.p, keys2.p, cur, nxt, n, 0, 32, st));
    Dev<char> sort_buf;
    GCN_R(sort_buf.alloc(sort_bytes));
    sort_tmp = sort_buf.p;
    for (; count > 0 && passes < max_passes; ++passes) {
      if (passes > 0) {                                  // a round of the vertices that came back, smallest community first
        rabbit_retry_keys_kernel<<<(count + 255) / 256, 256, 0, st>>>(cur, count, atom.p, keys.p);
        GCN_R(hipGetLastError());
        GCN_R(hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, keys.p, keys2.p, cur, nxt, (int)count, 0, 32, st));
        unsigned* t2 = cur; cur = nxt; nxt = t2;
      }
      a.list = cur; a.count = count; a.retry = nxt;
      // every pass has its own work counter and retry count, zeroed once before the first pass: a hipMemsetAsync
      // between two passes was seen to land AFTER the next kernel had started (the counter jumped back and vertices
      // were handed out twice — two waves on one vertex, a child pushed twice, a cyclic child list)
      a.counter = cnt.p + 8 + 2 * passes; a.nretry = a.counter + 1;
      const int launch = passes == 0 ? waves_for(count) : (retry_waves < waves_for(count) ? retry_waves : waves_for(count));
      if (verbose) { std::fprintf(stderr, "rabbit_device: pass %lld: launching %d waves for %u vertices\n", passes + 1, launch, count); std::fflush(stderr); }
      rabbit_pass_kernel<<<launch, 64, 0, st>>>(a);
      GCN_R(hipGetLastError());
      if (side && env_int("GCN_AMD_RABBIT_BEAT", 0) == 1) {
        std::vector<unsigned> hb((size_t)nwaves * 4);
        for (int tick = 0; tick < 5 && hipStreamQuery(st) == hipErrorNotReady; ++tick) {
          struct timespec ts = {1, 0};
          nanosleep(&ts, nullptr);
          if (hipMemcpyAsync(hb.data(), beat.p, sizeof(unsigned) * hb.size(), hipMemcpyDeviceToHost, side) != hipSuccess ||
              hipStreamSynchronize(side) != hipSuccess) break;
          std::fprintf(stderr, "rabbit_device: pass %lld tick %d:", passes + 1, tick);
          for (int w = 0; w < launch; ++w) if (hb[4 * w + 1] != 10u) std::fprintf(stderr, " [w%d u=%u stage=%u x=%u]", w, hb[4 * w], hb[4 * w + 1], hb[4 * w + 2]);
          std::fprintf(stderr, "\n");
          std::fflush(stderr);
        }
      }
      unsigned h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hp[2] = {0, 0};
      GCN_R(hipMemcpyAsync(h, cnt.p, sizeof(h), hipMemcpyDeviceToHost, st));
      GCN_R(hipMemcpyAsync(hp, a.counter, sizeof(hp), hipMemcpyDeviceToHost, st));
      GCN_R(hipStreamSynchronize(st));
      h[1] = hp[1];
      if (verbose) {
        std::fprintf(stderr, "rabbit_device: pass %lld: %u vertices, %u to retry, %u left top-level so far; guard trips: pointer chain %u, "
                             "child chain %u, table %u, index %u\n", passes + 1, count, h[1], h[2], h[4], h[5], h[6], h[7]);
        std::fflush(stderr);
      }
      for (int g = 0; g < 4; ++g) guards[g] = h[4 + g];
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
  if (stats_host) {
    stats_host[1] = passes; stats_host[2] = retried; stats_host[3] = (long long)h_skipped;
    for (int g = 0; g < 4; ++g) stats_host[4 + g] = (long long)guards[g];
  }
  if (guards[0] | guards[1] | guards[2] | guards[3]) {
    std::fprintf(stderr, "libgcnspmm: rabbit_device: guard trips (pointer chain %u, child chain %u, full table %u, bad index %u): "
                         "an aggregation ran on broken state; no ordering returned\n", guards[0], guards[1], guards[2], guards[3]);
    return hipErrorAssert;
  }
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
      for (unsigned c = (unsigned)h_atom[v]; c != kNone; c = h_sib[c]) {
        if (c >= (unsigned)n || stack.size() > (size_t)n) return hipErrorUnknown;     // (a broken child list: refuse)
        stack.push_back(c);
      }
    }
  }
  if (next != n) return hipErrorUnknown;                 // (a broken dendrogram: never observed; refuse rather than return a non-permutation)
  GCN_R(hipMemcpyAsync(rank_out_dev, rank.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
  if (community_out_dev) GCN_R(hipMemcpyAsync(community_out_dev, comm.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
  GCN_R(hipStreamSynchronize(st));
  if (stats_host) stats_host[0] = ncomm;
#undef GCN_R
  return hipSuccess;
}

}  // namespace gcn
