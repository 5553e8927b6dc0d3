// plan_policy.cpp — the dispatch policy of libgcnspmm's SpMM plan: chunk size, column tile, slice count, row stride of
// the re-laid feature table, and which kernel family / slice set a k-wide call takes.  Every threshold here is a measured
// one (the experiment is cited beside it); the alternate paths those experiments decided against are gone — what remains
// switchable is what a test needs (plan_policy.h).  Plan construction is plan_build.cpp, the launches api_spmm.cpp.
#include "plan_policy.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>

namespace gcn {

std::mutex g_plan_mu;

int cu_count_cached() {
  static int cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (dev < 0 || dev >= 64) return -1;
  if (cached[dev] > 0) return cached[dev];
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  cached[dev] = prop.multiProcessorCount;
  return cached[dev];
}

// chunk size: the largest power of two <= nnz / resident waves (8 blocks x 4 waves per CU), within
// [64, 2048].  Measured (profiles/r01_sweep_chunk_size.txt): every chunk boundary costs a partial
// row (slab write + fix-up read) and a row-pointer restart, and that outweighs the load imbalance of
// having only one or two chunks per wave — Reddit-shaped 1 GPU: T = 512 / 1024 / 2048 / 4096 ->
// 4.12 / 4.03 / 3.94 / 4.04 ms; rank of an 8-way partition: T = 64 / 512 / 2048 / 4096 ->
// 0.68 / 0.56 / 0.556 / 0.67 ms.
int auto_chunk_nnz(long long nnz, int cu) {
  if (cu <= 0) cu = 256;
  const long long waves = (long long)cu * 32;
  const long long per_wave = nnz / waves;
  long long t = 64;
  while (t * 2 <= per_wave && t < 2048) t *= 2;
  return (int)t;
}

// Feature-column tile per pass.  Measured on MI355X (profiles/r01_sweep_tiles_*.txt): when one
// 64-column slice of B (n x 256 B) sits well inside the 256 MiB Infinity Cache, k/64 narrow
// passes beat one wide pass by 3-6 % (Reddit-shaped, n = 233 k); when it does not (products-
// shaped, n = 2.4 M) the widest tile wins by 6-7 %.
int auto_tile_cols(long long n, int k) {
  if (k <= 64) return 0;
  const long long budget = 128LL << 20;          // half of the Infinity Cache
  if (n * 256 <= budget) return 64;
  if (n * 512 <= budget && k > 128) return 128;
  return 0;                                      // widest tile k allows (<= 256 columns)
}

namespace {
bool env_on(const char* name) { const char* e = std::getenv(name); return !e || e[0] != '0'; }
}  // namespace

bool group8_enabled() { static const bool v = env_on("GCN_AMD_GROUP8"); return v; }
bool group_fused_fixup() { static const bool v = env_on("GCN_AMD_GROUP_FUSED_FIXUP"); return v; }

// Entries per chunk of one 16-lane group.  A block walks 16 chunks and 4 blocks are resident per CU (114 VGPRs), so the
// chip holds cu*4 blocks per "round".  Large matrices run many rounds and 512 is the measured optimum
// (profiles/r02zg_chunk_length_slices.log); a matrix of a few rounds — a rank's row block of an 8-way partition: 1.7
// rounds at 512 — leaves the last round partly empty, so the length is picked from the multiples of 64 in [256, 1024]
// that fill whole rounds best (ties: the one closest to 512).
int group_chunk(long long entries, int cu) {
  if (cu <= 0) cu = 256;
  const double round = (double)cu * 4.0;
  if ((double)entries / (16.0 * 512.0) >= 6.0 * round) return 512;
  double fills[13], top = 0.0;                         // t = 256 + 64*i
  for (int i = 0; i < 13; ++i) {
    const double blocks = (double)entries / (16.0 * (256 + 64 * i));
    const double rounds = std::ceil(blocks / round);
    fills[i] = rounds > 0 ? blocks / (rounds * round) : 0.0;
    if (fills[i] > top) top = fills[i];
  }
  int best = 512;
  bool have = false;
  for (int i = 0; i < 13; ++i) {                       // among the lengths within 2 % of the best fill: the one closest to 512
    const int t = 256 + 64 * i;
    if (fills[i] >= top - 0.02 && (!have || std::abs(t - 512) < std::abs(best - 512))) { best = t; have = true; }
  }
  return best;
}

namespace {
// Expected 128-byte cache lines one gathered feature row costs, summed over its 64-column tiles, when B's
// rows are `ld` floats apart (the row start offsets cycle through the multiples of gcd(4*ld, 128)).
double lines_per_row(int k, int ld) {
  const long long row_bytes = 4LL * ld;
  long long g = row_bytes % 128;
  for (long long a = 128; g != 0;) { const long long t = a % g; a = g; g = t; if (g == 0) { g = a; break; } }
  if (g == 0) g = 128;                                // row_bytes % 128 == 0: every row starts on a line
  const int period = (int)(128 / g);
  double total = 0;
  for (int r = 0; r < period; ++r) {
    const long long off = (r * row_bytes) % 128;
    for (long long t0 = 0; t0 < 4LL * k; t0 += 256) {
      const long long w = (4LL * k - t0) < 256 ? (4LL * k - t0) : 256;
      const long long start = (off + t0) % 128;
      total += (double)((start + w - 1) / 128 + 1);
    }
  }
  return total / period;
}
}  // namespace

// Row stride (floats) B is gathered with: k itself, or k rounded up to whole 128-byte lines when that
// saves >= 15 % of the cache lines per gathered row and the re-laid table stays <= 768 MiB.  Measured
// (profiles/r01f_sweep_padded_feature_rows.log, whole SpMM, unpadded -> padded): Reddit-shaped k = 20:
// 2.19 -> 1.60 ms, 24: 2.26 -> 1.60, 47: 2.11 -> 2.00, 100: 4.43 -> 3.84, 172: 7.56 -> 5.73; no saving
// by the model and none measured for k = 40, 48 (rows of 160 / 192 B never straddle more lines than
// padded ones); products-shaped k = 47 (627 MB padded): 5.41 -> 4.86 ms, k = 100 (1.25 GB): 8.95 ->
// 9.73 ms — past the Infinity Cache the larger table and the copy cost more than the lines save.
int padded_ldb(long long n, int k) {
  if (k <= 16 || k % 32 == 0) return k;
  const int ld = (k + 31) / 32 * 32;
  if ((long long)sizeof(float) * n * ld > (768LL << 20)) return k;
  return lines_per_row(k, k) >= 1.15 * lines_per_row(k, ld) ? ld : k;
}

// Number of column slices for the XCD-aware slicing (slicing.hip), 0 = do not slice.
// Measured on MI355X with the r01f kernels (profiles/r01f_sweep_slices_scales.log; Reddit-shaped graphs
// of 14.5 k .. 1.86 M vertices, mean degree 493; whole SpMM, k = 128, best S in brackets):
//   n = 14.5 k (64-column table 3.7 MB): slicing buys nothing;  29 k (7.5 MB): [2] 0.352 vs 0.394 ms
//   unsliced;  58 k: [4] 0.84 vs 1.18;  116 k: [4/8] 1.76-1.80 vs 3.13;  233 k: [8] 3.62 vs 7.3;
//   466 k: [8] 9.20 vs 15.7 (16: 9.84);  932 k: [8] 24.4 vs 32.3;  1.86 M: [8] 56.5 vs 62.5.
// So, for matrices with a value stream on the four-per-gather kernel: as many slices as bring one slice of
// the table (n/S x 256 B) down to the 4 MiB of an XCD's L2, but never more than the 8 XCDs — beyond 8 every
// XCD walks several slices and the extra partial rows (S*m*k floats written and re-read) cost more than the
// higher hit rate returns.
// `value_free` (the group kernels of spmm_group.hip run, with or without the value stream): a partial row costs
// one non-temporal 256-byte store and no cross-lane work, so the count follows the table alone — one slice per
// 4 MiB of it (n = 233 k: 15), XCDs walking two slices each one after the other.  Measured
// (profiles/r02z5_nt_stores_slices.log, whole SpMM k = 128): S = 8 / 14 / 16 / 18 / 20 / 24 / 32:
// 3.19 / 3.08 / 3.09 / 3.13 / 3.19 / 3.36 / 3.68 ms — flat from 14 to 16, then the slab of partial rows
// (S*m*k floats, written and re-read by the reduction) takes over.
// Both need >= 16 non-zeros per virtual row; at mean degree 51 (products-shaped) slicing loses and stays off.
int auto_slices(long long m, long long n, long long nnz, bool value_free) {
  if (m <= 0 || nnz <= 0) return 0;
  if (nnz / m < 128) return 0;                        // low degree: partial rows outweigh the hits
  const long long table = n * 256;                    // bytes of one 64-column tile of B
  const long long l2 = 4LL << 20;
  if (table <= l2) return 0;                          // fits every L2 as it is
  if (value_free) {
    long long S = (table + l2 - 1) / l2;
    const long long narrow = (n + 32766) / 32767;     // the group kernel's 15-bit entries: slices <= 32 767 columns
    if (S < narrow) S = narrow;
    if (S > 8) {                                      // (up to 8 the rule below gives the same or better)
      if (S > nnz / m / 16) S = nnz / m / 16;         // keep >= 16 non-zeros per virtual row
      if (S > 8 && S <= 1024 && S >= narrow && table / S <= 2 * l2) return (int)S;
    }
  }
  int S = 2;
  while (S < 8 && table / S > l2) S *= 2;
  while (S > 1 && nnz / m / S < 16) S /= 2;           // keep >= 16 non-zeros per virtual row
  if (S < 2) return 0;
  // slices far larger than any cache (huge n): the partial rows cost traffic and buy no hits
  if (table / S > (64LL << 20)) return 0;
  return S;
}

void die(const char* what, hipError_t e) {
  std::fprintf(stderr, "libgcnspmm: %s failed: %s\n", what, hipGetErrorString(e));
  std::abort();
}

bool verbose() {
  const char* v = std::getenv("GCN_AMD_VERBOSE");
  return v && v[0] && v[0] != '0';
}

// The stateless entry points (oneshot / cuspmm / flexspmm) keep their partial slab and chunk rows in a
// scratch plan per (device, stream): two calls that can run concurrently never share buffers.  The plans
// are never freed (a static destructor would call hipFree after the runtime has shut down).
gcn_spmm_plan* scratch_plan(void* stream) {
  static auto* plans = new std::map<std::pair<int, void*>, gcn_spmm_plan*>();
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto& slot = (*plans)[{dev, stream}];
  if (!slot) { slot = new (std::nothrow) gcn_spmm_plan(); if (slot) slot->device = dev; }
  return slot;
}

size_t ws_elems(const gcn_spmm_plan* p, int k) {
  int chunks = std::max(p->nchunks, p->panels.out_nchunks);
  chunks = std::max(chunks, p->col16.nchunks16);
  chunks = std::max(chunks, p->group.nchunks);
  chunks = std::max(chunks, p->group_alt[0].nchunks);   // (whichever slice set a call picks: room for the larger)
  return 2 * (size_t)(chunks > 0 ? chunks : 1) * (size_t)k;
}

// Is a k-wide SpMM of this plan launched on the sliced copy?  The four-per-gather kernel pays from k = 33 (narrower
// rows gather 128 B or less per non-zero: the partial rows cost more than the L2 hits buy, 2.12 vs 2.02 ms at
// k = 32).  The group kernels pay from k = 12: their 64-column pass costs the same whatever k is, and beats the
// unsliced kernels there (Reddit-shaped, whole SpMM, profiles/r02zzg_narrow_widths_sliced.log: k = 12 / 16 / 20 /
// 32: 1.49 / 1.36 / 1.60 / 1.58 -> 1.23 / 1.02 / 1.23 / 1.10 ms; k = 8 a tie, k = 4 loses) — provided the width
// reaches them: a multiple of 4, or wide enough for the k' = ceil(k/4)*4 detour.
bool sliced_for(const gcn_spmm_plan* p, int k) {
  if (p->slicing.S <= 0 || p->nnz <= 0) return false;
  if (k >= kSliceMinK) return true;
  if (!p->group.ready() || p->panels.R != 0 || k < kGroupMinK) return false;
  if (k % 4 == 0) return true;
  const int kp = (k + 3) / 4 * 4, ldb = (kp + 31) / 32 * 32;         // (the conditions of odd_width_detour)
  return k > 16 && p->gather_width != 1 && (long long)sizeof(float) * p->n * ldb <= (768LL << 20);
}

// rows of the slice-by-slice copy of B the group kernels gather from on the plan's OWN slice set (decides their addressing mode)
long long group_table_rows(const gcn_spmm_plan* p) { return (long long)p->slicing.S * ((long long)p->group.w + 1); }

SliceSet own_slice_set(const gcn_spmm_plan* p) {
  SliceSet s;
  s.g = &p->group; s.S = p->slicing.S; s.alt = -1;
  return s;
}

// would the sliced launch of a k-wide SpMM run a value-free kernel (and is the scaled copy of B worth it)?
bool valless_pays(const gcn_spmm_plan* p, int k, int ldb) {
  // (the scaled copy of B costs 2*n*k*4 bytes of traffic whatever the matrix; the value stream it saves is
  //  4 bytes per non-zero plus instructions.  With the group kernel the rank-0 share of an 8-way partition of
  //  the Reddit-shaped graph, 61 non-zeros per column of the block, still gains: 0.460 against 0.511 ms,
  //  profiles/r02z7_rank_share_value_free.log; below 48 per column nothing has been measured, so it stays off)
  if (!sliced_for(p, k) || !p->factors.ready() || p->panels.R != 0 || p->nnz / p->n < kVallessMinPerCol) return false;
  if (p->group.vals) return false;                     // (the plan was built for the weighted pass: value-free did not pay)
  if (p->group.ready() && spmm_group_eligible(k, ldb, group_table_rows(p), nullptr, nullptr, nullptr)) return true;   // spmm_group.hip
  SpmmArgs t{};                                        // the launch as the sliced branch will issue it
  t.k = k; t.nnz = p->nnz; t.n = p->n; t.nchunks_grid = p->nchunks; t.T = p->T;
  t.m = p->slicing.S * p->m; t.ldb = ldb; t.tile_cols = p->tile_cols ? p->tile_cols : 64;
  t.gather_width = p->gather_width;
  return spmm_will_use_quad(t) && spmm_quad_lanes(k) == 16;
}

// will a sliced plan of this matrix run the group kernel value-free (known before the slicing exists)
bool value_free_plan(const gcn_spmm_plan* p) {
  return p->factors.ready() && p->panels.R == 0 && p->nnz / p->n >= kVallessMinPerCol;
}
// ... or the group kernel at all (value-free or weighted): it decides the automatic slice count
bool group_plan(const gcn_spmm_plan* p) { return p->panels.R == 0; }
// the sliced launch of a k-wide SpMM runs the WEIGHTED group kernel (values beside the stream)
bool weighted_pass(const gcn_spmm_plan* p, int k, int ldb) {
  return sliced_for(p, k) && p->panels.R == 0 && p->group.ready() && p->group.vals &&
         spmm_group_eligible(k, ldb, group_table_rows(p), nullptr, nullptr, nullptr);
}
// a launch decided as (valless, weighted) runs one of the group kernels: B is gathered from the slice-by-slice copy
bool group_launch(const gcn_spmm_plan* p, bool valless, bool weighted) {
  return weighted || (valless && p->group.ready() && !p->group.vals);
}

// Widths that are not a multiple of 4 take a detour over k' = k rounded up to 4 (gcn_spmm_csr_f32_epilogue); it
// exists to reach the 16-byte-per-lane kernels, so it follows their rule: the four-per-gather kernel only pays
// from ~48 non-zeros per (virtual) row up, the group kernel of the value-free pass does not mind short rows
bool odd_width_detour(const gcn_spmm_plan* p, int k) {
  if (!(k > 16 && k % 4 != 0 && p->nnz > 0 && p->panels.R == 0 && p->gather_width != 1)) return false;
  const int kp = (k + 3) / 4 * 4, ldb = (kp + 31) / 32 * 32;
  if ((long long)sizeof(float) * p->n * ldb > (768LL << 20)) return false;
  if (p->gather_width == 4) return true;
  const bool sliced = sliced_for(p, k);
  if (sliced && p->group.ready() && (p->group.vals || value_free_plan(p))) return true;
  const long long rows = sliced ? (long long)p->slicing.S * p->m : (long long)p->m;
  return rows > 0 && p->nnz / rows >= 48;
}

int alt_class(int k) { return k <= 32 ? 0 : -1; }

// Widths 33..48 on the five-engine kernel stay on the plan's own slices; where the row stride would have been padded to
// 64 floats (k = 44 and the odd widths' k' detour) they gather from rows of 48 instead (192 bytes: always two lines, a
// quarter less table and copy): *ldb = 48, *relay = the call lays that copy out itself (k = 41 / 47: 1.39 / 1.36 ->
// 1.34 / 1.30 ms; 36 / 40 keep their dense rows).  (A slice set of their own — 11..13 slices instead of 15 — was built
// and measured: +-1 %, profiles/r03ba_*; not kept.)  k <= 32: the narrow set (plan_build.cpp, maybe_build_alt).
SliceSet pick_slice_set(gcn_spmm_plan* p, int k, int* ldb, bool* relay, bool build, const int32_t* rowptr, const int32_t* col,
                        const float* val, hipStream_t st) {
  *relay = false;
  const SliceSet own = own_slice_set(p);
  if (p->nnz <= 0 || k % 4 != 0) return own;
  if (k > 32 && k <= 48 && *ldb > 48 && valless_pays(p, k, 48) && group_launch(p, true, false)) {
    GroupArgs probe{};
    probe.k = k; probe.ldb = 48; probe.table_rows = group_table_rows(p);
    if (spmm_group12_applies(probe)) { *ldb = 48; *relay = true; }
    return own;
  }
  const int cls = alt_class(k);
  if (cls < 0 || !valless_pays(p, k, *ldb) || !group_launch(p, true, false)) return own;
  if (build) maybe_build_alt(p, cls, rowptr, col, val, st);
  if (!p->group_alt[cls].ready()) return own;
  SliceSet s;
  s.g = &p->group_alt[cls]; s.S = p->alt_S[cls]; s.alt = cls;
  return s;
}

}  // namespace gcn

using namespace gcn;

extern "C" {

int32_t gcn_spmm_plan_num_passes(const gcn_spmm_plan_t* p, int32_t k) {
  if (!p || k <= 0) return -1;
  if (p->panels.R > 0 && k > 32) return (k + 63) / 64;
  {                                                    // the group kernels take every tile in one launch
    int kk = k, ldb = padded_ldb(p->n, k);
    if (odd_width_detour(p, k)) { kk = (k + 3) / 4 * 4; ldb = (kk + 31) / 32 * 32; }
    const bool vl = valless_pays(p, kk, ldb);
    if (group_launch(p, vl, !vl && weighted_pass(p, kk, ldb))) return 1;
  }
  const int tile = p->tile_cols ? p->tile_cols : (p->slicing.S > 0 && k > 32 ? 64 : auto_tile_cols(p->n, k));
  const int vec = pick_vec(k, tile, nullptr, nullptr, nullptr);   // 16-B aligned operands
  return (k + 64 * vec - 1) / (64 * vec);
}

int gcn_spmm_plan_main_kernel(const gcn_spmm_plan_t* p, int32_t k, int32_t epilogue, char* buf, int32_t buflen) {
  if (!p || k <= 0 || !buf || buflen <= 0) return GCN_ERR_INVALID_ARG;
  if (p->panels.R > 0 && k > 32) { snprintf(buf, (size_t)buflen, "gcn::spmm_panel_in_kernel"); return GCN_OK; }
  SpmmArgs a{};
  const bool sliced = sliced_for(p, k);
  a.k = k; a.n = p->n; a.m = sliced ? p->slicing.S * p->m : p->m; a.nnz = p->nnz;
  a.nchunks_grid = p->nchunks;
  a.relu = epilogue && !sliced ? 1 : 0;               // sliced: the epilogue runs in the slice reduction
  a.tile_cols = p->tile_cols ? p->tile_cols : (sliced ? 64 : auto_tile_cols(p->n, k));
  a.gather_width = p->gather_width;
  if (odd_width_detour(p, k)) {
    a.k = (k + 3) / 4 * 4;                             // odd widths run at k rounded up to 4 (see gcn_spmm_csr_f32_epilogue)
    a.ldb = (a.k + 31) / 32 * 32;
    a.relu = 0;
    a.valless = valless_pays(p, a.k, a.ldb);
  } else {
    if (const int ldb = padded_ldb(p->n, k); ldb != k) a.ldb = ldb;
    a.valless = valless_pays(p, k, a.ldb > 0 ? a.ldb : k);   // as spmm_impl decides
  }
  a.col16 = a.valless && p->col16.ready();
  int ld_eff = a.ldb > 0 ? a.ldb : a.k;
  if (a.valless && p->group.ready()) {
    // the slice set the call would run on (the narrow one only if it exists already: nothing is built here), and
    // everything reported — stride, table size, addressing mode, chunk count — from THAT set
    bool relay = false;
    const SliceSet ss = pick_slice_set(const_cast<gcn_spmm_plan*>(p), a.k, &ld_eff, &relay, /*build=*/false, nullptr, nullptr, nullptr, nullptr);
    const bool big = spmm_group_needs_big(ss.table_rows(), ld_eff);
    const char* bigs = big ? "true" : "false";
    GroupArgs probe{};
    probe.k = a.k; probe.ldb = ld_eff; probe.table_rows = ss.table_rows();
    if (group8_enabled() && a.k <= 32 && ss.g->nchunks % 64 == 0)
      snprintf(buf, (size_t)buflen, "gcn::spmm_group8_kernel<true, %s>", bigs);
    else if (spmm_group12_applies(probe))
      snprintf(buf, (size_t)buflen, "gcn::spmm_group12_kernel");
    else
      snprintf(buf, (size_t)buflen, "gcn::spmm_group_ring_kernel<2, %s>", bigs);
    return GCN_OK;
  }
  if (!a.valless && weighted_pass(p, a.k, ld_eff)) {
    const char* bigs = spmm_group_needs_big(group_table_rows(p), ld_eff) ? "true" : "false";
    if (group8_enabled() && a.k <= 32 && p->group.nchunks % 64 == 0)
      snprintf(buf, (size_t)buflen, "gcn::spmm_group8_weighted_kernel<%s>", bigs);
    else
      snprintf(buf, (size_t)buflen, "gcn::spmm_group_weighted_kernel<2, %s>", bigs);
    return GCN_OK;
  }
  describe_main_kernel(a, buf, (size_t)buflen);
  return GCN_OK;
}

int gcn_spmm_plan_prelaid_layout(const gcn_spmm_plan_t* p, int32_t k, int32_t* slices, int32_t* slice_cols,
                                 int64_t* table_rows, int32_t* ld) {
  if (!p || k <= 0 || k % 4 != 0) return GCN_ERR_INVALID_ARG;
  const int ldb = padded_ldb(p->n, k);
  // only the value-free group pass gathers from a scaled, slice-by-slice copy of B; the layout is that of the plan's
  // OWN slice set whatever the width (gcn_spmm_csr_f32_prelaid runs on it)
  if (!valless_pays(p, k, ldb) || !group_launch(p, true, false)) return GCN_ERR_INVALID_ARG;
  if (slices) *slices = p->slicing.S;
  if (slice_cols) *slice_cols = p->group.w;
  if (table_rows) *table_rows = group_table_rows(p);
  if (ld) *ld = ldb;
  return GCN_OK;
}

int32_t gcn_spmm_auto_slices(int64_t m, int64_t n, int64_t nnz, int32_t value_free) {
  return auto_slices(m, n, nnz, value_free != 0);
}

int32_t gcn_spmm_group_addressing(int64_t table_rows, int32_t ld_floats) {
  if (table_rows <= 0 || ld_floats <= 0) return -1;
  if (!spmm_group_eligible(ld_floats, ld_floats, table_rows, nullptr, nullptr, nullptr)) return -1;
  return spmm_group_needs_big(table_rows, ld_floats) ? 1 : 0;
}

}  // extern "C"
