// api_spmm.cpp — the native C ABI of libgcnspmm.so: the SpMM plan and the SpMM itself (see
// include/gcn_spmm.h for the contract and the reference interfaces each entry point replaces).  The
// reorderers' entry points are in api_reorder.cpp, the reference's own symbols (flexspmm, csr2tile, ...)
// in api_dropin.cpp.
#include "plan.h"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <new>

#define GCN_VERSION_STR "0.2.0"

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
int env_int(const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; }

// development knobs, read once per process (DESIGN.md lists them)
bool valless_enabled() { static const bool v = env_on("GCN_AMD_VALLESS"); return v; }   // value-free sliced pass
int valless_min_per_col() { static const int v = env_int("GCN_AMD_VALLESS_MIN_PER_COL", 48); return v; }   // non-zeros per column from which the scaled copy pays
bool col16_enabled() { static const bool v = env_on("GCN_AMD_COL16"); return v; }       // its 16-bit column stream (quad kernel)
bool group_enabled() { static const bool v = env_on("GCN_AMD_GROUP"); return v; }       // the group kernel (spmm_group.hip)
int group_store() {                                                                     // partial-row stores: 0 plain, 1 sc1 (write-through), 2 nt
  static const int v = [] { const int m = env_int("GCN_AMD_GROUP_STORE", 2); return (m >= 0 && m <= 2) ? m : 2; }();
  return v;
}
bool group_ring() { static const bool v = env_on("GCN_AMD_GROUP_RING"); return v; }   // finished rows through the LDS ring (value-free pass)
bool group_merge_tiles() { static const bool v = env_on("GCN_AMD_GROUP_MERGE_TILES"); return v; }   // all column tiles in one launch
bool group_fused_fixup() { static const bool v = env_on("GCN_AMD_GROUP_FUSED_FIXUP"); return v; }   // cut rows' pieces added by the slice reduction (no fix-up pass)
bool group8_enabled() { static const bool v = env_on("GCN_AMD_GROUP8"); return v; }   // k <= 32: eight 8-lane row engines per wave
bool group_weighted_enabled() { static const bool v = env_on("GCN_AMD_GROUP_WEIGHTED"); return v; }   // group kernel for values that do not factor
bool quad_stream_rows() { static const bool v = env_on("GCN_AMD_QUAD_NT"); return v; }   // sliced pass with values: nt partial-row stores
// Entries per chunk of one 16-lane group.  A block walks 16 chunks and 4 blocks are resident per CU (114 VGPRs), so the
// chip holds cu*4 blocks per "round".  Large matrices run many rounds and 512 is the measured optimum
// (profiles/r02zg_chunk_length_slices.log); a matrix of a few rounds — a rank's row block of an 8-way partition: 1.7
// rounds at 512 — leaves the last round partly empty, so the length is picked from the multiples of 64 in [256, 1024]
// that fill whole rounds best (ties: the one closest to 512).  GCN_AMD_GROUP_T pins it (development knob).
int group_chunk(long long entries, int cu) {
  static const int forced = [] { const int t = env_int("GCN_AMD_GROUP_T", 0); return (t >= 64 && t <= 4096 && t % 64 == 0) ? t : 0; }();
  if (forced) return forced;
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
bool panel_mfma_enabled() { static const bool v = env_on("GCN_AMD_PANEL_MFMA"); return v; }   // dense panels on the matrix cores
double panel_mfma_density() {                                                              // ... from this window density up
  static const double v = [] { const char* e = std::getenv("GCN_AMD_PANEL_MFMA_DENSITY"); const double d = e ? std::atof(e) : 0.25;
                               return d > 0.0 && d <= 1.0 ? d : 0.25; }();
  return v;
}
int slice_min_k() { static const int v = env_int("GCN_AMD_SLICE_MIN_K", 33); return v; }  // smallest k the sliced copy is used for (four-per-gather kernel)
int group_min_k() { static const int v = env_int("GCN_AMD_GROUP_MIN_K", 12); return v; }  // ... when the group kernels walk it

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

bool pad_b_enabled() { static const bool v = env_on("GCN_AMD_PAD_B"); return v; }

// Row stride (floats) B is gathered with: k itself, or k rounded up to whole 128-byte lines when that
// saves >= 15 % of the cache lines per gathered row and the re-laid table stays <= 768 MiB.  Measured
// (profiles/r01f_sweep_padded_feature_rows.log, whole SpMM, unpadded -> padded): Reddit-shaped k = 20:
// 2.19 -> 1.60 ms, 24: 2.26 -> 1.60, 47: 2.11 -> 2.00, 100: 4.43 -> 3.84, 172: 7.56 -> 5.73; no saving
// by the model and none measured for k = 40, 48 (rows of 160 / 192 B never straddle more lines than
// padded ones); products-shaped k = 47 (627 MB padded): 5.41 -> 4.86 ms, k = 100 (1.25 GB): 8.95 ->
// 9.73 ms — past the Infinity Cache the larger table and the copy cost more than the lines save.
int padded_ldb(long long n, int k) {
  if (k <= 16 || k % 32 == 0 || !pad_b_enabled()) return k;
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
// So, for matrices with a value stream (the four-per-gather kernel): as many slices as bring one slice of
// the table (n/S x 256 B) down to the 4 MiB of an XCD's L2, but never more than the 8 XCDs — beyond 8 every
// XCD walks several slices and the extra partial rows (S*m*k floats written and re-read) cost more than the
// higher hit rate returns.
// `value_free` (the values factor, the group kernel of spmm_group.hip runs): a partial row costs one
// non-temporal 256-byte store and no cross-lane work, so the count follows the table alone — one slice per
// 4 MiB of it (n = 233 k: 15), XCDs walking two slices each one after the other.  Measured
// (profiles/r02z5_nt_stores_slices.log, whole SpMM k = 128): S = 8 / 14 / 16 / 18 / 20 / 24 / 32:
// 3.19 / 3.08 / 3.09 / 3.13 / 3.19 / 3.36 / 3.68 ms — flat from 14 to 16, then the slab of partial rows
// (S*m*k floats, written and re-read by the reduction) takes over.
// Both need >= 16 non-zeros per virtual row; at mean degree 51 (products-shaped) slicing loses and stays off.
int auto_slices(long long m, long long n, long long nnz, bool value_free) {
  if (m <= 0 || nnz <= 0) return 0;
  static const int forced = env_int("GCN_AMD_SLICES", -1);
  if (forced >= 0) return forced;                     // development knob: the slice count "auto" resolves to
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

// the drop-in csr2tile / flexspmm pair packs and runs the group-kernel format (api_dropin.cpp)
bool dropin_group_format_enabled() {
  static const bool v = env_on("GCN_AMD_DROPIN_GROUP");
  return v && group_enabled();
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

}  // namespace gcn

using gcn::g_plan_mu;

namespace {

size_t ws_elems(const gcn_spmm_plan* p, int k) {
  int chunks = std::max(p->nchunks, p->panels.out_nchunks);
  chunks = std::max(chunks, p->col16.nchunks16);
  chunks = std::max(chunks, p->group.nchunks);
  chunks = std::max(chunks, p->group_alt[0].nchunks);
  return 2 * (size_t)(chunks > 0 ? chunks : 1) * (size_t)k;
}

// rows of a CSR without an entry (the main kernels skip them, launch_fill_empty_rows writes them); synchronises `st`
int count_empty(const int* rowptr, int m, int* out, hipStream_t st) {
  gcn::DevBuf<int> cnt;
  *out = -1;
  if (cnt.alloc(1) != hipSuccess) return GCN_ERR_ALLOC;
  int host = 0;
  if (hipMemsetAsync(cnt, 0, sizeof(int), st) != hipSuccess || gcn::launch_count_empty_rows(rowptr, m, cnt, st) != hipSuccess ||
      hipMemcpyAsync(&host, cnt, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return GCN_ERR_HIP;
  *out = host;
  return GCN_OK;
}

// grow-only scratch of a plan; plans may be shared between host threads, so growth is serialised
template <class T>
int grow(gcn::DevBuf<T>& buf, size_t count) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  return buf.grow(count) == hipSuccess ? GCN_OK : GCN_ERR_ALLOC;
}

// Is a k-wide SpMM of this plan launched on the sliced copy?  The four-per-gather kernel pays from k = 33 (narrower
// rows gather 128 B or less per non-zero: the partial rows cost more than the L2 hits buy, 2.12 vs 2.02 ms at
// k = 32).  The group kernels pay from k = 12: their 64-column pass costs the same whatever k is, and beats the
// unsliced kernels there (Reddit-shaped, whole SpMM, profiles/r02zzg_narrow_widths_sliced.log: k = 12 / 16 / 20 /
// 32: 1.49 / 1.36 / 1.60 / 1.58 -> 1.23 / 1.02 / 1.23 / 1.10 ms; k = 8 a tie, k = 4 loses) — provided the width
// reaches them: a multiple of 4, or wide enough for the k' = ceil(k/4)*4 detour.
bool sliced_for(const gcn_spmm_plan* p, int k) {
  if (p->slicing.S <= 0 || p->nnz <= 0) return false;
  if (k >= gcn::slice_min_k()) return true;
  if (!p->group.ready() || p->panels.R != 0 || k < gcn::group_min_k()) return false;
  if (k % 4 == 0) return true;
  const int kp = (k + 3) / 4 * 4, ldb = (kp + 31) / 32 * 32;         // (the conditions of odd_width_detour)
  return k > 16 && p->gather_width != 1 && gcn::pad_b_enabled() && (long long)sizeof(float) * p->n * ldb <= (768LL << 20);
}

// rows of the slice-by-slice copy of B the group kernels gather from (decides their addressing mode, spmm_group.hip)
long long group_table_rows(const gcn_spmm_plan* p) { return (long long)p->slicing.S * ((long long)p->group.w + 1); }
// the slice set of the call in progress (plan.h: group_alt for the narrow width classes once they exist, else the plan's own)
const gcn::GroupStream& cur_group(const gcn_spmm_plan* p) { return p->use_alt >= 0 ? p->group_alt[p->use_alt] : p->group; }
int cur_slices(const gcn_spmm_plan* p) { return p->use_alt >= 0 ? p->alt_S[p->use_alt] : p->slicing.S; }
long long cur_table_rows(const gcn_spmm_plan* p) { return (long long)cur_slices(p) * ((long long)cur_group(p).w + 1); }

// would the sliced launch of a k-wide SpMM run a value-free kernel (and is the scaled copy of B worth it)?
bool valless_pays(const gcn_spmm_plan* p, int k, int ldb) {
  // (the scaled copy of B costs 2*n*k*4 bytes of traffic whatever the matrix; the value stream it saves is
  //  4 bytes per non-zero plus instructions.  With the group kernel the rank-0 share of an 8-way partition of
  //  the Reddit-shaped graph, 61 non-zeros per column of the block, still gains: 0.460 against 0.511 ms,
  //  profiles/r02z7_rank_share_value_free.log; below 48 per column nothing has been measured, so it stays off)
  if (!sliced_for(p, k) || !p->factors.ready() || p->panels.R != 0 || p->nnz / p->n < gcn::valless_min_per_col()) return false;
  if (p->group.vals) return false;                     // (the plan was built for the weighted pass: value-free did not pay)
  if (p->group.ready() && gcn::spmm_group_eligible(k, ldb, group_table_rows(p), nullptr, nullptr, nullptr)) return true;   // spmm_group.hip
  gcn::SpmmArgs t{};                                   // the launch as the sliced branch will issue it
  t.k = k; t.nnz = p->nnz; t.n = p->n; t.nchunks_grid = p->nchunks; t.T = p->T;
  t.m = p->slicing.S * p->m; t.ldb = ldb; t.tile_cols = p->tile_cols ? p->tile_cols : 64;
  t.gather_width = p->gather_width;
  return gcn::spmm_will_use_quad(t) && gcn::spmm_quad_lanes(k) == 16;
}

// will a sliced plan of this matrix run the group kernel value-free (known before the slicing exists)
bool value_free_plan(const gcn_spmm_plan* p) {
  return p->factors.ready() && gcn::group_enabled() && p->panels.R == 0 && p->nnz / p->n >= gcn::valless_min_per_col();
}
// ... or the group kernel at all (value-free or weighted): it decides the automatic slice count
bool group_plan(const gcn_spmm_plan* p) {
  return value_free_plan(p) || (gcn::group_enabled() && gcn::group_weighted_enabled() && p->panels.R == 0);
}
// the sliced launch of a k-wide SpMM runs the WEIGHTED group kernel (values beside the stream)
bool weighted_pass(const gcn_spmm_plan* p, int k, int ldb) {
  return sliced_for(p, k) && p->panels.R == 0 && p->group.ready() && p->group.vals &&
         gcn::spmm_group_eligible(k, ldb, group_table_rows(p), nullptr, nullptr, nullptr);
}

// the value-free pass of this plan runs the group kernel (its scaled copy of B is then laid out slice by slice)
bool group_pass(const gcn_spmm_plan* p) { return p->group.ready(); }
// a launch decided as (valless, weighted) runs one of the group kernels: B is gathered from the slice-by-slice copy
bool group_launch(const gcn_spmm_plan* p, bool valless, bool weighted) {
  return weighted || (valless && p->group.ready() && !p->group.vals);
}

// Widths that are not a multiple of 4 take a detour over k' = k rounded up to 4 (gcn_spmm_csr_f32_epilogue); it
// exists to reach the 16-byte-per-lane kernels, so it follows their rule: the four-per-gather kernel only pays
// from ~48 non-zeros per (virtual) row up, the group kernel of the value-free pass does not mind short rows
bool odd_width_detour(const gcn_spmm_plan* p, int k) {
  if (!(k > 16 && k % 4 != 0 && p->nnz > 0 && p->panels.R == 0 && p->gather_width != 1 && gcn::pad_b_enabled())) return false;
  const int kp = (k + 3) / 4 * 4, ldb = (kp + 31) / 32 * 32;
  if ((long long)sizeof(float) * p->n * ldb > (768LL << 20)) return false;
  if (p->gather_width == 4) return true;
  const bool sliced = sliced_for(p, k);
  if (sliced && p->group.ready() && (p->group.vals || value_free_plan(p))) return true;
  const long long rows = sliced ? (long long)p->slicing.S * p->m : (long long)p->m;
  return rows > 0 && p->nnz / rows >= 48;
}

// Copy of B the sliced main pass gathers from: rows `ldb` floats apart (>= k, padding columns zero), scaled by
// u_col when `scaled`; one all-zero row more than B has (16-bit stream) or, for the group kernel, slice s at
// rows [s*(w+1), (s+1)*(w+1)) with row w of every slice zero.
int relay_B(gcn_spmm_plan* p, const float* B, int k, int ldb, bool scaled, bool group_layout, hipStream_t st) {
  if (group_layout) {                                  // (weighted pass: the same layout, rows not scaled)
    const size_t rows = (size_t)cur_table_rows(p);
    const int rc = grow(p->bpad, rows * (size_t)ldb);
    if (rc != GCN_OK) return rc;
    return gcn::launch_scale_rows_sliced(p->bpad, B, scaled ? p->factors.u_col : nullptr, p->n, k, ldb, cur_slices(p), cur_group(p).w,
                                         st) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
  }
  const int rc = grow(p->bpad, ((size_t)p->n + 1) * (size_t)ldb);
  if (rc != GCN_OK) return rc;
  if (gcn::launch_pad_rows(p->bpad, B, p->n, k, ldb, st, scaled ? p->factors.u_col : nullptr) != hipSuccess ||
      hipMemsetAsync(p->bpad + (size_t)p->n * ldb, 0, sizeof(float) * (size_t)ldb, st) != hipSuccess)
    return GCN_ERR_HIP;
  return GCN_OK;
}

struct Epilogue {                                      // C = dropout(act(A*B + bias))
  const float* bias = nullptr;
  int relu = 0;
  gcn::DropoutSpec drop;
  // pre-laid output (gcn_spmm_csr_f32_prelaid; group kernels only): row r is written to row r + r / gap_w of the
  // destination and multiplied by outscale[r] — the slice-by-slice, column-scaled layout a following SpMM gathers from
  const float* outscale = nullptr;
  int gap_w = 0;
};

// b_ld: row stride of B in floats when the caller of this function has already re-laid it, 0 = k;
// b_scaled: that copy's rows are already scaled by u_col (value-free pass).
// *dropped: set when the dropout mask has been applied by a pass of this function (the slice reduction carries
// it); otherwise the caller applies it in place afterwards.
int spmm_impl(gcn_spmm_plan* p, const int32_t* rowptr, const int32_t* col, const float* val, const float* B, int b_ld,
              bool b_scaled, float* C, const Epilogue& epi, int32_t k, hipStream_t st, bool* dropped) {
  *dropped = false;
  if (grow(p->ws, ws_elems(p, k)) != GCN_OK) return GCN_ERR_ALLOC;
  const float* bias = epi.bias;
  const int relu = epi.relu ? 1 : 0;
  gcn::SpmmArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val; a.B = B; a.C = C; a.P = p->ws;
  a.chunk_row = p->chunk_row; a.bias = bias; a.relu = relu;
  a.nchunks = p->nchunks; a.T = p->T; a.m = p->m; a.nnz = p->nnz; a.k = k; a.n = p->n;
  a.nnz_dev = nullptr; a.nchunks_grid = p->nchunks;
  a.empty_rows = p->empty_rows;
  // (which widths run on the sliced copy: sliced_for)
  const bool sliced = sliced_for(p, k);
  // Feature rows that are not a whole number of 128-byte cache lines straddle lines: a gathered row
  // then costs up to one extra L2 request per tile.  Where that matters (padded_ldb) B is first re-laid
  // with its rows padded to the next multiple of 32 floats (one streaming copy, ~45 us for 233 k x 100)
  // and gathered from there; C keeps the caller's layout.  The same copy carries the row scaling of the
  // value-free pass (values u[r]*u[c], sliced matrix): B' = diag(u) B.
  bool valless = false, weighted = false;
  if (b_ld > 0) {
    a.ldb = b_ld;                                      // already re-laid (and maybe scaled) by the caller (odd-width path)
    valless = b_scaled;
    weighted = !valless && weighted_pass(p, k, b_ld);
  } else if (p->nnz > 0) {
    const int ldb = gcn::padded_ldb(p->n, k);
    valless = valless_pays(p, k, ldb);
    weighted = !valless && weighted_pass(p, k, ldb);
    if (ldb != k || valless || weighted) {
      const int rc = relay_B(p, B, k, ldb, valless, group_launch(p, valless, weighted), st);
      if (rc != GCN_OK) return rc;
      a.B = p->bpad;
      a.ldb = ldb;
    }
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;             // live timing of the main kernel (gcn_spmm_profile_begin)
  if (p->prof.armed()) { const auto pr = p->prof.next(); ev0 = pr.first; ev1 = pr.second; }
  a.blocks_per_cu = p->blocks_per_cu;
  a.gather_width = p->gather_width;
  gcn::Panels& pn = p->panels;
  if (pn.R > 0 && p->nnz > 0 && k > 32) {
    // A = A_in + A_out: the staged part from LDS (raw sums into C), then the rest accumulated by the
    // chunk kernel, which also carries the epilogue
    if (ev0 && hipEventRecord(ev0, st) != hipSuccess) return GCN_ERR_HIP;
    const int tiles = (k + 63) / 64;
    for (int t = 0; t < tiles; ++t) {
      if (gcn::launch_panel_in(pn.in_rowptr, pn.in_off, pn.in_val, B, C, pn.w0, p->m, p->n, k, pn.R, t, st, pn.dense_slot) != hipSuccess)
        return GCN_ERR_HIP;
      if (gcn::launch_panel_dense(pn.adense, pn.dense_panel, pn.ndense, B, C, pn.w0, p->m, p->n, k, pn.R, t, st) != hipSuccess)
        return GCN_ERR_HIP;
    }
    if (pn.out_nnz == 0) {
      if (gcn::launch_panel_epilogue(C, bias, relu, p->m, k, st) != hipSuccess) return GCN_ERR_HIP;
      if (ev1 && hipEventRecord(ev1, st) != hipSuccess) return GCN_ERR_HIP;
      return GCN_OK;
    }
    a.nchunks = pn.out_nchunks; a.nchunks_grid = pn.out_nchunks;
    a.T = pn.out_T; a.nnz = pn.out_nnz;
    a.empty_rows = -1;                                 // (rows whose entries all sit inside their window: not counted)
    a.ev_start = nullptr; a.ev_stop = ev1;
    if (pn.out_S > 0) {
      // sliced: partial rows of the virtual CSR, then C += sum of the partials (+ epilogue)
      if (grow(p->cv, (size_t)pn.out_S * (size_t)p->m * (size_t)k) != GCN_OK) return GCN_ERR_ALLOC;
      a.rowptr = pn.out_vrowptr; a.col = pn.out_vcol; a.val = pn.out_vval; a.chunk_row = pn.out_vchunk_row;
      a.C = p->cv; a.m = pn.out_S * p->m; a.bias = nullptr; a.relu = 0; a.accumulate = 0;
      a.tile_cols = p->tile_cols ? p->tile_cols : 64;
      if (gcn::launch_spmm(a, p->cu_count, st) != hipSuccess) return GCN_ERR_HIP;
      *dropped = epi.drop.on();
      return gcn::launch_slice_reduce(p->cv, C, bias, relu, p->m, pn.out_S, k, st, 1, nullptr, epi.drop) == hipSuccess
                 ? GCN_OK : GCN_ERR_HIP;
    }
    a.rowptr = pn.out_rowptr; a.col = pn.out_col; a.val = pn.out_val;
    a.chunk_row = pn.out_chunk_row; a.accumulate = 1;
    a.tile_cols = p->tile_cols ? p->tile_cols : gcn::auto_tile_cols(p->n, k);
    return gcn::launch_spmm(a, p->cu_count, st) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
  }
  a.tile_cols = p->tile_cols ? p->tile_cols : (sliced ? 64 : gcn::auto_tile_cols(p->n, k));
  a.ev_start = ev0; a.ev_stop = ev1;
  if (!sliced) return gcn::launch_spmm(a, p->cu_count, st) == hipSuccess ? GCN_OK : GCN_ERR_HIP;

  // sliced: the slice-major virtual CSR (S*m rows) into the partial buffer, then the per-row reduction
  // over slices, which carries the whole epilogue (bias, ReLU, dropout mask, row factor)
  const gcn::Slicing& sl = p->slicing;
  const bool grp = group_launch(p, valless, weighted);
  const int S_run = grp ? cur_slices(p) : sl.S;
  if (grow(p->cv, (size_t)S_run * (size_t)p->m * (size_t)k) != GCN_OK) return GCN_ERR_ALLOC;
  *dropped = epi.drop.on();
  if (grp) {
    // four independent 16-lane row engines per wave on the 15-bit slice-major stream (spmm_group.hip)
    const gcn::GroupStream& G = cur_group(p);
    gcn::GroupArgs ga;
    ga.stream = G.stream; ga.chunk_meta = G.chunk_meta;
    ga.vals = weighted ? G.vals.get() : nullptr;
    ga.Bp = a.B; ga.Cv = p->cv; ga.P = p->ws;
    ga.nchunks = G.nchunks; ga.T = G.T; ga.k = k; ga.ldb = a.ldb;
    ga.table_rows = cur_table_rows(p);
    ga.store_policy = gcn::group_store();
    ga.ring = gcn::group_ring() ? 1 : 0;
    ga.merge_tiles = gcn::group_merge_tiles() ? 1 : 0;
    ga.narrow8 = gcn::group8_enabled() ? 1 : 0;
    if (ev0 && hipEventRecord(ev0, st) != hipSuccess) return GCN_ERR_HIP;
    if (gcn::launch_spmm_group(ga, st) != hipSuccess) return GCN_ERR_HIP;
    if (ev1 && hipEventRecord(ev1, st) != hipSuccess) return GCN_ERR_HIP;
    // rows cut by chunk ends: their later pieces are added by the reduction itself (cut lists per output row), or —
    // GCN_AMD_GROUP_FUSED_FIXUP=0, or a plan without the lists — by a pass of their own in front of it
    gcn::CutLists cuts;
    if (gcn::group_fused_fixup() && G.cutptr) { cuts.ptr = G.cutptr; cuts.chunk = G.cutchunk; cuts.P = p->ws; }
    else if (gcn::launch_group_fixup(G.fix, G.nfix, p->ws, p->cv, k, st) != hipSuccess) return GCN_ERR_HIP;
    return gcn::launch_slice_reduce(p->cv, C, bias, relu, p->m, S_run, k, st, 0, weighted ? nullptr : p->factors.u_row.get(),
                                    epi.drop, nullptr, epi.outscale, epi.gap_w, cuts) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
  }
  a.rowptr = sl.vrowptr; a.col = sl.vcol; a.val = sl.vval; a.chunk_row = sl.vchunk_row;
  a.C = p->cv; a.m = sl.S * p->m; a.bias = nullptr; a.relu = 0;
  a.empty_rows = sl.empty_vrows;
  a.stream_rows = gcn::quad_stream_rows() ? 1 : 0;
  const float* rowscale = nullptr;
  if (valless) {                                                          // B was scaled by u_col above
    a.valless = 1; a.val = nullptr; rowscale = p->factors.u_row;
    const gcn::Col16Stream& c16 = p->col16;
    if (c16.ready()) {                                                    // 16-bit column stream, slice-aligned chunks
      a.rowptr = c16.vrowptr16; a.col = reinterpret_cast<const int*>(c16.vcol16.get()); a.chunk_row = c16.vchunk_row16;
      a.nnz = c16.nnz16; a.nchunks = a.nchunks_grid = c16.nchunks16;
      a.col16 = 1; a.col16_S = sl.S; a.col16_w = (p->n + sl.S - 1) / sl.S;
      a.empty_rows = -1;                                                  // (its own row pointer: not counted)
      for (int i = 0; i < 9; ++i) a.col16_start[i] = c16.start16[i];
    }
  }
  if (gcn::launch_spmm(a, p->cu_count, st) != hipSuccess) return GCN_ERR_HIP;
  return gcn::launch_slice_reduce(p->cv, C, bias, relu, p->m, sl.S, k, st, 0, rowscale, epi.drop) == hipSuccess
             ? GCN_OK : GCN_ERR_HIP;
}

// the streams of a sliced plan beside its virtual CSR: the group kernel's (value-free when the values factor and
// the scaled copy pays, else with the values beside it), or the 16-bit columns of the value-free four-per-gather pass
void build_sliced_streams(gcn_spmm_plan* p, hipStream_t st) {
  gcn::Slicing& sl = p->slicing;
  if (sl.S <= 1 || p->group.ready() || p->col16.ready() || (!group_plan(p) && !p->factors.ready())) return;
  const long long vm = (long long)sl.S * p->m;
  const int w = (p->n + sl.S - 1) / sl.S;
  const bool value_free = value_free_plan(p);
  // 15-bit stream of the group kernel: slices at most 32 767 columns wide; best effort
  if (w <= 32767 && group_plan(p) && p->group.vrowptr.alloc((size_t)(vm + 1)) == hipSuccess) {
    unsigned short* stream = nullptr;
    float* vals = nullptr;
    int *chunk_row = nullptr, *chunk_meta = nullptr, *fix = nullptr, nch = 0, nfix = 0;
    int *cutptr = nullptr, *cutchunk = nullptr, ncut = 0;
    const int gT = gcn::group_chunk(p->nnz, p->cu_count);
    if (gcn::build_group_stream(sl.vrowptr, sl.vcol, p->m, p->n, sl.S, gT, p->group.vrowptr, &stream,
                                &chunk_row, &chunk_meta, &nch, &fix, &nfix, st, value_free ? nullptr : sl.vval.get(),
                                value_free ? nullptr : &vals, &cutptr, &cutchunk, &ncut) == hipSuccess && nch > 0) {
      p->group.fix.adopt(fix, 4 * (size_t)nfix); p->group.nfix = nfix;
      p->group.cutptr.adopt(cutptr, (size_t)p->m + 1); p->group.cutchunk.adopt(cutchunk, (size_t)(ncut > 0 ? ncut : 1)); p->group.ncut = ncut;
      p->group.stream.adopt(stream, (size_t)nch * (size_t)gT);
      if (vals) p->group.vals.adopt(vals, (size_t)nch * (size_t)gT);
      p->group.chunk_row.adopt(chunk_row, (size_t)nch);
      p->group.chunk_meta.adopt(chunk_meta, 2 * (size_t)nch);
      p->group.nchunks = nch; p->group.T = gT; p->group.w = w;
      p->group.chunk_row.reset();                      // (only the builder needed these two: the kernels read
      p->group.vrowptr.reset();                        //  chunk_meta and the fix list)
      return;
    }
    p->group = gcn::GroupStream{};
  for (int c_ = 0; c_ < 1; ++c_) { p->group_alt[c_] = gcn::GroupStream{}; p->alt_S[c_] = 0; p->alt_tried[c_] = false; }
  p->use_alt = -1;
  }
  if (!p->factors.ready()) return;
  // 16-bit column stream of the four-per-gather kernel (2 instead of 4 index bytes per non-zero): slices at
  // most 65 535 columns wide, at most 8 of them; best effort — without it the 32-bit stream is used
  if (sl.S <= 8 && w <= 65535 && gcn::col16_enabled() && p->col16.vrowptr16.alloc((size_t)(vm + 1)) == hipSuccess) {
    gcn::Col16Stream& c = p->col16;
    unsigned short* c16 = nullptr;
    int nnz16 = 0;
    if (gcn::build_col16_stream(sl.vrowptr, sl.vcol, p->m, p->n, sl.S, p->T, c.vrowptr16, &c16, &nnz16, c.start16, st) == hipSuccess &&
        nnz16 > 0) {
      c.vcol16.adopt(c16, (size_t)nnz16);
      c.nnz16 = nnz16;
      c.nchunks16 = nnz16 / p->T;
      if (c.vchunk_row16.alloc((size_t)c.nchunks16) == hipSuccess &&
          gcn::launch_plan_chunk_rows(c.vrowptr16, (int)vm, p->T, c.nchunks16, c.vchunk_row16, st) == hipSuccess &&
          hipStreamSynchronize(st) == hipSuccess)
        return;
    }
    p->col16 = gcn::Col16Stream{};                     // anything failed: drop the 16-bit stream
  }
}

}  // namespace

extern "C" {

const char* gcn_status_string(int s) {
  switch (s) {
    case GCN_OK: return "ok";
    case GCN_ERR_INVALID_ARG: return "invalid argument";
    case GCN_ERR_HIP: return "HIP runtime error";
    case GCN_ERR_NO_DEVICE: return "no HIP device";
    case GCN_ERR_CAPACITY: return "caller buffer too small";
    case GCN_ERR_ALLOC: return "device allocation failed";
    case GCN_ERR_NOT_FACTORED: return "values do not factor as u_row[r]*u_col[c]";
    case GCN_ERR_INTERNAL: return "internal consistency guard tripped";
    default: return "unknown status";
  }
}

const char* gcn_version(void) { return GCN_VERSION_STR; }

int gcn_device_cu_count(void) { return gcn::cu_count_cached(); }

// ---------------------------------------------------------------------------
int gcn_spmm_plan_create(gcn_spmm_plan_t** out, const int32_t* rowptr_dev, int32_t m, int32_t n,
                         int32_t nnz, int32_t chunk_nnz, void* stream) {
  if (!out || m < 0 || n < 0 || nnz < 0 || (m > 0 && !rowptr_dev)) return GCN_ERR_INVALID_ARG;
  if (chunk_nnz < 0 || (chunk_nnz % 64) != 0) return GCN_ERR_INVALID_ARG;
  const int cu = gcn::cu_count_cached();
  if (cu <= 0) return GCN_ERR_NO_DEVICE;
  gcn_spmm_plan* p = new (std::nothrow) gcn_spmm_plan();
  if (!p) return GCN_ERR_ALLOC;
  p->m = m; p->n = n; p->nnz = nnz; p->cu_count = cu;
  p->T = chunk_nnz ? chunk_nnz : gcn::auto_chunk_nnz(nnz, cu);
  p->nchunks = (int)(((long long)nnz + p->T - 1) / p->T);
  (void)hipGetDevice(&p->device);
  if (p->nchunks > 0) {
    if (p->chunk_row.alloc((size_t)p->nchunks) != hipSuccess) { delete p; return GCN_ERR_ALLOC; }
    // (synchronised: the header promises that rowptr_dev is only read during this call)
    if (gcn::launch_plan_chunk_rows(rowptr_dev, m, p->T, p->nchunks, p->chunk_row, (hipStream_t)stream) != hipSuccess ||
        count_empty(rowptr_dev, m, &p->empty_rows, (hipStream_t)stream) != GCN_OK) {
      delete p;
      return GCN_ERR_HIP;
    }
  }
  *out = p;
  return GCN_OK;
}

int gcn_spmm_plan_destroy(gcn_spmm_plan_t* p) {
  delete p;                                            // every buffer and event is owned by a member
  return GCN_OK;
}

int32_t gcn_spmm_plan_num_chunks(const gcn_spmm_plan_t* p) { return p ? p->nchunks : -1; }
int32_t gcn_spmm_plan_chunk_nnz(const gcn_spmm_plan_t* p) { return p ? p->T : -1; }
size_t gcn_spmm_plan_workspace_bytes(const gcn_spmm_plan_t* p, int32_t k) {
  return (!p || k <= 0) ? 0 : sizeof(float) * ws_elems(p, k);
}

namespace {

// The narrow slice set of a plan (plan.h, group_alt[0]): for k <= 32 a row of the table is 128 bytes, so an L2 holds a
// slice twice as wide and the matrix needs about half the slices — and every slice costs a partial row per matrix row.
// Reddit-shaped (profiles/r03az_*): 8 slices instead of 15; k = 16 / 32 whole SpMM 0.684 / 0.782 -> 0.655 / 0.746 ms.
// Built once, at the first such call of a value-free plan with an automatic slice count, from the CSR the call hands
// over (a transient virtual CSR;
// only the stream, its chunk table and cut lists are kept: 2 bytes per non-zero).  Anything that fails leaves the plan
// on its own slices.  GCN_AMD_GROUP_NARROW_SLICES=0: off.
int alt_class(int k) { return k <= 32 ? 0 : -1; }

void maybe_build_alt(gcn_spmm_plan* p, int cls, const int32_t* rowptr, const int32_t* col, const float* val, hipStream_t st) {
  if (cls < 0 || p->alt_tried[cls]) return;
  p->alt_tried[cls] = true;
  static const bool on = gcn::env_on("GCN_AMD_GROUP_NARROW_SLICES");
  if (!on || !p->slices_auto || !p->group.ready() || p->group.vals || !value_free_plan(p) || p->nnz <= 0) return;
  const long long l2 = 4LL << 20, row_bytes = 128;
  long long S2 = ((long long)p->n * row_bytes + l2 - 1) / l2;
  const long long by_entry = ((long long)p->n + 32766) / 32767;       // 15-bit entries: slices <= 32 767 columns
  if (S2 < by_entry) S2 = by_entry;
  if (S2 > (long long)p->nnz / p->m / 16) S2 = (long long)p->nnz / p->m / 16;
  if (S2 < 2 || S2 + 2 > p->slicing.S) return;                        // (not enough fewer to pay for another stream)
  const int S = (int)S2, w = (p->n + S - 1) / S;
  if (w > 32767) return;
  const long long vm = (long long)S * p->m;
  gcn::DevBuf<int> vrowptr, vcol, vrowptr_g;
  gcn::DevBuf<float> vval;
  if (vrowptr.alloc((size_t)vm + 1) != hipSuccess || vcol.alloc((size_t)p->nnz) != hipSuccess ||
      vval.alloc((size_t)p->nnz) != hipSuccess || vrowptr_g.alloc((size_t)vm + 1) != hipSuccess) return;
  int sorted = 0;
  if (gcn::build_sliced_csr(rowptr, col, val, p->m, p->n, p->nnz, S, vrowptr, vcol, vval, &sorted, st) != hipSuccess || !sorted) return;
  unsigned short* stream = nullptr;
  int *chunk_row = nullptr, *chunk_meta = nullptr, *fix = nullptr, *cutptr = nullptr, *cutchunk = nullptr, nch = 0, nfix = 0, ncut = 0;
  const int gT = gcn::group_chunk(p->nnz, p->cu_count);
  if (gcn::build_group_stream(vrowptr, vcol, p->m, p->n, S, gT, vrowptr_g, &stream, &chunk_row, &chunk_meta, &nch, &fix, &nfix, st,
                              nullptr, nullptr, &cutptr, &cutchunk, &ncut) != hipSuccess || nch <= 0) return;
  gcn::GroupStream& g = p->group_alt[cls];
  g.fix.adopt(fix, 4 * (size_t)nfix); g.nfix = nfix;
  g.cutptr.adopt(cutptr, (size_t)p->m + 1); g.cutchunk.adopt(cutchunk, (size_t)(ncut > 0 ? ncut : 1)); g.ncut = ncut;
  g.stream.adopt(stream, (size_t)nch * (size_t)gT);
  g.chunk_meta.adopt(chunk_meta, 2 * (size_t)nch);
  g.chunk_row.adopt(chunk_row, (size_t)nch); g.chunk_row.reset();
  g.nchunks = nch; g.T = gT; g.w = w;
  p->alt_S[cls] = S;
  if (gcn::verbose())
    std::fprintf(stderr, "libgcnspmm: slice set for k <= 32: %d slices of %d columns (the plan's own: %d)\n", S, w, p->slicing.S);
}

// Which slice set does a k-wide call run on (k already rounded up to a multiple of 4; *ldb the row stride it would
// gather with)?  Builds the narrow set on first use.  Widths 33..48 on the five-engine kernel stay on the plan's own
// slices; where the row stride would have been padded to 64 floats (k = 44 and the odd widths' k' detour) they gather
// from rows of 48 instead (192 bytes: always two lines, a quarter less table and copy): *ldb = 48, *relay = the call
// lays that copy out itself (k = 41 / 47: 1.39 / 1.36 -> 1.34 / 1.30 ms; 36 / 40 keep their dense rows).  (A slice set of their own — 11..13 slices
// instead of 15 — was built and measured: +-1 %, profiles/r03ba_*; not kept.)
int pick_slice_set(gcn_spmm_plan* p, int k, int* ldb, bool* relay, const int32_t* rowptr, const int32_t* col, const float* val,
                   hipStream_t st) {
  *relay = false;
  if (p->nnz <= 0 || k % 4 != 0) return -1;
  if (k > 32 && k <= 48 && *ldb > 48 && valless_pays(p, k, 48) && group_launch(p, true, false)) {
    gcn::GroupArgs probe{};
    probe.k = k; probe.ldb = 48; probe.table_rows = group_table_rows(p); probe.ring = gcn::group_ring() ? 1 : 0;
    if (gcn::spmm_group12_applies(probe)) { *ldb = 48; *relay = true; }
    return -1;
  }
  const int cls = alt_class(k);
  if (cls < 0 || !valless_pays(p, k, *ldb) || !group_launch(p, true, false)) return -1;
  maybe_build_alt(p, cls, rowptr, col, val, st);
  return p->group_alt[cls].ready() ? cls : -1;
}

}  // namespace

int gcn_spmm_csr_f32_epilogue(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col, const float* val,
                              const float* B, float* C, const float* bias, int32_t relu, float dropout_p,
                              uint64_t seed, uint64_t offset, int32_t k, void* stream) {
  if (!p || k < 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) return GCN_ERR_INVALID_ARG;
  if (p->m == 0 || k == 0) return GCN_OK;
  if (!C || !rowptr || (p->nnz > 0 && (!col || !val || !B))) return GCN_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  // which slice set does this call run on?  (decided here, once, for the re-laid copy of B and the kernels alike)
  const bool odd = odd_width_detour(p, k);
  int ld_call = odd ? ((k + 3) / 4 * 4 + 31) / 32 * 32 : gcn::padded_ldb(p->n, k);
  bool relay48 = false;
  p->use_alt = pick_slice_set(p, odd ? (k + 3) / 4 * 4 : k, &ld_call, &relay48, rowptr, col, val, st);
  Epilogue epi;
  epi.bias = bias; epi.relu = relu ? 1 : 0;
  epi.drop.p = dropout_p; epi.drop.seed = seed; epi.drop.offset = offset;
  bool dropped = false;
  int rc;
  // Widths that are not a multiple of 4 (class counts: 41, 47, ...) cannot use the 16-byte-per-lane
  // kernels on the caller's layout.  They are computed at k' = k rounded up to 4 on row-padded copies:
  // B re-laid with zero columns (stride a multiple of 32 floats), the product into a k'-wide scratch
  // result, and one pass that compacts it into C (and applies bias / ReLU).  Reddit-shaped k = 41:
  // 2.13 -> 1.87 ms.  Same limits as the B padding (tables <= 768 MiB), panels excluded.
  if (odd) {
    const int kp = (k + 3) / 4 * 4, ldb = ld_call;
    if (grow(p->cpad, (size_t)p->m * (size_t)kp) != GCN_OK) return GCN_ERR_ALLOC;
    const bool scaled = valless_pays(p, kp, ldb);      // the copy can carry the u_col scaling
    const bool weighted = !scaled && weighted_pass(p, kp, ldb);
    if ((rc = relay_B(p, B, k, ldb, scaled, group_launch(p, scaled, weighted), st)) != GCN_OK) return rc;
    if ((rc = spmm_impl(p, rowptr, col, val, p->bpad, ldb, scaled, p->cpad, Epilogue{}, kp, st, &dropped)) != GCN_OK) return rc;
    if (gcn::launch_unpad_rows(C, p->cpad, bias, epi.relu, p->m, k, kp, st) != hipSuccess) return GCN_ERR_HIP;
    dropped = false;
  } else if (relay48) {                                // k = 44 on the five-engine kernel: rows of 48 floats, scaled, slice by slice
    if ((rc = relay_B(p, B, k, ld_call, true, true, st)) != GCN_OK) return rc;
    if ((rc = spmm_impl(p, rowptr, col, val, p->bpad, ld_call, true, C, epi, k, st, &dropped)) != GCN_OK) return rc;
  } else {
    if ((rc = spmm_impl(p, rowptr, col, val, B, 0, false, C, epi, k, st, &dropped)) != GCN_OK) return rc;
  }
  if (epi.drop.on() && !dropped)                       // no epilogue pass carried the mask: one pass in place
    return gcn::launch_dropout(C, C, (long long)p->m * k, epi.drop, st) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
  return GCN_OK;
}

int gcn_spmm_csr_f32_bias_relu(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                               const float* val, const float* B, float* C, const float* bias,
                               int32_t relu, int32_t k, void* stream) {
  return gcn_spmm_csr_f32_epilogue(p, rowptr, col, val, B, C, bias, relu, 0.f, 0, 0, k, stream);
}

int gcn_spmm_csr_f32(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                     const float* val, const float* B, float* C, int32_t k, void* stream) {
  return gcn_spmm_csr_f32_epilogue(p, rowptr, col, val, B, C, nullptr, 0, 0.f, 0, 0, k, stream);
}

int gcn_spmm_plan_prelaid_layout(const gcn_spmm_plan_t* p, int32_t k, int32_t* slices, int32_t* slice_cols,
                                 int64_t* table_rows, int32_t* ld) {
  if (!p || k <= 0 || k % 4 != 0) return GCN_ERR_INVALID_ARG;
  const int ldb = gcn::padded_ldb(p->n, k);
  // only the value-free group pass gathers from a scaled, slice-by-slice copy of B
  if (!valless_pays(p, k, ldb) || !group_launch(p, true, false)) return GCN_ERR_INVALID_ARG;
  if (slices) *slices = p->slicing.S;
  if (slice_cols) *slice_cols = p->group.w;
  if (table_rows) *table_rows = group_table_rows(p);
  if (ld) *ld = ldb;
  return GCN_OK;
}

int gcn_spmm_csr_f32_prelaid(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col, const float* val,
                             const float* Bp, float* out, const float* out_scale, int32_t out_gap, int32_t k,
                             void* stream) {
  if (!p || k <= 0 || out_gap < 0) return GCN_ERR_INVALID_ARG;
  if (p->m == 0) return GCN_OK;
  if (!Bp || !out || !rowptr || !col || !val) return GCN_ERR_INVALID_ARG;
  int32_t ld = 0;
  const int rc = gcn_spmm_plan_prelaid_layout(p, k, nullptr, nullptr, nullptr, &ld);
  if (rc != GCN_OK) return rc;
  if ((((uintptr_t)Bp | (uintptr_t)out) & 15) != 0) return GCN_ERR_INVALID_ARG;
  Epilogue epi;
  epi.outscale = out_scale;
  epi.gap_w = out_gap;
  bool dropped = false;
  p->use_alt = -1;                                     // (the pre-laid layout is the plan's own slice set, whatever the width)
  return spmm_impl(p, rowptr, col, val, Bp, ld, /*b_scaled=*/true, out, epi, k, (hipStream_t)stream, &dropped);
}

int gcn_dropout_f32(float* dst, const float* src, int64_t count, float dropout_p, uint64_t seed, uint64_t offset,
                    void* stream) {
  if (count < 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) return GCN_ERR_INVALID_ARG;
  if (count == 0) return GCN_OK;
  if (!dst || !src) return GCN_ERR_INVALID_ARG;
  gcn::DropoutSpec d;
  d.p = dropout_p; d.seed = seed; d.offset = offset;
  if (!d.on()) {
    if (dst == src) return GCN_OK;
    return hipMemcpyAsync(dst, src, sizeof(float) * (size_t)count, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess
               ? GCN_OK : GCN_ERR_HIP;
  }
  return gcn::launch_dropout(dst, src, (long long)count, d, (hipStream_t)stream) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

int gcn_spmm_plan_enable_slicing(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                                 const float* val, int32_t slices, void* stream) {
  if (!p || slices < -1 || slices > 1024) return GCN_ERR_INVALID_ARG;
  p->slicing = gcn::Slicing{};
  p->col16 = gcn::Col16Stream{};
  p->group = gcn::GroupStream{};
  for (int c_ = 0; c_ < 1; ++c_) { p->group_alt[c_] = gcn::GroupStream{}; p->alt_S[c_] = 0; p->alt_tried[c_] = false; }
  p->use_alt = -1;
  p->cv.reset();
  const bool autom = slices == -1;
  p->slices_auto = autom;
  hipStream_t st = (hipStream_t)stream;
  if (p->nnz == 0 || p->m == 0 || slices == 0 || slices == 1) return GCN_OK;
  if (!rowptr || !col || !val) return GCN_ERR_INVALID_ARG;
  // Normalised adjacencies (D^-1/2 (A+I) D^-1/2) have values u[r]*u[c]: when every stored entry matches
  // that to 4 ulp the sliced main pass can run without its value stream, on a B whose rows were scaled by u,
  // with the row factor applied in the slice reduction.  (Factors handed over by the caller stay.)  Looked
  // for first: the automatic slice count depends on it.
  if (p->m == p->n && gcn::valless_enabled() && !p->factors.ready() &&
      (!autom || gcn::auto_slices(p->m, p->n, p->nnz, false) > 1)) {
    gcn::DevBuf<float> u;
    int ok = 0;
    if (u.alloc((size_t)p->n) == hipSuccess &&
        gcn::detect_rank1_values(rowptr, col, val, p->n, u, &ok, st) == hipSuccess && ok) {
      p->factors.u_row = std::move(u);
      p->factors.u_col = p->factors.u_row;
    }
  }
  // ... or depend on the row only / on the column only (r03): an unweighted adjacency (all ones), the row-normalised
  // D^-1 (A+I) of Kipf's pygcn, and its transpose (what the backward pass multiplies with) factor as u_row[r] * 1 and
  // 1 * u_col[c]; any shape.  Same 4-ulp check of every entry.
  if (gcn::valless_enabled() && !p->factors.ready() && (!autom || gcn::auto_slices(p->m, p->n, p->nnz, false) > 1)) {
    for (int mode = 1; mode <= 2 && !p->factors.ready(); ++mode) {
      gcn::Factors f;
      int ok = 0;
      if (f.u_row.alloc((size_t)p->m) == hipSuccess && f.u_col_own.alloc((size_t)p->n) == hipSuccess &&
          gcn::detect_constant_values(rowptr, col, val, p->m, p->n, p->nnz, mode, f.u_row, f.u_col_own, &ok, st) == hipSuccess && ok) {
        f.u_col = f.u_col_own;
        p->factors = std::move(f);
      }
    }
  }
  if (autom) slices = gcn::auto_slices(p->m, p->n, p->nnz, group_plan(p));
  if (slices <= 1) return GCN_OK;
  if ((long long)slices * p->m + 1 >= (1LL << 31)) return GCN_ERR_INVALID_ARG;
  const long long vm = (long long)slices * p->m;
  gcn::Slicing sl;
  if (sl.vrowptr.alloc((size_t)(vm + 1)) != hipSuccess || sl.vcol.alloc((size_t)p->nnz) != hipSuccess ||
      sl.vval.alloc((size_t)p->nnz) != hipSuccess || sl.vchunk_row.alloc((size_t)p->nchunks) != hipSuccess)
    return GCN_ERR_ALLOC;
  int sorted = 1;
  if (gcn::build_sliced_csr(rowptr, col, val, p->m, p->n, p->nnz, slices, sl.vrowptr, sl.vcol, sl.vval, &sorted, st) != hipSuccess)
    return GCN_ERR_HIP;
  if (!sorted) return autom ? GCN_OK : GCN_ERR_INVALID_ARG;   // needs column-sorted rows; auto mode just stays unsliced
  if (gcn::launch_plan_chunk_rows(sl.vrowptr, (int)vm, p->T, p->nchunks, sl.vchunk_row, st) != hipSuccess ||
      count_empty(sl.vrowptr, (int)vm, &sl.empty_vrows, st) != GCN_OK)
    return GCN_ERR_HIP;
  sl.S = slices;
  p->slicing = std::move(sl);
  build_sliced_streams(p, st);
  return GCN_OK;
}

int32_t gcn_spmm_plan_num_slices(const gcn_spmm_plan_t* p) { return p ? p->slicing.S : -1; }
int32_t gcn_spmm_plan_narrow_slices(const gcn_spmm_plan_t* p, int32_t k) {
  if (!p || k <= 0) return -1;
  const int cls = alt_class((k + 3) / 4 * 4);
  return cls >= 0 && p->group_alt[cls].ready() ? p->alt_S[cls] : 0;
}

int gcn_spmm_plan_set_value_factors(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                                    const float* val, const float* u_row, const float* u_col, void* stream) {
  if (!p) return GCN_ERR_INVALID_ARG;
  p->factors = gcn::Factors{};
  p->col16 = gcn::Col16Stream{};                       // (the value-free streams exist only beside factors)
  p->group = gcn::GroupStream{};
  for (int c_ = 0; c_ < 1; ++c_) { p->group_alt[c_] = gcn::GroupStream{}; p->alt_S[c_] = 0; p->alt_tried[c_] = false; }
  p->use_alt = -1;
  if (!u_row && !u_col) {                                           // (null, null): forget the factors;
    build_sliced_streams(p, (hipStream_t)stream);                   // the sliced plan goes back to its value stream
    return GCN_OK;
  }
  if (!u_row || !u_col || !rowptr || (p->nnz > 0 && (!col || !val))) return GCN_ERR_INVALID_ARG;
  if (p->m == 0 || p->nnz == 0 || !gcn::valless_enabled()) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  int ok = 0;
  if (gcn::verify_value_factors(rowptr, col, val, u_row, u_col, p->m, &ok, st) != hipSuccess) return GCN_ERR_HIP;
  if (!ok) {                                                        // some entry is not u_row[r]*u_col[c]
    build_sliced_streams(p, st);                                    // (the plan keeps working on its value stream)
    return GCN_ERR_NOT_FACTORED;
  }
  gcn::Factors f;
  if (f.u_row.alloc((size_t)p->m) != hipSuccess || f.u_col_own.alloc((size_t)p->n) != hipSuccess) return GCN_ERR_ALLOC;
  if (hipMemcpyAsync(f.u_row, u_row, sizeof(float) * (size_t)p->m, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemcpyAsync(f.u_col_own, u_col, sizeof(float) * (size_t)p->n, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return GCN_ERR_HIP;
  f.u_col = f.u_col_own;
  p->factors = std::move(f);
  // a slice count chosen automatically was chosen for a matrix WITH a value stream: choose again
  if (p->slices_auto && gcn::auto_slices(p->m, p->n, p->nnz, group_plan(p)) != p->slicing.S)
    return gcn_spmm_plan_enable_slicing(p, rowptr, col, val, -1, stream);
  build_sliced_streams(p, st);
  return GCN_OK;
}

int32_t gcn_spmm_plan_has_value_factors(const gcn_spmm_plan_t* p) { return p ? (p->factors.ready() ? 1 : 0) : -1; }

int gcn_spmm_plan_enable_panels(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                                const float* val, int32_t mode, void* stream) {
  if (!p || mode < -1 || mode > 1) return GCN_ERR_INVALID_ARG;
  p->panels = gcn::Panels{};
  if (mode == 0 || p->nnz == 0 || p->m == 0) return GCN_OK;
  if (!rowptr || !col || !val) return GCN_ERR_INVALID_ARG;
  const int R = 128;
  const int npanels = (p->m + R - 1) / R;
  hipStream_t st = (hipStream_t)stream;
  gcn::Panels pn;
  gcn::DevBuf<int> pcnt;                               // in-window non-zeros of every panel
  if (pn.w0.alloc((size_t)npanels) != hipSuccess || pcnt.alloc((size_t)npanels) != hipSuccess) return GCN_ERR_ALLOC;
  unsigned long long inside = 0;
  if (gcn::panel_plan(rowptr, col, p->m, p->n, R, pn.w0, &inside, st, pcnt) != hipSuccess) return GCN_ERR_HIP;
  pn.coverage = (double)inside / (double)p->nnz;
  // automatic: only when at least half of the non-zeros are served from the staged tile
  if (!(mode == 1 || pn.coverage >= 0.5)) { p->panels.coverage = pn.coverage; return GCN_OK; }
  // Panels whose 128 x 512 window is dense enough leave the sparse formats altogether: a dense fp32 tile in
  // MFMA fragment order, contracted on the matrix cores (spmm_panel_dense_mfma_kernel); break-even against one
  // LDS read per entry is near 13 % density, the default threshold 25 %.
  if (gcn::panel_mfma_enabled()) {
    std::vector<int> cnt((size_t)npanels), slot((size_t)npanels, -1), ids;
    if (hipMemcpyAsync(cnt.data(), pcnt, sizeof(int) * (size_t)npanels, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return GCN_ERR_HIP;
    const double thr = gcn::panel_mfma_density() * (double)R * 512.0;
    for (int q = 0; q < npanels; ++q)
      if ((double)cnt[(size_t)q] >= thr) { slot[(size_t)q] = (int)ids.size(); ids.push_back(q); }
    if (!ids.empty()) {
      const size_t tile = (size_t)R * 512;
      if (pn.dense_slot.alloc((size_t)npanels) != hipSuccess || pn.dense_panel.alloc(ids.size()) != hipSuccess ||
          pn.adense.alloc(ids.size() * tile) != hipSuccess)
        return GCN_ERR_ALLOC;
      if (hipMemcpyAsync(pn.dense_slot, slot.data(), sizeof(int) * (size_t)npanels, hipMemcpyHostToDevice, st) != hipSuccess ||
          hipMemcpyAsync(pn.dense_panel, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
          hipMemsetAsync(pn.adense, 0, sizeof(float) * ids.size() * tile, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess)               // (slot / ids are host vectors)
        return GCN_ERR_HIP;
      pn.ndense = (int)ids.size();
    }
  }
  // split A = A_in + A_out (+ the dense tiles) on the device
  if (pn.in_rowptr.alloc((size_t)p->m + 1) != hipSuccess || pn.out_rowptr.alloc((size_t)p->m + 1) != hipSuccess)
    return GCN_ERR_ALLOC;
  int nnz_in = 0;
  if (gcn::panel_split(rowptr, col, val, pn.w0, p->m, R, pn.in_rowptr, pn.out_rowptr, nullptr, nullptr, nullptr, nullptr,
                       &nnz_in, st, pn.dense_slot) != hipSuccess)
    return GCN_ERR_HIP;
  int nnz_out = 0;                                     // (dense-tile entries are neither staged nor rest)
  if (hipMemcpyAsync(&nnz_out, pn.out_rowptr + p->m, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return GCN_ERR_HIP;
  if (pn.in_off.alloc((size_t)nnz_in) != hipSuccess || pn.in_val.alloc((size_t)nnz_in) != hipSuccess ||
      pn.out_col.alloc((size_t)nnz_out) != hipSuccess || pn.out_val.alloc((size_t)nnz_out) != hipSuccess)
    return GCN_ERR_ALLOC;
  if (gcn::panel_split(rowptr, col, val, pn.w0, p->m, R, pn.in_rowptr, pn.out_rowptr, pn.in_off, pn.in_val, pn.out_col,
                       pn.out_val, &nnz_in, st, pn.dense_slot, pn.adense) != hipSuccess)
    return GCN_ERR_HIP;
  pn.out_nnz = nnz_out;
  pn.out_T = gcn::auto_chunk_nnz(nnz_out, p->cu_count);
  pn.out_nchunks = (int)(((long long)nnz_out + pn.out_T - 1) / pn.out_T);
  if (pn.out_nchunks > 0) {
    if (pn.out_chunk_row.alloc((size_t)pn.out_nchunks) != hipSuccess) return GCN_ERR_ALLOC;
    if (gcn::launch_plan_chunk_rows(pn.out_rowptr, p->m, pn.out_T, pn.out_nchunks, pn.out_chunk_row, st) != hipSuccess)
      return GCN_ERR_HIP;
  }
  // The out-of-window part is what is LEFT of the matrix once the local structure is staged: short
  // rows with columns all over the range, i.e. an unordered graph — the case XCD column slicing is
  // for (slicing.hip).  Measured on the 240 k-vertex planted-partition graph (22 M left-over entries,
  // 92 per row): slicing them 8-ways cuts the gather time only with the one-per-gather kernel (virtual
  // rows of 11 entries: 2.49 -> 2.07 ms of kernels) and then pays 0.23 ms for the reduction — no clear
  // win, so it stays off unless GCN_AMD_PANEL_OUT_SLICES asks for it.
  static const int out_slices = gcn::env_int("GCN_AMD_PANEL_OUT_SLICES", 0);
  const int S = out_slices;
  if (S > 1 && pn.out_nchunks > 0 && (long long)S * p->m + 1 < (1LL << 31)) {
    const long long vm = (long long)S * p->m;
    if (pn.out_vrowptr.alloc((size_t)(vm + 1)) != hipSuccess || pn.out_vcol.alloc((size_t)nnz_out) != hipSuccess ||
        pn.out_vval.alloc((size_t)nnz_out) != hipSuccess || pn.out_vchunk_row.alloc((size_t)pn.out_nchunks) != hipSuccess)
      return GCN_ERR_ALLOC;
    int sorted = 1;
    if (gcn::build_sliced_csr(pn.out_rowptr, pn.out_col, pn.out_val, p->m, p->n, nnz_out, S, pn.out_vrowptr, pn.out_vcol,
                              pn.out_vval, &sorted, st) != hipSuccess)
      return GCN_ERR_HIP;
    if (sorted && gcn::launch_plan_chunk_rows(pn.out_vrowptr, (int)vm, pn.out_T, pn.out_nchunks, pn.out_vchunk_row, st) != hipSuccess)
      return GCN_ERR_HIP;
    if (sorted) pn.out_S = S;                  // (unsorted rows: the unsliced out-of-window pass stays)
  }
  if (hipStreamSynchronize(st) != hipSuccess) return GCN_ERR_HIP;
  pn.R = R;
  p->panels = std::move(pn);
  return GCN_OK;
}

int32_t gcn_spmm_plan_panel_rows(const gcn_spmm_plan_t* p) { return p ? p->panels.R : -1; }
int32_t gcn_spmm_plan_dense_panels(const gcn_spmm_plan_t* p) { return p ? p->panels.ndense : -1; }
double gcn_spmm_plan_panel_coverage(const gcn_spmm_plan_t* p) { return p ? p->panels.coverage : -1.0; }

int gcn_spmm_plan_set_tile_cols(gcn_spmm_plan_t* p, int32_t cols) {
  if (!p || !(cols == 0 || cols == 64 || cols == 128 || cols == 256)) return GCN_ERR_INVALID_ARG;
  p->tile_cols = cols;
  return GCN_OK;
}

int gcn_spmm_plan_set_gather_width(gcn_spmm_plan_t* p, int32_t nz_per_gather) {
  if (!p || (nz_per_gather != 0 && nz_per_gather != 1 && nz_per_gather != 4)) return GCN_ERR_INVALID_ARG;
  p->gather_width = nz_per_gather;
  return GCN_OK;
}

int gcn_spmm_plan_set_blocks_per_cu(gcn_spmm_plan_t* p, int32_t blocks) {
  if (!p || blocks < 1 || blocks > 64) return GCN_ERR_INVALID_ARG;
  p->blocks_per_cu = blocks;
  return GCN_OK;
}

int32_t gcn_spmm_plan_num_passes(const gcn_spmm_plan_t* p, int32_t k) {
  if (!p || k <= 0) return -1;
  if (p->panels.R > 0 && k > 32) return (k + 63) / 64;
  if (gcn::group_merge_tiles()) {                      // the group kernels take every tile in one launch
    int kk = k, ldb = gcn::padded_ldb(p->n, k);
    if (odd_width_detour(p, k)) { kk = (k + 3) / 4 * 4; ldb = (kk + 31) / 32 * 32; }
    const bool vl = valless_pays(p, kk, ldb);
    if (group_launch(p, vl, !vl && weighted_pass(p, kk, ldb))) return 1;
  }
  const int tile = p->tile_cols ? p->tile_cols : (p->slicing.S > 0 && k > 32 ? 64 : gcn::auto_tile_cols(p->n, k));
  const int vec = gcn::pick_vec(k, tile, nullptr, nullptr, nullptr);   // 16-B aligned operands
  return (k + 64 * vec - 1) / (64 * vec);
}

int gcn_spmm_plan_main_kernel(const gcn_spmm_plan_t* p, int32_t k, int32_t epilogue, char* buf, int32_t buflen) {
  if (!p || k <= 0 || !buf || buflen <= 0) return GCN_ERR_INVALID_ARG;
  if (p->panels.R > 0 && k > 32) { snprintf(buf, (size_t)buflen, "gcn::spmm_panel_in_kernel"); return GCN_OK; }
  gcn::SpmmArgs a{};
  const bool sliced = sliced_for(p, k);
  a.k = k; a.n = p->n; a.m = sliced ? p->slicing.S * p->m : p->m; a.nnz = p->nnz;
  a.nchunks_grid = p->nchunks;
  a.relu = epilogue && !sliced ? 1 : 0;               // sliced: the epilogue runs in the slice reduction
  a.tile_cols = p->tile_cols ? p->tile_cols : (sliced ? 64 : gcn::auto_tile_cols(p->n, k));
  a.gather_width = p->gather_width;
  if (odd_width_detour(p, k)) {
    a.k = (k + 3) / 4 * 4;                             // odd widths run at k rounded up to 4 (see gcn_spmm_csr_f32_epilogue)
    a.ldb = (a.k + 31) / 32 * 32;
    a.relu = 0;
    a.valless = valless_pays(p, a.k, a.ldb);
  } else {
    if (const int ldb = gcn::padded_ldb(p->n, k); ldb != k) a.ldb = ldb;
    a.valless = valless_pays(p, k, a.ldb > 0 ? a.ldb : k);   // as spmm_impl decides
  }
  a.col16 = a.valless && p->col16.ready();
  const int ld_eff = a.ldb > 0 ? a.ldb : a.k;
  const bool big = gcn::spmm_group_needs_big(group_table_rows(p), ld_eff);
  const char* bigs = big ? "true" : "false";
  if (a.valless && group_pass(p)) {
    gcn::GroupArgs probe{};
    probe.k = a.k; probe.ldb = ld_eff; probe.table_rows = group_table_rows(p); probe.ring = gcn::group_ring() ? 1 : 0;
    const int nch8 = (a.k <= 32 && p->group_alt[0].ready()) ? p->group_alt[0].nchunks : p->group.nchunks;
    if (gcn::group8_enabled() && a.k <= 32 && nch8 % 64 == 0)
      snprintf(buf, (size_t)buflen, "gcn::spmm_group8_kernel<%s, %s>", gcn::group_ring() ? "true" : "false", bigs);
    else if (gcn::spmm_group12_applies(probe))
      snprintf(buf, (size_t)buflen, "gcn::spmm_group12_kernel");
    else
      snprintf(buf, (size_t)buflen, "gcn::spmm_group%s_kernel<%d, %s>", gcn::group_ring() ? "_ring" : "", big ? 2 : gcn::group_store(), bigs);
    return GCN_OK;
  }
  if (!a.valless && weighted_pass(p, a.k, ld_eff)) {
    if (gcn::group8_enabled() && a.k <= 32 && p->group.nchunks % 64 == 0)
      snprintf(buf, (size_t)buflen, "gcn::spmm_group8_weighted_kernel<%s>", bigs);
    else
      snprintf(buf, (size_t)buflen, "gcn::spmm_group_weighted_kernel<%d, %s>", big ? 2 : gcn::group_store(), bigs);
    return GCN_OK;
  }
  gcn::describe_main_kernel(a, buf, (size_t)buflen);
  return GCN_OK;
}

int gcn_spmm_profile_begin(gcn_spmm_plan_t* p, int32_t capacity) {
  if (!p || capacity <= 0 || p->prof.capacity() > 0) return GCN_ERR_INVALID_ARG;
  return p->prof.begin(capacity) == hipSuccess ? GCN_OK : GCN_ERR_HIP;   // (a failed create leaves no events behind)
}

int gcn_spmm_profile_end(gcn_spmm_plan_t* p, float* ms_out, int32_t* count_out) {
  if (!p || !count_out || p->prof.capacity() <= 0) return GCN_ERR_INVALID_ARG;
  int rc = GCN_OK;
  for (int i = 0; i < p->prof.recorded(); ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(p->prof.stop(i)) != hipSuccess ||
        hipEventElapsedTime(&ms, p->prof.start(i), p->prof.stop(i)) != hipSuccess) rc = GCN_ERR_HIP;
    if (ms_out) ms_out[i] = ms;
  }
  *count_out = p->prof.recorded();
  p->prof.clear();
  return rc;
}

// One-shot: the schedule is recomputed on the device every call (a few µs: one binary search per
// chunk) into the scratch plan of this (device, stream), so there is no cache that could go stale when
// the caller reuses device addresses, and calls on different streams or devices never share buffers.
int gcn_spmm_csr_f32_oneshot(const int32_t* rowptr, const int32_t* col, const float* val,
                             const float* B, float* C, int32_t m, int32_t n, int32_t nnz,
                             int32_t k, void* stream) {
  if (m < 0 || n < 0 || nnz < 0 || k < 0) return GCN_ERR_INVALID_ARG;
  const int cu = gcn::cu_count_cached();
  if (cu <= 0) return GCN_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(g_plan_mu);
  gcn_spmm_plan* p = gcn::scratch_plan(stream);
  if (!p) return GCN_ERR_ALLOC;
  p->m = m; p->n = n; p->nnz = nnz; p->cu_count = cu;
  p->T = gcn::auto_chunk_nnz(nnz, cu);
  p->nchunks = (int)(((long long)nnz + p->T - 1) / p->T);
  if (p->chunk_row.grow((size_t)p->nchunks) != hipSuccess) return GCN_ERR_ALLOC;
  if (m == 0 || k == 0) return GCN_OK;
  if (gcn::launch_plan_chunk_rows(rowptr, m, p->T, p->nchunks, p->chunk_row, (hipStream_t)stream) != hipSuccess) return GCN_ERR_HIP;
  if (p->ws.grow(ws_elems(p, k)) != hipSuccess) return GCN_ERR_ALLOC;
  gcn::SpmmArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val; a.B = B; a.C = C; a.P = p->ws;
  a.chunk_row = p->chunk_row; a.bias = nullptr; a.relu = 0;
  a.nchunks = p->nchunks; a.T = p->T; a.m = m; a.nnz = nnz; a.k = k; a.n = n;
  a.nnz_dev = nullptr; a.nchunks_grid = p->nchunks;
  a.tile_cols = gcn::auto_tile_cols(n, k);
  return gcn::launch_spmm(a, cu, (hipStream_t)stream) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

int32_t gcn_spmm_auto_slices(int64_t m, int64_t n, int64_t nnz, int32_t value_free) {
  return gcn::auto_slices(m, n, nnz, value_free != 0);
}

int32_t gcn_spmm_group_addressing(int64_t table_rows, int32_t ld_floats) {
  if (table_rows <= 0 || ld_floats <= 0) return -1;
  if (!gcn::spmm_group_eligible(ld_floats, ld_floats, table_rows, nullptr, nullptr, nullptr)) return -1;
  return gcn::spmm_group_needs_big(table_rows, ld_floats) ? 1 : 0;
}

int gcn_gather_rows_f32(float* dst, const float* src, const int32_t* idx, int32_t nrows, int32_t k,
                        void* stream) {
  if (nrows < 0 || k < 0) return GCN_ERR_INVALID_ARG;
  if (nrows == 0 || k == 0) return GCN_OK;
  if (!dst || !src || !idx || dst == src) return GCN_ERR_INVALID_ARG;
  return gcn::launch_gather_rows(dst, src, idx, nrows, k, (hipStream_t)stream) == hipSuccess
             ? GCN_OK : GCN_ERR_HIP;
}

}  // extern "C"
