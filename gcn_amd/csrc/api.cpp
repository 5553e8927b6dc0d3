// api.cpp — the C ABI of libgcnspmm.so (see include/gcn_spmm.h for the contract and
// the reference interfaces each entry point replaces).
#include "../../include/gcn_spmm.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "reorder.h"
#include "spmm_kernels.h"

#define GCN_VERSION_STR "0.1.0"

struct gcn_spmm_plan {
  int32_t m, n, nnz, T, nchunks;
  int* chunk_row;     // device [nchunks]
  float* ws;          // device partial slab, grow-only
  size_t ws_bytes;
  int cu_count;
  int device;
  // live kernel timing (gcn_spmm_profile_begin/_end)
  std::vector<hipEvent_t> ev;   // 2 per recorded launch
  int prof_cap, prof_n;
  int tile_cols;                // 0 = auto
  int gather_width;             // non-zeros per gather instruction of the 64-column kernel: 0 auto, 1, 4
  int blocks_per_cu;            // grid size: blocks of 4 waves per CU (default 32: oversubscribed, see header)
  // XCD-aware column slicing (slicing.hip): slice-major copy of the matrix with S*m virtual rows
  int S;                        // 0 = off
  int* vrowptr;                 // [S*m+1]
  int* vcol;                    // [nnz]
  float* vval;                  // [nnz]
  int* vchunk_row;              // [nchunks] rows of the virtual CSR
  float* cv;                    // partial outputs [S*m x k], grow-only
  size_t cv_bytes;
  int *vrowptr16, *vchunk_row16;  // 16-bit column stream of the sliced CSR (value-free pass): [S*m+1], [nchunks16]
  unsigned short* vcol16;       // [nnz16] offsets inside the slice, 0xFFFF = padding marker
  int nnz16, nchunks16, start16[9];
  // 15-bit slice-major stream of the group kernel (spmm_group.hip), value-free pass
  unsigned short* gstream;      // [gnchunks*gT]
  int *gchunk_row, *gvrowptr;   // [gnchunks], [S*m+1]
  int* gchunk_meta;             // int2 [gnchunks]
  int gnchunks, gT, gw;
  float* u_row;                 // [m], [n]: factors of rank-1 values (val[r,c] = u_row[r]*u_col[c]); null when
  float* u_col;                 // the values do not factor (u_col == u_row for a square normalised adjacency)
  float* bpad;                  // B re-laid with rows padded to whole 128-byte lines (odd k), grow-only
  size_t bpad_bytes;
  float* cpad;                  // result with k rounded up to a multiple of 4 (k % 4 != 0), grow-only
  size_t cpad_bytes;
  // LDS-staged row panels (spmm_panel.hip): rows per panel, 0 = off; measured window coverage
  int panel_R;
  int* panel_w0;                // device [ceil(m / panel_R)]: first column of each panel's window
  double panel_coverage;
  // A = A_in + A_out: staged entries (LDS byte offsets) / the rest (plain CSR + its chunk plan)
  int *pin_rowptr, *pin_off;    // [m+1], [nnz_in]
  float* pin_val;
  int *pout_rowptr, *pout_col;  // [m+1], [nnz_out]
  float* pout_val;
  int* pout_chunk_row;
  int pout_nnz, pout_T, pout_nchunks;
  int pout_S;                   // column slices of the out-of-window part (0: unsliced)
  int *pout_vrowptr, *pout_vcol, *pout_vchunk_row;
  float* pout_vval;
};

namespace {

std::mutex g_mu;

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

// value-free sliced main pass for matrices whose values factor as u[r]*u[c] (development knob GCN_AMD_VALLESS=0: off)
static bool valless_enabled() {
  static const bool v = [] { const char* e = std::getenv("GCN_AMD_VALLESS"); return !e || e[0] != '0'; }();
  return v;
}

// 16-bit column stream in the value-free pass (development knob GCN_AMD_COL16=0: off)
static bool col16_enabled() {
  static const bool v = [] { const char* e = std::getenv("GCN_AMD_COL16"); return !e || e[0] != '0'; }();
  return v;
}

// the group kernel (spmm_group.hip) for the value-free sliced pass (development knobs: GCN_AMD_GROUP=0 off,
// GCN_AMD_GROUP_T chunk size per 16-lane group, GCN_AMD_GROUP_SC1=0 plain instead of write-through partial-row stores)
static bool group_enabled() {
  static const bool v = [] { const char* e = std::getenv("GCN_AMD_GROUP"); return !e || e[0] != '0'; }();
  return v;
}
static int group_chunk() {
  static const int v = [] { const char* e = std::getenv("GCN_AMD_GROUP_T"); const int t = e ? std::atoi(e) : 512;
                            return (t == 256 || t == 512 || t == 1024 || t == 2048) ? t : 512; }();
  return v;
}
static bool group_sc1() {
  static const bool v = [] { const char* e = std::getenv("GCN_AMD_GROUP_SC1"); return !e || e[0] != '0'; }();
  return v;
}

// re-lay B with rows padded to whole cache lines for k % 32 != 0 (development knob GCN_AMD_PAD_B=0: off)
static bool pad_b_enabled() {
  static const bool v = [] { const char* e = std::getenv("GCN_AMD_PAD_B"); return !e || e[0] != '0'; }();
  return v;
}

// Expected 128-byte cache lines one gathered feature row costs, summed over its 64-column tiles, when B's
// rows are `ld` floats apart (the row start offsets cycle through the multiples of gcd(4*ld, 128)).
static double lines_per_row(int k, int ld) {
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

// Row stride (floats) B is gathered with: k itself, or k rounded up to whole 128-byte lines when that
// saves >= 15 % of the cache lines per gathered row and the re-laid table stays <= 768 MiB.  Measured
// (profiles/r01f_sweep_padded_feature_rows.log, whole SpMM, unpadded -> padded): Reddit-shaped k = 20:
// 2.19 -> 1.60 ms, 24: 2.26 -> 1.60, 47: 2.11 -> 2.00, 100: 4.43 -> 3.84, 172: 7.56 -> 5.73; no saving
// by the model and none measured for k = 40, 48 (rows of 160 / 192 B never straddle more lines than
// padded ones); products-shaped k = 47 (627 MB padded): 5.41 -> 4.86 ms, k = 100 (1.25 GB): 8.95 ->
// 9.73 ms — past the Infinity Cache the larger table and the copy cost more than the lines save.
static int padded_ldb(long long n, int k) {
  if (k <= 16 || k % 32 == 0 || !pad_b_enabled()) return k;
  const int ld = (k + 31) / 32 * 32;
  if ((long long)sizeof(float) * n * ld > (768LL << 20)) return k;
  return lines_per_row(k, k) >= 1.15 * lines_per_row(k, ld) ? ld : k;
}

// smallest k the sliced path is used for (development knob GCN_AMD_SLICE_MIN_K)
static int slice_min_k() {
  static const int v = [] { const char* e = std::getenv("GCN_AMD_SLICE_MIN_K"); return e ? std::atoi(e) : 33; }();
  return v;
}

// the four-per-gather kernel only pays from ~48 non-zeros per (virtual) row up (use_quad in
// spmm_kernels.hip); the odd-width path exists to reach that kernel, so it follows the same rule
static bool rows_long_enough_for_quad(const gcn_spmm_plan* p, int k) {
  if (p->gather_width == 4) return true;
  const bool sliced = p->S > 0 && k >= slice_min_k();
  // (the group kernel of the value-free pass does not mind short rows)
  if (sliced && p->gstream && p->u_row && p->panel_R == 0 && p->nnz / p->n >= 96) return true;
  const long long rows = sliced ? (long long)p->S * p->m : (long long)p->m;
  return rows > 0 && p->nnz / rows >= 48;
}

// Number of column slices for the XCD-aware slicing (slicing.hip), 0 = do not slice.
// Measured on MI355X with the r01f kernels (profiles/r01f_sweep_slices_scales.log; Reddit-shaped graphs
// of 14.5 k .. 1.86 M vertices, mean degree 493; whole SpMM, k = 128, best S in brackets):
//   n = 14.5 k (64-column table 3.7 MB): slicing buys nothing;  29 k (7.5 MB): [2] 0.352 vs 0.394 ms
//   unsliced;  58 k: [4] 0.84 vs 1.18;  116 k: [4/8] 1.76-1.80 vs 3.13;  233 k: [8] 3.62 vs 7.3;
//   466 k: [8] 9.20 vs 15.7 (16: 9.84);  932 k: [8] 24.4 vs 32.3;  1.86 M: [8] 56.5 vs 62.5.
// So: as many slices as bring one slice of the table (n/S x 256 B) down to the 4 MiB of an XCD's L2,
// but never more than the 8 XCDs — beyond 8 every XCD walks several slices and the extra partial rows
// (S*m*k floats written and re-read) cost more than the higher hit rate returns.  Needs >= 16
// non-zeros per virtual row; at mean degree 51 (products-shaped) slicing loses and stays off.
int auto_slices(long long m, long long n, long long nnz) {
  if (m <= 0 || nnz <= 0) return 0;
  static const int forced = [] { const char* e = std::getenv("GCN_AMD_SLICES"); return e ? std::atoi(e) : -1; }();
  if (forced >= 0) return forced;                     // development knob: the slice count "auto" resolves to
  if (nnz / m < 128) return 0;                        // low degree: partial rows outweigh the hits
  const long long table = n * 256;                    // bytes of one 64-column tile of B
  if (table <= (4LL << 20)) return 0;                 // fits every L2 as it is
  int S = 2;
  while (S < 8 && table / S > (4LL << 20)) S *= 2;
  while (S > 1 && nnz / m / S < 16) S /= 2;           // keep >= 16 non-zeros per virtual row
  if (S < 2) return 0;
  // slices far larger than any cache (huge n): the partial rows cost traffic and buy no hits
  if (table / S > (64LL << 20)) return 0;
  return S;
}

int ensure_ws(gcn_spmm_plan* p, int k) {
  const size_t need = gcn_spmm_plan_workspace_bytes(p, k);
  if (need <= p->ws_bytes) return GCN_OK;
  if (p->ws) { (void)hipFree(p->ws); p->ws = nullptr; p->ws_bytes = 0; }
  if (hipMalloc((void**)&p->ws, need) != hipSuccess) return GCN_ERR_ALLOC;
  p->ws_bytes = need;
  return GCN_OK;
}

void die(const char* what, hipError_t e) {
  std::fprintf(stderr, "libgcnspmm: %s failed: %s\n", what, hipGetErrorString(e));
  std::abort();
}

bool verbose() {
  const char* v = std::getenv("GCN_AMD_VERBOSE");
  return v && v[0] && v[0] != '0';
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
    default: return "unknown status";
  }
}

const char* gcn_version(void) { return GCN_VERSION_STR; }

int gcn_device_cu_count(void) { return cu_count_cached(); }

// ---------------------------------------------------------------------------
int gcn_spmm_plan_create(gcn_spmm_plan_t** out, const int32_t* rowptr_dev, int32_t m, int32_t n,
                         int32_t nnz, int32_t chunk_nnz, void* stream) {
  if (!out || m < 0 || n < 0 || nnz < 0 || (m > 0 && !rowptr_dev)) return GCN_ERR_INVALID_ARG;
  if (chunk_nnz < 0 || (chunk_nnz % 64) != 0) return GCN_ERR_INVALID_ARG;
  const int cu = cu_count_cached();
  if (cu <= 0) return GCN_ERR_NO_DEVICE;
  gcn_spmm_plan* p = new (std::nothrow) gcn_spmm_plan();
  if (!p) return GCN_ERR_ALLOC;
  p->m = m; p->n = n; p->nnz = nnz;
  p->T = chunk_nnz ? chunk_nnz : auto_chunk_nnz(nnz, cu);
  p->nchunks = (int)(((long long)nnz + p->T - 1) / p->T);
  p->chunk_row = nullptr; p->ws = nullptr; p->ws_bytes = 0; p->cu_count = cu;
  p->prof_cap = p->prof_n = 0;
  p->tile_cols = 0;
  p->u_row = p->u_col = nullptr;
  p->vrowptr16 = p->vchunk_row16 = nullptr; p->vcol16 = nullptr; p->nnz16 = p->nchunks16 = 0;
  p->bpad = nullptr; p->bpad_bytes = 0;
  p->cpad = nullptr; p->cpad_bytes = 0;
  p->blocks_per_cu = 32;
  p->gather_width = 0;
  (void)hipGetDevice(&p->device);
  if (p->nchunks > 0) {
    if (hipMalloc((void**)&p->chunk_row, sizeof(int) * (size_t)p->nchunks) != hipSuccess) {
      delete p; return GCN_ERR_ALLOC;
    }
    if (gcn::launch_plan_chunk_rows(rowptr_dev, m, p->T, p->nchunks, p->chunk_row,
                                    (hipStream_t)stream) != hipSuccess) {
      (void)hipFree(p->chunk_row); delete p; return GCN_ERR_HIP;
    }
  }
  *out = p;
  return GCN_OK;
}

static void free_factors(gcn_spmm_plan* p) {
  if (p->u_col && p->u_col != p->u_row) (void)hipFree(p->u_col);
  if (p->u_row) (void)hipFree(p->u_row);
  p->u_row = p->u_col = nullptr;
}

int gcn_spmm_plan_destroy(gcn_spmm_plan_t* p) {
  if (!p) return GCN_OK;
  if (p->chunk_row) (void)hipFree(p->chunk_row);
  if (p->ws) (void)hipFree(p->ws);
  if (p->vrowptr) (void)hipFree(p->vrowptr);
  if (p->vcol) (void)hipFree(p->vcol);
  if (p->vval) (void)hipFree(p->vval);
  if (p->vchunk_row) (void)hipFree(p->vchunk_row);
  if (p->cv) (void)hipFree(p->cv);
  if (p->vrowptr16) (void)hipFree(p->vrowptr16);
  if (p->vchunk_row16) (void)hipFree(p->vchunk_row16);
  if (p->vcol16) (void)hipFree(p->vcol16);
  if (p->gstream) (void)hipFree(p->gstream);
  if (p->gchunk_row) (void)hipFree(p->gchunk_row);
  if (p->gchunk_meta) (void)hipFree(p->gchunk_meta);
  if (p->gvrowptr) (void)hipFree(p->gvrowptr);
  free_factors(p);
  if (p->bpad) (void)hipFree(p->bpad);
  if (p->cpad) (void)hipFree(p->cpad);
  {
    void* ptrs[] = {p->panel_w0, p->pin_rowptr, p->pin_off, p->pin_val, p->pout_rowptr, p->pout_col,
                    p->pout_val, p->pout_chunk_row, p->pout_vrowptr, p->pout_vcol, p->pout_vchunk_row, p->pout_vval};
    for (void* q : ptrs) if (q) (void)hipFree(q);
  }
  for (auto& e : p->ev) (void)hipEventDestroy(e);
  delete p;
  return GCN_OK;
}

int32_t gcn_spmm_plan_num_chunks(const gcn_spmm_plan_t* p) { return p ? p->nchunks : -1; }
int32_t gcn_spmm_plan_chunk_nnz(const gcn_spmm_plan_t* p) { return p ? p->T : -1; }
size_t gcn_spmm_plan_workspace_bytes(const gcn_spmm_plan_t* p, int32_t k) {
  if (!p || k <= 0) return 0;
  int chunks = p->nchunks > p->pout_nchunks ? p->nchunks : p->pout_nchunks;
  if (p->nchunks16 > chunks) chunks = p->nchunks16;
  if (p->gnchunks > chunks) chunks = p->gnchunks;
  return sizeof(float) * 2 * (size_t)(chunks > 0 ? chunks : 1) * (size_t)k;
}

static int grow(float*& buf, size_t& have, size_t need) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (need <= have) return GCN_OK;
  if (buf) (void)hipFree(buf);
  buf = nullptr; have = 0;
  if (hipMalloc((void**)&buf, need) != hipSuccess) return GCN_ERR_ALLOC;
  have = need;
  return GCN_OK;
}

// b_ld: row stride of B in floats when the caller of this function has already re-laid it, 0 = k;
// b_scaled: that copy's rows are already scaled by u_col (value-free pass)
static int spmm_impl(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col, const float* val,
                     const float* B, int b_ld, bool b_scaled, float* C, const float* bias, int32_t relu, int32_t k,
                     void* stream);
static bool valless_pays(const gcn_spmm_plan_t* p, const gcn::SpmmArgs& base, int k, int ldb);
static int relay_B(gcn_spmm_plan_t* p, const float* B, int k, int ldb, bool scaled, hipStream_t st);

int gcn_spmm_csr_f32_bias_relu(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                               const float* val, const float* B, float* C, const float* bias,
                               int32_t relu, int32_t k, void* stream) {
  if (!p || k < 0) return GCN_ERR_INVALID_ARG;
  if (p->m == 0 || k == 0) return GCN_OK;
  if (!C || !rowptr || (p->nnz > 0 && (!col || !val || !B))) return GCN_ERR_INVALID_ARG;
  // Widths that are not a multiple of 4 (class counts: 41, 47, ...) cannot use the 16-byte-per-lane
  // kernel on the caller's layout.  They are computed at k' = k rounded up to 4 on row-padded copies:
  // B re-laid with zero columns (stride a multiple of 32 floats), the product into a k'-wide scratch
  // result, and one pass that compacts it into C (and applies the epilogue).  Reddit-shaped k = 41:
  // 2.13 -> 1.87 ms.  Same limits as the B padding above (tables <= 768 MiB), panels excluded.
  if (k > 16 && k % 4 != 0 && p->nnz > 0 && p->panel_R == 0 && p->gather_width != 1 && pad_b_enabled() &&
      rows_long_enough_for_quad(p, k)) {
    const int kp = (k + 3) / 4 * 4, ldb = (kp + 31) / 32 * 32;
    if ((long long)sizeof(float) * p->n * ldb <= (768LL << 20)) {
      int st = grow(p->cpad, p->cpad_bytes, sizeof(float) * (size_t)p->m * (size_t)kp);
      if (st != GCN_OK) return st;
      const bool scaled = valless_pays(p, gcn::SpmmArgs{}, kp, ldb);       // the copy can carry the u_col scaling
      st = relay_B(p, B, k, ldb, scaled, (hipStream_t)stream);
      if (st != GCN_OK) return st;
      st = spmm_impl(p, rowptr, col, val, p->bpad, ldb, scaled, p->cpad, nullptr, 0, kp, stream);
      if (st != GCN_OK) return st;
      return gcn::launch_unpad_rows(C, p->cpad, bias, relu ? 1 : 0, p->m, k, kp, (hipStream_t)stream) == hipSuccess
                 ? GCN_OK : GCN_ERR_HIP;
    }
  }
  return spmm_impl(p, rowptr, col, val, B, 0, false, C, bias, relu, k, stream);
}

// would the sliced launch of a k-wide SpMM run the value-free quad kernel (and is the scaled copy worth it)?
static bool valless_pays(const gcn_spmm_plan_t* p, const gcn::SpmmArgs& base, int k, int ldb) {
  const bool sliced = p->S > 0 && p->nnz > 0 && k >= slice_min_k();
  // (the scaled copy of B costs 2*n*k*4 bytes of traffic whatever the matrix; the value stream it saves is
  //  4 bytes per non-zero plus instructions — measured break-even near 65 non-zeros per column of the
  //  block: the rank-0 share of an 8-way partition of the Reddit-shaped graph (62 per column) does not gain)
  if (!sliced || !p->u_row || p->panel_R != 0 || p->nnz / p->n < 96) return false;
  if (p->gstream && gcn::spmm_group_eligible(k, ldb, nullptr, nullptr, nullptr)) return true;   // spmm_group.hip
  gcn::SpmmArgs t = base;                              // the launch as the sliced branch will issue it
  t.B = nullptr; t.C = nullptr; t.bias = nullptr; t.relu = 0; t.k = k; t.nnz = p->nnz; t.n = p->n;
  t.nchunks_grid = p->nchunks; t.T = p->T; t.nnz_dev = nullptr;
  t.m = p->S * p->m; t.ldb = ldb; t.tile_cols = p->tile_cols ? p->tile_cols : 64;
  t.gather_width = p->gather_width;
  return gcn::spmm_will_use_quad(t) && gcn::spmm_quad_lanes(k) == 16;
}

// the value-free pass of this plan runs the group kernel (its scaled copy of B is then laid out slice by slice)
static bool group_pass(const gcn_spmm_plan_t* p) { return p->gstream != nullptr; }

// Copy of B the sliced main pass gathers from: rows `ldb` floats apart (>= k, padding columns zero), scaled by
// u_col when `scaled`; one all-zero row more than B has (16-bit stream) or, for the group kernel, slice s at
// rows [s*(w+1), (s+1)*(w+1)) with row w of every slice zero.
static int relay_B(gcn_spmm_plan_t* p, const float* B, int k, int ldb, bool scaled, hipStream_t st) {
  if (scaled && group_pass(p)) {
    const size_t rows = (size_t)p->S * (size_t)(p->gw + 1);
    const int rc = grow(p->bpad, p->bpad_bytes, sizeof(float) * rows * (size_t)ldb);
    if (rc != GCN_OK) return rc;
    return gcn::launch_scale_rows_sliced(p->bpad, B, p->u_col, p->n, k, ldb, p->S, p->gw, st) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
  }
  const int rc = grow(p->bpad, p->bpad_bytes, sizeof(float) * ((size_t)p->n + 1) * (size_t)ldb);
  if (rc != GCN_OK) return rc;
  if (gcn::launch_pad_rows(p->bpad, B, p->n, k, ldb, st, scaled ? p->u_col : nullptr) != hipSuccess ||
      hipMemsetAsync(p->bpad + (size_t)p->n * ldb, 0, sizeof(float) * (size_t)ldb, st) != hipSuccess)
    return GCN_ERR_HIP;
  return GCN_OK;
}

static int spmm_impl(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col, const float* val,
                     const float* B, int b_ld, bool b_scaled, float* C, const float* bias, int32_t relu, int32_t k,
                     void* stream) {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    const int st = ensure_ws(p, k);
    if (st != GCN_OK) return st;
  }
  gcn::SpmmArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val; a.B = B; a.C = C; a.P = p->ws;
  a.chunk_row = p->chunk_row; a.bias = bias; a.relu = relu ? 1 : 0;
  a.nchunks = p->nchunks; a.T = p->T; a.m = p->m; a.nnz = p->nnz; a.k = k; a.n = p->n;
  a.nnz_dev = nullptr; a.nchunks_grid = p->nchunks;
  // narrow feature widths (k <= 32, the GCN hidden/class sizes) gather 128 B or less per
  // non-zero: there the extra partial rows cost more than the L2 hits buy (measured 2.12 vs
  // 2.02 ms at k = 32), so the sliced copy is used for k > 32 only
  const bool sliced = p->S > 0 && p->nnz > 0 && k >= slice_min_k();
  // Feature rows that are not a whole number of 128-byte cache lines straddle lines: a gathered row
  // then costs up to one extra L2 request per tile.  Where that matters (padded_ldb) B is first re-laid
  // with its rows padded to the next multiple of 32 floats (one streaming copy, ~45 us for 233 k x 100)
  // and gathered from there; C keeps the caller's layout.  The same copy carries the row scaling of the
  // value-free pass (values u[r]*u[c], sliced matrix, four-per-gather kernel): B' = diag(u) B.
  bool valless = false;
  if (b_ld > 0) {
    a.ldb = b_ld;                                      // already re-laid (and maybe scaled) by the caller (odd-width path)
    valless = b_scaled;
  } else if (p->nnz > 0) {
    const int ldb = padded_ldb(p->n, k);
    valless = valless_pays(p, a, k, ldb);
    if (ldb != k || valless) {
      const int st = relay_B(p, B, k, ldb, valless, (hipStream_t)stream);
      if (st != GCN_OK) return st;
      a.B = p->bpad;
      a.ldb = ldb;
    }
  }
  if (p->panel_R > 0 && p->nnz > 0 && k > 32) {
    // A = A_in + A_out: the staged part from LDS (raw sums into C), then the rest accumulated by the
    // chunk kernel, which also carries the epilogue
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (p->prof_cap > 0 && p->prof_n < p->prof_cap) {
      ev0 = p->ev[2 * p->prof_n];
      ev1 = p->ev[2 * p->prof_n + 1];
      ++p->prof_n;
    }
    if (ev0 && hipEventRecord(ev0, st) != hipSuccess) return GCN_ERR_HIP;
    const int tiles = (k + 63) / 64;
    for (int t = 0; t < tiles; ++t)
      if (gcn::launch_panel_in(p->pin_rowptr, p->pin_off, p->pin_val, B, C, p->panel_w0, p->m, p->n, k,
                               p->panel_R, t, st) != hipSuccess) return GCN_ERR_HIP;
    if (p->pout_nnz == 0) {
      if (gcn::launch_panel_epilogue(C, bias, relu ? 1 : 0, p->m, k, st) != hipSuccess) return GCN_ERR_HIP;
      if (ev1 && hipEventRecord(ev1, st) != hipSuccess) return GCN_ERR_HIP;
      return GCN_OK;
    }
    a.nchunks = p->pout_nchunks; a.nchunks_grid = p->pout_nchunks;
    a.T = p->pout_T; a.nnz = p->pout_nnz;
    a.blocks_per_cu = p->blocks_per_cu;
    a.gather_width = p->gather_width;
    a.ev_start = nullptr; a.ev_stop = ev1;
    if (p->pout_S > 0) {
      // sliced: partial rows of the virtual CSR, then C += sum of the partials (+ epilogue)
      const int st2 = grow(p->cv, p->cv_bytes, sizeof(float) * (size_t)p->pout_S * (size_t)p->m * (size_t)k);
      if (st2 != GCN_OK) return st2;
      a.rowptr = p->pout_vrowptr; a.col = p->pout_vcol; a.val = p->pout_vval; a.chunk_row = p->pout_vchunk_row;
      a.C = p->cv; a.m = p->pout_S * p->m; a.bias = nullptr; a.relu = 0; a.accumulate = 0;
      a.tile_cols = p->tile_cols ? p->tile_cols : 64;
      if (gcn::launch_spmm(a, p->cu_count, st) != hipSuccess) return GCN_ERR_HIP;
      return gcn::launch_slice_reduce(p->cv, C, bias, relu ? 1 : 0, p->m, p->pout_S, k, st, 1) == hipSuccess
                 ? GCN_OK : GCN_ERR_HIP;
    }
    a.rowptr = p->pout_rowptr; a.col = p->pout_col; a.val = p->pout_val;
    a.chunk_row = p->pout_chunk_row; a.accumulate = 1;
    a.tile_cols = p->tile_cols ? p->tile_cols : auto_tile_cols(p->n, k);
    return gcn::launch_spmm(a, p->cu_count, st) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
  }
  a.tile_cols = p->tile_cols ? p->tile_cols : (sliced ? 64 : auto_tile_cols(p->n, k));
  a.blocks_per_cu = p->blocks_per_cu;
  a.gather_width = p->gather_width;
  if (p->prof_cap > 0 && p->prof_n < p->prof_cap) {
    a.ev_start = p->ev[2 * p->prof_n];
    a.ev_stop = p->ev[2 * p->prof_n + 1];
    ++p->prof_n;
  }
  if (sliced) {
    // sliced: same kernel on the slice-major virtual CSR (S*m rows) into the partial buffer,
    // then the per-row reduction over slices (which also carries the epilogue)
    const size_t need = sizeof(float) * (size_t)p->S * (size_t)p->m * (size_t)k;
    {
      std::lock_guard<std::mutex> lk(g_mu);
      if (need > p->cv_bytes) {
        if (p->cv) (void)hipFree(p->cv);
        p->cv = nullptr; p->cv_bytes = 0;
        if (hipMalloc((void**)&p->cv, need) != hipSuccess) return GCN_ERR_ALLOC;
        p->cv_bytes = need;
      }
    }
    a.rowptr = p->vrowptr; a.col = p->vcol; a.val = p->vval; a.chunk_row = p->vchunk_row;
    a.C = p->cv; a.m = p->S * p->m; a.bias = nullptr; a.relu = 0;
    const float* rowscale = nullptr;
    if (valless && group_pass(p)) {
      // four independent 16-lane row engines per wave on the 15-bit slice-major stream (spmm_group.hip)
      gcn::GroupArgs ga;
      ga.stream = p->gstream; ga.chunk_meta = p->gchunk_meta;
      ga.Bp = a.B; ga.Cv = p->cv; ga.P = p->ws;
      ga.nchunks = p->gnchunks; ga.T = p->gT; ga.k = k; ga.ldb = a.ldb;
      ga.write_through = group_sc1() ? 1 : 0;
      hipStream_t st = (hipStream_t)stream;
      if (a.ev_start && hipEventRecord(a.ev_start, st) != hipSuccess) return GCN_ERR_HIP;
      if (gcn::launch_spmm_group(ga, st) != hipSuccess) return GCN_ERR_HIP;
      if (a.ev_stop && hipEventRecord(a.ev_stop, st) != hipSuccess) return GCN_ERR_HIP;
      if (gcn::launch_spmm_fixup(p->gvrowptr, p->ws, p->cv, p->gchunk_row, p->gnchunks, p->gT, k, st) != hipSuccess)
        return GCN_ERR_HIP;
      return gcn::launch_slice_reduce(p->cv, C, bias, relu ? 1 : 0, p->m, p->S, k, st, 0, p->u_row) == hipSuccess
                 ? GCN_OK : GCN_ERR_HIP;
    }
    if (valless) {                                                          // B was scaled by u_col above
      a.valless = 1; a.val = nullptr; rowscale = p->u_row;
      if (p->vcol16) {                                                      // 16-bit column stream, slice-aligned chunks
        a.rowptr = p->vrowptr16; a.col = reinterpret_cast<const int*>(p->vcol16); a.chunk_row = p->vchunk_row16;
        a.nnz = p->nnz16; a.nchunks = a.nchunks_grid = p->nchunks16;
        a.col16 = 1; a.col16_S = p->S; a.col16_w = (p->n + p->S - 1) / p->S;
        for (int i = 0; i < 9; ++i) a.col16_start[i] = p->start16[i];
      }
    }
    if (gcn::launch_spmm(a, p->cu_count, (hipStream_t)stream) != hipSuccess) return GCN_ERR_HIP;
    return gcn::launch_slice_reduce(p->cv, C, bias, relu ? 1 : 0, p->m, p->S, k,
                                    (hipStream_t)stream, 0, rowscale) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
  }
  return gcn::launch_spmm(a, p->cu_count, (hipStream_t)stream) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

static void free_slicing(gcn_spmm_plan* p) {
  if (p->vrowptr) (void)hipFree(p->vrowptr);
  if (p->vcol) (void)hipFree(p->vcol);
  if (p->vval) (void)hipFree(p->vval);
  if (p->vchunk_row) (void)hipFree(p->vchunk_row);
  if (p->cv) (void)hipFree(p->cv);
  if (p->vrowptr16) (void)hipFree(p->vrowptr16);
  if (p->vchunk_row16) (void)hipFree(p->vchunk_row16);
  if (p->vcol16) (void)hipFree(p->vcol16);
  if (p->gstream) (void)hipFree(p->gstream);
  if (p->gchunk_row) (void)hipFree(p->gchunk_row);
  if (p->gchunk_meta) (void)hipFree(p->gchunk_meta);
  if (p->gvrowptr) (void)hipFree(p->gvrowptr);
  p->gstream = nullptr; p->gchunk_row = p->gvrowptr = p->gchunk_meta = nullptr; p->gnchunks = p->gT = p->gw = 0;
  p->vrowptr16 = p->vchunk_row16 = nullptr; p->vcol16 = nullptr; p->nnz16 = p->nchunks16 = 0;
  p->vrowptr = p->vcol = p->vchunk_row = nullptr; p->vval = nullptr; p->cv = nullptr;
  p->cv_bytes = 0; p->S = 0;
}

int gcn_spmm_plan_enable_slicing(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                                 const float* val, int32_t slices, void* stream) {
  if (!p || slices < -1 || slices > 1024) return GCN_ERR_INVALID_ARG;
  free_slicing(p);
  const bool autom = slices == -1;
  if (autom) slices = auto_slices(p->m, p->n, p->nnz);
  if (slices <= 1 || p->nnz == 0 || p->m == 0) return GCN_OK;
  if (!rowptr || !col || !val) return GCN_ERR_INVALID_ARG;
  if ((long long)slices * p->m + 1 >= (1LL << 31)) return GCN_ERR_INVALID_ARG;
  const long long vm = (long long)slices * p->m;
  if (hipMalloc((void**)&p->vrowptr, sizeof(int) * (size_t)(vm + 1)) != hipSuccess ||
      hipMalloc((void**)&p->vcol, sizeof(int) * (size_t)p->nnz) != hipSuccess ||
      hipMalloc((void**)&p->vval, sizeof(float) * (size_t)p->nnz) != hipSuccess ||
      hipMalloc((void**)&p->vchunk_row, sizeof(int) * (size_t)p->nchunks) != hipSuccess) {
    free_slicing(p);
    return GCN_ERR_ALLOC;
  }
  int sorted = 1;
  if (gcn::build_sliced_csr(rowptr, col, val, p->m, p->n, p->nnz, slices, p->vrowptr, p->vcol,
                            p->vval, &sorted, (hipStream_t)stream) != hipSuccess) {
    free_slicing(p);
    return GCN_ERR_HIP;
  }
  if (!sorted) {                          // needs column-sorted rows; auto mode just stays unsliced
    free_slicing(p);
    return autom ? GCN_OK : GCN_ERR_INVALID_ARG;
  }
  if (gcn::launch_plan_chunk_rows(p->vrowptr, (int)vm, p->T, p->nchunks, p->vchunk_row,
                                  (hipStream_t)stream) != hipSuccess) {
    free_slicing(p);
    return GCN_ERR_HIP;
  }
  p->S = slices;
  // 15-bit stream of the group kernel (value-free pass): slices at most 32 767 columns wide; best effort
  if ((p->n + slices - 1) / slices <= 32767 && group_enabled()) {
    if (hipMalloc((void**)&p->gvrowptr, sizeof(int) * (size_t)(vm + 1)) == hipSuccess) {
      int nch = 0;
      if (gcn::build_group_stream(p->vrowptr, p->vcol, p->m, p->n, slices, group_chunk(), p->gvrowptr, &p->gstream,
                                  &p->gchunk_row, &p->gchunk_meta, &nch, (hipStream_t)stream) == hipSuccess && nch > 0) {
        p->gnchunks = nch; p->gT = group_chunk(); p->gw = (p->n + slices - 1) / slices;
      } else {
        (void)hipFree(p->gvrowptr); p->gvrowptr = nullptr;
      }
    }
  }
  // 16-bit column stream for the value-free pass (2 instead of 4 index bytes per non-zero across the fabric):
  // slices at most 65 535 columns wide, at most 8 of them; best effort — without it the 32-bit stream is used
  if (!p->gstream && slices <= 8 && (p->n + slices - 1) / slices <= 65535 && col16_enabled()) {
    if (hipMalloc((void**)&p->vrowptr16, sizeof(int) * (size_t)(vm + 1)) == hipSuccess) {
      int nnz16 = 0;
      if (gcn::build_col16_stream(p->vrowptr, p->vcol, p->m, p->n, slices, p->T, p->vrowptr16, &p->vcol16, &nnz16,
                                  p->start16, (hipStream_t)stream) == hipSuccess && nnz16 > 0) {
        p->nnz16 = nnz16;
        p->nchunks16 = nnz16 / p->T;
        if (hipMalloc((void**)&p->vchunk_row16, sizeof(int) * (size_t)p->nchunks16) != hipSuccess ||
            gcn::launch_plan_chunk_rows(p->vrowptr16, (int)vm, p->T, p->nchunks16, p->vchunk_row16,
                                        (hipStream_t)stream) != hipSuccess ||
            hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
          if (p->vchunk_row16) (void)hipFree(p->vchunk_row16);
          p->vchunk_row16 = nullptr;
        }
      }
      if (!p->vchunk_row16) {                           // anything failed: drop the 16-bit stream
        if (p->vcol16) (void)hipFree(p->vcol16);
        (void)hipFree(p->vrowptr16);
        p->vcol16 = nullptr; p->vrowptr16 = nullptr; p->nnz16 = p->nchunks16 = 0;
      }
    }
  }
  // Normalised adjacencies (D^-1/2 (A+I) D^-1/2) have values u[r]*u[c]: when every stored entry matches
  // that to 4 ulp the sliced main pass can run without its value stream (spmm_quad.hip, VALLESS) on a
  // B whose rows were scaled by u, with the row factor applied in the slice reduction.
  if (p->m == p->n && valless_enabled() && !p->u_row) {          // (factors given by the caller stay)
    float* u = nullptr;
    if (hipMalloc((void**)&u, sizeof(float) * (size_t)p->n) != hipSuccess) return GCN_OK;
    int ok = 0;
    if (gcn::detect_rank1_values(rowptr, col, val, p->n, u, &ok, (hipStream_t)stream) != hipSuccess || !ok)
      (void)hipFree(u);
    else
      p->u_row = p->u_col = u;
  }
  return GCN_OK;
}

int32_t gcn_spmm_plan_num_slices(const gcn_spmm_plan_t* p) { return p ? p->S : -1; }

int gcn_spmm_plan_set_value_factors(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                                    const float* val, const float* u_row, const float* u_col, void* stream) {
  if (!p) return GCN_ERR_INVALID_ARG;
  free_factors(p);
  if (!u_row && !u_col) return GCN_OK;                              // (null, null): forget the factors
  if (!u_row || !u_col || !rowptr || (p->nnz > 0 && (!col || !val))) return GCN_ERR_INVALID_ARG;
  if (p->m == 0 || p->nnz == 0 || !valless_enabled()) return GCN_OK;
  int ok = 0;
  if (gcn::verify_value_factors(rowptr, col, val, u_row, u_col, p->m, &ok, (hipStream_t)stream) != hipSuccess)
    return GCN_ERR_HIP;
  if (!ok) return GCN_ERR_INVALID_ARG;                              // some entry is not u_row[r]*u_col[c]
  float *ur = nullptr, *uc = nullptr;
  if (hipMalloc((void**)&ur, sizeof(float) * (size_t)p->m) != hipSuccess) return GCN_ERR_ALLOC;
  if (hipMalloc((void**)&uc, sizeof(float) * (size_t)p->n) != hipSuccess) { (void)hipFree(ur); return GCN_ERR_ALLOC; }
  if (hipMemcpyAsync(ur, u_row, sizeof(float) * (size_t)p->m, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess ||
      hipMemcpyAsync(uc, u_col, sizeof(float) * (size_t)p->n, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    (void)hipFree(ur); (void)hipFree(uc);
    return GCN_ERR_HIP;
  }
  p->u_row = ur; p->u_col = uc;
  return GCN_OK;
}

int32_t gcn_spmm_plan_has_value_factors(const gcn_spmm_plan_t* p) { return p ? (p->u_row != nullptr) : -1; }

static void free_panels(gcn_spmm_plan* p) {
  void* ptrs[] = {p->panel_w0, p->pin_rowptr, p->pin_off, p->pin_val, p->pout_rowptr, p->pout_col,
                  p->pout_val, p->pout_chunk_row, p->pout_vrowptr, p->pout_vcol, p->pout_vchunk_row, p->pout_vval};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  p->panel_w0 = p->pin_rowptr = p->pin_off = p->pout_rowptr = p->pout_col = p->pout_chunk_row = nullptr;
  p->pout_vrowptr = p->pout_vcol = p->pout_vchunk_row = nullptr;
  p->pin_val = p->pout_val = p->pout_vval = nullptr;
  p->panel_R = 0; p->pout_nnz = p->pout_T = p->pout_nchunks = p->pout_S = 0;
}

int gcn_spmm_plan_enable_panels(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                                const float* val, int32_t mode, void* stream) {
  if (!p || mode < -1 || mode > 1) return GCN_ERR_INVALID_ARG;
  free_panels(p);
  p->panel_coverage = 0.0;
  if (mode == 0 || p->nnz == 0 || p->m == 0) return GCN_OK;
  if (!rowptr || !col || !val) return GCN_ERR_INVALID_ARG;
  const int R = 128;
  const int panels = (p->m + R - 1) / R;
  if (hipMalloc((void**)&p->panel_w0, sizeof(int) * (size_t)panels) != hipSuccess) return GCN_ERR_ALLOC;
  unsigned long long inside = 0;
  if (gcn::panel_plan(rowptr, col, p->m, p->n, R, p->panel_w0, &inside, (hipStream_t)stream) != hipSuccess) {
    (void)hipFree(p->panel_w0); p->panel_w0 = nullptr;
    return GCN_ERR_HIP;
  }
  p->panel_coverage = (double)inside / (double)p->nnz;
  // automatic: only when at least half of the non-zeros are served from the staged tile
  if (!(mode == 1 || p->panel_coverage >= 0.5)) { free_panels(p); return GCN_OK; }
  // split A = A_in + A_out on the device
  hipStream_t st = (hipStream_t)stream;
  const size_t rp_bytes = sizeof(int) * (size_t)(p->m + 1);
  if (hipMalloc((void**)&p->pin_rowptr, rp_bytes) != hipSuccess ||
      hipMalloc((void**)&p->pout_rowptr, rp_bytes) != hipSuccess) { free_panels(p); return GCN_ERR_ALLOC; }
  int nnz_in = 0;
  if (gcn::panel_split(rowptr, col, val, p->panel_w0, p->m, R, p->pin_rowptr, p->pout_rowptr, nullptr,
                       nullptr, nullptr, nullptr, &nnz_in, st) != hipSuccess) { free_panels(p); return GCN_ERR_HIP; }
  const int nnz_out = p->nnz - nnz_in;
  if (hipMalloc((void**)&p->pin_off, sizeof(int) * (size_t)(nnz_in > 0 ? nnz_in : 1)) != hipSuccess ||
      hipMalloc((void**)&p->pin_val, sizeof(float) * (size_t)(nnz_in > 0 ? nnz_in : 1)) != hipSuccess ||
      hipMalloc((void**)&p->pout_col, sizeof(int) * (size_t)(nnz_out > 0 ? nnz_out : 1)) != hipSuccess ||
      hipMalloc((void**)&p->pout_val, sizeof(float) * (size_t)(nnz_out > 0 ? nnz_out : 1)) != hipSuccess) {
    free_panels(p); return GCN_ERR_ALLOC;
  }
  if (gcn::panel_split(rowptr, col, val, p->panel_w0, p->m, R, p->pin_rowptr, p->pout_rowptr, p->pin_off,
                       p->pin_val, p->pout_col, p->pout_val, &nnz_in, st) != hipSuccess) {
    free_panels(p); return GCN_ERR_HIP;
  }
  p->pout_nnz = nnz_out;
  p->pout_T = auto_chunk_nnz(nnz_out, p->cu_count);
  p->pout_nchunks = (int)(((long long)nnz_out + p->pout_T - 1) / p->pout_T);
  if (p->pout_nchunks > 0) {
    if (hipMalloc((void**)&p->pout_chunk_row, sizeof(int) * (size_t)p->pout_nchunks) != hipSuccess) {
      free_panels(p); return GCN_ERR_ALLOC;
    }
    if (gcn::launch_plan_chunk_rows(p->pout_rowptr, p->m, p->pout_T, p->pout_nchunks, p->pout_chunk_row,
                                    st) != hipSuccess) { free_panels(p); return GCN_ERR_HIP; }
  }
  // The out-of-window part is what is LEFT of the matrix once the local structure is staged: short
  // rows with columns all over the range, i.e. an unordered graph — the case XCD column slicing is
  // for (slicing.hip).  Measured on the 240 k-vertex planted-partition graph (22 M left-over entries,
  // 92 per row): slicing them 8-ways cuts the gather time only with the one-per-gather kernel (virtual
  // rows of 11 entries: 2.49 -> 2.07 ms of kernels) and then pays 0.23 ms for the reduction — no clear
  // win, so it stays off unless GCN_AMD_PANEL_OUT_SLICES asks for it.
  static const int out_slices = [] { const char* v = std::getenv("GCN_AMD_PANEL_OUT_SLICES"); return v ? std::atoi(v) : 0; }();
  const int S = out_slices;
  if (S > 1 && p->pout_nchunks > 0 && (long long)S * p->m + 1 < (1LL << 31)) {
    const long long vm = (long long)S * p->m;
    if (hipMalloc((void**)&p->pout_vrowptr, sizeof(int) * (size_t)(vm + 1)) != hipSuccess ||
        hipMalloc((void**)&p->pout_vcol, sizeof(int) * (size_t)nnz_out) != hipSuccess ||
        hipMalloc((void**)&p->pout_vval, sizeof(float) * (size_t)nnz_out) != hipSuccess ||
        hipMalloc((void**)&p->pout_vchunk_row, sizeof(int) * (size_t)p->pout_nchunks) != hipSuccess) {
      free_panels(p); return GCN_ERR_ALLOC;
    }
    int sorted = 1;
    if (gcn::build_sliced_csr(p->pout_rowptr, p->pout_col, p->pout_val, p->m, p->n, nnz_out, S, p->pout_vrowptr,
                              p->pout_vcol, p->pout_vval, &sorted, st) != hipSuccess) { free_panels(p); return GCN_ERR_HIP; }
    if (sorted && gcn::launch_plan_chunk_rows(p->pout_vrowptr, (int)vm, p->pout_T, p->pout_nchunks,
                                              p->pout_vchunk_row, st) != hipSuccess) { free_panels(p); return GCN_ERR_HIP; }
    if (sorted) p->pout_S = S;                 // (unsorted rows: the unsliced out-of-window pass stays)
  }
  p->panel_R = R;
  return GCN_OK;
}

int32_t gcn_spmm_plan_panel_rows(const gcn_spmm_plan_t* p) { return p ? p->panel_R : -1; }
double gcn_spmm_plan_panel_coverage(const gcn_spmm_plan_t* p) { return p ? p->panel_coverage : -1.0; }

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
  if (p->panel_R > 0 && k > 32) return (k + 63) / 64;
  const int tile = p->tile_cols ? p->tile_cols : (p->S > 0 && k > 32 ? 64 : auto_tile_cols(p->n, k));
  const int vec = gcn::pick_vec(k, tile, nullptr, nullptr, nullptr);   // 16-B aligned operands
  return (k + 64 * vec - 1) / (64 * vec);
}

int gcn_spmm_plan_main_kernel(const gcn_spmm_plan_t* p, int32_t k, int32_t epilogue, char* buf, int32_t buflen) {
  if (!p || k <= 0 || !buf || buflen <= 0) return GCN_ERR_INVALID_ARG;
  if (p->panel_R > 0 && k > 32) { snprintf(buf, (size_t)buflen, "gcn::spmm_panel_in_kernel"); return GCN_OK; }
  gcn::SpmmArgs a{};
  const bool sliced = p->S > 0 && p->nnz > 0 && k >= slice_min_k();
  a.k = k; a.n = p->n; a.m = sliced ? p->S * p->m : p->m; a.nnz = p->nnz;
  a.nchunks_grid = p->nchunks;
  a.relu = epilogue && !sliced ? 1 : 0;               // sliced: the epilogue runs in the slice reduction
  a.tile_cols = p->tile_cols ? p->tile_cols : (sliced ? 64 : auto_tile_cols(p->n, k));
  a.gather_width = p->gather_width;
  if (k > 16 && k % 4 != 0 && p->panel_R == 0 && p->gather_width != 1 && pad_b_enabled() &&
      rows_long_enough_for_quad(p, k) && (long long)sizeof(float) * p->n * (((k + 3) / 4 * 4 + 31) / 32 * 32) <= (768LL << 20)) {
    a.k = (k + 3) / 4 * 4;                             // odd widths run at k rounded up to 4 (see gcn_spmm_csr_f32_bias_relu)
    a.ldb = (a.k + 31) / 32 * 32;
    a.relu = 0;
    a.valless = valless_pays(p, a, a.k, a.ldb);
    a.col16 = a.valless && p->vcol16 != nullptr;
  } else {
    if (const int ldb = padded_ldb(p->n, k); ldb != k) a.ldb = ldb;
    a.valless = valless_pays(p, a, k, a.ldb > 0 ? a.ldb : k);   // as spmm_impl decides
    a.col16 = a.valless && p->vcol16 != nullptr;
  }
  if (a.valless && group_pass(p)) {
    snprintf(buf, (size_t)buflen, "gcn::spmm_group_kernel<%d, %s>", p->gT, group_sc1() ? "true" : "false");
    return GCN_OK;
  }
  gcn::describe_main_kernel(a, buf, (size_t)buflen);
  return GCN_OK;
}

int gcn_spmm_profile_begin(gcn_spmm_plan_t* p, int32_t capacity) {
  if (!p || capacity <= 0 || p->prof_cap > 0) return GCN_ERR_INVALID_ARG;
  p->ev.resize(2 * (size_t)capacity);
  for (auto& e : p->ev)
    if (hipEventCreate(&e) != hipSuccess) return GCN_ERR_HIP;
  p->prof_cap = capacity;
  p->prof_n = 0;
  return GCN_OK;
}

int gcn_spmm_profile_end(gcn_spmm_plan_t* p, float* ms_out, int32_t* count_out) {
  if (!p || !count_out || p->prof_cap <= 0) return GCN_ERR_INVALID_ARG;
  int st = GCN_OK;
  for (int i = 0; i < p->prof_n; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(p->ev[2 * i + 1]) != hipSuccess ||
        hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]) != hipSuccess) st = GCN_ERR_HIP;
    if (ms_out) ms_out[i] = ms;
  }
  *count_out = p->prof_n;
  for (auto& e : p->ev) (void)hipEventDestroy(e);
  p->ev.clear();
  p->prof_cap = p->prof_n = 0;
  return st;
}

int gcn_spmm_csr_f32(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                     const float* val, const float* B, float* C, int32_t k, void* stream) {
  return gcn_spmm_csr_f32_bias_relu(p, rowptr, col, val, B, C, nullptr, 0, k, stream);
}

// One-shot: the schedule is recomputed on the device every call (a few µs: one
// binary search per chunk) into a process-wide scratch plan, so there is no cache
// that could go stale when the caller reuses device addresses.
int gcn_spmm_csr_f32_oneshot(const int32_t* rowptr, const int32_t* col, const float* val,
                             const float* B, float* C, int32_t m, int32_t n, int32_t nnz,
                             int32_t k, void* stream) {
  static gcn_spmm_plan scratch{};
  static size_t chunk_cap = 0;
  if (m < 0 || n < 0 || nnz < 0 || k < 0) return GCN_ERR_INVALID_ARG;
  const int cu = cu_count_cached();
  if (cu <= 0) return GCN_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(g_mu);
  gcn_spmm_plan* p = &scratch;
  p->m = m; p->n = n; p->nnz = nnz; p->cu_count = cu;
  p->T = auto_chunk_nnz(nnz, cu);
  p->nchunks = (int)(((long long)nnz + p->T - 1) / p->T);
  if ((size_t)p->nchunks > chunk_cap) {
    if (p->chunk_row) (void)hipFree(p->chunk_row);
    p->chunk_row = nullptr; chunk_cap = 0;
    if (hipMalloc((void**)&p->chunk_row, sizeof(int) * (size_t)p->nchunks) != hipSuccess)
      return GCN_ERR_ALLOC;
    chunk_cap = (size_t)p->nchunks;
  }
  if (m == 0 || k == 0) return GCN_OK;
  if (gcn::launch_plan_chunk_rows(rowptr, m, p->T, p->nchunks, p->chunk_row,
                                  (hipStream_t)stream) != hipSuccess) return GCN_ERR_HIP;
  const int st = ensure_ws(p, k);
  if (st != GCN_OK) return st;
  gcn::SpmmArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val; a.B = B; a.C = C; a.P = p->ws;
  a.chunk_row = p->chunk_row; a.bias = nullptr; a.relu = 0;
  a.nchunks = p->nchunks; a.T = p->T; a.m = m; a.nnz = nnz; a.k = k; a.n = n;
  a.nnz_dev = nullptr; a.nchunks_grid = p->nchunks;
  a.tile_cols = auto_tile_cols(n, k);
  return gcn::launch_spmm(a, cu, (hipStream_t)stream) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

int gcn_gather_rows_f32(float* dst, const float* src, const int32_t* idx, int32_t nrows, int32_t k,
                        void* stream) {
  if (nrows < 0 || k < 0) return GCN_ERR_INVALID_ARG;
  if (nrows == 0 || k == 0) return GCN_OK;
  if (!dst || !src || !idx || dst == src) return GCN_ERR_INVALID_ARG;
  return gcn::launch_gather_rows(dst, src, idx, nrows, k, (hipStream_t)stream) == hipSuccess
             ? GCN_OK : GCN_ERR_HIP;
}

// ---------------------------------------------------------------------------
// host reorderers
// ---------------------------------------------------------------------------
static bool csr_ok(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t nnz) {
  if (n < 0 || nnz < 0 || !rowptr || (nnz > 0 && !col)) return false;
  if (rowptr[0] != 0 || rowptr[n] != nnz) return false;
  for (int32_t i = 0; i < n; ++i) if (rowptr[i + 1] < rowptr[i]) return false;
  for (int32_t e = 0; e < nnz; ++e) if (col[e] < 0 || col[e] >= n) return false;
  return true;
}

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

// ---------------------------------------------------------------------------
// drop-in symbols: renumber.so
// ---------------------------------------------------------------------------
static void apply_and_emit(int* rowPtr, int* col, float* vals, int* vomp, int n, int nnz,
                           const std::vector<gcn::reorder::u64>& rank) {
  gcn::reorder::csr_apply_rank(rowPtr, col, vals, n, nnz, rank.data());
  for (int i = 0; i < n; ++i) vomp[rank[i]] = i;     // C ABI returns new -> old
}

static void check_csr_or_die(const char* fn, int* rowPtr, int* col, int n, int nnz) {
  if (!csr_ok(rowPtr, col, n, nnz)) {
    std::fprintf(stderr, "libgcnspmm: %s: malformed CSR input (n=%d nnz=%d)\n", fn, n, nnz);
    std::abort();
  }
}

void dfs(int* rowPtr, int* col, float* vals, int* vomp, int m, int n, int nnz) {
  (void)m;
  check_csr_or_die("dfs", rowPtr, col, n, nnz);
  gcn::reorder::Csr g{rowPtr, col, n, nnz};
  apply_and_emit(rowPtr, col, vals, vomp, n, nnz, gcn::reorder::order_dfs(g));
}

void gorder(int* rowPtr, int* col, float* vals, int* vomp, int m, int n, int nnz) {
  check_csr_or_die("gorder", rowPtr, col, m, nnz);
  (void)n;
  gcn::reorder::Csr g{rowPtr, col, m, nnz};
  bool ok = true;
  auto rank = gcn::reorder::order_gorder_complete(g, 3, &ok);      // window 3: renumber.cu:176
  if (!ok) {
    std::fprintf(stderr, "libgcnspmm: gorder: graph has isolated vertices in the heap index "
                         "range; the reference's behaviour is undefined for it\n");
    std::abort();
  }
  apply_and_emit(rowPtr, col, vals, vomp, m, nnz, rank);
}

void perm_apply(int* rowPtr, int* col, float* vals, int* vomp, int m, int n, int nnz) {
  (void)m;
  check_csr_or_die("perm_apply", rowPtr, col, n, nnz);
  std::vector<gcn::reorder::u64> rank(n, (gcn::reorder::u64)n);
  for (int v = 0; v < n; ++v) {
    const int old = vomp[v];
    if (old < 0 || old >= n || rank[old] != (gcn::reorder::u64)n) {       // renumber.cu:251
      std::fprintf(stderr, "libgcnspmm: perm_apply: vomp is not a permutation\n");
      std::abort();
    }
    rank[old] = v;
  }
  gcn::reorder::csr_apply_rank(rowPtr, col, vals, n, nnz, rank.data());
}

void rabbit(int* rowPtr, int* col, float* vals, int* vomp, int m, int n, int nnz) {
  check_csr_or_die("rabbit", rowPtr, col, n, nnz);
  gcn::reorder::Csr g{rowPtr, col, n, nnz};
  auto vo = gcn::reorder::order_rabbit_vomp(g, verbose());
  for (int i = 0; i < n; ++i) vomp[i] = vo[i];
  perm_apply(rowPtr, col, vals, vomp, m, n, nnz);                 // renumber.cu:521
}

// ---------------------------------------------------------------------------
// drop-in symbols: tile.so / flexspmm.so
//
// Packed layout written by csr2tile into the caller's buffers (gcn6.py:334-339;
// after the call gcn6 shrinks seg_rowPtr to 9*n_segs and segVoMap to 8*n_segs
// ints and copies everything to the device, gcn6.py:353-366):
//   n_segs[0]      = nnz / 9                      (so 9*n_segs <= nnz capacity)
//   seg_rowPtr     = rowPtr[0..m]                 (needs m+1 <= 9*n_segs) — or, when the graph
//                    qualifies for XCD-aware slicing (dropin_slices), the slice-major virtual
//                    row pointer [0..S*m] with col/val reordered to match
//   segVoMap       = chunk_row[0..nchunks)        (needs nchunks <= 8*n_segs)
//   segNzCV[0..nnz)      = column indices, int32 bit patterns (exact for any n,
//                          unlike the reference's float(col), tile.cu:67)
//   segNzCV[nnz..2nnz)   = values
//   grouped_tailSeg / next_seg: 256 zeros (never 257 entries — defect D2)
// The chunk size T is a pure function of n_segs (auto_chunk_nnz(9*n_segs, 256)),
// so flexspmm() can recover the whole schedule from its scalar arguments; the
// exact nnz is read on the device from seg_rowPtr[m].
// ---------------------------------------------------------------------------
static int dropin_T(int n_segs) { return auto_chunk_nnz(9LL * n_segs, 256); }

// Column slices used by the drop-in pair — a pure function of what BOTH csr2tile (host) and
// flexspmm (device pointers only) know: m, n and n_segs.  Slicing is dropped when the virtual
// row pointer (S*m+1 ints) would not fit into seg_rowPtr after gcn6 shrinks it to 9*n_segs.
static int dropin_slices(int m, int n, int n_segs) {
  const int S = auto_slices(m, n, 9LL * n_segs);
  if (S <= 1) return 0;
  if ((long long)S * m + 1 > 9LL * n_segs) return 0;
  return S;
}

void csr2tile(int* rowPtr, int* colIdx, float* vals, int m, int n, int nnz, int* vo_mp,
              int* segVoMap, int* seg_rowPtr, float* segNzCV, int* grouped_tailSeg, int* next_seg,
              int tm, int* n_segs) {
  (void)vo_mp;
  if (tm != 8) {
    std::fprintf(stderr, "libgcnspmm: csr2tile: tm must be 8 (got %d)\n", tm);
    std::abort();
  }
  check_csr_or_die("csr2tile", rowPtr, colIdx, m, nnz);
  const int ns = nnz / 9;
  const int T = dropin_T(ns);
  const int nchunks = (int)(((long long)nnz + T - 1) / T);
  if (m + 1 > 9 * ns || nchunks > 8 * ns) {
    std::fprintf(stderr, "libgcnspmm: csr2tile: graph too sparse to pack into the caller's "
                         "buffers (m=%d nnz=%d); need nnz >= m+10\n", m, nnz);
    std::abort();
  }
  const int S = dropin_slices(m, n, ns);
  int* cols = reinterpret_cast<int*>(segNzCV);
  float* vs = segNzCV + nnz;
  int vm = m;                                   // rows of the CSR that is packed
  if (S == 0) {
    std::memcpy(seg_rowPtr, rowPtr, sizeof(int) * (size_t)(m + 1));
    std::memcpy(cols, colIdx, sizeof(int) * (size_t)nnz);
    std::memcpy(vs, vals, sizeof(float) * (size_t)nnz);
  } else {
    // slice-major virtual CSR (slicing.hip describes the device-side twin): virtual row
    // s*m + r = the entries of row r with column in [s*w, (s+1)*w), in ascending column order
    vm = S * m;
    const int w = (n + S - 1) / S;
    std::vector<int> cnt((size_t)vm + 1, 0);
    for (int r = 0; r < m; ++r)
      for (int e = rowPtr[r]; e < rowPtr[r + 1]; ++e) ++cnt[(size_t)(colIdx[e] / w) * m + r];
    int run = 0;
    for (int i = 0; i < vm; ++i) { seg_rowPtr[i] = run; run += cnt[i]; }
    seg_rowPtr[vm] = run;
    std::vector<int> fill(seg_rowPtr, seg_rowPtr + vm);
    std::vector<std::pair<int, float>> row;
    for (int r = 0; r < m; ++r) {
      row.clear();
      for (int e = rowPtr[r]; e < rowPtr[r + 1]; ++e) row.emplace_back(colIdx[e], vals[e]);
      // the reference's pipeline hands over column-sorted rows (renumber.cu:105-117); sort if not
      if (!std::is_sorted(row.begin(), row.end(),
                          [](const auto& x, const auto& y) { return x.first < y.first; }))
        std::stable_sort(row.begin(), row.end(),
                         [](const auto& x, const auto& y) { return x.first < y.first; });
      for (const auto& [c, v] : row) {
        const int dst = fill[(size_t)(c / w) * m + r]++;
        cols[dst] = c;
        vs[dst] = v;
      }
    }
  }
  for (int i = vm + 1; i < 9 * ns; ++i) seg_rowPtr[i] = nnz;
  // chunk_row[c] = (virtual) row holding non-zero c*T (first row for c = 0)
  int r = 0;
  for (int c = 0; c < nchunks; ++c) {
    const long long target = (long long)c * T;
    while (r < vm && seg_rowPtr[r + 1] <= target) ++r;
    segVoMap[c] = (c == 0) ? 0 : r;
  }
  for (int i = nchunks; i < 8 * ns; ++i) segVoMap[i] = 0;
  for (int i = 0; i < 256; ++i) { grouped_tailSeg[i] = 0; next_seg[i] = 0; }
  n_segs[0] = ns;
}

void flexspmm(int* seg_rowPtr, float* segNzCV, int* segVoMap, int* grouped_tailSeg, int* next_seg,
              int m, int n, int k, int n_segs, float* B, float* C) {
  (void)grouped_tailSeg; (void)next_seg;
  static gcn_spmm_plan scratch{};
  if (m <= 0 || k <= 0) return;
  const int cu = cu_count_cached();
  if (cu <= 0) { std::fprintf(stderr, "libgcnspmm: flexspmm: no HIP device\n"); std::abort(); }
  const int T = dropin_T(n_segs);
  const int S = dropin_slices(m, n, n_segs);
  const int vm = S > 0 ? S * m : m;
  const long long nnz_ub = 9LL * n_segs + 8;
  const int nchunks_ub = (int)((nnz_ub + T - 1) / T);
  // odd widths: computed at k' = k rounded up to 4 on row-padded copies, as in gcn_spmm_csr_f32_bias_relu
  const bool odd = k > 16 && k % 4 != 0 && pad_b_enabled() &&
                   (long long)sizeof(float) * n * (((k + 3) / 4 * 4 + 31) / 32 * 32) <= (768LL << 20);
  const int kc = odd ? (k + 3) / 4 * 4 : k;                      // width the kernels compute at
  const int ldb = odd ? (kc + 31) / 32 * 32 : padded_ldb(n, k);  // row stride B is gathered with
  std::lock_guard<std::mutex> lk(g_mu);
  scratch.nchunks = nchunks_ub;
  if (ensure_ws(&scratch, kc) != GCN_OK) die("flexspmm workspace", hipErrorOutOfMemory);
  auto grow_or_die = [](float*& buf, size_t& have, size_t need, const char* what) {
    if (need <= have) return;
    if (buf) (void)hipFree(buf);
    buf = nullptr; have = 0;
    if (hipMalloc((void**)&buf, need) != hipSuccess) die(what, hipErrorOutOfMemory);
    have = need;
  };
  if (S > 0) grow_or_die(scratch.cv, scratch.cv_bytes, sizeof(float) * (size_t)vm * (size_t)kc, "flexspmm slice buffer");
  if (odd) grow_or_die(scratch.cpad, scratch.cpad_bytes, sizeof(float) * (size_t)m * (size_t)kc, "flexspmm padded result");
  gcn::SpmmArgs a;
  a.rowptr = seg_rowPtr;
  a.col = reinterpret_cast<const int*>(segNzCV);
  a.val = nullptr;                       // = segNzCV + nnz, resolved on the device
  float* Cc = odd ? scratch.cpad : C;    // compact-width or padded-width result
  a.B = B; a.C = S > 0 ? scratch.cv : Cc; a.P = scratch.ws; a.chunk_row = segVoMap;   // (the packed
  // layout is fixed by csr2tile, so the drop-in pair slices for every k once the graph qualifies)
  a.bias = nullptr; a.relu = 0;
  a.nchunks = 0; a.T = T; a.m = vm; a.nnz = 0; a.k = kc; a.n = n;
  a.nnz_dev = seg_rowPtr + vm;           // exact nnz lives at the end of the (virtual) row pointer
  a.nchunks_grid = nchunks_ub;
  a.tile_cols = S > 0 ? 64 : auto_tile_cols(n, kc);
  hipError_t e;
  if (ldb != k) {                        // rows on whole cache lines (and zero columns up to k')
    grow_or_die(scratch.bpad, scratch.bpad_bytes, sizeof(float) * (size_t)n * (size_t)ldb, "flexspmm padded features");
    e = gcn::launch_pad_rows(scratch.bpad, B, n, k, ldb, (hipStream_t) nullptr);
    if (e != hipSuccess) die("flexspmm feature padding", e);
    a.B = scratch.bpad;
    a.ldb = ldb;
  }
  e = gcn::launch_spmm(a, cu, (hipStream_t) nullptr);                    // legacy default stream
  if (e != hipSuccess) die("flexspmm launch", e);
  if (S > 0) {
    e = gcn::launch_slice_reduce(scratch.cv, Cc, nullptr, 0, m, S, kc, (hipStream_t) nullptr);
    if (e != hipSuccess) die("flexspmm slice reduction", e);
  }
  if (odd) {
    e = gcn::launch_unpad_rows(C, scratch.cpad, nullptr, 0, m, k, kc, (hipStream_t) nullptr);
    if (e != hipSuccess) die("flexspmm result compaction", e);
  }
}

// ---------------------------------------------------------------------------
// drop-in symbols: permutate.so / cuspmm.so
// ---------------------------------------------------------------------------
void permutate(float* B, int* voMp, int* labels, int m, int n, int k) {
  (void)labels; (void)m;                 // labels are NOT permuted: permutate.cu:17,35
  if (n <= 0 || k <= 0) return;
  float* shadow = nullptr;
  const size_t bytes = sizeof(float) * (size_t)n * (size_t)k;
  hipError_t e = hipMalloc((void**)&shadow, bytes);
  if (e != hipSuccess) die("permutate hipMalloc", e);
  e = gcn::launch_gather_rows(shadow, B, voMp, n, k, nullptr);
  if (e != hipSuccess) die("permutate gather", e);
  e = hipMemcpyAsync(B, shadow, bytes, hipMemcpyDeviceToDevice, nullptr);
  if (e != hipSuccess) die("permutate copy-back", e);
  e = hipStreamSynchronize(nullptr);     // the reference synchronises too (permutate.cu:56)
  if (e != hipSuccess) die("permutate sync", e);
  (void)hipFree(shadow);
}

void cuspmm(float* rowPtr, int* col, float* vals, float* X, float* C, int m, int n, int nnz,
            int dim) {
  const int st = gcn_spmm_csr_f32_oneshot(reinterpret_cast<const int32_t*>(rowPtr), col, vals, X,
                                          C, m, n, nnz, dim, nullptr);
  if (st != GCN_OK) {
    std::fprintf(stderr, "libgcnspmm: cuspmm failed: %s\n", gcn_status_string(st));
    std::abort();
  }
}

}  // extern "C"
