// api_dropin.cpp — the reference's own symbols with identical argument lists (pygcn/gcn6.py:21-25 binds them by
// ctypes): dfs / gorder / perm_apply / rabbit (renumber.so), csr2tile (tile.so), flexspmm (flexspmm.so), permutate
// (permutate.so), cuspmm (cuspmm.so).  All `void`, no status: a failure prints to stderr and aborts, so it is
// never silent (the reference only prints cuSPARSE errors, cuspmm.cu:3-21).  Contract: include/gcn_spmm.h (2).
#include "plan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "reorder.h"

namespace gcn { bool csr_ok(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t nnz); }
using gcn::csr_ok;
using gcn::auto_chunk_nnz;
using gcn::auto_slices;
using gcn::auto_tile_cols;
using gcn::cu_count_cached;
using gcn::die;
using gcn::pad_b_enabled;
using gcn::padded_ldb;
using gcn::verbose;

extern "C" {

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
  std::lock_guard<std::mutex> lk(gcn::g_plan_mu);
  // scratch of the legacy default stream on this device (the reference launches there, flexspmm.cu:512)
  gcn_spmm_plan* sp = gcn::scratch_plan(nullptr);
  if (!sp) die("flexspmm scratch plan", hipErrorOutOfMemory);
  gcn_spmm_plan& scratch = *sp;
  auto grow_or_die = [](gcn::DevBuf<float>& buf, size_t count, const char* what) {
    if (buf.grow(count) != hipSuccess) die(what, hipErrorOutOfMemory);
  };
  grow_or_die(scratch.ws, 2 * (size_t)(nchunks_ub > 0 ? nchunks_ub : 1) * (size_t)kc, "flexspmm workspace");
  if (S > 0) grow_or_die(scratch.cv, (size_t)vm * (size_t)kc, "flexspmm slice buffer");
  if (odd) grow_or_die(scratch.cpad, (size_t)m * (size_t)kc, "flexspmm padded result");
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
    grow_or_die(scratch.bpad, (size_t)n * (size_t)ldb, "flexspmm padded features");
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
  gcn::DevBuf<float> shadow;
  const size_t bytes = sizeof(float) * (size_t)n * (size_t)k;
  hipError_t e = shadow.alloc((size_t)n * (size_t)k);
  if (e != hipSuccess) die("permutate hipMalloc", e);
  e = gcn::launch_gather_rows(shadow, B, voMp, n, k, nullptr);
  if (e != hipSuccess) die("permutate gather", e);
  e = hipMemcpyAsync(B, shadow, bytes, hipMemcpyDeviceToDevice, nullptr);
  if (e != hipSuccess) die("permutate copy-back", e);
  e = hipStreamSynchronize(nullptr);     // the reference synchronises too (permutate.cu:56)
  if (e != hipSuccess) die("permutate sync", e);
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
