// api_dropin.cpp — the reference's own symbols with identical argument lists (pygcn/gcn6.py:21-25 binds them by
// ctypes): dfs / gorder / perm_apply / rabbit (renumber.so), csr2tile (tile.so), flexspmm (flexspmm.so), permutate
// (permutate.so), cuspmm (cuspmm.so).  All `void`, no status: a failure prints to stderr and aborts, so it is
// never silent (the reference only prints cuSPARSE errors, cuspmm.cu:3-21).  Contract: include/gcn_spmm.h (2).
#include "plan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
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
//   seg_rowPtr     = rowPtr[0..m]                 (needs m+1 <= 9*n_segs)
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

// ---------------------------------------------------------------------------
// The group-kernel format of the drop-in pair (spmm_group.hip): when the graph qualifies — the same rule as
// the plan API's automatic slicing on the group path, evaluated on what BOTH csr2tile and flexspmm know — csr2tile
// packs the slice-major 15-bit stream, its chunk metadata and the list of cut rows straight into the caller's
// buffers, and flexspmm runs the same kernels as gcn_spmm_csr_f32 on a plan (value-free when the values are
// u[r]*u[c], which csr2tile checks on the host with the plan API's 4-ulp rule; with the values beside the stream
// otherwise).
//   seg_rowPtr[0..16)   header {magic, S, T, w, nchunks, nfix, value_free, total entries, nnz}
//   seg_rowPtr[16..)    chunk_meta (int2 per chunk), then the fix list (int4 per cut row, 16-byte aligned)
//   segNzCV             stream (u16 per entry, lane-major runs of 64), then — 16-byte aligned — the values
//                       (fp32 per entry, same order; absent when value-free)
//   segVoMap[0..n)      u (fp32 bit patterns) when value-free
// flexspmm reads the 64-byte header back (one small synchronous copy per call) — the chunk count depends on the
// padding and cannot be derived from (m, n, n_segs) alone.
// ---------------------------------------------------------------------------
namespace {
constexpr int kDropinMagic = 0x47434E47;            // "GCNG"
constexpr int kDropinT = 512;
struct DropinGroup { int S, w; long long total_ub, nchunks_ub; };

long long dropin_phys(long long p) {                // group_phys of slicing.hip: lane-major runs of 64 entries
  const int r = (int)(p & 63);
  return (p & ~63LL) + (r & 15) * 4 + (r >> 4);
}

// pure function of (m, n, n_segs): does the pair use the group format, and with how many slices?
bool dropin_group(int m, int n, int n_segs, DropinGroup* g) {
  if (!gcn::dropin_group_format_enabled() || m != n) return false;
  const long long nnz_lb = 9LL * n_segs, nnz_ub = 9LL * n_segs + 8;
  const int S = auto_slices(m, n, nnz_lb, true);
  if (S <= 1) return false;
  const int w = (n + S - 1) / S;
  if (w > 32767) return false;
  // entries: the non-zeros, one padding entry per empty virtual row (at most S*m), every slice padded to whole
  // chunks and the total to 64 chunks
  const long long total_ub = (nnz_ub + (long long)S * m + (long long)(S + 64) * kDropinT + 63) / 64 * 64;
  const long long nchunks_ub = total_ub / kDropinT + 1;
  if (total_ub >= (1LL << 31)) return false;
  if (((total_ub * 2 + 15) / 16 * 16) + total_ub * 4 > 8 * nnz_lb) return false;      // segNzCV: 2*nnz floats
  if (16 + 2 * nchunks_ub + 4 + 4 * nchunks_ub > 9LL * n_segs) return false;            // seg_rowPtr: 9*n_segs ints
  if (n > 8LL * n_segs) return false;                                                   // segVoMap: 8*n_segs ints
  g->S = S; g->w = w; g->total_ub = total_ub; g->nchunks_ub = nchunks_ub;
  return true;
}

// the host twin of build_group_stream (slicing.hip); rows must be column-sorted (sorted here if not)
void dropin_pack_group(const int* rowPtr, const int* colIdx, const float* vals, int m, int n, int nnz,
                       const DropinGroup& g, int* segVoMap, int* seg_rowPtr, float* segNzCV, int n_segs) {
  const int S = g.S, w = g.w, T = kDropinT;
  const long long vm = (long long)S * m;
  // the reference's pipeline hands over column-sorted rows (renumber.cu:105-117); work on a sorted copy if not
  std::vector<int> col_sorted;
  std::vector<float> val_sorted;
  {
    bool sorted = true;
    for (int r = 0; r < m && sorted; ++r)
      for (int e = rowPtr[r] + 1; e < rowPtr[r + 1]; ++e)
        if (colIdx[e] < colIdx[e - 1]) { sorted = false; break; }
    if (!sorted) {
      col_sorted.assign(colIdx, colIdx + nnz);
      val_sorted.assign(vals, vals + nnz);
      std::vector<std::pair<int, float>> row;
      for (int r = 0; r < m; ++r) {
        row.clear();
        for (int e = rowPtr[r]; e < rowPtr[r + 1]; ++e) row.emplace_back(colIdx[e], vals[e]);
        std::stable_sort(row.begin(), row.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        for (size_t i = 0; i < row.size(); ++i) { col_sorted[(size_t)rowPtr[r] + i] = row[i].first; val_sorted[(size_t)rowPtr[r] + i] = row[i].second; }
      }
      colIdx = col_sorted.data();
      vals = val_sorted.data();
    }
  }
  // 1. values u[r]*u[c]?  (rank1_diag_kernel / rank1_check_kernel of slicing.hip)
  std::vector<float> u((size_t)n, 0.f);
  bool value_free = true;
  for (int r = 0; r < n && value_free; ++r) {
    const int* lo = std::lower_bound(colIdx + rowPtr[r], colIdx + rowPtr[r + 1], r);
    if (lo < colIdx + rowPtr[r + 1] && *lo == r && vals[lo - colIdx] > 0.f) u[r] = (float)std::sqrt((double)vals[lo - colIdx]);
    else value_free = false;
  }
  for (int r = 0; r < n && value_free; ++r)
    for (int e = rowPtr[r]; e < rowPtr[r + 1]; ++e) {
      const float want = u[r] * u[colIdx[e]];
      if (!(std::fabs(vals[e] - want) <= 4.8e-7f * std::fabs(vals[e]))) { value_free = false; break; }
    }
  // 2. entries per virtual row (at least one), positions in the padded stream
  std::vector<int> cnt((size_t)vm, 0);
  for (int r = 0; r < m; ++r)
    for (int e = rowPtr[r]; e < rowPtr[r + 1]; ++e) ++cnt[(size_t)(colIdx[e] / w) * m + r];
  std::vector<long long> vrp((size_t)vm + 1);
  long long pos = 0;
  for (int s = 0; s < S; ++s) {
    for (int r = 0; r < m; ++r) {
      const size_t vr = (size_t)s * m + r;
      vrp[vr] = pos;
      pos += cnt[vr] > 0 ? cnt[vr] : 1;
    }
    pos = (pos + T - 1) / T * T;                                  // every slice in whole chunks
  }
  const long long total = (pos + 64LL * T - 1) / (64LL * T) * (64LL * T);
  vrp[(size_t)vm] = total;
  if (total > g.total_ub) { std::fprintf(stderr, "libgcnspmm: csr2tile: internal capacity bound violated\n"); std::abort(); }
  const int nchunks = (int)(total / T);
  // 3. stream (+ values): zero-row entries everywhere, then the rows; the last position a virtual row owns (for the
  //    last row of a slice: the end of the slice's padding) carries the row-end bit
  unsigned short* stream = reinterpret_cast<unsigned short*>(segNzCV);
  float* vs = segNzCV + ((size_t)total * 2 + 15) / 16 * 4;        // (16-byte aligned, in floats)
  for (long long i = 0; i < total; ++i) stream[i] = (unsigned short)w;
  if (!value_free) std::memset(vs, 0, sizeof(float) * (size_t)total);
  {
    std::vector<long long> fill(vrp.begin(), vrp.end() - 1);
    for (int r = 0; r < m; ++r)
      for (int e = rowPtr[r]; e < rowPtr[r + 1]; ++e) {
        const int c = colIdx[e], s = c / w;
        const long long p = fill[(size_t)s * m + r]++;
        stream[dropin_phys(p)] = (unsigned short)(c - s * w);
        if (!value_free) vs[dropin_phys(p)] = vals[e];
      }
    for (long long vr = 0; vr < vm; ++vr) {
      // the row owns [vrp[vr], next row's start) — for the last row of a slice that includes the slice's padding
      const long long last = vrp[(size_t)vr + 1] - 1;
      stream[dropin_phys(last)] |= 0x8000;
    }
  }
  // (vrp[vr+1] of a slice's last row is the next slice's start, i.e. already behind this slice's padding)
  // 4. chunk metadata and the list of cut rows
  int* meta = seg_rowPtr + 16;
  size_t vr = 0;
  std::vector<int> chunk_row((size_t)nchunks);
  for (int c = 0; c < nchunks; ++c) {
    const long long target = (long long)c * T;
    while (vr + 1 < (size_t)vm && vrp[vr + 1] <= target) ++vr;
    chunk_row[(size_t)c] = (int)vr;
    meta[2 * c] = 2 * (int)vr + (vrp[vr] < target ? 1 : 0);
    meta[2 * c + 1] = (int)(vr / (size_t)m) * (w + 1);
  }
  int* fix = seg_rowPtr + (16 + 2 * (size_t)nchunks + 3) / 4 * 4;
  int nfix = 0;
  for (int c = 1; c < nchunks; ++c) {
    if (!(meta[2 * c] & 1)) continue;
    const size_t r = (size_t)chunk_row[(size_t)c];
    if (vrp[r] / T != c - 1) continue;
    fix[4 * nfix + 0] = (int)r; fix[4 * nfix + 1] = c; fix[4 * nfix + 2] = (int)((vrp[r + 1] - 1) / T); fix[4 * nfix + 3] = 0;
    ++nfix;
  }
  for (int* q = fix + 4 * (size_t)nfix; q < seg_rowPtr + 9 * (size_t)n_segs; ++q) *q = 0;
  if (value_free) std::memcpy(segVoMap, u.data(), sizeof(float) * (size_t)n);
  for (int i = value_free ? n : 0; i < 8 * n_segs; ++i) segVoMap[i] = 0;
  const int header[16] = {kDropinMagic, S, T, w, nchunks, nfix, value_free ? 1 : 0, (int)total, nnz, 0, 0, 0, 0, 0, 0, 0};
  std::memcpy(seg_rowPtr, header, sizeof(header));
}
}  // namespace

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
  DropinGroup gg;
  if (dropin_group(m, n, ns, &gg)) {                   // the group-kernel format (see above)
    dropin_pack_group(rowPtr, colIdx, vals, m, n, nnz, gg, segVoMap, seg_rowPtr, segNzCV, ns);
    for (int i = 0; i < 256; ++i) { grouped_tailSeg[i] = 0; next_seg[i] = 0; }
    n_segs[0] = ns;
    return;
  }
  // everything else: the plain CSR (unsliced kernels)
  int* cols = reinterpret_cast<int*>(segNzCV);
  float* vs = segNzCV + nnz;
  const int vm = m;
  std::memcpy(seg_rowPtr, rowPtr, sizeof(int) * (size_t)(m + 1));
  std::memcpy(cols, colIdx, sizeof(int) * (size_t)nnz);
  std::memcpy(vs, vals, sizeof(float) * (size_t)nnz);
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

// flexspmm on the group-kernel format: the kernels of the plan API on the caller's packed buffers
static void flexspmm_group(const int* seg_rowPtr, const float* segNzCV, const int* segVoMap, int m, int n, int k,
                           const DropinGroup& g, const float* B, float* C) {
  int h[16];
  hipError_t e = hipMemcpy(h, seg_rowPtr, sizeof(h), hipMemcpyDeviceToHost);      // (synchronous: legacy stream)
  if (e != hipSuccess) die("flexspmm header copy", e);
  const int nchunks = h[4], nfix = h[5], value_free = h[6];
  const long long total = h[7];
  if (h[0] != kDropinMagic || h[1] != g.S || h[2] != kDropinT || h[3] != g.w || nchunks <= 0 || nchunks % 64 != 0 ||
      total != (long long)nchunks * kDropinT || total > g.total_ub || nfix < 0 || nfix > nchunks) {
    std::fprintf(stderr, "libgcnspmm: flexspmm: the buffers were not packed by this library's csr2tile "
                         "for m=%d n=%d (header mismatch)\n", m, n);
    std::abort();
  }
  const int kc = (k + 3) / 4 * 4;                                 // the group kernels compute at a multiple of 4
  const int ldb = kc != k ? (kc + 31) / 32 * 32 : padded_ldb(n, k);
  std::lock_guard<std::mutex> lk(gcn::g_plan_mu);
  gcn_spmm_plan* sp = gcn::scratch_plan(nullptr);
  if (!sp) die("flexspmm scratch plan", hipErrorOutOfMemory);
  gcn_spmm_plan& scratch = *sp;
  auto grow_or_die = [](gcn::DevBuf<float>& buf, size_t count, const char* what) {
    if (buf.grow(count) != hipSuccess) die(what, hipErrorOutOfMemory);
  };
  grow_or_die(scratch.ws, 2 * (size_t)nchunks * (size_t)kc, "flexspmm workspace");
  grow_or_die(scratch.cv, (size_t)g.S * (size_t)m * (size_t)kc, "flexspmm slice buffer");
  grow_or_die(scratch.bpad, (size_t)g.S * (size_t)(g.w + 1) * (size_t)ldb, "flexspmm feature copy");
  if (kc != k) grow_or_die(scratch.cpad, (size_t)m * (size_t)kc, "flexspmm padded result");
  hipStream_t st = nullptr;                                        // legacy default stream (flexspmm.cu:512)
  const float* u = value_free ? reinterpret_cast<const float*>(segVoMap) : nullptr;
  e = gcn::launch_scale_rows_sliced(scratch.bpad, B, u, n, k, ldb, g.S, g.w, st);
  if (e != hipSuccess) die("flexspmm feature copy", e);
  gcn::GroupArgs ga;
  ga.stream = reinterpret_cast<const unsigned short*>(segNzCV);
  ga.vals = value_free ? nullptr : segNzCV + ((size_t)total * 2 + 15) / 16 * 4;
  ga.chunk_meta = seg_rowPtr + 16;
  ga.Bp = scratch.bpad; ga.Cv = scratch.cv; ga.P = scratch.ws;
  ga.nchunks = nchunks; ga.T = kDropinT; ga.k = kc; ga.ldb = ldb;
  ga.table_rows = (long long)g.S * ((long long)g.w + 1);
  e = gcn::launch_spmm_group(ga, st);
  if (e != hipSuccess) die("flexspmm launch", e);
  const int* fix = seg_rowPtr + (16 + 2 * (size_t)nchunks + 3) / 4 * 4;
  e = gcn::launch_group_fixup(fix, nfix, scratch.ws, scratch.cv, kc, st);
  if (e != hipSuccess) die("flexspmm fix-up", e);
  float* Cc = kc != k ? scratch.cpad.get() : C;
  e = gcn::launch_slice_reduce(scratch.cv, Cc, nullptr, 0, m, g.S, kc, st, 0, u, gcn::DropoutSpec{});
  if (e != hipSuccess) die("flexspmm slice reduction", e);
  if (kc != k) {
    e = gcn::launch_unpad_rows(C, scratch.cpad, nullptr, 0, m, k, kc, st);
    if (e != hipSuccess) die("flexspmm result compaction", e);
  }
}

void flexspmm(int* seg_rowPtr, float* segNzCV, int* segVoMap, int* grouped_tailSeg, int* next_seg,
              int m, int n, int k, int n_segs, float* B, float* C) {
  (void)grouped_tailSeg; (void)next_seg;
  if (m <= 0 || k <= 0) return;
  const int cu = cu_count_cached();
  if (cu <= 0) { std::fprintf(stderr, "libgcnspmm: flexspmm: no HIP device\n"); std::abort(); }
  DropinGroup gg;
  if (dropin_group(m, n, n_segs, &gg)) { flexspmm_group(seg_rowPtr, segNzCV, segVoMap, m, n, k, gg, B, C); return; }
  const int T = dropin_T(n_segs);
  const int vm = m;                                  // (graphs that qualify for slicing take the group format above)
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
  if (odd) grow_or_die(scratch.cpad, (size_t)m * (size_t)kc, "flexspmm padded result");
  gcn::SpmmArgs a;
  a.rowptr = seg_rowPtr;
  a.col = reinterpret_cast<const int*>(segNzCV);
  a.val = nullptr;                       // = segNzCV + nnz, resolved on the device
  float* Cc = odd ? scratch.cpad : C;    // compact-width or padded-width result
  a.B = B; a.C = Cc; a.P = scratch.ws; a.chunk_row = segVoMap;
  a.bias = nullptr; a.relu = 0;
  a.nchunks = 0; a.T = T; a.m = vm; a.nnz = 0; a.k = kc; a.n = n;
  a.nnz_dev = seg_rowPtr + vm;           // exact nnz lives at the end of the (virtual) row pointer
  a.nchunks_grid = nchunks_ub;
  a.tile_cols = auto_tile_cols(n, kc);
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
