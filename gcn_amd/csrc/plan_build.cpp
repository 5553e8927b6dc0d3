// plan_build.cpp — construction of the SpMM plan (gcn_spmm_plan_t): the chunk table, the XCD-aware column slicing with its
// streams (15-bit group stream, 16-bit column stream, the narrow slice set for k <= 32), value factors, LDS / MFMA panels.
// Everything here runs once per graph (or once per width class) and may synchronise; the launches are api_spmm.cpp, the
// rules that decide what gets built plan_policy.cpp.
#include "plan_policy.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

namespace gcn {

// rows of a CSR without an entry (the main kernels skip them, launch_fill_empty_rows writes them); synchronises `st`
int count_empty(const int* rowptr, int m, int* out, hipStream_t st) {
  DevBuf<int> cnt;
  *out = -1;
  if (cnt.alloc(1) != hipSuccess) return GCN_ERR_ALLOC;
  int host = 0;
  if (hipMemsetAsync(cnt, 0, sizeof(int), st) != hipSuccess || launch_count_empty_rows(rowptr, m, cnt, st) != hipSuccess ||
      hipMemcpyAsync(&host, cnt, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return GCN_ERR_HIP;
  *out = host;
  return GCN_OK;
}

void drop_streams(gcn_spmm_plan* p) {
  p->col16 = Col16Stream{};
  p->group = GroupStream{};
  std::lock_guard<std::mutex> lk(g_plan_mu);
  for (int c = 0; c < 1; ++c) { p->group_alt[c] = GroupStream{}; p->alt_S[c] = 0; p->alt_tried[c] = false; }
}


// the streams of a sliced plan beside its virtual CSR: the group kernel's (value-free when the values factor and
// the scaled copy pays, else with the values beside it), or the 16-bit columns of the value-free four-per-gather pass
void build_sliced_streams(gcn_spmm_plan* p, hipStream_t st) {
  Slicing& sl = p->slicing;
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
    const int gT = group_chunk(p->nnz, p->cu_count);
    if (build_group_stream(sl.vrowptr, sl.vcol, p->m, p->n, sl.S, gT, p->group.vrowptr, &stream,
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
    drop_streams(p);
  }
  if (!p->factors.ready()) return;
  // 16-bit column stream of the four-per-gather kernel (2 instead of 4 index bytes per non-zero): slices at
  // most 65 535 columns wide, at most 8 of them; best effort — without it the 32-bit stream is used
  if (sl.S <= 8 && w <= 65535 && p->col16.vrowptr16.alloc((size_t)(vm + 1)) == hipSuccess) {
    Col16Stream& c = p->col16;
    unsigned short* c16 = nullptr;
    int nnz16 = 0;
    if (build_col16_stream(sl.vrowptr, sl.vcol, p->m, p->n, sl.S, p->T, c.vrowptr16, &c16, &nnz16, c.start16, st) == hipSuccess &&
        nnz16 > 0) {
      c.vcol16.adopt(c16, (size_t)nnz16);
      c.nnz16 = nnz16;
      c.nchunks16 = nnz16 / p->T;
      if (c.vchunk_row16.alloc((size_t)c.nchunks16) == hipSuccess &&
          launch_plan_chunk_rows(c.vrowptr16, (int)vm, p->T, c.nchunks16, c.vchunk_row16, st) == hipSuccess &&
          hipStreamSynchronize(st) == hipSuccess)
        return;
    }
    p->col16 = Col16Stream{};                     // anything failed: drop the 16-bit stream
  }
}

// The narrow slice set of a plan (plan.h, group_alt[0]): for k <= 32 a row of the table is 128 bytes, so an L2 holds a
// slice twice as wide and the matrix needs about half the slices — and every slice costs a partial row per matrix row.
// Reddit-shaped (profiles/r03az_*): 8 slices instead of 15; k = 16 / 32 whole SpMM 0.684 / 0.782 -> 0.655 / 0.746 ms.
// Built once, at the first such call of a value-free plan with an automatic slice count, from the CSR the call hands
// over or by gcn_spmm_plan_prepare_width (a transient virtual CSR; only the stream, its chunk table and cut lists are
// kept: 2 bytes per non-zero).  Anything that fails leaves the plan on its own slices.

void maybe_build_alt(gcn_spmm_plan* p, int cls, const int32_t* rowptr, const int32_t* col, const float* val, hipStream_t st) {
  if (cls < 0) return;
  std::lock_guard<std::mutex> lk(g_plan_mu);           // (two host threads at their first narrow call: one builds)
  if (p->alt_tried[cls]) return;
  p->alt_tried[cls] = true;
  if (!p->slices_auto || !p->group.ready() || p->group.vals || !value_free_plan(p) || p->nnz <= 0) return;
  const long long l2 = 4LL << 20, row_bytes = 128;
  long long S2 = ((long long)p->n * row_bytes + l2 - 1) / l2;
  const long long by_entry = ((long long)p->n + 32766) / 32767;       // 15-bit entries: slices <= 32 767 columns
  if (S2 < by_entry) S2 = by_entry;
  if (S2 > (long long)p->nnz / p->m / 16) S2 = (long long)p->nnz / p->m / 16;
  if (S2 < 2 || S2 + 2 > p->slicing.S) return;                        // (not enough fewer to pay for another stream)
  const int S = (int)S2, w = (p->n + S - 1) / S;
  if (w > 32767) return;
  const long long vm = (long long)S * p->m;
  DevBuf<int> vrowptr, vcol, vrowptr_g;
  DevBuf<float> vval;
  if (vrowptr.alloc((size_t)vm + 1) != hipSuccess || vcol.alloc((size_t)p->nnz) != hipSuccess ||
      vval.alloc((size_t)p->nnz) != hipSuccess || vrowptr_g.alloc((size_t)vm + 1) != hipSuccess) return;
  int sorted = 0;
  if (build_sliced_csr(rowptr, col, val, p->m, p->n, p->nnz, S, vrowptr, vcol, vval, &sorted, st) != hipSuccess || !sorted) return;
  unsigned short* stream = nullptr;
  int *chunk_row = nullptr, *chunk_meta = nullptr, *fix = nullptr, *cutptr = nullptr, *cutchunk = nullptr, nch = 0, nfix = 0, ncut = 0;
  const int gT = group_chunk(p->nnz, p->cu_count);
  if (build_group_stream(vrowptr, vcol, p->m, p->n, S, gT, vrowptr_g, &stream, &chunk_row, &chunk_meta, &nch, &fix, &nfix, st,
                              nullptr, nullptr, &cutptr, &cutchunk, &ncut) != hipSuccess || nch <= 0) return;
  GroupStream& g = p->group_alt[cls];
  g.fix.adopt(fix, 4 * (size_t)nfix); g.nfix = nfix;
  g.cutptr.adopt(cutptr, (size_t)p->m + 1); g.cutchunk.adopt(cutchunk, (size_t)(ncut > 0 ? ncut : 1)); g.ncut = ncut;
  g.stream.adopt(stream, (size_t)nch * (size_t)gT);
  g.chunk_meta.adopt(chunk_meta, 2 * (size_t)nch);
  g.chunk_row.adopt(chunk_row, (size_t)nch); g.chunk_row.reset();
  g.nchunks = nch; g.T = gT; g.w = w;
  p->alt_S[cls] = S;
  if (verbose())
    std::fprintf(stderr, "libgcnspmm: slice set for k <= 32: %d slices of %d columns (the plan's own: %d)\n", S, w, p->slicing.S);
}

}  // namespace gcn

using namespace gcn;

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
  p->m = m; p->n = n; p->nnz = nnz; p->cu_count = cu;
  p->T = chunk_nnz ? chunk_nnz : auto_chunk_nnz(nnz, cu);
  p->nchunks = (int)(((long long)nnz + p->T - 1) / p->T);
  (void)hipGetDevice(&p->device);
  if (p->nchunks > 0) {
    if (p->chunk_row.alloc((size_t)p->nchunks) != hipSuccess) { delete p; return GCN_ERR_ALLOC; }
    // (synchronised: the header promises that rowptr_dev is only read during this call)
    if (launch_plan_chunk_rows(rowptr_dev, m, p->T, p->nchunks, p->chunk_row, (hipStream_t)stream) != hipSuccess ||
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

int gcn_spmm_plan_enable_slicing(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                                 const float* val, int32_t slices, void* stream) {
  if (!p || slices < -1 || slices > 1024) return GCN_ERR_INVALID_ARG;
  p->slicing = Slicing{};
  drop_streams(p);
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
  if (p->m == p->n && !p->factors.ready() &&
      (!autom || auto_slices(p->m, p->n, p->nnz, false) > 1)) {
    DevBuf<float> u;
    int ok = 0;
    if (u.alloc((size_t)p->n) == hipSuccess &&
        detect_rank1_values(rowptr, col, val, p->n, u, &ok, st) == hipSuccess && ok) {
      p->factors.u_row = std::move(u);
      p->factors.u_col = p->factors.u_row;
    }
  }
  // ... or depend on the row only / on the column only (r03): an unweighted adjacency (all ones), the row-normalised
  // D^-1 (A+I) of Kipf's pygcn, and its transpose (what the backward pass multiplies with) factor as u_row[r] * 1 and
  // 1 * u_col[c]; any shape.  Same 4-ulp check of every entry.
  if (!p->factors.ready() && (!autom || auto_slices(p->m, p->n, p->nnz, false) > 1)) {
    for (int mode = 1; mode <= 2 && !p->factors.ready(); ++mode) {
      Factors f;
      int ok = 0;
      if (f.u_row.alloc((size_t)p->m) == hipSuccess && f.u_col_own.alloc((size_t)p->n) == hipSuccess &&
          detect_constant_values(rowptr, col, val, p->m, p->n, p->nnz, mode, f.u_row, f.u_col_own, &ok, st) == hipSuccess && ok) {
        f.u_col = f.u_col_own;
        p->factors = std::move(f);
      }
    }
  }
  if (autom) slices = auto_slices(p->m, p->n, p->nnz, group_plan(p));
  if (slices <= 1) return GCN_OK;
  if ((long long)slices * p->m + 1 >= (1LL << 31)) return GCN_ERR_INVALID_ARG;
  const long long vm = (long long)slices * p->m;
  Slicing sl;
  if (sl.vrowptr.alloc((size_t)(vm + 1)) != hipSuccess || sl.vcol.alloc((size_t)p->nnz) != hipSuccess ||
      sl.vval.alloc((size_t)p->nnz) != hipSuccess || sl.vchunk_row.alloc((size_t)p->nchunks) != hipSuccess)
    return GCN_ERR_ALLOC;
  int sorted = 1;
  if (build_sliced_csr(rowptr, col, val, p->m, p->n, p->nnz, slices, sl.vrowptr, sl.vcol, sl.vval, &sorted, st) != hipSuccess)
    return GCN_ERR_HIP;
  if (!sorted) return autom ? GCN_OK : GCN_ERR_INVALID_ARG;   // needs column-sorted rows; auto mode just stays unsliced
  if (launch_plan_chunk_rows(sl.vrowptr, (int)vm, p->T, p->nchunks, sl.vchunk_row, st) != hipSuccess ||
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
  p->factors = Factors{};
  drop_streams(p);                                     // (the value-free streams exist only beside factors)
  if (!u_row && !u_col) {                                           // (null, null): forget the factors;
    build_sliced_streams(p, (hipStream_t)stream);                   // the sliced plan goes back to its value stream
    return GCN_OK;
  }
  if (!u_row || !u_col || !rowptr || (p->nnz > 0 && (!col || !val))) return GCN_ERR_INVALID_ARG;
  if (p->m == 0 || p->nnz == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  int ok = 0;
  if (verify_value_factors(rowptr, col, val, u_row, u_col, p->m, &ok, st) != hipSuccess) return GCN_ERR_HIP;
  if (!ok) {                                                        // some entry is not u_row[r]*u_col[c]
    build_sliced_streams(p, st);                                    // (the plan keeps working on its value stream)
    return GCN_ERR_NOT_FACTORED;
  }
  Factors f;
  if (f.u_row.alloc((size_t)p->m) != hipSuccess || f.u_col_own.alloc((size_t)p->n) != hipSuccess) return GCN_ERR_ALLOC;
  if (hipMemcpyAsync(f.u_row, u_row, sizeof(float) * (size_t)p->m, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemcpyAsync(f.u_col_own, u_col, sizeof(float) * (size_t)p->n, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return GCN_ERR_HIP;
  f.u_col = f.u_col_own;
  p->factors = std::move(f);
  // a slice count chosen automatically was chosen for a matrix WITH a value stream: choose again
  if (p->slices_auto && auto_slices(p->m, p->n, p->nnz, group_plan(p)) != p->slicing.S)
    return gcn_spmm_plan_enable_slicing(p, rowptr, col, val, -1, stream);
  build_sliced_streams(p, st);
  return GCN_OK;
}

int32_t gcn_spmm_plan_has_value_factors(const gcn_spmm_plan_t* p) { return p ? (p->factors.ready() ? 1 : 0) : -1; }

int gcn_spmm_plan_enable_panels(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col,
                                const float* val, int32_t mode, void* stream) {
  if (!p || mode < -1 || mode > 1) return GCN_ERR_INVALID_ARG;
  p->panels = Panels{};
  if (mode == 0 || p->nnz == 0 || p->m == 0) return GCN_OK;
  if (!rowptr || !col || !val) return GCN_ERR_INVALID_ARG;
  const int R = 128;
  const int npanels = (p->m + R - 1) / R;
  hipStream_t st = (hipStream_t)stream;
  Panels pn;
  DevBuf<int> pcnt;                               // in-window non-zeros of every panel
  if (pn.w0.alloc((size_t)npanels) != hipSuccess || pcnt.alloc((size_t)npanels) != hipSuccess) return GCN_ERR_ALLOC;
  unsigned long long inside = 0;
  if (panel_plan(rowptr, col, p->m, p->n, R, pn.w0, &inside, st, pcnt) != hipSuccess) return GCN_ERR_HIP;
  pn.coverage = (double)inside / (double)p->nnz;
  // automatic: only when at least half of the non-zeros are served from the staged tile
  if (!(mode == 1 || pn.coverage >= 0.5)) { p->panels.coverage = pn.coverage; return GCN_OK; }
  // Panels whose 128 x 512 window is dense enough leave the sparse formats altogether: a dense fp32 tile in
  // MFMA fragment order, contracted on the matrix cores (spmm_panel_dense_mfma_kernel); break-even against one
  // LDS read per entry is near 13 % density, the default threshold 25 %.
  {
    std::vector<int> cnt((size_t)npanels), slot((size_t)npanels, -1), ids;
    if (hipMemcpyAsync(cnt.data(), pcnt, sizeof(int) * (size_t)npanels, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return GCN_ERR_HIP;
    const double thr = kPanelMfmaDensity * (double)R * 512.0;
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
  if (panel_split(rowptr, col, val, pn.w0, p->m, R, pn.in_rowptr, pn.out_rowptr, nullptr, nullptr, nullptr, nullptr,
                       &nnz_in, st, pn.dense_slot) != hipSuccess)
    return GCN_ERR_HIP;
  int nnz_out = 0;                                     // (dense-tile entries are neither staged nor rest)
  if (hipMemcpyAsync(&nnz_out, pn.out_rowptr + p->m, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return GCN_ERR_HIP;
  if (pn.in_off.alloc((size_t)nnz_in) != hipSuccess || pn.in_val.alloc((size_t)nnz_in) != hipSuccess ||
      pn.out_col.alloc((size_t)nnz_out) != hipSuccess || pn.out_val.alloc((size_t)nnz_out) != hipSuccess)
    return GCN_ERR_ALLOC;
  if (panel_split(rowptr, col, val, pn.w0, p->m, R, pn.in_rowptr, pn.out_rowptr, pn.in_off, pn.in_val, pn.out_col,
                       pn.out_val, &nnz_in, st, pn.dense_slot, pn.adense) != hipSuccess)
    return GCN_ERR_HIP;
  pn.out_nnz = nnz_out;
  pn.out_T = auto_chunk_nnz(nnz_out, p->cu_count);
  pn.out_nchunks = (int)(((long long)nnz_out + pn.out_T - 1) / pn.out_T);
  if (pn.out_nchunks > 0) {
    if (pn.out_chunk_row.alloc((size_t)pn.out_nchunks) != hipSuccess) return GCN_ERR_ALLOC;
    if (launch_plan_chunk_rows(pn.out_rowptr, p->m, pn.out_T, pn.out_nchunks, pn.out_chunk_row, st) != hipSuccess)
      return GCN_ERR_HIP;
  }
  // (The out-of-window rest is what is LEFT of the matrix once the local structure is staged: short rows with columns all
  //  over the range.  Slicing it 8 ways was measured on the 240 k-vertex planted-partition graph — 2.49 -> 2.07 ms of
  //  kernels, then 0.23 ms for the reduction: no clear win — and is not built.)
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

}  // extern "C"
