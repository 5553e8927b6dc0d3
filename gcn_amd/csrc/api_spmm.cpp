// api_spmm.cpp — the native C ABI of libgcnspmm.so: the SpMM launches (see include/gcn_spmm.h for the contract and the
// reference interfaces each entry point replaces).  The plan behind them is built in plan_build.cpp, the rules that pick
// kernel family, tile, stride and slice set are plan_policy.cpp; the reorderers' entry points are in api_reorder.cpp, the
// reference's own symbols (flexspmm, csr2tile, ...) in api_dropin.cpp.
#include "plan_policy.h"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>

#define GCN_VERSION_STR "0.3.0"

using namespace gcn;

namespace {

// Copy of B the sliced main pass gathers from: rows `ldb` floats apart (>= k, padding columns zero), scaled by
// u_col when `scaled`; one all-zero row more than B has (16-bit stream) or, for the group kernel, slice s at
// rows [s*(w+1), (s+1)*(w+1)) with row w of every slice zero.
int relay_B(gcn_spmm_plan* p, const SliceSet& ss, const float* B, int k, int ldb, bool scaled, bool group_layout, hipStream_t st) {
  if (group_layout) {                                  // (weighted pass: the same layout, rows not scaled)
    const size_t rows = (size_t)ss.table_rows();
    const int rc = grow(p->bpad, rows * (size_t)ldb);
    if (rc != GCN_OK) return rc;
    return launch_scale_rows_sliced(p->bpad, B, scaled ? p->factors.u_col : nullptr, p->n, k, ldb, ss.S, ss.g->w,
                                         st) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
  }
  const int rc = grow(p->bpad, ((size_t)p->n + 1) * (size_t)ldb);
  if (rc != GCN_OK) return rc;
  if (launch_pad_rows(p->bpad, B, p->n, k, ldb, st, scaled ? p->factors.u_col : nullptr) != hipSuccess ||
      hipMemsetAsync(p->bpad + (size_t)p->n * ldb, 0, sizeof(float) * (size_t)ldb, st) != hipSuccess)
    return GCN_ERR_HIP;
  return GCN_OK;
}

struct Epilogue {                                      // C = dropout(act(A*B + bias))
  const float* bias = nullptr;
  int relu = 0;
  DropoutSpec drop;
  // pre-laid output (gcn_spmm_csr_f32_prelaid; group kernels only): row r is written to row r + r / gap_w of the
  // destination and multiplied by outscale[r] — the slice-by-slice, column-scaled layout a following SpMM gathers from
  const float* outscale = nullptr;
  int gap_w = 0;
};

// b_ld: row stride of B in floats when the caller of this function has already re-laid it, 0 = k;
// b_scaled: that copy's rows are already scaled by u_col (value-free pass).
// *dropped: set when the dropout mask has been applied by a pass of this function (the slice reduction carries
// it); otherwise the caller applies it in place afterwards.
// ss: the slice set this call runs on when it takes a group kernel (pick_slice_set; the caller decided it once, for the
// re-laid copy of B and the kernels alike).
int spmm_impl(gcn_spmm_plan* p, const SliceSet& ss, const int32_t* rowptr, const int32_t* col, const float* val, const float* B,
              int b_ld, bool b_scaled, float* C, const Epilogue& epi, int32_t k, hipStream_t st, bool* dropped) {
  *dropped = false;
  if (grow(p->ws, ws_elems(p, k)) != GCN_OK) return GCN_ERR_ALLOC;
  const float* bias = epi.bias;
  const int relu = epi.relu ? 1 : 0;
  SpmmArgs a;
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
    const int ldb = padded_ldb(p->n, k);
    valless = valless_pays(p, k, ldb);
    weighted = !valless && weighted_pass(p, k, ldb);
    if (ldb != k || valless || weighted) {
      const int rc = relay_B(p, ss, B, k, ldb, valless, group_launch(p, valless, weighted), st);
      if (rc != GCN_OK) return rc;
      a.B = p->bpad;
      a.ldb = ldb;
    }
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;             // live timing of the main kernel (gcn_spmm_profile_begin)
  if (p->prof.armed()) { const auto pr = p->prof.next(); ev0 = pr.first; ev1 = pr.second; }
  a.blocks_per_cu = p->blocks_per_cu;
  a.gather_width = p->gather_width;
  Panels& pn = p->panels;
  if (pn.R > 0 && p->nnz > 0 && k > 32) {
    // A = A_in + A_out: the staged part from LDS (raw sums into C), then the rest accumulated by the
    // chunk kernel, which also carries the epilogue
    if (ev0 && hipEventRecord(ev0, st) != hipSuccess) return GCN_ERR_HIP;
    const int tiles = (k + 63) / 64;
    for (int t = 0; t < tiles; ++t) {
      if (launch_panel_in(pn.in_rowptr, pn.in_off, pn.in_val, B, C, pn.w0, p->m, p->n, k, pn.R, t, st, pn.dense_slot) != hipSuccess)
        return GCN_ERR_HIP;
      if (launch_panel_dense(pn.adense, pn.dense_panel, pn.ndense, B, C, pn.w0, p->m, p->n, k, pn.R, t, st) != hipSuccess)
        return GCN_ERR_HIP;
    }
    if (pn.out_nnz == 0) {
      if (launch_panel_epilogue(C, bias, relu, p->m, k, st) != hipSuccess) return GCN_ERR_HIP;
      if (ev1 && hipEventRecord(ev1, st) != hipSuccess) return GCN_ERR_HIP;
      return GCN_OK;
    }
    a.nchunks = pn.out_nchunks; a.nchunks_grid = pn.out_nchunks;
    a.T = pn.out_T; a.nnz = pn.out_nnz;
    a.empty_rows = -1;                                 // (rows whose entries all sit inside their window: not counted)
    a.ev_start = nullptr; a.ev_stop = ev1;
    a.rowptr = pn.out_rowptr; a.col = pn.out_col; a.val = pn.out_val;
    a.chunk_row = pn.out_chunk_row; a.accumulate = 1;
    a.tile_cols = p->tile_cols ? p->tile_cols : auto_tile_cols(p->n, k);
    return launch_spmm(a, p->cu_count, st) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
  }
  a.tile_cols = p->tile_cols ? p->tile_cols : (sliced ? 64 : auto_tile_cols(p->n, k));
  a.ev_start = ev0; a.ev_stop = ev1;
  if (!sliced) return launch_spmm(a, p->cu_count, st) == hipSuccess ? GCN_OK : GCN_ERR_HIP;

  // sliced: the slice-major virtual CSR (S*m rows) into the partial buffer, then the per-row reduction
  // over slices, which carries the whole epilogue (bias, ReLU, dropout mask, row factor)
  const Slicing& sl = p->slicing;
  const bool grp = group_launch(p, valless, weighted);
  const int S_run = grp ? ss.S : sl.S;
  if (grow(p->cv, (size_t)S_run * (size_t)p->m * (size_t)k) != GCN_OK) return GCN_ERR_ALLOC;
  *dropped = epi.drop.on();
  if (grp) {
    // four independent 16-lane row engines per wave on the 15-bit slice-major stream (spmm_group.hip)
    const GroupStream& G = *ss.g;
    GroupArgs ga;
    ga.stream = G.stream; ga.chunk_meta = G.chunk_meta;
    ga.vals = weighted ? G.vals.get() : nullptr;
    ga.Bp = a.B; ga.Cv = p->cv; ga.P = p->ws;
    ga.nchunks = G.nchunks; ga.T = G.T; ga.k = k; ga.ldb = a.ldb;
    ga.table_rows = ss.table_rows();
    ga.narrow8 = group8_enabled() ? 1 : 0;
    if (ev0 && hipEventRecord(ev0, st) != hipSuccess) return GCN_ERR_HIP;
    if (launch_spmm_group(ga, st) != hipSuccess) return GCN_ERR_HIP;
    if (ev1 && hipEventRecord(ev1, st) != hipSuccess) return GCN_ERR_HIP;
    // rows cut by chunk ends: their later pieces are added by the reduction itself (cut lists per output row), or —
    // GCN_AMD_GROUP_FUSED_FIXUP=0, or a plan without the lists — by a pass of their own in front of it
    // (Both passes on a high-priority stream of their own, and the main kernels of two plans taking turns, were built and
    //  measured for the two planes of the multi-GPU layer in r04: 0.417 / 0.450 ms against 0.377 — DESIGN §6; removed again.)
    CutLists cuts;
    if (group_fused_fixup() && G.cutptr) { cuts.ptr = G.cutptr; cuts.chunk = G.cutchunk; cuts.P = p->ws; }
    else if (launch_group_fixup(G.fix, G.nfix, p->ws, p->cv, k, st) != hipSuccess) return GCN_ERR_HIP;
    return launch_slice_reduce(p->cv, C, bias, relu, p->m, S_run, k, st, 0, weighted ? nullptr : p->factors.u_row.get(),
                               epi.drop, nullptr, epi.outscale, epi.gap_w, cuts) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
  }
  a.rowptr = sl.vrowptr; a.col = sl.vcol; a.val = sl.vval; a.chunk_row = sl.vchunk_row;
  a.C = p->cv; a.m = sl.S * p->m; a.bias = nullptr; a.relu = 0;
  a.empty_rows = sl.empty_vrows;
  a.stream_rows = 1;                                   // partial rows leave with non-temporal stores (3.667 -> 3.646 ms, profiles/r02zi_*)
  const float* rowscale = nullptr;
  if (valless) {                                                          // B was scaled by u_col above
    a.valless = 1; a.val = nullptr; rowscale = p->factors.u_row;
    const Col16Stream& c16 = p->col16;
    if (c16.ready()) {                                                    // 16-bit column stream, slice-aligned chunks
      a.rowptr = c16.vrowptr16; a.col = reinterpret_cast<const int*>(c16.vcol16.get()); a.chunk_row = c16.vchunk_row16;
      a.nnz = c16.nnz16; a.nchunks = a.nchunks_grid = c16.nchunks16;
      a.col16 = 1; a.col16_S = sl.S; a.col16_w = (p->n + sl.S - 1) / sl.S;
      a.empty_rows = -1;                                                  // (its own row pointer: not counted)
      for (int i = 0; i < 9; ++i) a.col16_start[i] = c16.start16[i];
    }
  }
  if (launch_spmm(a, p->cu_count, st) != hipSuccess) return GCN_ERR_HIP;
  return launch_slice_reduce(p->cv, C, bias, relu, p->m, sl.S, k, st, 0, rowscale, epi.drop) == hipSuccess
             ? GCN_OK : GCN_ERR_HIP;
}

}  // namespace

extern "C" {

const char* gcn_version(void) { return GCN_VERSION_STR; }

int gcn_spmm_csr_f32_epilogue(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col, const float* val,
                              const float* B, float* C, const float* bias, int32_t relu, float dropout_p,
                              uint64_t seed, uint64_t offset, int32_t k, void* stream) {
  if (!p || k < 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) return GCN_ERR_INVALID_ARG;
  if (p->m == 0 || k == 0) return GCN_OK;
  if (!C || !rowptr || (p->nnz > 0 && (!col || !val || !B))) return GCN_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  // which slice set does this call run on?  (decided here, once, for the re-laid copy of B and the kernels alike)
  const bool odd = odd_width_detour(p, k);
  int ld_call = odd ? ((k + 3) / 4 * 4 + 31) / 32 * 32 : padded_ldb(p->n, k);
  bool relay48 = false;
  const SliceSet ss = pick_slice_set(p, odd ? (k + 3) / 4 * 4 : k, &ld_call, &relay48, /*build=*/true, rowptr, col, val, st);
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
    if ((rc = relay_B(p, ss, B, k, ldb, scaled, group_launch(p, scaled, weighted), st)) != GCN_OK) return rc;
    if ((rc = spmm_impl(p, ss, rowptr, col, val, p->bpad, ldb, scaled, p->cpad, Epilogue{}, kp, st, &dropped)) != GCN_OK) return rc;
    if (launch_unpad_rows(C, p->cpad, bias, epi.relu, p->m, k, kp, st) != hipSuccess) return GCN_ERR_HIP;
    dropped = false;
  } else if (relay48) {                                // k = 44 on the five-engine kernel: rows of 48 floats, scaled, slice by slice
    if ((rc = relay_B(p, ss, B, k, ld_call, true, true, st)) != GCN_OK) return rc;
    if ((rc = spmm_impl(p, ss, rowptr, col, val, p->bpad, ld_call, true, C, epi, k, st, &dropped)) != GCN_OK) return rc;
  } else {
    if ((rc = spmm_impl(p, ss, rowptr, col, val, B, 0, false, C, epi, k, st, &dropped)) != GCN_OK) return rc;
  }
  if (epi.drop.on() && !dropped)                       // no epilogue pass carried the mask: one pass in place
    return launch_dropout(C, C, (long long)p->m * k, epi.drop, st) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
  return GCN_OK;
}

// Everything a k-wide call builds lazily, built NOW (the narrow slice set of k <= 32: device allocations and stream
// synchronisation that a stream capture would refuse): afterwards the first k-wide call only grows workspaces.
int gcn_spmm_plan_prepare_width(gcn_spmm_plan_t* p, const int32_t* rowptr, const int32_t* col, const float* val, int32_t k,
                                void* stream) {
  if (!p || k <= 0) return GCN_ERR_INVALID_ARG;
  if (p->nnz > 0 && (!rowptr || !col || !val)) return GCN_ERR_INVALID_ARG;
  const bool odd = odd_width_detour(p, k);
  int ld_call = odd ? ((k + 3) / 4 * 4 + 31) / 32 * 32 : padded_ldb(p->n, k);
  bool relay48 = false;
  (void)pick_slice_set(p, odd ? (k + 3) / 4 * 4 : k, &ld_call, &relay48, /*build=*/true, rowptr, col, val, (hipStream_t)stream);
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
  // (the pre-laid layout is the plan's own slice set, whatever the width)
  return spmm_impl(p, own_slice_set(p), rowptr, col, val, Bp, ld, /*b_scaled=*/true, out, epi, k, (hipStream_t)stream, &dropped);
}

int gcn_dropout_f32(float* dst, const float* src, int64_t count, float dropout_p, uint64_t seed, uint64_t offset,
                    void* stream) {
  if (count < 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) return GCN_ERR_INVALID_ARG;
  if (count == 0) return GCN_OK;
  if (!dst || !src) return GCN_ERR_INVALID_ARG;
  DropoutSpec d;
  d.p = dropout_p; d.seed = seed; d.offset = offset;
  if (!d.on()) {
    if (dst == src) return GCN_OK;
    return hipMemcpyAsync(dst, src, sizeof(float) * (size_t)count, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess
               ? GCN_OK : GCN_ERR_HIP;
  }
  return launch_dropout(dst, src, (long long)count, d, (hipStream_t)stream) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
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
  const int cu = cu_count_cached();
  if (cu <= 0) return GCN_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(g_plan_mu);
  gcn_spmm_plan* p = scratch_plan(stream);
  if (!p) return GCN_ERR_ALLOC;
  p->m = m; p->n = n; p->nnz = nnz; p->cu_count = cu;
  p->T = auto_chunk_nnz(nnz, cu);
  p->nchunks = (int)(((long long)nnz + p->T - 1) / p->T);
  if (p->chunk_row.grow((size_t)p->nchunks) != hipSuccess) return GCN_ERR_ALLOC;
  if (m == 0 || k == 0) return GCN_OK;
  if (launch_plan_chunk_rows(rowptr, m, p->T, p->nchunks, p->chunk_row, (hipStream_t)stream) != hipSuccess) return GCN_ERR_HIP;
  if (p->ws.grow(ws_elems(p, k)) != hipSuccess) return GCN_ERR_ALLOC;
  SpmmArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val; a.B = B; a.C = C; a.P = p->ws;
  a.chunk_row = p->chunk_row; a.bias = nullptr; a.relu = 0;
  a.nchunks = p->nchunks; a.T = p->T; a.m = m; a.nnz = nnz; a.k = k; a.n = n;
  a.nnz_dev = nullptr; a.nchunks_grid = p->nchunks;
  a.tile_cols = auto_tile_cols(n, k);
  return launch_spmm(a, cu, (hipStream_t)stream) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

int gcn_gather_rows_f32(float* dst, const float* src, const int32_t* idx, int32_t nrows, int32_t k,
                        void* stream) {
  if (nrows < 0 || k < 0) return GCN_ERR_INVALID_ARG;
  if (nrows == 0 || k == 0) return GCN_OK;
  if (!dst || !src || !idx || dst == src) return GCN_ERR_INVALID_ARG;
  return launch_gather_rows(dst, src, idx, nrows, k, (hipStream_t)stream) == hipSuccess
             ? GCN_OK : GCN_ERR_HIP;
}

}  // extern "C"
