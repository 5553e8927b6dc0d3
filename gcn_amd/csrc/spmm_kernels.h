// Internal interface between the C-ABI layer (plan_policy.cpp, plan_build.cpp, api_spmm.cpp, api_dropin.cpp) and the device code
// (spmm_kernels.hip).  Not installed; the public contract is include/gcn_spmm.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "philox.h"

namespace gcn {

struct SpmmArgs {
  const int*   rowptr;     // [m+1] device
  const int*   col;        // [nnz] device
  const float* val;        // [nnz] device
  const float* B;          // [n x k] device, row-major, ld = k
  float*       C;          // [m x k] device, row-major, ld = k
  float*       P;          // partial slab [2*nchunks x k] device
  const int*   chunk_row;  // [nchunks] device
  const float* bias;       // [k] or nullptr
  int relu;
  int nchunks, T, m, nnz, k;
  int n = 0x7fffffff;      // rows of B (columns of A); decides 32-bit buffer addressing
  int ldb = 0;             // row stride of B in floats, 0 = k (api_spmm.cpp pads rows to 128-byte lines for odd k)
  // drop-in mode (flexspmm symbol): nnz is only known on the device (nnz_dev =
  // &rowptr[m]); the kernels then derive nchunks and the value pointer themselves
  // and the host sizes its grids with the upper bound nchunks_grid.
  const int* nnz_dev;
  int nchunks_grid;
  // optional HIP events recorded on the launch stream right before / after the MAIN
  // kernel (not the fix-up) — the live kernel timing bench.py reports (null = off)
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  int tile_cols = 0;       // feature-column tile per pass: 0 auto, else 64 / 128 / 256
  int accumulate = 0;      // 1: C += A*B (C already holds another part of the product); epilogue after the add
  int col16 = 0;           // 1 (value-free pass only): a.col is a 16-bit stream of offsets inside the slice, see spmm_quad.hip
  int col16_S = 0, col16_w = 0, col16_start[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int valless = 0;         // 1: ignore val (every entry counts 1): the caller pre-scaled B and post-scales the rows
  int stream_rows = 0;     // 1: finished rows are written with non-temporal stores (partial rows of a sliced pass: read back
                           // only by the slice reduction; spmm_quad_kernel)
  int gather_width = 0;    // 64-column tile: non-zeros per gather instruction, 0 auto (4 when eligible), 1, 4
  // Empty rows are never written by the main kernels (a run of them would be walked by ONE wave, row by row: the
  // 6.5 s of profiles/r03d_hub_split_probe_rmat24.log) — they skip whole runs (next_nonempty_row) and a pass of its own
  // writes every empty row in parallel (launch_fill_empty_rows).  empty_rows: how many the matrix has when the caller
  // knows (0: the pass is skipped), -1: unknown (the pass always runs).
  int empty_rows = -1;
  int blocks_per_cu = 32;  // grid size in 256-thread blocks per CU (1..64).  Up to 8 (4 for the 108-VGPR
                           // quad kernel) are resident; more = later blocks start as earlier ones end, i.e.
                           // the hardware dispatcher balances the load (Reddit-shaped, k=128: 3.96 ms at 8,
                           // 3.68 ms at 32).  Less than the resident count leaves wave slots and registers
                           // free for a concurrent kernel (the RCCL all-gather).
};

hipError_t launch_plan_chunk_rows(const int* rowptr, int m, int T, int nchunks,
                                  int* chunk_row, hipStream_t s);
// rows r with rowptr[r] == rowptr[r+1]: *count_dev += their number (count_dev zeroed by the caller)
hipError_t launch_count_empty_rows(const int* rowptr, int m, int* count_dev, hipStream_t s);
// C[r, :] = act((accumulate ? C[r, :] : 0) + bias) for every EMPTY row r — what the main kernels leave out
hipError_t launch_fill_empty_rows(const int* rowptr, float* C, const float* bias, int relu, int accumulate, int m, int k,
                                  hipStream_t s);

// The main kernels' row walk has just stepped onto row r and found it empty (rowptr[r + 1] == pos): the first row
// i > r that holds an entry (rowptr[i + 1] > pos), or m.  Gallop, then bisect — wave-uniform scalar loads, O(log run)
// where the walk used to spend one dependent load and one store per empty row.
template <class RowPtr>
__device__ __forceinline__ int next_nonempty_row(RowPtr rowptr, int r, int m, int pos) {
  int lo = r, hi = r + 1;                               // row lo is empty; is row hi?
  unsigned step = 1;
  while (hi < m && rowptr[hi + 1] == pos) {
    lo = hi;
    step <<= 1;
    hi = (step >= (unsigned)(m - hi)) ? m : hi + (int)step;
  }
  while (hi - lo > 1) {                                 // row lo empty, row hi not (or hi == m)
    const int mid = lo + ((hi - lo) >> 1);
    if (rowptr[mid + 1] == pos) lo = mid; else hi = mid;
  }
  return hi;
}
hipError_t launch_spmm(const SpmmArgs& a, int cu_count, hipStream_t s);
hipError_t launch_gather_rows(float* dst, const float* src, const int* idx, int nrows, int k,
                              hipStream_t s);
// dst[r, 0:k] = src[r, 0:k], dst[r, k:ld] = 0 for r < rows (dst row stride ld >= k)
hipError_t launch_pad_rows(float* dst, const float* src, long long rows, int k, int ld, hipStream_t s,
                           const float* rowscale = nullptr);   // dst[r, :] = rowscale[r] * src[r, :] when given
// true when launch_spmm will run the four-per-gather kernel for these arguments
bool spmm_will_use_quad(const SpmmArgs& a);
// dst[r, 0:k] = act(src[r, 0:k] + bias), src row stride ld >= k
hipError_t launch_unpad_rows(float* dst, const float* src, const float* bias, int relu, long long rows, int k,
                             int ld, hipStream_t s, const int* guard = nullptr);   // guard: skip when *guard == 0
int pick_vec(int k, int tile_cols, const void* B, const void* C, const void* P);
void describe_main_kernel(const SpmmArgs& a, char* buf, size_t len);

// spmm_narrow.hip — k <= 32: several non-zeros per gather instruction
hipError_t launch_spmm_narrow(const SpmmArgs& a, int nblocks, bool epi, hipStream_t s);

// spmm_quad.hip — 64-column tile, four non-zeros per 16-byte-per-lane gather instruction
bool spmm_quad_eligible(const SpmmArgs& a);
int spmm_quad_lanes(int k);
hipError_t launch_spmm_quad(const SpmmArgs& a, int nblocks, bool epi, hipStream_t s);

// spmm_group.hip — value-free sliced main pass, four independent 16-lane row engines per wave
struct GroupArgs {
  const unsigned short* stream;  // [nchunks*T]: bits 0..14 column offset inside the slice, bit 15 = row end
  const float* vals = nullptr;   // [nchunks*T] matrix values in stream order (0 at padding entries); nullptr: every entry counts 1
  const int* chunk_meta;         // int2 [nchunks]: {2 * (virtual row holding entry c*T) + (it began in an earlier chunk),
                                 //                  first row of the chunk's slice in Bp}
  const float* Bp;               // scaled copy of B: slice s at rows [s*(w+1), (s+1)*(w+1)), row w all zero
  float* Cv;                     // partial outputs [S*m x k]
  float* P;                      // partial slab [2*nchunks x k]
  int nchunks, T, k, ldb;        // T = entries per chunk (of ONE 16-lane group), ldb = row stride of Bp (0 = k)
  const int* dyn = nullptr;      // drop-in flexspmm: device words {buffers recognised, chunk count, cut rows}; nchunks is then an
                                 // upper bound that sizes the grid (dropin_guard_kernel, api_dropin.cpp)
  long long table_rows = 0;      // rows of Bp, S * (w + 1): decides 32-bit or 64-bit (BIG) slice-base addressing
  int narrow8 = 1;               // k <= 32 on the eight-engine kernel (spmm_group8_kernel) when nchunks % 64 == 0
  int narrow12 = 1;              // 33 <= k <= 48, value-free: the five-engine kernel (spmm_group12_kernel)
  // (partial rows leave with non-temporal stores, finished rows of the value-free pass through the LDS ring, every
  //  64-column tile in one launch: the alternatives were measured in round 2 and are no longer built)
};
bool spmm_group_eligible(int k, int ldb, long long table_rows, const void* B, const void* C, const void* P);
bool spmm_group_needs_big(long long table_rows, int ldb);
hipError_t launch_spmm_group(const GroupArgs& a, hipStream_t s);
bool spmm_group8_applies(const GroupArgs& a);
bool spmm_group12_applies(const GroupArgs& a);
// the slice-major 15-bit stream the group kernel walks (S slices of width w = ceil(n/S) <= 32767): every virtual
// row gets >= 1 entry, every slice is padded to whole chunks and the total to a multiple of 32 chunks with
// entries that gather the slice's zero row.  Outputs: vrowptr_g [S*m+1] (caller-allocated; the fix-up pass needs
// it), *stream_out, *chunk_row_out [nchunks] and *chunk_meta_out (int2 [nchunks], see GroupArgs) — allocated here,
// the caller frees —, *nchunks_host.
// *fix_out (int4 [*nfix_host], allocated here): the rows whose pieces lie in the partial slab, see launch_group_fixup.
// vval + vals_out (optional): the slice-major values are laid out beside the stream (*vals_out [nchunks*T], 0 at padding).
hipError_t build_group_stream(const int* vrowptr, const int* vcol, int m, int n, int S, int T, int* vrowptr_g,
                              unsigned short** stream_out, int** chunk_row_out, int** chunk_meta_out,
                              int* nchunks_host, int** fix_out, int* nfix_host, hipStream_t st,
                              const float* vval = nullptr, float** vals_out = nullptr,
                              int** cutptr_out = nullptr, int** cutchunk_out = nullptr, int* ncut_host = nullptr);
// cutptr_out / cutchunk_out / ncut_host (optional; allocated here): the same cut rows keyed by OUTPUT row, for the slice
// reduction — pieces cutchunk[cutptr[r] .. cutptr[r+1]) (chunk numbers; the piece of chunk c is P[2c]) belong to row r.
// Cv[row, :] += the row's head pieces in the partial slab P, in chunk order, for every row of the list (k % 4 == 0)
// dyn (drop-in flexspmm): nfix is an upper bound, the count is dyn[2] and nothing runs when dyn[0] == 0
hipError_t launch_group_fixup(const int* fix, int nfix, const float* P, float* Cv, int k, hipStream_t s,
                              const int* dyn = nullptr);
// dst[(c / w)*(w+1) + c % w, :] = rowscale[c] * src[c, :] (rowscale nullptr: 1; row stride ld >= k, padding columns zero), row w of
// every slice zero: the layout GroupArgs::Bp describes
hipError_t launch_scale_rows_sliced(float* dst, const float* src, const float* rowscale, int n, int k, int ld,
                                    int S, int w, hipStream_t s);

// spmm_panel.hip — LDS-staged feature tiles per row panel (near-diagonal matrices)
// cnt_dev (optional, [panels]): in-window non-zeros of every panel
hipError_t panel_plan(const int* rowptr, const int* col, int m, int n, int R, int* w0_dev,
                      unsigned long long* inside_host, hipStream_t st, int* cnt_dev = nullptr);
// dense_slot (optional, [panels]): slot of the panel's dense tile, -1 for an ordinary panel; the in-window entries of
// dense panels go into `adense` (MFMA fragment order, 128 x 512 floats per slot, zeroed by the caller) instead
hipError_t panel_split(const int* rowptr, const int* col, const float* val, const int* w0_dev, int m,
                       int R, int* in_rowptr, int* out_rowptr, int* in_off, float* in_val,
                       int* out_col, float* out_val, int* nnz_in_host, hipStream_t st,
                       const int* dense_slot = nullptr, float* adense = nullptr);
hipError_t launch_panel_in(const int* in_rowptr, const int* in_off, const float* in_val, const float* B,
                           float* C, const int* panel_w0, int m, int n, int k, int R, int tile,
                           hipStream_t s, const int* dense_slot = nullptr);
hipError_t launch_panel_dense(const float* adense, const int* dense_panel, int ndense, const float* B, float* C,
                              const int* panel_w0, int m, int n, int k, int R, int tile, hipStream_t s);
hipError_t launch_panel_epilogue(float* C, const float* bias, int relu, int m, int k, hipStream_t s);

// reorder_device.hip — degree / RCM orderings and the CSR rewrite on the device (same integers as reorder.cpp)
hipError_t device_order_deg(const int* rowptr, const int* col, int n, int nnz, int which, int desc,
                            int* rank_out, hipStream_t st);
hipError_t device_order_rcm(const int* rowptr, const int* col, int n, int nnz, int* rank_out,
                            int* levels_out, hipStream_t st);
hipError_t device_csr_apply_rank(const int* rowptr, const int* col, const float* val, const int* rank, int n,
                                 int nnz, int* out_rowptr, int* out_col, float* out_val, int* vomp_out,
                                 int* bad_rank_host, hipStream_t st);

// rabbit_device.hip — Rabbit ordering by parallel incremental aggregation (Arai et al., IPDPS 2016; renumber.cu:328-330);
// rowptr/col: SYMMETRIC pattern (self-loops ignored); stats_host[4] = {communities, passes, retried, left top-level}
hipError_t device_order_rabbit(const int* rowptr, const int* col, int n, int nnz, int* rank_out_dev, int* community_out_dev,
                               long long* stats_host, hipStream_t st);

// slicing.hip
// 16-bit column stream of a sliced CSR (S <= 8, slice width <= 65 535): every slice padded to a multiple of T
// with 0xFFFF markers; vrowptr16 [S*m+1] (the last row of a slice owns its markers), col16 [*nnz16_host],
// start_host[S+1] = where each slice starts in the padded stream.  col16 is allocated here (caller frees).
hipError_t build_col16_stream(const int* vrowptr, const int* vcol, int m, int n, int S, int T, int* vrowptr16,
                              unsigned short** col16_out, int* nnz16_host, int* start_host, hipStream_t st);
hipError_t build_sliced_csr(const int* rowptr, const int* col, const float* val, int m, int n,
                            int nnz, int S, int* vrowptr, int* vcol, float* vval,
                            int* sorted_out, hipStream_t st);
// rows of the group kernels' stream cut by chunk ends, per output row (build_group_stream): the reduction adds the pieces
// P[2 * chunk[q]], q in [ptr[r], ptr[r+1]), behind the S partial rows of row r — the fix-up pass folded into the reduction
struct CutLists {
  const int* ptr = nullptr;      // [m+1]; null: no pieces to add
  const int* chunk = nullptr;    // [ptr[m]]
  const float* P = nullptr;      // the piece slab of the launch (GroupArgs::P)
};
hipError_t launch_slice_reduce(const float* Cv, float* C, const float* bias, int relu, int m, int S,
                               int k, hipStream_t st, int accumulate = 0, const float* rowscale = nullptr,
                               const DropoutSpec& drop = DropoutSpec{}, const int* guard = nullptr,   // guard: skip when *guard == 0
                               const float* outscale = nullptr, int gap_w = 0,    // pre-laid output: row r -> r + r / gap_w, times outscale[r]
                               const CutLists& cuts = CutLists{});
// dst[i] = dropout(src[i]) for i < total, mask from the flat index i (dst may be src)
hipError_t launch_dropout(float* dst, const float* src, long long total, const DropoutSpec& drop, hipStream_t st);
// values factor as u[r]*u[c]?  u_out[n] (device), *ok_host = 1 when every stored entry matches within 4 ulp
hipError_t verify_value_factors(const int* rowptr, const int* col, const float* val, const float* u_row,
                                const float* u_col, int m, int* ok_host, hipStream_t st);
hipError_t detect_rank1_values(const int* rowptr, const int* col, const float* val, int n, float* u_out,
                               int* ok_host, hipStream_t st);
// values that depend on the row only (mode 1: u_col = 1) or on the column only (mode 2: u_row = 1)?
hipError_t detect_constant_values(const int* rowptr, const int* col, const float* val, int m, int n, int nnz, int mode,
                                  float* u_row_out, float* u_col_out, int* ok_host, hipStream_t st);

}  // namespace gcn
