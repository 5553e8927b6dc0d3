/*
 * gcn_spmm.h — C ABI of libgcnspmm.so, the MI355X (gfx950) GCN aggregation library.
 *
 * Everything here is `extern "C"`, plain pointers and sizes; no torch types.
 * Two groups of entry points:
 *
 *  (1) NATIVE API (gcn_*): what the Python host layer (gcn_amd/) binds.  Explicit
 *      stream, explicit status codes, cached plan object.
 *  (2) DROP-IN API: the exact symbols / signatures the reference's ctypes call
 *      sites bind (pygcn/gcn6.py:21-25).  They are exported both from
 *      libgcnspmm.so and from five tiny shared objects that carry the reference's
 *      file names (flexspmm.so, cuspmm.so, tile.so, permutate.so, renumber.so),
 *      so that gcn6.py loads them unchanged.  See INTEGRATION.md.
 *
 * Each declaration cites the reference interface it replaces (file:line under
 * the reference tree guohaoqiang/gcn @ v1).
 */
#ifndef GCN_SPMM_H
#define GCN_SPMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* status codes (native API).  The reference has no error convention at all   */
/* (void functions, cuspmm.cu:3-21 only prints) — the drop-in symbols keep    */
/* `void`, print ONE line to stderr and return with the caller's outputs      */
/* untouched (section (2) below): never silent, never fatal.                  */
/* ------------------------------------------------------------------------- */
#define GCN_OK                 0
#define GCN_ERR_INVALID_ARG    1
#define GCN_ERR_HIP            2
#define GCN_ERR_NO_DEVICE      3
#define GCN_ERR_CAPACITY       4   /* caller buffer too small for the packed plan */
#define GCN_ERR_ALLOC          5
#define GCN_ERR_NOT_FACTORED   6   /* set_value_factors: some stored entry is not u_row[r]*u_col[c]; the plan keeps its value stream */
#define GCN_ERR_INTERNAL       7   /* a consistency guard inside the library tripped (gcn_order_rabbit_device: stats_host[4..7]); outputs not written */

const char* gcn_status_string(int status);
/* library version, "major.minor.patch" */
const char* gcn_version(void);
/* number of compute units of the current device (replaces the per-call
 * cudaGetDeviceProperties of flexspmm.cu:506-508 / tile.cu:118-122); <0 on error */
int gcn_device_cu_count(void);

/* ------------------------------------------------------------------------- */
/* (1a) SpMM plan + launch:   C[m x k] = A[m x n, CSR fp32] * B[n x k]        */
/*      replaces  cuspmm()   cuspmm.cu:23-68   (the math: op(A)=A, op(B)=B,   */
/*                row-major B and C, alpha=1, beta=0)                         */
/*      and       flexspmm() flexspmm.cu:499-544 (the launcher + 5 kernels)   */
/* ------------------------------------------------------------------------- */
typedef struct gcn_spmm_plan gcn_spmm_plan_t;

/* Build the per-graph schedule (equal-nnz chunks + first row of each chunk) on
 * the device.  `rowptr_dev` is the int32 CSR row pointer [m+1] in device memory;
 * it is only read during this call (the call synchronises `stream` before it returns).  `chunk_nnz` = 0 picks a size automatically
 * (multiple of 64).  The plan owns a small device buffer and a grow-only
 * workspace for the partial sums of rows that straddle chunk boundaries. */
int gcn_spmm_plan_create(gcn_spmm_plan_t** plan, const int32_t* rowptr_dev,
                         int32_t m, int32_t n, int32_t nnz, int32_t chunk_nnz,
                         void* stream);
int gcn_spmm_plan_destroy(gcn_spmm_plan_t* plan);
/* introspection (tests, bench) */
int32_t gcn_spmm_plan_num_chunks(const gcn_spmm_plan_t* plan);
int32_t gcn_spmm_plan_chunk_nnz(const gcn_spmm_plan_t* plan);
/* bytes of workspace the plan needs for feature width k */
size_t  gcn_spmm_plan_workspace_bytes(const gcn_spmm_plan_t* plan, int32_t k);

/* C = A*B.  All pointers are device pointers; C is fully overwritten (no need
 * to pre-zero, unlike gcn6.py:37).  Asynchronous on `stream` (NULL = the legacy
 * default stream, which is what the reference launches on, flexspmm.cu:512).
 * Deterministic: no atomics, each row is summed in CSR order within a chunk
 * and chunk partials are added in chunk order.
 * Stream capture: the first call for a width allocates the plan's workspaces
 * (and builds the streams of a sliced plan); every later call with the same width
 * only enqueues kernels on `stream`, so it can be captured in a HIP graph and
 * replayed on new operand contents (tests/test_spmm_gpu.py,
 * test_spmm_can_be_captured_in_a_hip_graph_and_replayed). */
int gcn_spmm_csr_f32(gcn_spmm_plan_t* plan,
                     const int32_t* rowptr_dev, const int32_t* col_dev,
                     const float* val_dev, const float* B_dev, float* C_dev,
                     int32_t k, void* stream);

/* Same, plus a fused epilogue  C = act(A*B + bias)  (bias may be NULL;
 * relu = 0/1).  Covers gcn6.py:141-142,245 (bias add, ReLU) — SURVEY §8(f).1 */
int gcn_spmm_csr_f32_bias_relu(gcn_spmm_plan_t* plan,
                     const int32_t* rowptr_dev, const int32_t* col_dev,
                     const float* val_dev, const float* B_dev, float* C_dev,
                     const float* bias_dev, int32_t relu,
                     int32_t k, void* stream);

/* The full fused epilogue of a GCN layer (pygcn/gcn6.py:141-142, 245-246: bias add, ReLU and
 * dropout are three separate PyTorch ops there):  C = dropout(act(A*B + bias)).  dropout_p in [0, 1) is the drop
 * probability (0 = none); kept elements are scaled by 1/(1-p).  The mask is NOT stored: element i = r*k + c is kept
 * iff word (i mod 4) of Philox4x32-10(counter = (i / 4, offset), key = seed) >= p * 2^32, so the backward pass
 * regenerates it with gcn_dropout_f32 on the gradient (same p, seed, offset) and every kernel family agrees.  Where
 * the plan has an epilogue pass (column slicing: the slice reduction) the mask rides in it; otherwise one in-place
 * pass over C applies it.  (torch's own dropout stream cannot be reproduced: the semantic — Bernoulli(1-p), scaled
 * — is what is matched.) */
int gcn_spmm_csr_f32_epilogue(gcn_spmm_plan_t* plan,
                     const int32_t* rowptr_dev, const int32_t* col_dev,
                     const float* val_dev, const float* B_dev, float* C_dev,
                     const float* bias_dev, int32_t relu, float dropout_p, uint64_t seed, uint64_t offset,
                     int32_t k, void* stream);
/* dst[i] = dropout(src[i]) for i < count with the mask defined above (dst may equal src): the backward of the
 * fused epilogue, and the forward where no fused pass exists. */
int gcn_dropout_f32(float* dst_dev, const float* src_dev, int64_t count, float dropout_p, uint64_t seed,
                    uint64_t offset, void* stream);

/* Feature-column tile per kernel pass: 0 = automatic, else 64, 128 or 256 columns.  A k-wide
 * SpMM runs as ceil(k/tile) back-to-back passes, each gathering only its column slice of B
 * (smaller per-pass working set -> more of it stays in L2 / Infinity Cache). */
int gcn_spmm_plan_set_tile_cols(gcn_spmm_plan_t* plan, int32_t cols);
/* Grid size of the main kernel in 256-thread blocks per CU, 1..64 (default 32).  At most 8 blocks
 * (4 for the four-per-gather kernel) are resident on a CU; the default oversubscribes on purpose so
 * that the hardware dispatcher hands out the remaining blocks as earlier ones finish (load balance:
 * 3.96 -> 3.68 ms on the Reddit-shaped graph).  A value below the resident count leaves wave slots
 * and registers free so that a kernel on another stream — the RCCL all-gather of the multi-GPU
 * path — can run beside the SpMM instead of behind it. */
int gcn_spmm_plan_set_blocks_per_cu(gcn_spmm_plan_t* plan, int32_t blocks);
/* Non-zeros per gather instruction of the 64-column-tile kernel: 0 = automatic — 4 (16 lanes x 16 bytes
 * per feature row, spmm_quad.hip) when k % 4 == 0 (odd widths are rounded up internally), operands are
 * 16-byte aligned, n < 2^24, n*k*4 < 4 GiB AND rows are long (>= 48 non-zeros per row, or per virtual
 * row when sliced: every finished row costs that layout a cross-lane reduction), else 1; 1 = always the
 * one-row-per-instruction kernel (52 VGPRs: leaves more room for a concurrent kernel); 4 = the
 * four-per-gather kernel wherever its layout applies, whatever the row length.  Any other value:
 * GCN_ERR_INVALID_ARG. */
int gcn_spmm_plan_set_gather_width(gcn_spmm_plan_t* plan, int32_t nz_per_gather);
/* number of main-kernel launches (column passes) one k-wide SpMM issues with the current tile */
int32_t gcn_spmm_plan_num_passes(const gcn_spmm_plan_t* plan, int32_t k);
/* name (as rocprofv3 --kernel-trace prints it, without the argument list) of the main kernel a
 * k-wide SpMM on this plan launches with the current settings and 16-byte aligned operands;
 * epilogue != 0: the bias/ReLU variant.  For benchmarks that report which kernel they timed. */
int gcn_spmm_plan_main_kernel(const gcn_spmm_plan_t* plan, int32_t k, int32_t epilogue, char* buf, int32_t buflen);

/* XCD-aware column slicing (optional, off by default).  Builds, on the device, a slice-major copy
 * of the matrix (`slices` equal column ranges; virtual row s*m+r = the part of row r in slice s)
 * that later gcn_spmm_csr_f32* calls on this plan use instead of the caller's col/val: every XCD
 * then gathers from only ~slices/8 column slices of B, sized to stay in its 4 MiB L2, and a
 * reduction over slices (in slice order, deterministic) produces C.  Needs column-sorted rows
 * (GCN_ERR_INVALID_ARG otherwise); slices = 0/1 turns it off; slices = -1 picks the count from
 * (m, n, nnz) — off for low-degree graphs and tables that fit an L2 anyway, 2..8 otherwise — and silently stays off for
 * unsorted rows.  The matrix passed here must be the
 * one the plan was created for.  Costs one extra copy of col/val plus slices*m*k floats.
 * Values that factor as u_row[r]*u_col[c] — found here on the device, every entry checked to 4 ulp: the symmetric GCN
 * normalisation D^-1/2 (A+I) D^-1/2 (square matrices), values that depend on the row only (an unweighted adjacency,
 * the row-normalised D^-1 (A+I)) or on the column only (its transpose) — let the sliced pass run WITHOUT its value
 * stream (B scaled by u_col in the copy it gathers from, rows scaled by u_row in the reduction; results within the
 * 1e-5 contract).  gcn_spmm_plan_set_value_factors hands factors over for matrices that cannot see them (row blocks). */
int gcn_spmm_plan_enable_slicing(gcn_spmm_plan_t* plan, const int32_t* rowptr_dev,
                                 const int32_t* col_dev, const float* val_dev,
                                 int32_t slices, void* stream);
int32_t gcn_spmm_plan_num_slices(const gcn_spmm_plan_t* plan);
/* Slices of the slice set a k-wide call runs on when it is NOT the plan's own (0: it is).  Value-free plans whose slice
 * count was automatic cut the matrix again at their first call with k <= 32: a table row is 128 bytes there, half as
 * many slices fill an L2, and the partial rows — whose cost goes with the slice count — halve (Reddit-shaped: 8
 * instead of 15). */
int32_t gcn_spmm_plan_narrow_slices(const gcn_spmm_plan_t* plan, int32_t k);
/* Build NOW whatever a k-wide call of this plan would build at its first use (the narrow slice set above: device
 * allocations and a stream synchronisation), e.g. before a stream capture whose first k-wide call must only enqueue
 * kernels.  The matrix arrays must be the ones the plan was created for.  Idempotent; GCN_OK also when nothing is to build. */
int gcn_spmm_plan_prepare_width(gcn_spmm_plan_t* plan, const int32_t* rowptr_dev, const int32_t* col_dev,
                                const float* val_dev, int32_t k, void* stream);

/* Rank-1 values.  When every stored value is u_row[r] * u_col[c] — the GCN normalisation
 * D^-1/2 (A+I) D^-1/2 has u = D^-1/2 — the sliced main pass runs WITHOUT its value stream (5 % of the
 * bytes it moves, and they are its time): B is gathered from a copy whose rows were scaled by u_col and
 * the finished rows are scaled by u_row in the slice reduction; results stay within the 1e-5 contract
 * (each term carries one more rounding).  gcn_spmm_plan_enable_slicing detects this by itself for SQUARE
 * matrices with a stored diagonal (u = sqrt(diag)); for anything else — e.g. a row block of such a
 * matrix with renumbered columns — hand the factors over here (device arrays [m] and [n], copied).
 * Every entry is checked (4 ulp): GCN_ERR_NOT_FACTORED if one does not factor (the plan then keeps working on
 * its value stream; every other status is a real failure).  (NULL, NULL) forgets the factors.  The matrix arrays must be the ones the plan was created for. */
int gcn_spmm_plan_set_value_factors(gcn_spmm_plan_t* plan, const int32_t* rowptr_dev, const int32_t* col_dev,
                                    const float* val_dev, const float* u_row_dev, const float* u_col_dev,
                                    void* stream);
int32_t gcn_spmm_plan_has_value_factors(const gcn_spmm_plan_t* plan);   /* 1 / 0 */

/* LDS-staged row panels (optional): for matrices whose non-zeros sit near the diagonal (community
 * graphs after Rabbit / RCM / Gorder renumbering) a workgroup stages the feature rows of its panel's
 * column window (512 rows x 64 columns = 128 KiB of LDS) once and sums the in-window non-zeros
 * from LDS; the matrix is split on the device into that staged part and the rest, which the chunk
 * kernel adds in accumulate mode (the split copy costs one extra copy of col/val).  mode 0 = off, 1 = on, -1 = on iff at least half of the
 * non-zeros fall inside their panel's window (measured here, on the device).  When on it is used for
 * k > 32 and takes precedence over slicing; results stay deterministic (rows are summed in CSR order). */
int gcn_spmm_plan_enable_panels(gcn_spmm_plan_t* plan, const int32_t* rowptr_dev,
                                const int32_t* col_dev, const float* val_dev, int32_t mode,
                                void* stream);
int32_t gcn_spmm_plan_panel_rows(const gcn_spmm_plan_t* plan);       /* rows per panel, 0 = off */
/* Panels whose 128 x 512 window holds >= 25 % non-zeros are stored as dense fp32 tiles and contracted on the matrix
 * cores (v_mfma_f32_32x32x2_f32: exact fp32) instead of entry by entry — the BASELINE north_star's "MFMA on the
 * dense row-panel x feature-tile product where nnz-per-panel forms a dense contraction"; a window that holds a
 * non-finite feature value falls back to entry-by-entry sums so that NaN/Inf never reach rows that do not reference
 * them.  Returns how many panels of the plan take that path (0 when panels are off). */
int32_t gcn_spmm_plan_dense_panels(const gcn_spmm_plan_t* plan);
double gcn_spmm_plan_panel_coverage(const gcn_spmm_plan_t* plan);    /* fraction of nnz inside windows */

/* Live kernel timing for bench.py: between _begin and _end every gcn_spmm_csr_f32*
 * call on this plan records a HIP event pair on its launch stream right around the
 * MAIN kernel passes of one SpMM (up to `capacity` SpMM calls).  _end synchronises the events and returns
 * the per-launch durations in milliseconds.  (The reference times with CUDA events
 * from Python, pygcn/perf/dmk.py:71-117.) */
int gcn_spmm_profile_begin(gcn_spmm_plan_t* plan, int32_t capacity);
int gcn_spmm_profile_end(gcn_spmm_plan_t* plan, float* ms_out, int32_t* count_out);

/* One-shot convenience: the schedule is rebuilt on the device at every call into scratch buffers that belong to
 * the calling (device, stream) pair, so concurrent calls on different streams or devices never share state.
 * This is the body of the drop-in `cuspmm` symbol. */
int gcn_spmm_csr_f32_oneshot(const int32_t* rowptr_dev, const int32_t* col_dev,
                     const float* val_dev, const float* B_dev, float* C_dev,
                     int32_t m, int32_t n, int32_t nnz, int32_t k, void* stream);

/* Chains of aggregations (H <- A·H per layer: the multi-GPU pipeline, SGC-style models) without the per-call copy
 * of B.  A sliced plan whose values factor (u_row[r]*u_col[c]) gathers from B' = diag(u_col)·B laid out slice by
 * slice: slice s (columns [s*w, (s+1)*w)) at rows [s*(w+1), (s+1)*(w+1)) of a [table_rows x ld] array, row w of every
 * slice all zero.  gcn_spmm_csr_f32 builds that copy at every call (one pass over B, O(n*k) whatever the matrix
 * holds); `_prelaid_layout` describes it (GCN_ERR_INVALID_ARG when a k-wide SpMM of this plan does not take that
 * pass: k % 4 != 0, no factors, no slicing, ...), and `_prelaid` multiplies with a B' the caller already holds:
 *   out[r + r / out_gap, 0:k] = out_scale[r] * (A·B)[r, 0:k]          (out_gap > 0; out_scale NULL = 1)
 * i.e. the result is written straight INTO the B' layout of a consumer whose slices hold `out_gap` of these rows
 * each (the zero row behind every slice is never touched) and already carries the consumer's column factor — the
 * next layer's call needs no copy at all.  out_gap = 0 writes a plain [m x k] result.  The reference has no
 * counterpart: its flexspmm re-reads B as handed over on every call (flexspmm.cu:499-544). */
int gcn_spmm_plan_prelaid_layout(const gcn_spmm_plan_t* plan, int32_t k, int32_t* slices, int32_t* slice_cols,
                                 int64_t* table_rows, int32_t* ld_floats);
int gcn_spmm_csr_f32_prelaid(gcn_spmm_plan_t* plan, const int32_t* rowptr_dev, const int32_t* col_dev,
                             const float* val_dev, const float* Bp_dev, float* out_dev, const float* out_scale_dev,
                             int32_t out_gap, int32_t k, void* stream);

/* The column-slice count gcn_spmm_plan_enable_slicing(-1) would choose for an m x n matrix with nnz entries
 * (value_free: its values factor, the group kernels run) — host arithmetic; 0 = no slicing.  A caller that must
 * align its own buffers with the slices (the multi-GPU row shards: slices = whole fractions of a rank's rows) asks
 * here first and then passes an explicit count. */
int32_t gcn_spmm_auto_slices(int64_t m, int64_t n, int64_t nnz, int32_t value_free);

/* How the group kernels (csrc/spmm_group.hip) address a slice-by-slice copy of B with `table_rows` rows of
 * `ld_floats` floats: 0 = 32-bit byte offsets (table below 4 GiB and 2^24 rows), 1 = the slice base is added in
 * 64 bits (any size), -1 = not served by them (rows of 128 KiB or more in a table that needs 64 bits, or a
 * stride that is no multiple of 4 floats): such a plan runs the four-per-gather / one-per-gather kernels.
 * Host arithmetic only; the reference has no counterpart (its `int addr = row*k`, flexspmm.cu:67, wraps). */
int32_t gcn_spmm_group_addressing(int64_t table_rows, int32_t ld_floats);

/* ------------------------------------------------------------------------- */
/* (1b) feature-row permutation  dst[r,:] = src[idx[r],:]                     */
/*      replaces flexspmm_v9_permuteX / put_back / permutate()                */
/*      permutate.cu:3-59                                                     */
/* ------------------------------------------------------------------------- */
int gcn_gather_rows_f32(float* dst_dev, const float* src_dev, const int32_t* idx_dev,
                        int32_t nrows, int32_t k, void* stream);

/* ------------------------------------------------------------------------- */
/* (1c) host-side reorderers (CPU, bit-exact with the reference's integer      */
/*      vectors).  rank_out[old] = new, n entries, computed from a CSR graph   */
/*      (one directed edge per stored entry, self-loops included —            */
/*      edgelist.cuh:16-25).                                                  */
/*      order_deg   order_deg.cu:19-45    which: 0 total(in+out) 1 out 2 in   */
/*      order_rcm   order_rcm.cu:15-33 + algo_bfs.cu:11-39                    */
/*      gorder      order_gorder.cu:13-143 + unitheap.cu (RCM∘Gorder)         */
/* ------------------------------------------------------------------------- */
int gcn_order_deg(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t nnz,
                  int32_t which, int32_t desc, int64_t* rank_out);
int gcn_order_rcm(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t nnz,
                  int32_t directed, int64_t* rank_out);
int gcn_order_gorder(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t nnz,
                     int32_t window, int64_t* rank_out);
/* rabbit (renumber.cu:319-520: serial modularity merging, rounds in degree order) WITHOUT the CSR rewrite the
 * drop-in symbol does: vomp_out[new] = old (bit-exact with the reference) and, optionally, the surviving top-level
 * vertex every vertex ended under (community_out; what the parallel version's quality is measured against). */
int gcn_order_rabbit(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t nnz, int32_t* vomp_out,
                     int32_t* community_out);
/* CSR rewrite under a rank: rows/cols relabelled, each row's columns sorted
 * ascending with values carried along (renumber.cu:190-217); vomp_out[new]=old. */
int gcn_csr_apply_rank(int32_t* rowptr, int32_t* col, float* vals, int32_t n, int32_t nnz,
                       const int64_t* rank, int32_t* vomp_out);

/* ------------------------------------------------------------------------- */
/* (1d) the same orderings ON THE DEVICE: device pointers in,   */
/*      int32 rank_out[old] = new on the device, identical integers to the    */
/*      host versions above.  order_rcm_device works on the symmetrised       */
/*      pattern A ∪ Aᵀ (= gcn_order_rcm with directed = 0; for symmetric      */
/*      patterns also directed = 1): component labelling + multi-source       */
/*      level-synchronous BFS whose per-level radix sort reproduces the       */
/*      serial queue order of algo_bfs.cu:11-39 exactly (reorder_device.hip). */
/*      csr_apply_rank_device = gcn_csr_apply_rank out of place (outputs must */
/*      not alias inputs); GCN_ERR_INVALID_ARG if rank is not a permutation.  */
/*      All three synchronise the stream before returning.                    */
/* ------------------------------------------------------------------------- */
int gcn_order_deg_device(const int32_t* rowptr_dev, const int32_t* col_dev, int32_t n, int32_t nnz,
                         int32_t which, int32_t desc, int32_t* rank_out_dev, void* stream);
int gcn_order_rcm_device(const int32_t* rowptr_dev, const int32_t* col_dev, int32_t n, int32_t nnz,
                         int32_t* rank_out_dev, int32_t* bfs_levels_out /* host, may be NULL */, void* stream);
int gcn_csr_apply_rank_device(const int32_t* rowptr_dev, const int32_t* col_dev, const float* val_dev,
                              const int32_t* rank_dev, int32_t n, int32_t nnz, int32_t* out_rowptr_dev,
                              int32_t* out_col_dev, float* out_val_dev, int32_t* vomp_out_dev, void* stream);
/* Rabbit ON THE DEVICE: the parallel algorithm the reference's serial code cites as "Rabbit properly" (renumber.cu:
 * 328-330: Arai et al., IPDPS 2016) — every vertex once, in ascending degree order, one wave per vertex; lazy
 * aggregation of the merged vertices' (community, weight) lists; the merge itself is one compare-and-swap on the
 * target's {lock, degree, child} word (csrc/rabbit_device.hip).  NOT the serial code's integers (gcn_order_rabbit /
 * the `rabbit` symbol give those) and not bit-reproducible from run to run (which merges race differs): what is
 * guaranteed is a permutation whose communities reach the serial version's modularity to a few percent (tests).
 * Input: a SYMMETRIC pattern (A = Aᵀ; self-loops ignored; values play no part, as in the reference).
 * rank_out_dev[old] = new; community_out_dev (may be NULL): top-level vertex of every vertex; stats_host (may be NULL,
 * else 8 words): {communities, passes, vertices retried, vertices left top-level for lack of table or pool room, and
 * four guard counters — pointer chain, child chain, full table, bad index}.  Every loop of the kernel is bounded by
 * such a guard; all four are 0 in a sound run, and a non-zero one means an aggregation ran on broken state: the call
 * then returns GCN_ERR_INTERNAL, prints one line and writes NO ordering (the counters are still reported). */
int gcn_order_rabbit_device(const int32_t* rowptr_dev, const int32_t* col_dev, int32_t n, int32_t nnz,
                            int32_t* rank_out_dev, int32_t* community_out_dev, int64_t* stats_host, void* stream);

/* ------------------------------------------------------------------------- */
/* (1e) the "push" form of the multi-GPU layer exchange (gcn_amd/dist.py,      */
/*      exchange="push"; csrc/exchange.hip).  The reference is single-GPU      */
/*      (device 0 hard-coded, flexspmm.cu:507): no counterpart.  Every rank    */
/*      maps its peers' exchange buffers once (IPC handles), writes its shard  */
/*      of a layer output straight into them with the runtime's copy path      */
/*      (no compute units), raises one flag per peer and layer behind the      */
/*      data, and waits for its own flags with one wave.                       */
/* ------------------------------------------------------------------------- */
/* `count` (<= 64) int32 flags, zeroed, in FINE-GRAINED device memory of the current device (a peer's copy engine writes
 * them, a wave of this GPU polls them: coherent at system scope, which ordinary device memory is not promised to be),
 * and the 64-byte IPC handle under which another process maps them (gcn_exchange_flags_open; _close unmaps, _destroy
 * frees — after every peer has closed). */
int gcn_exchange_flags_create(int32_t count, int32_t** flags_dev_out, void* ipc_handle_out_64);
int gcn_exchange_flags_open(const void* ipc_handle_64, int32_t** flags_peer_out);
int gcn_exchange_flags_close(int32_t* flags_peer);
int gcn_exchange_flags_destroy(int32_t* flags_dev);
/* dst_peer[0:bytes] = src[0:bytes]; dst_peer is a pointer into a peer's buffer mapped into this process; asynchronous */
int gcn_exchange_push(void* dst_peer, const void* src, size_t bytes, void* stream);
/* *flag_peer = *value_dev (4 bytes), enqueued behind the pushes on the same stream: "my shard has landed" */
int gcn_exchange_signal(int32_t* flag_peer, const int32_t* value_dev, void* stream);
/* Enqueue ONE wave that returns when flags_dev[i] == value for every i < count (<= 64) except i == skip (-1: none).
 * Bounded: after timeout_seconds a lane gives up and writes 1 + (the flag it waited for) to *status_dev (0 otherwise
 * untouched) — the stream goes on, the caller checks the status word at its next synchronisation point. */
int gcn_exchange_wait(const int32_t* flags_dev, int32_t count, int32_t skip, int32_t value, int32_t* status_dev,
                      double timeout_seconds, void* stream);

/* ------------------------------------------------------------------------- */
/* (2) DROP-IN symbols — identical names and argument lists to the reference.  */
/* ------------------------------------------------------------------------- */

/* Error convention of every drop-in symbol: the reference's (void, print — cuspmm.cu:3-21).  On malformed input,
 * foreign buffers or a HIP failure ONE line goes to stderr ("libgcnspmm: <symbol>: ...") and the call RETURNS with
 * the caller's outputs untouched; nothing aborts the host process. */

/* renumber.so — renumber.cu:23 (dfs), :157 (gorder), :233 (perm_apply), :319 (rabbit).
 * All pointers HOST, CSR rewritten in place, vomp[new] = old. */
void dfs(int* rowPtr, int* col, float* vals, int* vomp, int m, int n, int nnz);
void gorder(int* rowPtr, int* col, float* vals, int* vomp, int m, int n, int nnz);
void perm_apply(int* rowPtr, int* col, float* vals, int* vomp, int m, int n, int nnz);
void rabbit(int* rowPtr, int* col, float* vals, int* vomp, int m, int n, int nnz);

/* tile.so — tile.cu:104-106.  All pointers HOST.  Packs THIS library's plan
 * (not the reference's defective tile-seg arrays, SURVEY defects D1-D3) into the
 * caller's buffers; capacities honoured: seg_rowPtr nnz ints, segNzCV 2*nnz
 * floats, segVoMap nnz ints, grouped_tailSeg/next_seg 256 ints (gcn6.py:334-339).
 * n_segs[0] = nnz / 9, or one less so that its lowest bit tells flexspmm whether the values are u[r]*u[c] (n_segs is the
 * only scalar gcn6 carries from csr2tile to flexspmm, gcn6.py:353-366); needs nnz >= m + 19, else nothing is packed
 * and n_segs[0] = 0.  Encoding documented in INTEGRATION.md. */
void csr2tile(int* rowPtr, int* colIdx, float* vals, int m, int n, int nnz,
              int* vo_mp, int* segVoMap, int* seg_rowPtr, float* segNzCV,
              int* grouped_tailSeg, int* next_seg, int tm, int* n_segs);

/* tile.so — tile.cu:11-12: the per-panel helper csr2tile loops over in the reference, exported there as well.  No call
 * site binds it (gcn6.py:341-352 calls csr2tile only) and this library's packing has no per-panel step: the symbol
 * resolves, prints one line to stderr and returns with every buffer (n_segs included) untouched. */
void csr2seg_Cmajor(int ridx, int* rowPtr, int* colIdx, float* vals, int m, int n, int nnz,
                    int* voMp, int* segVoMap, int* seg_rowPtr, float* segNzCV, int tm, int* n_segs);

/* flexspmm.so — flexspmm.cu:499-502.  All pointers DEVICE.  Consumes the arrays
 * written by this library's csr2tile (plain CSR, or — square graphs that qualify for XCD-aware slicing — the group
 * kernels' stream format).  Like the reference's (flexspmm.cu:497-540) the call only ENQUEUES
 * kernels on the legacy default stream (flexspmm.cu:512): in the group format the chunk and cut-row counts are read
 * from the packed header ON THE DEVICE by a one-thread guard kernel (grids sized from upper bounds) — no layer drains
 * the stream.  Only the first call on a given set of buffers reads the 64-byte header once, to report buffers this
 * library did not pack (message, return, C as handed over); a header that disappears later is caught by the guard. */
void flexspmm(int* seg_rowPtr, float* segNzCV, int* segVoMap,
              int* grouped_tailSeg, int* next_seg,
              int m, int n, int k, int n_segs, float* B, float* C);

/* permutate.so — permutate.cu:40-41.  Device pointers; B permuted in place
 * (B[r,:] <- B[voMp[r],:]); labels untouched exactly like the reference
 * (`if (false && lane_id==0)`, permutate.cu:17,35). */
void permutate(float* B, int* voMp, int* labels, int m, int n, int k);

/* cuspmm.so — cuspmm.cu:23-24 (first parameter is declared float* there although
 * it carries the int32 row pointer; kept for signature parity). */
void cuspmm(float* rowPtr, int* col, float* vals, float* X, float* C,
            int m, int n, int nnz, int dim);

#ifdef __cplusplus
}
#endif
#endif /* GCN_SPMM_H */
