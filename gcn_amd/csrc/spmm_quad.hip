// spmm_quad.hip — the 64-column-tile SpMM kernel with FOUR non-zeros per gather instruction.
//
// Why: on a dense band matrix (every gather an L1/L2 hit) the one-non-zero-per-instruction chunk
// kernel (spmm_kernels.hip, VEC = 1) tops out at 67 G gathers/s — 9 CU-cycles per 256-byte row —
// and the Reddit-shaped graph already runs at 92 % of that, so its time is set by what a gather
// costs in the CU's vector-memory path, not by HBM.  16-byte-per-lane loads move the same cache
// lines in fewer cycles (the 256-column tile reached 24.7 TB/s against 17.2 TB/s), but a
// 256-column tile quadruples the gathered working set.  This kernel keeps the 64-column tile and
// still issues 16-byte loads:
//
//   lane = sub*16 + f :  sub = 0..3 is which of the step's four non-zeros, f = 0..15 is which
//   float4 of the 64-column slice.  One global_load_dwordx4 fetches FOUR different feature rows
//   (4 x 256 B); each lane keeps a float4 partial sum for its (sub, f).
//   The 64-entry (col, val) block is loaded transposed (lane s*16 + u holds entry 4u + s), so at
//   step u a DPP row broadcast (row_newbcast:u, a modifier on a full-rate VALU op) hands every
//   16-lane row exactly the non-zero it gathers for: no v_readlane / SGPR round trip, and a
//   quarter of the vector instructions per non-zero of the VEC = 1 kernel.
//   A row that ends — anywhere inside a step — is reduced over sub (two xor-shuffles per
//   component) and written by the lanes sub == 0 as one 256-byte store.
//
// Chunk schedule, partial slab, fix-up, accumulate mode and epilogue are those of
// spmm_kernels.hip (same plan, same arguments); the order of the additions inside a row differs
// (4 interleaved chains + a tree), within the 1e-5 contract, and is fixed — results are
// bit-reproducible run to run.  Needs k % 4 == 0, 16-byte aligned B/C/P, n < 2^24 and
// n*k*4 < 4 GiB (24-bit multiply, 32-bit byte offsets); launch_spmm falls back to the VEC = 1
// kernel otherwise.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spmm_kernels.h"

namespace gcn {

typedef float f32x4_q __attribute__((ext_vector_type(4)));


typedef const int __attribute__((address_space(4)))* const_int_ptr;

__device__ __forceinline__ int qsgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// The value held by the lane that loaded non-zero (step UU, this lane's sub) of the transposed block.
//   LPE = 16: lane s*16+u holds entry 4u+s   -> DPP row_newbcast:UU (lane UU of every 16-lane row)
//   LPE =  4: lane s*4+u  holds entry 16u+s  -> DPP quad_perm [UU,UU,UU,UU]
// (every lane has a source lane, so bound_ctrl is irrelevant)
template <int LPE, int UU>
__device__ __forceinline__ int quad_bcast(int v) {
  static_assert(LPE == 16 || LPE == 4, "layouts: 16 or 4 lanes per non-zero");
  if constexpr (LPE == 16) {
    return __builtin_amdgcn_mov_dpp(v, 0x150 + UU, 0xf, 0xf, true);
  } else {
    return __builtin_amdgcn_mov_dpp(v, UU | (UU << 2) | (UU << 4) | (UU << 6), 0xf, 0xf, true);
  }
}

// sum over the subs (lane bits >= log2(LPE))
template <int LPE>
__device__ __forceinline__ float4 quad_reduce(float4 t) {
#pragma unroll
  for (int o = LPE; o < 64; o <<= 1) {
    t.x += __shfl_xor(t.x, o); t.y += __shfl_xor(t.y, o);
    t.z += __shfl_xor(t.z, o); t.w += __shfl_xor(t.w, o);
  }
  return t;
}

// LPE = lanes per non-zero (each lane a float4): 16 -> 64-column tile, 4 non-zeros per gather
// instruction; 4 -> 16-column tile (k <= 16), 16 per instruction.  A 64-entry block takes LPE steps of
// 64/LPE non-zeros.  (An 8-lane / 32-column layout was measured for k = 17..32 and bought nothing over
// the 16-lane layout with half its lanes idle — 1.61 vs 1.59 ms on the Reddit-shaped graph: narrow
// widths are bound by L2 requests per non-zero, not by instructions — so it is not instantiated.)
// VALLESS: the matrix values are not read at all — every stored entry counts 1.  For adjacencies whose
// values factor as u[r]*u[c] (the GCN normalisation D^-1/2 (A+I) D^-1/2) the caller pre-scales B's rows by
// u and scales the finished rows by u[r] (api_spmm.cpp, slice_reduce_kernel): the 4-byte value stream is
// 5 % of what the sliced kernel moves across the fabric, and fabric bytes are its time.
// COL16 (value-free pass only): the column stream is 16 bits per non-zero — the column's offset inside its
// slice (slices <= 65 535 columns wide).  The slice-major stream is laid out so that no chunk straddles two
// slices (every slice padded to a multiple of the chunk size with 0xFFFF markers, which gather the all-zero
// row n of the scaled copy of B), so the slice base is one scalar per chunk (QuadSlices: where each slice
// starts in the padded stream, its width).  Saves 2 of the 4 index bytes per non-zero of the fabric traffic.
struct QuadSlices { int S, w, start[9]; };           // start[s] .. start[s+1]: padded stream range of slice s (S <= 8)

template <int LPE, bool EPI, bool VALLESS, bool COL16>
__global__ void __launch_bounds__(256)
spmm_quad_kernel(const int* __restrict__ g_rowptr, const int* __restrict__ g_col,
                 const float* __restrict__ g_val, const float* __restrict__ g_B,
                 float* __restrict__ g_C, float* __restrict__ g_P,
                 const int* __restrict__ g_chunk_row, const float* __restrict__ g_bias,
                 const int* __restrict__ nnz_dev,
                 int relu, int nchunks, int T, int m, int nnz, int k, int col_tile, int flags, int ldb,
                 QuadSlices sl, int n_rows_b) {
  static_assert(!COL16 || VALLESS, "16-bit columns are only built for the value-free pass");
  const bool accumulate = flags & 1;                // C += ...
  const bool stream_rows = flags & 2;               // finished rows leave with non-temporal stores (SpmmArgs::stream_rows)
  if (nnz_dev) {                                    // drop-in (flexspmm) mode, see spmm_kernels.hip
    nnz = *nnz_dev;
    nchunks = (int)(((long long)nnz + T - 1) / T);
    g_val = reinterpret_cast<const float*>(g_col) + nnz;
  }
  // Row pointers and chunk rows are read through the CONSTANT address space: wave-uniform loads
  // from it always go through the scalar cache (s_load_dword), which keeps row_end / pos in SGPRs.
  // (The kernel is too large for the compiler to prove by itself that its stores to C / P never
  // clobber them.)  Nothing writes these arrays while the kernel runs.
  const struct {
    const_int_ptr rowptr; const int* __restrict__ col; const float* __restrict__ val;
    const float* __restrict__ B; float* __restrict__ C; float* __restrict__ P;
    const_int_ptr chunk_row; const float* __restrict__ bias;
  } a = {(const_int_ptr)(uintptr_t)g_rowptr, g_col, g_val, g_B, g_C, g_P,
         (const_int_ptr)(uintptr_t)g_chunk_row, g_bias};
  const int lane = threadIdx.x & 63;
  const int wib  = qsgpr(threadIdx.x >> 6);
  constexpr int EPS = 64 / LPE;                     // non-zeros per step (= per gather instruction)
  const int sub  = lane / LPE;
  const int f    = lane % LPE;
  const int fcol = col_tile * (4 * LPE) + f * 4;    // first of this lane's four feature columns
  const bool fok = fcol < k;                        // (k % 4 == 0: a float4 is all in or all out)
  const bool writer = fok && sub == 0;
  const int tl   = f * EPS + sub;                   // transposed position this lane loads

  const int xcd           = blockIdx.x & 7;
  const int wave_in_xcd   = (blockIdx.x >> 3) * 4 + wib;
  const int waves_per_xcd = (gridDim.x >> 3) * 4;
  const int c_lo = (int)(((long long)nchunks * xcd) >> 3);
  const int c_hi = (int)(((long long)nchunks * (xcd + 1)) >> 3);

  const unsigned row_bytes = (unsigned)ldb * 4u;    // B row stride (>= k: rows may be padded to 128-byte lines)
  // lanes past k gather the tile's first four columns of the same rows (valid memory, a cache line
  // the active lanes fetch anyway) and never store
  const unsigned foff = (unsigned)(fok ? fcol : col_tile * (4 * LPE)) * 4u;
  const char* __restrict__ Bb = reinterpret_cast<const char*>(a.B);
  const size_t kk = (size_t)k;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI) {
    if (a.bias && fok) bias4 = *reinterpret_cast<const float4*>(a.bias + fcol);
  }

  for (int c = c_lo + wave_in_xcd; c < c_hi; c += waves_per_xcd) {
    const int start = c * T;
    const int end   = (int)min((long long)start + T, (long long)nnz);
    int r = a.chunk_row[c];
    int row_end    = a.rowptr[r + 1];
    int row_end_nx = (r + 1 < m) ? a.rowptr[r + 2] : -1;
    bool head = a.rowptr[r] < start;
    int pos = start;
    int last_flush = start;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);   // this lane's (sub, f) partial of the current row

    auto flush = [&]() {
      float4 t = quad_reduce<LPE>(acc);
      if (head) {
        if (writer) *reinterpret_cast<float4*>(a.P + (size_t)(2 * c) * kk + fcol) = t;
      } else if (last_flush != pos) {               // (empty rows belong to launch_fill_empty_rows, spmm_kernels.hip)
        float4* dst = reinterpret_cast<float4*>(a.C + (size_t)r * kk + fcol);
        if (accumulate) {
          if (writer) { const float4 o = *dst; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
        }
        if (EPI) {
          t.x += bias4.x; t.y += bias4.y; t.z += bias4.z; t.w += bias4.w;
          if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
        }
        if (writer) {
          if (stream_rows) __builtin_nontemporal_store(f32x4_q{t.x, t.y, t.z, t.w}, reinterpret_cast<f32x4_q*>(dst));
          else *dst = t;
        }
      }
      acc = make_float4(0.f, 0.f, 0.f, 0.f);
      head = false;
      last_flush = pos;
      ++r;
      row_end    = row_end_nx;
      if (row_end == pos) {                         // row r is empty: jump over the whole run of empty rows
        r = next_nonempty_row(a.rowptr, r, m, pos);
        row_end = (r < m) ? a.rowptr[r + 1] : -1;
      }
      row_end_nx = (r + 1 < m) ? a.rowptr[r + 2] : -1;
    };
    while (pos == row_end) flush();                 // leading empty rows (chunk 0 only)

    int   cj_nx = 0;
    float vj_nx = 0.f;
    // 16-bit columns: this chunk's slice (chunks never straddle slices) -> its first column
    int col_base = 0;
    if (COL16) {
#pragma unroll
      for (int i = 1; i < 8; ++i) col_base += (i < sl.S && start >= sl.start[i]) ? sl.w : 0;
    }
    auto load_col = [&](int idx) -> int {
      if (COL16) {
        const int c16 = reinterpret_cast<const unsigned short*>(a.col)[idx];
        return c16 == 0xFFFF ? n_rows_b : col_base + c16;          // marker -> the all-zero row behind B'
      }
      return a.col[idx];
    };
    if (start + tl < end) { cj_nx = load_col(start + tl); if (!VALLESS) vj_nx = a.val[start + tl]; }
    for (int base = start; base < end; base += 64) {
      const int cnt = min(64, end - base);
      const int cj = cj_nx;
      const int vj = __builtin_bit_cast(int, vj_nx);
      cj_nx = 0; vj_nx = 0.f;
      if (base + 64 + tl < end) { cj_nx = load_col(base + 64 + tl); if (!VALLESS) vj_nx = a.val[base + 64 + tl]; }

      float4 b[LPE];
#define GCN_Q_GATHER(UU)                                                                         \
      if constexpr (UU < LPE)                                                                    \
        b[UU] = *reinterpret_cast<const float4*>(                                                \
            Bb + (size_t)(__umul24((unsigned)quad_bcast<LPE, UU>(cj), row_bytes) + foff));
#define GCN_Q_VAL(UU) (VALLESS ? 1.0f : __builtin_bit_cast(float, quad_bcast<LPE, UU>(vj)))
#define GCN_Q_ALL(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
      // entries past `cnt` carry col = 0, val = 0: they gather row 0 (valid memory) and are
      // masked out below, never multiplied in
      GCN_Q_ALL(GCN_Q_GATHER)
      if (cnt == 64 && (row_end < 0 || row_end - pos >= 64)) {
        // fast path: all 64 non-zeros belong to the current row
#define GCN_Q_FMA(UU)                                                                            \
        if constexpr (UU < LPE) {                                                                \
          const float v = GCN_Q_VAL(UU);                                                         \
          acc.x = fmaf(v, b[UU].x, acc.x); acc.y = fmaf(v, b[UU].y, acc.y);                      \
          acc.z = fmaf(v, b[UU].z, acc.z); acc.w = fmaf(v, b[UU].w, acc.w); }
        GCN_Q_ALL(GCN_Q_FMA)
#undef GCN_Q_FMA
        pos += 64;
        while (pos == row_end) flush();
      } else {
        // rows end inside this block (or it is the ragged last block): one pass over the steps that
        // overlap it per row segment [q0, q1) of the block, each lane adding only the products of its own
        // non-zeros that lie in the segment (masked on the product, so a NaN/Inf in a feature row
        // never leaks into a row of A that does not reference it)
        int q0 = 0;
        while (true) {
          const int q1 = row_end < 0 ? cnt : min(cnt, row_end - base);
          const unsigned lo = (unsigned)(q0 - sub), len = (unsigned)(q1 - q0);
#define GCN_Q_SEG(UU)                                                                            \
          if constexpr (UU < LPE) if (UU * EPS < q1 && (UU + 1) * EPS > q0) {   /* step overlaps the segment (scalar test) */ \
            const bool in = (unsigned)(UU * EPS) - lo < len;      /* entry UU*EPS + sub in [q0, q1) */ \
            const float v = GCN_Q_VAL(UU);                                                       \
            acc.x = in ? fmaf(v, b[UU].x, acc.x) : acc.x; acc.y = in ? fmaf(v, b[UU].y, acc.y) : acc.y; \
            acc.z = in ? fmaf(v, b[UU].z, acc.z) : acc.z; acc.w = in ? fmaf(v, b[UU].w, acc.w) : acc.w; }
          GCN_Q_ALL(GCN_Q_SEG)
#undef GCN_Q_SEG
          pos = base + q1;
          if (pos != row_end) break;                // the row goes on past this block
          while (pos == row_end) flush();           // row complete (+ any empty rows after it)
          q0 = q1;
          if (q0 >= cnt) break;
        }
      }
#undef GCN_Q_ALL
#undef GCN_Q_GATHER
#undef GCN_Q_VAL
    }

    if (last_flush != end) {                        // the row piece that sticks out of the chunk
      const float4 t = quad_reduce<LPE>(acc);
      const int slot = head ? 2 * c : 2 * c + 1;
      if (writer) *reinterpret_cast<float4*>(a.P + (size_t)slot * kk + fcol) = t;
    }
  }
}

bool spmm_quad_eligible(const SpmmArgs& a) {
  const uintptr_t al = (uintptr_t)a.B | (uintptr_t)a.C | (uintptr_t)a.P | (uintptr_t)a.bias;
  const int ldb = a.ldb > 0 ? a.ldb : a.k;
  return a.k % 4 == 0 && ldb % 4 == 0 && (al & 15) == 0 && a.n < (1 << 24) && ldb * 4 < (1 << 24) &&
         (unsigned long long)a.n * (unsigned long long)ldb * 4ull < 0xFFFFFFF0ull;
}

template <int LPE>
static hipError_t launch_quad(const SpmmArgs& a, int nblocks, bool epi, hipStream_t s) {
  const int tiles = (a.k + 4 * LPE - 1) / (4 * LPE);
  QuadSlices sl{};
  if (a.col16) {
    sl.S = a.col16_S; sl.w = a.col16_w;
    for (int i = 0; i < 9; ++i) sl.start[i] = a.col16_start[i];
  }
  for (int t = 0; t < tiles; ++t) {
#define GCN_QUAD_ARGS a.rowptr, a.col, a.val, a.B, a.C, a.P, a.chunk_row, a.bias, a.nnz_dev, \
                      a.relu, a.nchunks, a.T, a.m, a.nnz, a.k, t, (a.accumulate ? 1 : 0) | (a.stream_rows ? 2 : 0), (a.ldb > 0 ? a.ldb : a.k), sl, a.n
    if constexpr (LPE == 16) {
      if (a.valless && !epi) {                       // (the value-free variants are only built for the sliced main pass)
        if (a.col16) spmm_quad_kernel<LPE, false, true, true><<<dim3(nblocks), dim3(256), 0, s>>>(GCN_QUAD_ARGS);
        else         spmm_quad_kernel<LPE, false, true, false><<<dim3(nblocks), dim3(256), 0, s>>>(GCN_QUAD_ARGS);
        continue;
      }
    }
    if (a.valless || a.col16) return hipErrorInvalidValue;   // a caller bug: B was prepared for a kernel that is not there
    if (epi) spmm_quad_kernel<LPE, true, false, false><<<dim3(nblocks), dim3(256), 0, s>>>(GCN_QUAD_ARGS);
    else     spmm_quad_kernel<LPE, false, false, false><<<dim3(nblocks), dim3(256), 0, s>>>(GCN_QUAD_ARGS);
#undef GCN_QUAD_ARGS
  }
  return hipGetLastError();
}

// lanes per non-zero for a k-wide SpMM: the narrowest layout whose tile (4*LPE columns) covers k
int spmm_quad_lanes(int k) { return k <= 16 ? 4 : 16; }

hipError_t launch_spmm_quad(const SpmmArgs& a, int nblocks, bool epi, hipStream_t s) {
  switch (spmm_quad_lanes(a.k)) {
    case 4:  return launch_quad<4>(a, nblocks, epi, s);
    default: return launch_quad<16>(a, nblocks, epi, s);
  }
}

}  // namespace gcn
