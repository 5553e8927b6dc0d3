// spmm_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels for the GCN
// aggregation SpMM  C[m x k] = A[m x n, CSR fp32] * B[n x k]  (row-major B, C).
//
// Replaces the five CUDA kernels of the reference's flexspmm.cu:17-498 and the
// cuSPARSE call of cuspmm.cu:57-61.  It is NOT a translation of either: the
// reference walks 8-row "tile segs" with 4/8 lanes per row and fp32 atomics on
// split rows; this file uses an equal-nnz chunk schedule with a wavefront-level
// segmented sum and a deterministic fix-up pass:
//
//   * the nnz stream is cut into chunks of T non-zeros (T multiple of 64); one
//     wave64 owns one chunk at a time (persistent grid, XCD-aware chunk ranges);
//   * the wave loads 64 (col,val) pairs with one coalesced load each, then walks
//     them with v_readlane (col/val become SGPRs), issuing U whole-row gathers
//     of B back to back: one global_load per non-zero covers the full feature
//     row (64 lanes x VEC floats = 256/512/1024 B contiguous), so every HBM /
//     Infinity-Cache request is a full line and U*VEC*256 B are in flight per wave;
//   * rows are summed in registers in CSR order; a row that ends inside the
//     chunk is stored with one coalesced row store (no atomics, C need not be
//     zeroed); the (at most two) row pieces that stick out of the chunk go to a
//     partial slab P and are added in chunk order by spmm_fixup_kernel, so the
//     result is bitwise reproducible and independent of the launch geometry.
//
// Roofline: HBM/cache-bandwidth bound (0.49 flop/B); algorithmic bytes per
// non-zero = 8 + 4k.  MFMA is deliberately not used: fp32 MFMA
// peak equals the fp32 VALU peak on gfx950 and the contraction is a gather.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include "spmm_kernels.h"

namespace gcn {

template <int VEC> struct VecOf;
template <> struct VecOf<1> { typedef float  type; };
template <> struct VecOf<2> { typedef float2 type; };
template <> struct VecOf<4> { typedef float4 type; };

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&out)[VEC]) {
  typedef typename VecOf<VEC>::type V;
  V v = *reinterpret_cast<const V*>(p);
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) out[i] = f[i];
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&in)[VEC]) {
  typedef typename VecOf<VEC>::type V;
  V v;
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) f[i] = in[i];
  *reinterpret_cast<V*>(p) = v;
}

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One gathered feature-row slice: 64 lanes x VEC floats.  BUF = buffer addressing: the row base
// goes into the instruction's SGPR offset (col * row_bytes, 32-bit) and the lane part into a
// constant VGPR offset, so a gather costs v_readlane + s_mul + buffer_load and no 64-bit vector
// address arithmetic (requires n*k*4 < 4 GiB); otherwise flat 64-bit addressing.
template <int VEC, bool BUF>
__device__ __forceinline__ void gather_row(const float* __restrict__ Bl, __amdgpu_buffer_rsrc_t rsrc,
                                           int voff, int cu, size_t k, unsigned row_bytes,
                                           float (&out)[VEC]) {
  if (BUF) {
    // VEC == 1 only: hipcc 7.2 lowers __builtin_amdgcn_raw_buffer_load_b64/_b128 to a single
    // dword load (checked in the ISA), so the 2- and 4-float tiles keep flat addressing.
    static_assert(!BUF || VEC == 1, "buffer addressing is only instantiated for VEC == 1");
    const unsigned soff = (unsigned)cu * row_bytes;
    out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
  } else {
    load_vec<VEC>(Bl + (size_t)cu * k, out);
  }
}

// ---------------------------------------------------------------------------
// plan kernel: chunk_row[c] = the row that contains non-zero c*T, i.e. the r with
// rowptr[r] <= c*T < rowptr[r+1];  chunk_row[0] = 0 so that chunk 0 also emits
// leading empty rows.
// ---------------------------------------------------------------------------
__global__ void plan_chunk_rows_kernel(const int* __restrict__ rowptr, int m, int T,
                                       int nchunks, int* __restrict__ chunk_row) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  if (c == 0) { chunk_row[0] = 0; return; }
  const long long target = (long long)c * T;
  // first index i in [0, m] with rowptr[i] > target
  int lo = 0, hi = m + 1;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if ((long long)rowptr[mid] > target) hi = mid; else lo = mid + 1;
  }
  chunk_row[c] = lo - 1;
}

// ---------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------
template <int VEC, int U, bool EPI, bool BUF>
__global__ void __launch_bounds__(256)
spmm_chunk_kernel(const int* __restrict__ g_rowptr, const int* __restrict__ g_col,
                  const float* __restrict__ g_val, const float* __restrict__ g_B,
                  float* __restrict__ g_C, float* __restrict__ g_P,
                  const int* __restrict__ g_chunk_row, const float* __restrict__ g_bias,
                  const int* __restrict__ g_nnz_dev,
                  int relu, int nchunks, int T, int m, int nnz, int kk, int col_tile, int accumulate, int ldb) {
  // drop-in (flexspmm) mode: the host does not know nnz; it lives in rowptr[m] and
  // the values follow the column indices in one buffer (api_dropin.cpp, csr2tile layout)
  if (g_nnz_dev) {
    nnz = *g_nnz_dev;
    nchunks = (int)(((long long)nnz + T - 1) / T);
    g_val = reinterpret_cast<const float*>(g_col) + nnz;
  }
  // plain-struct view of the arguments (all loads below are from __restrict__
  // const pointers, so row pointers / chunk rows come in through the scalar cache)
  const struct {
    const int* __restrict__ rowptr; const int* __restrict__ col; const float* __restrict__ val;
    const float* __restrict__ B; float* __restrict__ C; float* __restrict__ P;
    const int* __restrict__ chunk_row; const float* __restrict__ bias;
    int relu, nchunks, T, m, nnz, k;
  } a = {g_rowptr, g_col, g_val, g_B, g_C, g_P, g_chunk_row, g_bias, relu, nchunks, T, m, nnz, kk};
  const int lane = threadIdx.x & 63;
  const int wib  = sgpr(threadIdx.x >> 6);

  // XCD-aware chunk ranges: blocks b and b+8 share an XCD (round-robin dispatch),
  // so XCD x walks the contiguous chunk range [x*N/8, (x+1)*N/8) — neighbouring
  // rows (which share neighbours after RCM/Gorder) meet in one 4 MiB L2.
  // Placement only affects speed, never the result.
  const int xcd           = blockIdx.x & 7;
  const int wave_in_xcd   = (blockIdx.x >> 3) * 4 + wib;
  const int waves_per_xcd = (gridDim.x >> 3) * 4;
  const int c_lo = (int)(((long long)a.nchunks * xcd) >> 3);
  const int c_hi = (int)(((long long)a.nchunks * (xcd + 1)) >> 3);

  const int  fcol   = col_tile * (64 * VEC) + lane * VEC;   // first feature column of this lane
  const bool active = fcol < a.k;
  const float* __restrict__ Bl = a.B + (active ? fcol : 0);
  const size_t k = (size_t)a.k;
  // buffer descriptor over B (wave-uniform inputs only: kernel arguments)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g_B), 0, 0xFFFFFFFFu, 0x00020000);
  const int voff = (active ? fcol : 0) * 4;
  const unsigned row_bytes = (unsigned)ldb * 4u;     // B row stride (>= k: rows may be padded to 128-byte lines)
  const size_t ldB = (size_t)ldb;

  float bias[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) bias[i] = 0.f;
  if (EPI) {
    if (active && a.bias) load_vec<VEC>(a.bias + fcol, bias);
  }

  for (int c = c_lo + wave_in_xcd; c < c_hi; c += waves_per_xcd) {
    const int start = c * a.T;
    const int end   = (int)min((long long)start + a.T, (long long)a.nnz);
    int r = a.chunk_row[c];
    int row_end    = a.rowptr[r + 1];
    int row_end_nx = (r + 1 < a.m) ? a.rowptr[r + 2] : -1;
    bool head = a.rowptr[r] < start;      // row r began in an earlier chunk
    int pos = start;
    int last_flush = start;

    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

    // row r is finished at `pos`: write it out and step to the next row
    // (an EMPTY row — nothing consumed since the last flush — is not written here: launch_fill_empty_rows owns those,
    //  and the run of empty rows behind it is jumped over, not walked)
    auto flush = [&]() {
      if (head) {
        if (active) store_vec<VEC>(a.P + (size_t)(2 * c) * k + fcol, acc);
      } else if (last_flush != pos) {
        if (accumulate) {                   // C already holds another part of the product (beta = 1)
          float old[VEC];
#pragma unroll
          for (int i = 0; i < VEC; ++i) old[i] = 0.f;
          if (active) load_vec<VEC>(a.C + (size_t)r * k + fcol, old);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += old[i];
        }
        if (EPI) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            acc[i] += bias[i];
            if (a.relu) acc[i] = fmaxf(acc[i], 0.f);
          }
        }
        if (active) store_vec<VEC>(a.C + (size_t)r * k + fcol, acc);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      head = false;
      last_flush = pos;
      ++r;
      row_end    = row_end_nx;
      if (row_end == pos) {                 // row r is empty: on to the next row that holds an entry
        r = next_nonempty_row(a.rowptr, r, a.m, pos);
        row_end = (r < a.m) ? a.rowptr[r + 1] : -1;
      }
      row_end_nx = (r + 1 < a.m) ? a.rowptr[r + 2] : -1;
    };

    while (pos == row_end) flush();        // leading empty rows (chunk 0 only)

    // (col, val) of the next 64 non-zeros are fetched one block ahead, so the coalesced index
    // load is never on the critical path of the gathers that depend on it
    int   cj_nx = 0;
    float vj_nx = 0.f;
    if (start + lane < end) { cj_nx = a.col[start + lane]; vj_nx = a.val[start + lane]; }
    for (int base = start; base < end; base += 64) {
      const int cnt = min(64, end - base);
      const int   cj = cj_nx;
      const float vj = vj_nx;
      if (base + 64 + lane < end) { cj_nx = a.col[base + 64 + lane]; vj_nx = a.val[base + 64 + lane]; }

      int j = 0;
      for (; j + U <= cnt; j += U) {
        float b[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int cu = __builtin_amdgcn_readlane(cj, j + u);
          gather_row<VEC, BUF>(Bl, rsrc, voff, cu, ldB, row_bytes, b[u]);
        }
        if (row_end - pos >= U || row_end < 0) {
          // fast path: the current row does not end strictly inside this batch
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const float vu = __builtin_bit_cast(float,
                __builtin_amdgcn_readlane(__builtin_bit_cast(int, vj), j + u));
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(vu, b[u][i], acc[i]);
          }
          pos += U;
          while (pos == row_end) flush();
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const float vu = __builtin_bit_cast(float,
                __builtin_amdgcn_readlane(__builtin_bit_cast(int, vj), j + u));
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(vu, b[u][i], acc[i]);
            ++pos;
            while (pos == row_end) flush();
          }
        }
      }
      for (; j < cnt; ++j) {               // ragged tail (last chunk of the matrix only)
        float b1[VEC];
        const int cu = __builtin_amdgcn_readlane(cj, j);
        const float vu = __builtin_bit_cast(float,
            __builtin_amdgcn_readlane(__builtin_bit_cast(int, vj), j));
        gather_row<VEC, BUF>(Bl, rsrc, voff, cu, ldB, row_bytes, b1);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(vu, b1[i], acc[i]);
        ++pos;
        while (pos == row_end) flush();
      }
    }

    // the row piece that sticks out of the chunk's end
    if (last_flush != end) {
      const int slot = head ? 2 * c : 2 * c + 1;
      if (active) store_vec<VEC>(a.P + (size_t)slot * k + fcol, acc);
    }
  }
}

// ---------------------------------------------------------------------------
// fix-up: one wave per chunk boundary c (1..nchunks-1).  The boundary that is the
// FIRST one inside a row owns that row: C[r] = P[2*c0+1] + P[2*(c0+1)] + ... +
// P[2*c1] in chunk order (c0 = chunk holding the row's first non-zero).
// ---------------------------------------------------------------------------
template <bool EPI>
__global__ void __launch_bounds__(256)
spmm_fixup_kernel(const int* __restrict__ g_rowptr, const float* __restrict__ g_P,
                  float* __restrict__ g_C, const int* __restrict__ g_chunk_row,
                  const float* __restrict__ g_bias, const int* __restrict__ g_nnz_dev,
                  int relu, int nchunks, int T, int kk, int accumulate) {
  if (g_nnz_dev) nchunks = (int)(((long long)(*g_nnz_dev) + T - 1) / T);
  const struct {
    const int* __restrict__ rowptr; const float* __restrict__ P; float* __restrict__ C;
    const int* __restrict__ chunk_row; const float* __restrict__ bias; int relu, nchunks, T, k;
  } a = {g_rowptr, g_P, g_C, g_chunk_row, g_bias, relu, nchunks, T, kk};
  const int lane = threadIdx.x & 63;
  const int c = sgpr(blockIdx.x * 4 + (threadIdx.x >> 6)) + 1;
  if (c >= a.nchunks) return;
  const int r  = a.chunk_row[c];
  const int rs = a.rowptr[r];
  const long long start = (long long)c * a.T;
  if (!(rs < start && rs / a.T == c - 1)) return;
  const int re = a.rowptr[r + 1];
  const int c1 = (re - 1) / a.T;
  const size_t k = (size_t)a.k;
  for (int x = lane; x < a.k; x += 64) {
    float s = a.P[(size_t)(2 * (c - 1) + 1) * k + x];
    for (int cc = c; cc <= c1; ++cc) s += a.P[(size_t)(2 * cc) * k + x];
    if (accumulate) s += a.C[(size_t)r * k + x];
    if (EPI) {
      if (a.bias) s += a.bias[x];
      if (a.relu) s = fmaxf(s, 0.f);
    }
    a.C[(size_t)r * k + x] = s;
  }
}

// ---------------------------------------------------------------------------
// empty rows: counted once per plan, written by a pass of their own (every main kernel skips them)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
count_empty_rows_kernel(const int* __restrict__ rowptr, int m, int* __restrict__ count) {
  int mine = 0;
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (long long)gridDim.x * blockDim.x)
    mine += rowptr[r] == rowptr[r + 1];
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}

hipError_t launch_count_empty_rows(const int* rowptr, int m, int* count_dev, hipStream_t s) {
  if (m <= 0) return hipSuccess;
  int nb = (m + 255) / 256;
  if (nb > 4096) nb = 4096;
  count_empty_rows_kernel<<<nb, 256, 0, s>>>(rowptr, m, count_dev);
  return hipGetLastError();
}

// one wave looks at 64 rows at a time and writes the empty ones among them, a whole row per step
template <int VEC>
__global__ void __launch_bounds__(256)
fill_empty_rows_kernel(const int* __restrict__ rowptr, float* __restrict__ C, const float* __restrict__ bias,
                       int relu, int accumulate, int m, int k) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  for (long long base = wave * 64; base < m; base += nwaves * 64) {
    const long long r = base + lane;
    unsigned long long mask = __ballot(r < m && rowptr[r] == rowptr[r + 1]);
    while (mask) {
      const int j = __builtin_ctzll(mask);
      mask &= mask - 1;
      float* row = C + (size_t)(base + j) * (size_t)k;
      for (int x = lane * VEC; x < k; x += 64 * VEC) {
        float v[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = 0.f;
        if (accumulate) load_vec<VEC>(row + x, v);
        if (bias) {
          float b[VEC];
          load_vec<VEC>(bias + x, b);
#pragma unroll
          for (int i = 0; i < VEC; ++i) v[i] += b[i];
        }
        if (relu) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        store_vec<VEC>(row + x, v);
      }
    }
  }
}

hipError_t launch_fill_empty_rows(const int* rowptr, float* C, const float* bias, int relu, int accumulate, int m, int k,
                                  hipStream_t s) {
  if (m <= 0 || k <= 0) return hipSuccess;
  if (accumulate && !bias && !relu) return hipSuccess;         // C += 0: nothing to write
  long long nb = ((long long)m + 255) / 256;                   // a wave per 64 rows
  if (nb > 16384) nb = 16384;
  const uintptr_t al = (uintptr_t)C | (uintptr_t)bias;
  if (k % 4 == 0 && (al & 15) == 0) fill_empty_rows_kernel<4><<<(int)nb, 256, 0, s>>>(rowptr, C, bias, relu, accumulate, m, k);
  else                              fill_empty_rows_kernel<1><<<(int)nb, 256, 0, s>>>(rowptr, C, bias, relu, accumulate, m, k);
  return hipGetLastError();
}

// nnz == 0: C = act(bias) (or zeros)
__global__ void spmm_empty_kernel(float* C, const float* bias, int relu, long long total, int k) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    float s = bias ? bias[i % k] : 0.f;
    if (relu) s = fmaxf(s, 0.f);
    C[i] = s;
  }
}

// ---------------------------------------------------------------------------
// row gather  dst[r,:] = src[idx[r],:]   (permutate.cu:3-21 counterpart)
// one wave per row, float4 when k%4==0, grid-stride over rows.
// ---------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256)
gather_rows_kernel(float* __restrict__ dst, const float* __restrict__ src,
                   const int* __restrict__ idx, int nrows, int k) {
  const int lane = threadIdx.x & 63;
  const int wave = sgpr(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int nwaves = gridDim.x * 4;
  for (int r = wave; r < nrows; r += nwaves) {
    const int s = idx[r];
    const float* sp = src + (size_t)s * k;
    float* dp = dst + (size_t)r * k;
    for (int x = lane * VEC; x < k; x += 64 * VEC) {
      float t[VEC];
      load_vec<VEC>(sp + x, t);
      store_vec<VEC>(dp + x, t);
    }
  }
}

// ---------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------
hipError_t launch_plan_chunk_rows(const int* rowptr, int m, int T, int nchunks,
                                  int* chunk_row, hipStream_t s) {
  if (nchunks <= 0) return hipSuccess;
  const int bs = 256;
  plan_chunk_rows_kernel<<<(nchunks + bs - 1) / bs, bs, 0, s>>>(rowptr, m, T, nchunks, chunk_row);
  return hipGetLastError();
}

template <int VEC, int U>
static hipError_t launch_main(const SpmmArgs& a, int nblocks, bool epi, hipStream_t s) {
  // One launch per 64*VEC-column tile, back to back on the stream: a pass only touches its
  // own column slice of B, so the gathered working set per pass is n*64*VEC*4 bytes — the
  // narrower the slice, the larger the share of it that stays in L2 / Infinity Cache.
  const int tiles = (a.k + 64 * VEC - 1) / (64 * VEC);
  dim3 grid(nblocks), block(256);
#define GCN_MAIN_ARGS a.rowptr, a.col, a.val, a.B, a.C, a.P, a.chunk_row, a.bias, a.nnz_dev, \
                      a.relu, a.nchunks, a.T, a.m, a.nnz, a.k, t, a.accumulate, ldb
  const int ldb = a.ldb > 0 ? a.ldb : a.k;
  // buffer addressing needs every byte offset into B to fit 32 bits
  const bool buf = VEC == 1 && (unsigned long long)a.n * (unsigned long long)ldb * 4ull < 0xFFFFFFF0ull;
  for (int t = 0; t < tiles; ++t) {
    if constexpr (VEC == 1) {
      if (buf) {
        if (epi) spmm_chunk_kernel<VEC, U, true, true><<<grid, block, 0, s>>>(GCN_MAIN_ARGS);
        else     spmm_chunk_kernel<VEC, U, false, true><<<grid, block, 0, s>>>(GCN_MAIN_ARGS);
        continue;
      }
    }
    {
      if (epi) spmm_chunk_kernel<VEC, U, true, false><<<grid, block, 0, s>>>(GCN_MAIN_ARGS);
      else     spmm_chunk_kernel<VEC, U, false, false><<<grid, block, 0, s>>>(GCN_MAIN_ARGS);
    }
  }
#undef GCN_MAIN_ARGS
  return hipGetLastError();
}

// floats per lane (1, 2 or 4): the column tile is 64*VEC wide.  tile_cols = 0 picks the
// widest tile that k fills; otherwise the requested tile width (64 / 128 / 256) is honoured
// when k and the pointers allow the vector width.
int pick_vec(int k, int tile_cols, const void* B, const void* C, const void* P) {
  const uintptr_t al = (uintptr_t)B | (uintptr_t)C | (uintptr_t)P;
  int want = tile_cols > 0 ? tile_cols / 64 : (k > 128 ? 4 : (k > 64 ? 2 : 1));
  if (want >= 4 && k % 4 == 0 && (al & 15) == 0) return 4;
  if (want >= 2 && k % 2 == 0 && (al & 7) == 0) return 2;
  return 1;
}

// the quad kernel (spmm_quad.hip) runs whenever its layout applies: k % 4 == 0, 16-byte aligned operands,
// 32-bit byte offsets; for k > 32 only where the 64-column tile is the chosen tile width
constexpr int kQuadMinRowLen = 48;

static bool use_quad(const SpmmArgs& a) {
  constexpr int min_k = 12;               // (below: the narrow kernels' 4-byte-per-lane gathers, spmm_narrow.hip)
  if (a.gather_width == 1 || a.k < min_k || !spmm_quad_eligible(a)) return false;
  // Short rows: every finished row costs the quad layout a cross-lane reduction (8-16 shuffles) where
  // the one-per-gather kernel just stores.  R-MAT, n = 1 M, k = 64 (profiles/r01f_lowdeg_probe.log), one-
  // vs four-per-gather: mean degree 4.9: 0.229 vs 0.508 ms; 8.8: 0.33 vs 0.56; 16: 0.54 vs 0.64;
  // 31: 0.87 vs 0.89; 59: 1.573 vs 1.564 -> the quad kernel from ~48 non-zeros per (virtual) row up.
  const long long nnz = a.nnz_dev ? (long long)a.nchunks_grid * a.T : (long long)a.nnz;
  if (a.gather_width != 4 && a.m > 0 && nnz / a.m < kQuadMinRowLen) return false;
  return a.k <= 32 || pick_vec(a.k, a.tile_cols, a.B, a.C, a.P) == 1;
}

bool spmm_will_use_quad(const SpmmArgs& a) { return a.nchunks_grid > 0 && use_quad(a); }

hipError_t launch_spmm(const SpmmArgs& a, int cu_count, hipStream_t s) {
  const bool epi = (a.bias != nullptr) || a.relu;
  if (a.m <= 0 || a.k <= 0) return hipSuccess;
  const int ng = a.nchunks_grid;          // chunk count the grids are sized for (>= actual)
  if (ng == 0) {
    const long long total = (long long)a.m * a.k;
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    spmm_empty_kernel<<<nb, 256, 0, s>>>(a.C, a.bias, a.relu, total, a.k);
    return hipGetLastError();
  }
  // grid: blocks_per_cu blocks of 4 waves per CU (chunks strided over the waves of an XCD), a multiple
  // of 8 blocks (XCDs); the default 32 oversubscribes the CUs on purpose (spmm_kernels.h)
  int nblocks = (ng + 3) / 4;
  const int cap = cu_count * (a.blocks_per_cu > 0 && a.blocks_per_cu <= 64 ? a.blocks_per_cu : 32);
  if (nblocks > cap) nblocks = cap;
  nblocks = (nblocks + 7) & ~7;
  hipError_t e;
  // the empty rows first (the main kernels never write them), outside the timed interval of the main kernel
  if (a.empty_rows != 0 && (e = launch_fill_empty_rows(a.rowptr, a.C, a.bias, a.relu, a.accumulate, a.m, a.k, s)) != hipSuccess) return e;
  if (a.ev_start && (e = hipEventRecord(a.ev_start, s)) != hipSuccess) return e;
  if (a.valless && !use_quad(a)) return hipErrorInvalidValue;
  if (use_quad(a)) {
    // 16-byte-per-lane gathers, 4 (k > 16) or 16 (k <= 16) non-zeros per instruction (spmm_quad.hip)
    e = launch_spmm_quad(a, nblocks, epi, s);
  } else if (a.k <= 16 && (a.ldb == 0 || a.ldb == a.k)) {
    // k <= 16 without the alignment the quad kernel needs: several non-zeros per 4-byte-per-lane gather
    e = launch_spmm_narrow(a, nblocks, epi, s);
  } else switch (pick_vec(a.k, a.tile_cols, a.B, a.C, a.P)) {
    case 4:  e = launch_main<4, 4>(a, nblocks, epi, s); break;
    case 2:  e = launch_main<2, 8>(a, nblocks, epi, s); break;
    default:
      // 32 gathers in flight per wave for the 64-column tile (measured on the sliced Reddit-shaped case, whole SpMM:
      // U = 4 / 8 / 16 / 32 -> 5.16 / 4.33 / 4.18 / 4.09 ms; 52 VGPRs at U = 32, still 8 waves per SIMD)
      e = launch_main<1, 32>(a, nblocks, epi, s);
      break;
  }
  if (e != hipSuccess) return e;
  if (a.ev_stop && (e = hipEventRecord(a.ev_stop, s)) != hipSuccess) return e;
  if (ng > 1) {
    const int nb = (ng - 1 + 3) / 4;
    if (epi) spmm_fixup_kernel<true><<<nb, 256, 0, s>>>(a.rowptr, a.P, a.C, a.chunk_row, a.bias,
                                                        a.nnz_dev, a.relu, a.nchunks, a.T, a.k, a.accumulate);
    else     spmm_fixup_kernel<false><<<nb, 256, 0, s>>>(a.rowptr, a.P, a.C, a.chunk_row, a.bias,
                                                         a.nnz_dev, a.relu, a.nchunks, a.T, a.k, a.accumulate);
    e = hipGetLastError();
  }
  return e;
}

// Name (as rocprofv3 prints it) of the main kernel launch_spmm picks for these arguments; mirrors the
// selection above.  Used by gcn_spmm_plan_main_kernel so that a benchmark reports the kernel that runs.
void describe_main_kernel(const SpmmArgs& a, char* buf, size_t len) {
  const bool epi = (a.bias != nullptr) || a.relu;
  const char* e = epi ? "true" : "false";
  const bool buf32 = (unsigned long long)a.n * (unsigned long long)(a.ldb > 0 ? a.ldb : a.k) * 4ull < 0xFFFFFFF0ull;
  if (a.nchunks_grid == 0) { snprintf(buf, len, "gcn::spmm_empty_kernel"); return; }
  if (use_quad(a)) {
    snprintf(buf, len, "gcn::spmm_quad_kernel<%d, %s, %s, %s>", spmm_quad_lanes(a.k), e, a.valless ? "true" : "false",
             a.col16 ? "true" : "false");
    return;
  }
  if (a.k <= 16 && (a.ldb == 0 || a.ldb == a.k)) {
    if (a.k > 8 && buf32 && a.n < (1 << 24)) snprintf(buf, len, "gcn::spmm_narrow16_dpp_kernel<%s>", e);
    else {
      const int g = a.k <= 4 ? 4 : (a.k <= 8 ? 8 : 16);
      snprintf(buf, len, "gcn::spmm_narrow_kernel<%d, %d, %s, %s>", g, g, e, buf32 ? "true" : "false");
    }
    return;
  }
  const int vec = pick_vec(a.k, a.tile_cols, a.B, a.C, a.P);
  if (vec == 4) { snprintf(buf, len, "gcn::spmm_chunk_kernel<4, 4, %s, false>", e); return; }
  if (vec == 2) { snprintf(buf, len, "gcn::spmm_chunk_kernel<2, 8, %s, false>", e); return; }
  snprintf(buf, len, "gcn::spmm_chunk_kernel<1, 32, %s, %s>", e, buf32 ? "true" : "false");
}

// dst[r, 0:k] = src[r, 0:k], dst[r, k:ld] = 0: feature rows re-laid on whole 128-byte lines
__global__ void __launch_bounds__(256)
pad_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, long long rows, int k, int ld,
                const float* __restrict__ rowscale) {
  const long long total = rows * ld;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / ld;
    const int c = (int)(i - r * ld);
    dst[i] = c < k ? (rowscale ? rowscale[r] * src[r * k + c] : src[r * k + c]) : 0.f;
  }
}

hipError_t launch_pad_rows(float* dst, const float* src, long long rows, int k, int ld, hipStream_t s,
                           const float* rowscale) {
  if (rows <= 0 || k <= 0) return hipSuccess;
  long long nb = (rows * ld + 255) / 256;
  if (nb > 65536) nb = 65536;
  pad_rows_kernel<<<(int)nb, 256, 0, s>>>(dst, src, rows, k, ld, rowscale);
  return hipGetLastError();
}

// dst[r, 0:k] = act(src[r, 0:k] + bias): the compact result out of a row-padded one (src row stride ld)
__global__ void __launch_bounds__(256)
unpad_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, const float* __restrict__ bias,
                  int relu, long long rows, int k, int ld, const int* __restrict__ guard) {
  if (guard && *guard == 0) return;
  const long long total = rows * k;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / k;
    const int c = (int)(i - r * k);
    float v = src[r * ld + c];
    if (bias) v += bias[c];
    if (relu) v = fmaxf(v, 0.f);
    dst[i] = v;
  }
}

hipError_t launch_unpad_rows(float* dst, const float* src, const float* bias, int relu, long long rows, int k,
                             int ld, hipStream_t s, const int* guard) {
  if (rows <= 0 || k <= 0) return hipSuccess;
  long long nb = (rows * k + 255) / 256;
  if (nb > 65536) nb = 65536;
  unpad_rows_kernel<<<(int)nb, 256, 0, s>>>(dst, src, bias, relu, rows, k, ld, guard);
  return hipGetLastError();
}

// dst[i] = dropout(src[i]): the mask of the fused epilogue (philox.h) as a pass of its own — used where no
// epilogue pass exists to carry it (unsliced matrices) and for the backward pass (the same mask on the gradient)
__global__ void __launch_bounds__(256)
dropout_kernel(float* __restrict__ dst, const float* __restrict__ src, long long total, DropoutSpec drop, int vec4) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (vec4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < total; i += stride)
      reinterpret_cast<float4*>(dst)[i] = dropout_apply4(drop, (unsigned long long)i * 4, reinterpret_cast<const float4*>(src)[i]);
  } else {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
      dst[i] = dropout_apply(drop, (unsigned long long)i, src[i]);
  }
}

hipError_t launch_dropout(float* dst, const float* src, long long total, const DropoutSpec& drop, hipStream_t st) {
  if (total <= 0) return hipSuccess;
  const int vec4 = (total % 4 == 0) && ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0);
  long long nb = ((vec4 ? total / 4 : total) + 255) / 256;
  if (nb > 65536) nb = 65536;
  dropout_kernel<<<(int)nb, 256, 0, st>>>(dst, src, total, drop, vec4);
  return hipGetLastError();
}

hipError_t launch_gather_rows(float* dst, const float* src, const int* idx, int nrows, int k,
                              hipStream_t s) {
  if (nrows <= 0 || k <= 0) return hipSuccess;
  int nb = (nrows + 3) / 4;
  if (nb > 8192) nb = 8192;
  const uintptr_t al = (uintptr_t)dst | (uintptr_t)src;
  if (k % 4 == 0 && (al & 15) == 0) gather_rows_kernel<4><<<nb, 256, 0, s>>>(dst, src, idx, nrows, k);
  else                              gather_rows_kernel<1><<<nb, 256, 0, s>>>(dst, src, idx, nrows, k);
  return hipGetLastError();
}

}  // namespace gcn
