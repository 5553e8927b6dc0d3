// spmm_narrow.hip — the SpMM kernel for narrow feature widths, k <= 32 (GCN hidden / class sizes;
// the reference dedicates four of its five kernels to them: flexspmm.cu:17-422, k4/k8/k16/k32).
//
// With k <= 32 a gathered feature row is only 16..128 bytes, so the wide kernel's "one wave
// instruction per non-zero" spends its time on instruction issue, not on bytes (measured: every
// k <= 32 took the same 1.66 ms on the Reddit-shaped graph).  Here a wave instruction covers
// NPI = 64/G non-zeros at once (G = 4/8/16/32 lanes per non-zero, the power of two >= k):
//
//   lane = sub*G + f:  sub = which of the NPI non-zeros of this step, f = feature column.
//   The column index and value of non-zero (j + sub) come from the coalesced 64-entry block by
//   one cross-lane read each; the gather is ONE load instruction with a per-lane offset
//   col*k*4 + f*4 (NPI rows of k floats); each lane accumulates its (sub, f) partial sum.
//   When a row ends — anywhere inside a step — the lanes that belong to it are reduced across
//   `sub` (log2(NPI) xor-shuffles) and lanes sub == 0 store the k floats.
//
// Chunk schedule, partial slab and fix-up are those of spmm_kernels.hip (same plan), so results
// are deterministic; the summation order inside a row differs from the wide kernel (a tree over
// `sub` instead of a chain), within the 1e-5 contract.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spmm_kernels.h"

namespace gcn {

__device__ __forceinline__ int nsgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int G, int U, bool EPI, bool BUF>
__global__ void __launch_bounds__(256)
spmm_narrow_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                   const float* __restrict__ val, const float* __restrict__ B,
                   float* __restrict__ C, float* __restrict__ P,
                   const int* __restrict__ chunk_row, const float* __restrict__ bias,
                   const int* __restrict__ nnz_dev,
                   int relu, int nchunks, int T, int m, int nnz, int k) {
  constexpr int NPI = 64 / G;                       // non-zeros per gather instruction
  if (nnz_dev) {                                    // drop-in (flexspmm) mode, see spmm_kernels.hip
    nnz = *nnz_dev;
    nchunks = (int)(((long long)nnz + T - 1) / T);
    val = reinterpret_cast<const float*>(col) + nnz;
  }
  const int lane = threadIdx.x & 63;
  const int wib  = nsgpr(threadIdx.x >> 6);
  const int sub  = lane / G;
  const int f    = lane % G;
  const bool fok = f < k;

  const int xcd           = blockIdx.x & 7;
  const int wave_in_xcd   = (blockIdx.x >> 3) * 4 + wib;
  const int waves_per_xcd = (gridDim.x >> 3) * 4;
  const int c_lo = (int)(((long long)nchunks * xcd) >> 3);
  const int c_hi = (int)(((long long)nchunks * (xcd + 1)) >> 3);

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(B), 0, 0xFFFFFFFFu, 0x00020000);
  const unsigned row_bytes = (unsigned)k * 4u;
  const float bias_f = (EPI && bias && fok) ? bias[f] : 0.f;

  for (int c = c_lo + wave_in_xcd; c < c_hi; c += waves_per_xcd) {
    const int start = c * T;
    const int end   = (int)min((long long)start + T, (long long)nnz);
    int r = chunk_row[c];
    int row_end    = rowptr[r + 1];
    int row_end_nx = (r + 1 < m) ? rowptr[r + 2] : -1;
    bool head = rowptr[r] < start;
    int pos = start;                                // position of the first non-zero not yet summed
    int last_flush = start;
    float acc = 0.f;                                // this lane's (sub, f) partial of the current row

    // the current row is complete at `pos`: reduce over sub, write, step to the next row
    auto flush = [&]() {
      float t = acc;
#pragma unroll
      for (int off = G; off < 64; off <<= 1) t += __shfl_xor(t, off);
      if (head) {
        if (sub == 0 && fok) P[(size_t)(2 * c) * k + f] = t;
      } else if (last_flush != pos) {               // (empty rows belong to launch_fill_empty_rows, spmm_kernels.hip)
        if (EPI) {
          t += bias_f;
          if (relu) t = fmaxf(t, 0.f);
        }
        if (sub == 0 && fok) C[(size_t)r * k + f] = t;
      }
      acc = 0.f;
      head = false;
      last_flush = pos;
      ++r;
      row_end    = row_end_nx;
      if (row_end == pos) {                         // row r is empty: jump over the whole run of empty rows
        r = next_nonempty_row(rowptr, r, m, pos);
        row_end = (r < m) ? rowptr[r + 1] : -1;
      }
      row_end_nx = (r + 1 < m) ? rowptr[r + 2] : -1;
    };

    while (pos == row_end) flush();                 // leading empty rows (chunk 0 only)

    int   cj_nx = 0;
    float vj_nx = 0.f;
    if (start + lane < end) { cj_nx = col[start + lane]; vj_nx = val[start + lane]; }
    for (int base = start; base < end; base += 64) {
      const int cnt = min(64, end - base);
      const int   cj = cj_nx;
      const float vj = vj_nx;
      if (base + 64 + lane < end) { cj_nx = col[base + 64 + lane]; vj_nx = val[base + 64 + lane]; }

      for (int j = 0; j < cnt; j += U * NPI) {
        float prod[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {               // U gather instructions back to back
          const int e = j + u * NPI + sub;          // this lane's non-zero inside the 64-block
          const int   cu = __shfl(cj, e & 63);
          const float vu = __shfl(vj, e & 63);
          float b = 0.f;
          if (e < cnt && fok) {
            if (BUF) {
              b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                      rsrc, (int)((unsigned)cu * row_bytes + (unsigned)f * 4u), 0, 0));
            } else {
              b = B[(size_t)cu * (size_t)k + f];
            }
          }
          prod[u] = (e < cnt && fok) ? vu * b : 0.f;  // lanes past the block's end / past k add nothing
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int n_step = min(NPI, cnt - (j + u * NPI));     // non-zeros this step really holds
          if (n_step <= 0) break;
          if (row_end < 0 || row_end - pos > n_step) {          // fast path: no row ends in this step
            acc += prod[u];
            pos += n_step;
            continue;
          }
          int done = 0;                             // sub-slots of this step already summed
          while (true) {
            const int rel = row_end - pos;          // non-zeros the current row still has from `pos`
            if (row_end < 0 || rel > n_step - done) {           // row continues past this step
              acc += (sub >= done) ? prod[u] : 0.f;
              pos += n_step - done;
              break;
            }
            acc += (sub >= done && sub < done + rel) ? prod[u] : 0.f;
            pos += rel;
            done += rel;
            flush();                                // (may run again at once for empty rows)
            if (done == n_step && pos != row_end) break;
          }
        }
      }
    }

    if (last_flush != end) {                        // the row piece that sticks out of the chunk
      float t = acc;
#pragma unroll
      for (int off = G; off < 64; off <<= 1) t += __shfl_xor(t, off);
      const int slot = head ? 2 * c : 2 * c + 1;
      if (sub == 0 && fok) P[(size_t)slot * k + f] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k in 9..16 (G = 16, four non-zeros per gather) with the cross-lane traffic on DPP instead of
// ds_bpermute: the 64-entry (col, val) block is loaded TRANSPOSED — lane s*16 + u holds entry
// u*4 + s — so that at step u every 16-lane DPP row broadcasts its lane u (`row_newbcast:u`,
// a modifier on a full-rate v_mov) and lane (s, f) receives exactly the non-zero 4u + s it
// gathers for.  Per-lane byte offsets use the 24-bit multiplier (needs n < 2^24 and buffer
// addressing; otherwise the generic kernel above runs).
// ---------------------------------------------------------------------------------------------
template <int UU>
__device__ __forceinline__ int dpp_row_bcast(int v) {
  return __builtin_amdgcn_mov_dpp(v, 0x150 + UU, 0xf, 0xf, true);               // row_newbcast:UU
}

template <bool EPI>
__global__ void __launch_bounds__(256)
spmm_narrow16_dpp_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                         const float* __restrict__ val, const float* __restrict__ B,
                         float* __restrict__ C, float* __restrict__ P,
                         const int* __restrict__ chunk_row, const float* __restrict__ bias,
                         const int* __restrict__ nnz_dev,
                         int relu, int nchunks, int T, int m, int nnz, int k) {
  constexpr int NPI = 4;
  if (nnz_dev) {
    nnz = *nnz_dev;
    nchunks = (int)(((long long)nnz + T - 1) / T);
    val = reinterpret_cast<const float*>(col) + nnz;
  }
  const int lane = threadIdx.x & 63;
  const int wib  = nsgpr(threadIdx.x >> 6);
  const int sub  = lane >> 4;
  const int f    = lane & 15;
  const bool fok = f < k;
  const int tl   = (lane & 15) * 4 + (lane >> 4);   // transposed position this lane loads

  const int xcd           = blockIdx.x & 7;
  const int wave_in_xcd   = (blockIdx.x >> 3) * 4 + wib;
  const int waves_per_xcd = (gridDim.x >> 3) * 4;
  const int c_lo = (int)(((long long)nchunks * xcd) >> 3);
  const int c_hi = (int)(((long long)nchunks * (xcd + 1)) >> 3);

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(B), 0, 0xFFFFFFFFu, 0x00020000);
  const unsigned row_bytes = (unsigned)k * 4u;
  const unsigned foff = (unsigned)f * 4u;
  const float bias_f = (EPI && bias && fok) ? bias[f] : 0.f;

  for (int c = c_lo + wave_in_xcd; c < c_hi; c += waves_per_xcd) {
    const int start = c * T;
    const int end   = (int)min((long long)start + T, (long long)nnz);
    int r = chunk_row[c];
    int row_end    = rowptr[r + 1];
    int row_end_nx = (r + 1 < m) ? rowptr[r + 2] : -1;
    bool head = rowptr[r] < start;
    int pos = start;
    int last_flush = start;
    float acc = 0.f;

    auto flush = [&]() {
      float t = acc;
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      if (head) {
        if (sub == 0 && fok) P[(size_t)(2 * c) * k + f] = t;
      } else if (last_flush != pos) {               // (empty rows belong to launch_fill_empty_rows, spmm_kernels.hip)
        if (EPI) {
          t += bias_f;
          if (relu) t = fmaxf(t, 0.f);
        }
        if (sub == 0 && fok) C[(size_t)r * k + f] = t;
      }
      acc = 0.f;
      head = false;
      last_flush = pos;
      ++r;
      row_end    = row_end_nx;
      if (row_end == pos) {                         // row r is empty: jump over the whole run of empty rows
        r = next_nonempty_row(rowptr, r, m, pos);
        row_end = (r < m) ? rowptr[r + 1] : -1;
      }
      row_end_nx = (r + 1 < m) ? rowptr[r + 2] : -1;
    };
    // one step = the 4 non-zeros at [pos, pos + n_step)
    auto step = [&](float prod, int n_step) {
      if (row_end < 0 || row_end - pos > n_step) { acc += prod; pos += n_step; return; }
      int done = 0;
      while (true) {
        const int rel = row_end - pos;
        if (row_end < 0 || rel > n_step - done) {
          acc += (sub >= done) ? prod : 0.f;
          pos += n_step - done;
          return;
        }
        acc += (sub >= done && sub < done + rel) ? prod : 0.f;
        pos += rel;
        done += rel;
        flush();
        if (done == n_step && pos != row_end) return;
      }
    };

    while (pos == row_end) flush();

    int   cj_nx = 0;
    float vj_nx = 0.f;
    if (start + tl < end) { cj_nx = col[start + tl]; vj_nx = val[start + tl]; }
    for (int base = start; base < end; base += 64) {
      const int cnt = min(64, end - base);
      const int cj = cj_nx;
      const int vj = __builtin_bit_cast(int, vj_nx);
      if (base + 64 + tl < end) { cj_nx = col[base + 64 + tl]; vj_nx = val[base + 64 + tl]; }

#define GCN_N16_GATHER(UU)                                                                      \
      {                                                                                         \
        const int cu = dpp_row_bcast<UU>(cj);                                                   \
        const float vu = __builtin_bit_cast(float, dpp_row_bcast<UU>(vj));                      \
        const bool ok = (UU * 4 + sub < cnt) && fok;                                            \
        float b = 0.f;                                                                          \
        if (ok) b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(             \
                        rsrc, (int)(__umul24((unsigned)cu, row_bytes) + foff), 0, 0));          \
        prod[UU] = ok ? vu * b : 0.f;                                                       \
      }
#define GCN_N16_STEP(UU)                                                                        \
      { const int ns = min(NPI, cnt - UU * 4); if (ns > 0) step(prod[UU], ns); }
      float prod[16];
      GCN_N16_GATHER(0) GCN_N16_GATHER(1) GCN_N16_GATHER(2) GCN_N16_GATHER(3)
      GCN_N16_GATHER(4) GCN_N16_GATHER(5) GCN_N16_GATHER(6) GCN_N16_GATHER(7)
      GCN_N16_GATHER(8) GCN_N16_GATHER(9) GCN_N16_GATHER(10) GCN_N16_GATHER(11)
      GCN_N16_GATHER(12) GCN_N16_GATHER(13) GCN_N16_GATHER(14) GCN_N16_GATHER(15)
      GCN_N16_STEP(0) GCN_N16_STEP(1) GCN_N16_STEP(2) GCN_N16_STEP(3)
      GCN_N16_STEP(4) GCN_N16_STEP(5) GCN_N16_STEP(6) GCN_N16_STEP(7)
      GCN_N16_STEP(8) GCN_N16_STEP(9) GCN_N16_STEP(10) GCN_N16_STEP(11)
      GCN_N16_STEP(12) GCN_N16_STEP(13) GCN_N16_STEP(14) GCN_N16_STEP(15)
#undef GCN_N16_GATHER
#undef GCN_N16_STEP
    }

    if (last_flush != end) {
      float t = acc;
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      const int slot = head ? 2 * c : 2 * c + 1;
      if (sub == 0 && fok) P[(size_t)slot * k + f] = t;
    }
  }
}

template <int G, int U>
static hipError_t launch_g(const SpmmArgs& a, int nblocks, bool epi, bool buf, hipStream_t s) {
#define GCN_NARROW_ARGS a.rowptr, a.col, a.val, a.B, a.C, a.P, a.chunk_row, a.bias, a.nnz_dev, \
                        a.relu, a.nchunks, a.T, a.m, a.nnz, a.k
  dim3 grid(nblocks), block(256);
  if (buf) {
    if (epi) spmm_narrow_kernel<G, U, true, true><<<grid, block, 0, s>>>(GCN_NARROW_ARGS);
    else     spmm_narrow_kernel<G, U, false, true><<<grid, block, 0, s>>>(GCN_NARROW_ARGS);
  } else {
    if (epi) spmm_narrow_kernel<G, U, true, false><<<grid, block, 0, s>>>(GCN_NARROW_ARGS);
    else     spmm_narrow_kernel<G, U, false, false><<<grid, block, 0, s>>>(GCN_NARROW_ARGS);
  }
#undef GCN_NARROW_ARGS
  return hipGetLastError();
}

// k <= 32 only
hipError_t launch_spmm_narrow(const SpmmArgs& a, int nblocks, bool epi, hipStream_t s) {
  const bool buf = (unsigned long long)a.n * (unsigned long long)a.k * 4ull < 0xFFFFFFF0ull;
  // U gather instructions per batch: one 64-entry (col, val) block per batch where that fits
  if (a.k <= 4)  return launch_g<4, 4>(a, nblocks, epi, buf, s);
  if (a.k <= 8)  return launch_g<8, 8>(a, nblocks, epi, buf, s);
  if (a.k <= 16) {
    if (a.k > 8 && buf && a.n < (1 << 24) && a.k * 4 < (1 << 24)) {   // DPP + 24-bit offsets
#define GCN_N16_ARGS a.rowptr, a.col, a.val, a.B, a.C, a.P, a.chunk_row, a.bias, a.nnz_dev, \
                     a.relu, a.nchunks, a.T, a.m, a.nnz, a.k
      if (epi) spmm_narrow16_dpp_kernel<true><<<dim3(nblocks), dim3(256), 0, s>>>(GCN_N16_ARGS);
      else     spmm_narrow16_dpp_kernel<false><<<dim3(nblocks), dim3(256), 0, s>>>(GCN_N16_ARGS);
#undef GCN_N16_ARGS
      return hipGetLastError();
    }
    return launch_g<16, 16>(a, nblocks, epi, buf, s);
  }
  return launch_g<32, 16>(a, nblocks, epi, buf, s);
}

}  // namespace gcn
