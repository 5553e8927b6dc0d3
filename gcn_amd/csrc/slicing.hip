// slicing.hip — XCD-aware column slicing of the adjacency (plan-time) + the reduction that
// follows a sliced SpMM.
//
// Why: on an unordered graph ≈92 % of the feature-row gathers miss the 4 MiB per-XCD L2 and are
// served by the Infinity Cache, whose random-row bandwidth (≈7.2 TB/s) is what bounds
// spmm_chunk_kernel (profiles/r01_pmc_*.json).  Slicing cuts the column range [0, n) into S
// equal slices and reorders the non-zero stream SLICE-MAJOR:
//
//     virtual row  R = s·m + r   holds the non-zeros of row r whose column lies in slice s.
//
// The SpMM kernel then runs unchanged on this virtual CSR (S·m rows).  Because each XCD walks a
// CONTIGUOUS range of the chunk stream (spmm_chunk_kernel, XCD-aware ranges), an XCD gathers
// from only ≈ S/8 slices of B, one after the other — a slice of n/S rows × 256 B (64-column
// tile) is sized to sit in that XCD's L2.  The kernel writes one partial output row per virtual
// row (plain stores, each exactly once); slice_reduce_kernel adds the S partials of every real
// row in slice order (deterministic) and applies the optional bias/ReLU epilogue.
//
// Requires columns sorted ascending inside each row (true for the reference's pipeline:
// renumber.cu:105-117 sorts, torch's to_sparse_csr sorts); rows' slice segments are then
// contiguous, so the reordering is a segment copy.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>
#include "spmm_kernels.h"

namespace gcn {

hipError_t verify_value_factors(const int* rowptr, const int* col, const float* val, const float* u_row,
                                const float* u_col, int m, int* ok_host, hipStream_t st);

// cnt[s*m + r] = number of non-zeros of row r with column in [s*w, (s+1)*w);  *unsorted is set
// when a row's columns are not ascending.  One thread per (row, slice boundary).
__global__ void slice_count_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                   int m, int S, int w, int* __restrict__ split,
                                   int* __restrict__ unsorted) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)m * (S + 1)) return;
  const int r = (int)(t / (S + 1)), s = (int)(t % (S + 1));
  const int lo0 = rowptr[r], hi0 = rowptr[r + 1];
  int lo = lo0, hi = hi0;
  const long long bound = (long long)s * w;
  while (lo < hi) {                       // first entry with col >= s*w
    const int mid = (lo + hi) >> 1;
    if ((long long)col[mid] < bound) lo = mid + 1; else hi = mid;
  }
  split[t] = (s == S) ? hi0 : lo;         // the last boundary is the row end (cols < n <= S*w)
}

__global__ void slice_check_sorted_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                          int m, int* __restrict__ unsorted) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nw = gridDim.x * (blockDim.x >> 6);
  for (int r = wave; r < m; r += nw)
    for (int e = rowptr[r] + 1 + lane; e < rowptr[r + 1]; e += 64)
      if (col[e] < col[e - 1]) *unsorted = 1;
}

__global__ void slice_sizes_kernel(const int* __restrict__ split, int m, int S, int* __restrict__ cnt) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)m * S) return;
  const int s = (int)(t / m), r = (int)(t % m);
  cnt[t] = split[(long long)r * (S + 1) + s + 1] - split[(long long)r * (S + 1) + s];
}

// copy every (row, slice) segment to its slice-major position; one wave per row
__global__ void __launch_bounds__(256)
slice_scatter_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                     const float* __restrict__ val, const int* __restrict__ split,
                     const int* __restrict__ vrowptr, int m, int S, int w,
                     int* __restrict__ vcol, float* __restrict__ vval) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int r = wave; r < m; r += nw) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    for (int e = beg + lane; e < end; e += 64) {
      const int c = col[e];
      const int s = c / w;
      const int dst = vrowptr[(long long)s * m + r] + (e - split[(long long)r * (S + 1) + s]);
      vcol[dst] = c;
      vval[dst] = val[e];
    }
  }
}

// C[r, :] = act( [C[r, :] +] sum_s Cv[s*m + r, :] + bias ), partials added in slice order; wave per row
template <int VEC>
__global__ void __launch_bounds__(256)
slice_reduce_kernel(const float* __restrict__ Cv, float* __restrict__ C,
                    const float* __restrict__ bias, int relu, int m, int S, int k, int accumulate,
                    const float* __restrict__ rowscale, DropoutSpec drop, const int* __restrict__ guard,
                    const float* __restrict__ outscale, int gap_w, CutLists cuts) {
  if (guard && *guard == 0) return;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int r = wave; r < m; r += nw) {
    const int q0 = cuts.ptr ? cuts.ptr[r] : 0, q1 = cuts.ptr ? cuts.ptr[r + 1] : 0;   // head pieces of this row's cut slices
    const float rs = (rowscale ? rowscale[r] : 1.f);
    const float os = outscale ? outscale[r] : 1.f;             // the consumer's column factor (pre-laid output, see below)
    const size_t orow = gap_w > 0 ? (size_t)r + (size_t)(r / gap_w) : (size_t)r;    // value-free main pass: the row's own factor u[r]
    for (int x = lane * VEC; x < k; x += 64 * VEC) {
      float acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      for (int s = 0; s < S; ++s) {
        const float* p = Cv + ((size_t)s * m + r) * (size_t)k + x;
        if (VEC == 4) {
          const float4 v = *reinterpret_cast<const float4*>(p);
          acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += p[i];
        }
      }
      for (int q = q0; q < q1; ++q) {               // (row, chunk) order: slice by slice, as the pieces lie in the stream
        const float* p = cuts.P + (size_t)(2 * cuts.chunk[q]) * (size_t)k + x;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += p[i];
      }
      if (rowscale) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] *= rs;
      }
      if (accumulate) {                             // C already holds another part of the product
        const float* o = C + orow * (size_t)k + x;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += o[i];
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        if (bias) acc[i] += bias[x + i];
        if (relu) acc[i] = fmaxf(acc[i], 0.f);
      }
      if (drop.on()) {                                  // the dropout mask of the fused epilogue (philox.h)
        const unsigned long long idx = (unsigned long long)r * (unsigned long long)k + (unsigned long long)x;
        if (VEC == 4) {
          const float4 t = dropout_apply4(drop, idx, make_float4(acc[0], acc[1], acc[2], acc[3]));
          acc[0] = t.x; acc[1] = t.y; acc[2] = t.z; acc[3] = t.w;
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = dropout_apply(drop, idx + i, acc[i]);
        }
      }
      if (outscale) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] *= os;
      }
      float* o = C + orow * (size_t)k + x;
      if (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = acc[i];
      }
    }
  }
}

// The same reduction for row widths up to 256 floats (k % 4 == 0; the widths of the 64-column tiles and the padded odd ones):
// a wave covers up to 1 KiB of C at a time — floor(256 / k) consecutive rows, every lane a float4 — so (nearly) all 64 lanes load
// whatever k is (the kernel above leaves half of them idle at k = 128), and the S partial rows are fetched
// eight at a time with non-temporal loads (they are read exactly once) before they are added in slice order.
typedef float slice_f32x4 __attribute__((ext_vector_type(4)));

// OFF32: a plane of partial rows is smaller than 4 GiB, so a lane's place in it is a 32-bit byte offset beside a
// wave-uniform plane pointer (one address register per lane instead of a 64-bit pointer per load in flight): the
// kernel lives on loads in flight, and with the cut-row pieces it dropped from five to four waves per SIMD without this.
template <bool DROP, bool OFF32>
__global__ void __launch_bounds__(256)
slice_reduce_wide_kernel(const float* __restrict__ Cv, float* __restrict__ C,
                         const float* __restrict__ bias, int relu, int m, int S, int k, int accumulate,
                         const float* __restrict__ rowscale, DropoutSpec drop, const int* __restrict__ guard,
                         const float* __restrict__ outscale, int gap_w, CutLists cuts) {
  if (guard && *guard == 0) return;
  typedef typename std::conditional<OFF32, unsigned, size_t>::type off_t;
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * 4;
  const int rpw = 256 / k;                            // rows per wave and step (k = 48: five rows on sixty lanes)
  const int lr = lane * 4 / k, x = lane * 4 % k;
  if (lr >= rpw) return;                              // (widths that do not divide 256 floats leave the last lanes without a row)
  const size_t slab = (size_t)m * (size_t)k;          // floats between two slices' partial rows of one row
  const char* const Cvb = reinterpret_cast<const char*>(Cv);
  const char* const Pb = reinterpret_cast<const char*>(cuts.P);
  const off_t xb = (off_t)x * 4;
  // rows cut by chunk ends of the group kernels' stream: Cv holds the first piece, the following chunks' head pieces lie in
  // the slab P — about one per output row at 15 slices.  Two dependent index loads stand in front of a piece: they are
  // issued one row step AHEAD (nq0 / nq1 / nc0), so that a step only waits for its slices
  int nq0 = 0, nq1 = 0, nc0 = 0;
  if (cuts.ptr && wave * rpw + lr < m) {
    nq0 = cuts.ptr[wave * rpw + lr]; nq1 = cuts.ptr[wave * rpw + lr + 1];
    nc0 = nq0 < nq1 ? cuts.chunk[nq0] : 0;
  }
  for (long long r0 = wave * rpw; r0 < m; r0 += nw * rpw) {
    const long long r = r0 + lr;
    if (r >= m) continue;
    const off_t off = ((off_t)r * (off_t)k + (off_t)x) * 4;      // bytes inside a plane
    const int q0 = nq0, q1 = nq1;
    slice_f32x4 pv = {0.f, 0.f, 0.f, 0.f};             // the first piece: on its way while the slices are added
    if (q0 < q1) pv = __builtin_nontemporal_load(reinterpret_cast<const slice_f32x4*>(Pb + (size_t)(2 * nc0) * (size_t)k * 4 + xb));
    const long long rn = r + nw * rpw;
    if (cuts.ptr && rn < m) {
      nq0 = cuts.ptr[rn]; nq1 = cuts.ptr[rn + 1];
      nc0 = nq0 < nq1 ? cuts.chunk[nq0] : 0;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      slice_f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const slice_f32x4*>(Cvb + (size_t)(s + u) * slab * 4 + off));
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; s < S; ++s) {
      const slice_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const slice_f32x4*>(Cvb + (size_t)s * slab * 4 + off));
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (q0 < q1) { acc.x += pv.x; acc.y += pv.y; acc.z += pv.z; acc.w += pv.w; }
    for (int q = q0 + 1; q < q1; ++q) {               // (row, chunk) order: slice by slice, as the pieces lie in the stream
      const slice_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const slice_f32x4*>(Pb + (size_t)(2 * cuts.chunk[q]) * (size_t)k * 4 + xb));
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (rowscale) { const float rs = rowscale[r]; acc.x *= rs; acc.y *= rs; acc.z *= rs; acc.w *= rs; }
    // pre-laid output (gcn_spmm_csr_f32_prelaid): row r lands in the consumer's slice-by-slice feature layout — one
    // (all-zero, never written) row behind every gap_w rows — already multiplied by the consumer's column factor
    const size_t orow = gap_w > 0 ? (size_t)r + (size_t)(r / gap_w) : (size_t)r;
    float* o = C + orow * (size_t)k + x;
    if (accumulate) {                                 // C already holds another part of the product
      const float4 c = *reinterpret_cast<const float4*>(o);
      acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
    }
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + x);
      acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    if constexpr (DROP)                               // the dropout mask of the fused epilogue (philox.h)
      acc = dropout_apply4(drop, (unsigned long long)r * (unsigned long long)k + (unsigned long long)x, acc);
    if (outscale) { const float os = outscale[r]; acc.x *= os; acc.y *= os; acc.z *= os; acc.w *= os; }
    *reinterpret_cast<float4*>(o) = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
#define GCN_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

// Builds the slice-major CSR.  Outputs (device, caller-allocated): vrowptr [S*m+1], vcol/vval
// [nnz].  `*sorted_out` = 0 if the input rows are not column-sorted (nothing else is valid then).
hipError_t build_sliced_csr(const int* rowptr, const int* col, const float* val, int m, int n,
                            int nnz, int S, int* vrowptr, int* vcol, float* vval,
                            int* sorted_out, hipStream_t st) {
  *sorted_out = 1;
  if (m <= 0 || S <= 0) return hipSuccess;
  const int w = (n + S - 1) / S;          // slice width in columns
  int *split = nullptr, *cnt = nullptr, *flag = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  const long long nsplit = (long long)m * (S + 1), ncnt = (long long)m * S;
  hipError_t err = hipSuccess;
  auto cleanup = [&]() {
    if (split) (void)hipFree(split);
    if (cnt) (void)hipFree(cnt);
    if (flag) (void)hipFree(flag);
    if (tmp) (void)hipFree(tmp);
  };
#define GCN_GO(x) do { err = (x); if (err != hipSuccess) { cleanup(); return err; } } while (0)
  GCN_GO(hipMalloc((void**)&split, sizeof(int) * (size_t)nsplit));
  GCN_GO(hipMalloc((void**)&cnt, sizeof(int) * (size_t)(ncnt + 1)));
  GCN_GO(hipMalloc((void**)&flag, sizeof(int)));
  GCN_GO(hipMemsetAsync(flag, 0, sizeof(int), st));
  slice_check_sorted_kernel<<<2048, 256, 0, st>>>(rowptr, col, m, flag);
  GCN_GO(hipGetLastError());
  int unsorted = 0;
  GCN_GO(hipMemcpyAsync(&unsorted, flag, sizeof(int), hipMemcpyDeviceToHost, st));
  GCN_GO(hipStreamSynchronize(st));
  if (unsorted) { *sorted_out = 0; cleanup(); return hipSuccess; }
  slice_count_kernel<<<(unsigned)((nsplit + 255) / 256), 256, 0, st>>>(rowptr, col, m, S, w, split, flag);
  GCN_GO(hipGetLastError());
  slice_sizes_kernel<<<(unsigned)((ncnt + 255) / 256), 256, 0, st>>>(split, m, S, cnt);
  GCN_GO(hipGetLastError());
  GCN_GO(hipMemsetAsync(cnt + ncnt, 0, sizeof(int), st));
  GCN_GO(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt, vrowptr, (int)(ncnt + 1), st));
  GCN_GO(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  GCN_GO(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, cnt, vrowptr, (int)(ncnt + 1), st));
  int nb = (m + 3) / 4;
  if (nb > 16384) nb = 16384;
  slice_scatter_kernel<<<nb, 256, 0, st>>>(rowptr, col, val, split, vrowptr, m, S, w, vcol, vval);
  GCN_GO(hipGetLastError());
  GCN_GO(hipStreamSynchronize(st));
#undef GCN_GO
  (void)nnz;
  cleanup();
  return hipSuccess;
}

hipError_t launch_slice_reduce(const float* Cv, float* C, const float* bias, int relu, int m, int S,
                               int k, hipStream_t st, int accumulate, const float* rowscale, const DropoutSpec& drop,
                               const int* guard, const float* outscale, int gap_w, const CutLists& cuts) {
  if (m <= 0 || k <= 0) return hipSuccess;
  int nb = (m + 3) / 4;
  if (nb > 8192) nb = 8192;
  const uintptr_t al = (uintptr_t)Cv | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)cuts.P;
  if (k % 4 == 0 && k <= 256 && (al & 15) == 0) {
    const long long steps = ((long long)m + 256 / k - 1) / (256 / k);   // wave steps of 256 / k rows (1 KiB when k divides 256)
    const int cap = 16384;
    const int nbw = (int)(steps / 4 + 1 < cap ? steps / 4 + 1 : cap);
    const bool off32 = (size_t)m * (size_t)k * 4 < (1ull << 32);   // a lane's byte offset inside a plane of partial rows
    // (62 VGPRs: eight waves per SIMD, and the kernel wants them all — 344 / 357 / 374 / 399 us at 8 / 7 / 5 / 4, profiles/r03am_*)
#define GCN_RW(D, O) slice_reduce_wide_kernel<D, O><<<nbw, 256, 0, st>>>(Cv, C, bias, relu, m, S, k, accumulate, rowscale, drop, guard, outscale, gap_w, cuts)
    if (drop.on()) { if (off32) GCN_RW(true, true); else GCN_RW(true, false); }
    else           { if (off32) GCN_RW(false, true); else GCN_RW(false, false); }
#undef GCN_RW
  } else if (k % 4 == 0 && (al & 15) == 0) slice_reduce_kernel<4><<<nb, 256, 0, st>>>(Cv, C, bias, relu, m, S, k, accumulate, rowscale, drop, guard, outscale, gap_w, cuts);
  else                              slice_reduce_kernel<1><<<nb, 256, 0, st>>>(Cv, C, bias, relu, m, S, k, accumulate, rowscale, drop, guard, outscale, gap_w, cuts);
  return hipGetLastError();
}

// ---- 16-bit column stream (spmm_quad.hip, COL16) ----
__global__ void col16_rowptr_kernel(const int* __restrict__ vrowptr, int m, int S, const int* __restrict__ pad_before,
                                    int* __restrict__ vrowptr16) {
  const long long vr = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vr > (long long)S * m) return;
  const int s = (int)(vr / m);                       // vr == S*m -> s == S: pad_before[S] = all the padding
  vrowptr16[vr] = vrowptr[vr] + pad_before[s];
}

// one wave per virtual row: offsets inside the slice, 16 bits each
__global__ void __launch_bounds__(256)
col16_scatter_kernel(const int* __restrict__ vrowptr, const int* __restrict__ vcol, const int* __restrict__ vrowptr16,
                     int m, int S, int w, unsigned short* __restrict__ col16) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * 4;
  for (long long vr = wave; vr < (long long)S * m; vr += nw) {
    const int base = (int)(vr / m) * w;
    const int src = vrowptr[vr], dst = vrowptr16[vr], len = vrowptr[vr + 1] - src;
    for (int i = lane; i < len; i += 64) col16[dst + i] = (unsigned short)(vcol[src + i] - base);
  }
}

hipError_t build_col16_stream(const int* vrowptr, const int* vcol, int m, int n, int S, int T, int* vrowptr16,
                              unsigned short** col16_out, int* nnz16_host, int* start_host, hipStream_t st) {
  *col16_out = nullptr;
  *nnz16_host = 0;
  const int w = (n + S - 1) / S;
  if (S < 1 || S > 8 || w > 65535 || m <= 0) return hipErrorInvalidValue;
  // slice boundaries of the unpadded stream
  int bounds[9];
  hipError_t e = hipSuccess;
  for (int s = 0; s <= S && e == hipSuccess; ++s)
    e = hipMemcpyAsync(&bounds[s], vrowptr + (size_t)s * m, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  int pad_before[9];
  long long total = 0;
  for (int s = 0; s < S; ++s) {
    pad_before[s] = (int)(total - bounds[s]);
    start_host[s] = (int)total;
    const long long len = bounds[s + 1] - bounds[s];
    total += (len + T - 1) / T * T;
    if (total >= (1LL << 31)) return hipErrorInvalidValue;
  }
  pad_before[S] = (int)(total - bounds[S]);
  start_host[S] = (int)total;
  for (int s = S + 1; s < 9; ++s) start_host[s] = (int)total;
  int* d_pad = nullptr;
  unsigned short* c16 = nullptr;
  if ((e = hipMalloc((void**)&d_pad, sizeof(int) * 9)) != hipSuccess) return e;
  if ((e = hipMalloc((void**)&c16, sizeof(unsigned short) * (size_t)(total > 0 ? total : 1))) != hipSuccess) {
    (void)hipFree(d_pad);
    return e;
  }
  e = hipMemcpyAsync(d_pad, pad_before, sizeof(int) * (S + 1), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(c16, 0xFF, sizeof(unsigned short) * (size_t)total, st);   // markers everywhere
  if (e == hipSuccess) {
    const long long vm1 = (long long)S * m + 1;
    col16_rowptr_kernel<<<(unsigned)((vm1 + 255) / 256), 256, 0, st>>>(vrowptr, m, S, d_pad, vrowptr16);
    int nb = (int)(((long long)S * m + 3) / 4);
    if (nb > 16384) nb = 16384;
    col16_scatter_kernel<<<nb, 256, 0, st>>>(vrowptr, vcol, vrowptr16, m, S, w, c16);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_pad);
  if (e != hipSuccess) { (void)hipFree(c16); return e; }
  *col16_out = c16;
  *nnz16_host = (int)total;
  return hipSuccess;
}

// ---- 15-bit slice-major stream of the group kernel (spmm_group.hip) ----
// len[vr] = max(1, entries of virtual row vr): empty virtual rows get one padding entry
__global__ void group_len_kernel(const int* __restrict__ vrowptr, long long vm, int* __restrict__ len) {
  const long long vr = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vr > vm) return;
  const int l = vr < vm ? vrowptr[vr + 1] - vrowptr[vr] : 0;
  len[vr] = vr < vm ? (l > 0 ? l : 1) : 0;
}

// pos[vr] (exclusive scan of len) -> position in the padded stream; pos[vm] -> total
__global__ void group_rowptr_kernel(const int* __restrict__ pos, int m, int S, const int* __restrict__ pad_before,
                                    int* __restrict__ vrowptr_g) {
  const long long vr = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vr > (long long)S * m) return;
  vrowptr_g[vr] = pos[vr] + pad_before[(int)(vr / m)];       // vr == S*m -> pad_before[S] = all the padding
}

// Where logical stream position p lives in memory: every run of 64 entries (four 16-entry blocks of one group) is
// stored transposed, lane-major — [lane f][block j] — so that lane f of the group fetches ITS entries of four
// consecutive blocks with one 8-byte load (spmm_group.hip: one stream load per 64 gathers instead of four).
__device__ __forceinline__ long long group_phys(long long p) {
  const int r = (int)(p & 63);
  return (p & ~63LL) + (r & 15) * 4 + (r >> 4);
}

// one wave per virtual row: entries = column offset inside the slice; the LAST stream position the row owns
// (which for the last row of a slice is the end of the slice's padding) carries the row-end bit
__global__ void __launch_bounds__(256)
group_scatter_kernel(const int* __restrict__ vrowptr, const int* __restrict__ vcol, const int* __restrict__ vrowptr_g,
                     int m, int S, int w, unsigned short* __restrict__ stream,
                     const float* __restrict__ vval, float* __restrict__ vals) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * 4;
  for (long long vr = wave; vr < (long long)S * m; vr += nw) {
    const int base = (int)(vr / m) * w;
    const int src = vrowptr[vr], len = vrowptr[vr + 1] - src;
    const int dst = vrowptr_g[vr], last = vrowptr_g[vr + 1] - 1;
    for (int i = lane; i < len; i += 64) {
      stream[group_phys(dst + i)] = (unsigned short)((vcol[src + i] - base) | (dst + i == last ? 0x8000 : 0));
      if (vals) vals[group_phys(dst + i)] = vval[src + i];
    }
    if (lane == 0 && last >= dst + len) stream[group_phys(last)] = (unsigned short)(w | 0x8000);   // padding entry ends the row
  }
}

// meta[c] = {2 * chunk_row[c] + (the row began before the chunk), first row of the chunk's slice in B'}
__global__ void group_meta_kernel(const int* __restrict__ chunk_row, const int* __restrict__ vrowptr_g, int nchunks,
                                  int T, int m, int w, int2* __restrict__ meta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const int vr = chunk_row[c];
  meta[c] = make_int2(2 * vr + (vrowptr_g[vr] < (long long)c * T ? 1 : 0), (vr / m) * (w + 1));
}

// fix[i] = {virtual row, c, c1, 0} for every row that begins in chunk c-1 and runs on into chunks c .. c1: the rows
// whose pieces lie in the partial slab (group_fixup_kernel adds them up).  Order of the list: as the atomics fall —
// every entry owns its row, so the results do not depend on it.
__global__ void group_fix_list_kernel(const int2* __restrict__ meta, const int* __restrict__ vrowptr_g, int nchunks, int T,
                                      int4* __restrict__ fix, int* __restrict__ nfix) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x + 1;
  if (c >= nchunks) return;
  const int2 mt = meta[c];
  if (!(mt.x & 1)) return;
  const int vr = mt.x >> 1;
  if (vrowptr_g[vr] / T != c - 1) return;                        // began earlier still: the entry of that chunk covers it
  fix[atomicAdd(nfix, 1)] = make_int4(vr, c, (vrowptr_g[vr + 1] - 1) / T, 0);
}

// Cut lists for the slice reduction: every chunk c that continues a row begun earlier holds one head piece P[2c] of
// output row r = (virtual row) % m.  keys[c] = (r << 32) | c, all ones for the other chunks (they sort to the end);
// cnt[r] = pieces of row r.  Sorted keys = the pieces in (row, chunk) order — chunks ascend with the slice, so a row's
// pieces follow each other as its slices do.
__global__ void group_cut_keys_kernel(const int2* __restrict__ meta, int nchunks, int m, unsigned long long* __restrict__ keys,
                                      int* __restrict__ cnt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const int2 mt = meta[c];
  if (c == 0 || !(mt.x & 1)) { keys[c] = ~0ull; return; }
  const int r = (mt.x >> 1) % m;
  keys[c] = ((unsigned long long)(unsigned)r << 32) | (unsigned)c;
  atomicAdd(cnt + r, 1);
}

__global__ void group_cut_chunks_kernel(const unsigned long long* __restrict__ keys, int ncut, int* __restrict__ chunk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncut) chunk[i] = (int)(unsigned)keys[i];
}

// Cv[row, :] += head pieces of chunks c .. c1, in chunk order (Cv[row] holds the row's first piece: the group kernels
// write the piece that sticks out of chunk c-1 there).  One thread per float4 of FOUR list entries (the loads of the
// four are in flight together: the pass is latency-bound otherwise).  The plans of api_spmm.cpp do not run this pass:
// their slice reduction adds the same pieces in the same order (CutLists); the drop-in flexspmm does.
__global__ void __launch_bounds__(256)
group_fixup_kernel(const int4* __restrict__ fix, int nfix, const float* __restrict__ P, float* __restrict__ Cv, int k,
                   const int* __restrict__ dyn) {
  if (dyn) { if (dyn[0] == 0) return; nfix = dyn[2]; }          // (drop-in flexspmm: the launch was sized from an upper bound)
  const int k4 = k >> 2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long e0 = t / k4 * 4;
  const int x = (int)(t % k4) * 4;
  if (e0 >= nfix) return;
  int4 f[4];
  float4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = fix[e0 + i < nfix ? e0 + i : e0];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const float4*>(Cv + (size_t)f[i].x * (size_t)k + x);
    b[i] = *reinterpret_cast<const float4*>(P + (size_t)(2 * f[i].y) * (size_t)k + x);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (e0 + i >= nfix) break;
    float4 s = make_float4(a[i].x + b[i].x, a[i].y + b[i].y, a[i].z + b[i].z, a[i].w + b[i].w);
    for (int cc = f[i].y + 1; cc <= f[i].z; ++cc) {              // rows longer than a chunk: whole chunks in between
      const float4 v = *reinterpret_cast<const float4*>(P + (size_t)(2 * cc) * (size_t)k + x);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(Cv + (size_t)f[i].x * (size_t)k + x) = s;
  }
}

hipError_t launch_group_fixup(const int* fix, int nfix, const float* P, float* Cv, int k, hipStream_t s, const int* dyn) {
  if (nfix <= 0 || k <= 0) return hipSuccess;
  if (k % 4 != 0) return hipErrorInvalidValue;
  const long long threads = ((long long)nfix + 3) / 4 * (k / 4);
  group_fixup_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(reinterpret_cast<const int4*>(fix), nfix, P, Cv, k, dyn);
  return hipGetLastError();
}

hipError_t build_group_stream(const int* vrowptr, const int* vcol, int m, int n, int S, int T, int* vrowptr_g,
                              unsigned short** stream_out, int** chunk_row_out, int** chunk_meta_out,
                              int* nchunks_host, int** fix_out, int* nfix_host, hipStream_t st,
                              const float* vval, float** vals_out, int** cutptr_out, int** cutchunk_out, int* ncut_host) {
  if (vals_out) *vals_out = nullptr;
  if (cutptr_out) *cutptr_out = nullptr;
  if (cutchunk_out) *cutchunk_out = nullptr;
  if (ncut_host) *ncut_host = 0;
  *stream_out = nullptr; *chunk_row_out = nullptr; *chunk_meta_out = nullptr; *nchunks_host = 0;
  *fix_out = nullptr; *nfix_host = 0;
  const int w = (n + S - 1) / S;
  if (S < 1 || S > 256 || w > 32767 || m <= 0 || T < 16 || T % 16) return hipErrorInvalidValue;
  const long long vm = (long long)S * m;
  int *len = nullptr, *pos = nullptr, *d_pad = nullptr, *chunk_row = nullptr, *nfix_dev = nullptr;
  int2* meta = nullptr;
  int4* fix = nullptr;
  float* vals = nullptr;
  unsigned short* stream = nullptr;
  int *cutptr = nullptr, *cutchunk = nullptr, *cutcnt = nullptr;
  unsigned long long *ckeys = nullptr, *ckeys2 = nullptr;
  void* ctmp = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  hipError_t err = hipSuccess;
  auto cleanup = [&](bool all) {
    if (cutcnt) (void)hipFree(cutcnt);
    if (ckeys) (void)hipFree(ckeys);
    if (ckeys2) (void)hipFree(ckeys2);
    if (ctmp) (void)hipFree(ctmp);
    if (all) {
      if (cutptr) (void)hipFree(cutptr);
      if (cutchunk) (void)hipFree(cutchunk);
    }
    if (len) (void)hipFree(len);
    if (pos) (void)hipFree(pos);
    if (d_pad) (void)hipFree(d_pad);
    if (tmp) (void)hipFree(tmp);
    if (nfix_dev) (void)hipFree(nfix_dev);
    if (all) {
      if (stream) (void)hipFree(stream);
      if (chunk_row) (void)hipFree(chunk_row);
      if (meta) (void)hipFree(meta);
      if (fix) (void)hipFree(fix);
      if (vals) (void)hipFree(vals);
    }
  };
#define GCN_GO(x) do { err = (x); if (err != hipSuccess) { cleanup(true); return err; } } while (0)
  GCN_GO(hipMalloc((void**)&len, sizeof(int) * (size_t)(vm + 1)));
  GCN_GO(hipMalloc((void**)&pos, sizeof(int) * (size_t)(vm + 1)));
  GCN_GO(hipMalloc((void**)&d_pad, sizeof(int) * (size_t)(S + 1)));
  group_len_kernel<<<(unsigned)((vm + 256) / 256), 256, 0, st>>>(vrowptr, vm, len);
  GCN_GO(hipGetLastError());
  GCN_GO(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, len, pos, (int)(vm + 1), st));
  GCN_GO(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  GCN_GO(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, len, pos, (int)(vm + 1), st));
  // slice boundaries of the unpadded stream -> where each slice starts once padded to whole chunks
  std::vector<int> bounds(S + 1), pad_before(S + 1);
  for (int s = 0; s <= S; ++s)
    GCN_GO(hipMemcpyAsync(&bounds[s], pos + (size_t)s * m, sizeof(int), hipMemcpyDeviceToHost, st));
  GCN_GO(hipStreamSynchronize(st));
  long long total = 0;
  for (int s = 0; s < S; ++s) {
    pad_before[s] = (int)(total - bounds[s]);
    const long long l = bounds[s + 1] - bounds[s];
    total += (l + T - 1) / T * T;
  }
  total = (total + 64LL * T - 1) / (64LL * T) * (64LL * T);     // whole waves for every XCD (four chunks per wave in
                                                                // spmm_group_kernel, eight in spmm_group8_kernel)
  if (total >= (1LL << 31)) { cleanup(true); return hipErrorInvalidValue; }
  pad_before[S] = (int)(total - bounds[S]);
  const int nchunks = (int)(total / T);
  GCN_GO(hipMalloc((void**)&stream, sizeof(unsigned short) * (size_t)total));
  GCN_GO(hipMalloc((void**)&chunk_row, sizeof(int) * (size_t)nchunks));
  GCN_GO(hipMemcpyAsync(d_pad, pad_before.data(), sizeof(int) * (size_t)(S + 1), hipMemcpyHostToDevice, st));
  GCN_GO(hipMemsetD16Async((hipDeviceptr_t)stream, (unsigned short)w, (size_t)total, st));   // zero-row entries everywhere
  group_rowptr_kernel<<<(unsigned)((vm + 256) / 256), 256, 0, st>>>(pos, m, S, d_pad, vrowptr_g);
  {
    int nb = (int)((vm + 3) / 4);
    if (nb > 16384) nb = 16384;
    if (vval && vals_out) {
      GCN_GO(hipMalloc((void**)&vals, sizeof(float) * (size_t)total));
      GCN_GO(hipMemsetAsync(vals, 0, sizeof(float) * (size_t)total, st));     // padding entries weigh nothing
    }
    group_scatter_kernel<<<nb, 256, 0, st>>>(vrowptr, vcol, vrowptr_g, m, S, w, stream, vval, vals);
  }
  GCN_GO(hipGetLastError());
  GCN_GO(launch_plan_chunk_rows(vrowptr_g, (int)vm, T, nchunks, chunk_row, st));
  GCN_GO(hipMalloc((void**)&meta, sizeof(int2) * (size_t)nchunks));
  group_meta_kernel<<<(nchunks + 255) / 256, 256, 0, st>>>(chunk_row, vrowptr_g, nchunks, T, m, w, meta);
  GCN_GO(hipGetLastError());
  GCN_GO(hipMalloc((void**)&fix, sizeof(int4) * (size_t)nchunks));
  GCN_GO(hipMalloc((void**)&nfix_dev, sizeof(int)));
  GCN_GO(hipMemsetAsync(nfix_dev, 0, sizeof(int), st));
  group_fix_list_kernel<<<(nchunks + 255) / 256, 256, 0, st>>>(meta, vrowptr_g, nchunks, T, fix, nfix_dev);
  GCN_GO(hipGetLastError());
  int nfix = 0;
  GCN_GO(hipMemcpyAsync(&nfix, nfix_dev, sizeof(int), hipMemcpyDeviceToHost, st));
  int ncut = 0;
  if (cutptr_out && cutchunk_out && ncut_host) {                 // the same pieces per OUTPUT row, for the slice reduction
    GCN_GO(hipMalloc((void**)&cutcnt, sizeof(int) * (size_t)(m + 1)));
    GCN_GO(hipMalloc((void**)&cutptr, sizeof(int) * (size_t)(m + 1)));
    GCN_GO(hipMalloc((void**)&ckeys, sizeof(unsigned long long) * (size_t)nchunks));
    GCN_GO(hipMalloc((void**)&ckeys2, sizeof(unsigned long long) * (size_t)nchunks));
    GCN_GO(hipMemsetAsync(cutcnt, 0, sizeof(int) * (size_t)(m + 1), st));
    group_cut_keys_kernel<<<(nchunks + 255) / 256, 256, 0, st>>>(meta, nchunks, m, ckeys, cutcnt);
    GCN_GO(hipGetLastError());
    size_t b1 = 0, b2 = 0;
    GCN_GO(hipcub::DeviceScan::ExclusiveSum(nullptr, b1, cutcnt, cutptr, m + 1, st));
    GCN_GO(hipcub::DeviceRadixSort::SortKeys(nullptr, b2, ckeys, ckeys2, nchunks, 0, 64, st));
    GCN_GO(hipMalloc(&ctmp, (b1 > b2 ? b1 : b2) + 16));
    GCN_GO(hipcub::DeviceScan::ExclusiveSum(ctmp, b1, cutcnt, cutptr, m + 1, st));
    GCN_GO(hipcub::DeviceRadixSort::SortKeys(ctmp, b2, ckeys, ckeys2, nchunks, 0, 64, st));
    GCN_GO(hipMemcpyAsync(&ncut, cutptr + m, sizeof(int), hipMemcpyDeviceToHost, st));
  }
  GCN_GO(hipStreamSynchronize(st));                              // (pad_before, nfix and ncut are host buffers)
  if (cutptr) {
    GCN_GO(hipMalloc((void**)&cutchunk, sizeof(int) * (size_t)(ncut > 0 ? ncut : 1)));
    if (ncut > 0) {
      group_cut_chunks_kernel<<<(ncut + 255) / 256, 256, 0, st>>>(ckeys2, ncut, cutchunk);
      GCN_GO(hipGetLastError());
      GCN_GO(hipStreamSynchronize(st));
    }
    *cutptr_out = cutptr; *cutchunk_out = cutchunk; *ncut_host = ncut;
  }
#undef GCN_GO
  cleanup(false);
  *stream_out = stream; *chunk_row_out = chunk_row; *chunk_meta_out = reinterpret_cast<int*>(meta); *nchunks_host = nchunks;
  *fix_out = reinterpret_cast<int*>(fix); *nfix_host = nfix;
  if (vals_out) *vals_out = vals;
  return hipSuccess;
}

// scaled copy of B in the group kernel's layout: slice s at rows [s*(w+1), (s+1)*(w+1)), the last one zero
__global__ void __launch_bounds__(256)
scale_rows_sliced_kernel(float* __restrict__ dst, const float* __restrict__ src, const float* __restrict__ rowscale,
                         int n, int k, int ld, int S, int w, int src_vec) {
  const int ld4 = ld >> 2;                                       // float4 per destination row
  const long long total = (long long)S * (w + 1) * ld4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long R = i / ld4;
    const int x = (int)(i - R * ld4) * 4;
    const int s = (int)(R / (w + 1)), j = (int)(R - (long long)s * (w + 1));
    const long long c = (long long)s * w + j;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < w && c < n && x < k) {
      const float u = rowscale ? rowscale[c] : 1.f;
      const float* p = src + c * k + x;
      if (src_vec) { const float4 t = *reinterpret_cast<const float4*>(p); v = make_float4(u * t.x, u * t.y, u * t.z, u * t.w); }
      else { v.x = u * p[0]; if (x + 1 < k) v.y = u * p[1]; if (x + 2 < k) v.z = u * p[2]; if (x + 3 < k) v.w = u * p[3]; }
    }
    *reinterpret_cast<float4*>(dst + R * ld + x) = v;
  }
}

hipError_t launch_scale_rows_sliced(float* dst, const float* src, const float* rowscale, int n, int k, int ld,
                                    int S, int w, hipStream_t s) {
  if (n <= 0 || k <= 0) return hipSuccess;
  if (ld % 4 != 0 || ((uintptr_t)dst & 15) != 0) return hipErrorInvalidValue;
  // 16-byte loads of the source rows when they are 16-byte aligned (k % 4 == 0 and an aligned B); scalar loads otherwise
  const int src_vec = ((k & 3) == 0 && ((uintptr_t)src & 15) == 0) ? 1 : 0;
  long long nb = ((long long)S * (w + 1) * (ld / 4) + 255) / 256;
  if (nb > 65536) nb = 65536;
  scale_rows_sliced_kernel<<<(int)nb, 256, 0, s>>>(dst, src, rowscale, n, k, ld, S, w, src_vec);
  return hipGetLastError();
}

// ---- do the values factor as u[r] * u[c]?  (the GCN normalisation D^-1/2 (A+I) D^-1/2: u = D^-1/2) ----
// u[r] = sqrt(A[r, r]) from the stored diagonal (binary search in the column-sorted row)
__global__ void rank1_diag_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                  const float* __restrict__ val, int n, float* __restrict__ u, int* __restrict__ fail) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int lo = rowptr[r], hi = rowptr[r + 1];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[mid] < r) lo = mid + 1; else hi = mid;
  }
  if (lo < rowptr[r + 1] && col[lo] == r && val[lo] > 0.f) u[r] = (float)sqrt((double)val[lo]);
  else { u[r] = 0.f; *fail = 1; }
}

// every stored entry within 4 ulp of u[r] * u[c]; one wave per row
__global__ void __launch_bounds__(256)
rank1_check_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
                   const float* __restrict__ u, const float* __restrict__ ucol, int n, int* __restrict__ fail) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int r = wave; r < n; r += nw) {
    const float ur = u[r];
    for (int e = rowptr[r] + lane; e < rowptr[r + 1]; e += 64) {
      const float want = ur * ucol[col[e]];
      if (!(fabsf(val[e] - want) <= 4.8e-7f * fabsf(val[e]))) *fail = 1;
    }
  }
}

hipError_t detect_rank1_values(const int* rowptr, const int* col, const float* val, int n, float* u_out,
                               int* ok_host, hipStream_t st) {
  *ok_host = 0;
  if (n <= 0) return hipSuccess;
  int* fail = nullptr;
  hipError_t e = hipMalloc((void**)&fail, sizeof(int));
  if (e != hipSuccess) return e;
  int h = 1;
  e = hipMemsetAsync(fail, 0, sizeof(int), st);
  if (e == hipSuccess) {
    rank1_diag_kernel<<<(n + 255) / 256, 256, 0, st>>>(rowptr, col, val, n, u_out, fail);
    int nb = (n + 3) / 4;
    if (nb > 16384) nb = 16384;
    rank1_check_kernel<<<nb, 256, 0, st>>>(rowptr, col, val, u_out, u_out, n, fail);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&h, fail, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(fail);
  if (e == hipSuccess) *ok_host = h ? 0 : 1;
  return e;
}

// ---- values that depend on the row only, or on the column only (an unweighted adjacency: all ones; D^-1 (A+I), the
// row-normalised adjacency of Kipf's pygcn; its transpose, which the backward pass multiplies with) ----
__global__ void const_fill_kernel(float* __restrict__ p, int n, float v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void row_first_value_kernel(const int* __restrict__ rowptr, const float* __restrict__ val, int m, float* __restrict__ u_row) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < m) u_row[r] = rowptr[r] < rowptr[r + 1] ? val[rowptr[r]] : 0.f;
}
// u_col[c] = the value of SOME entry of column c (all of them are equal when the values are column-constant: the
// racing stores then write the same bits; the check below decides)
__global__ void col_any_value_kernel(const int* __restrict__ col, const float* __restrict__ val, int nnz, float* __restrict__ u_col) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) u_col[col[e]] = val[e];
}

// mode 1: val[r, c] == u_row[r] (u_col = 1);  mode 2: val[r, c] == u_col[c] (u_row = 1).  u_row_out [m], u_col_out [n]
// (device, caller-allocated); *ok_host = 1 when every stored entry matches within 4 ulp.
hipError_t detect_constant_values(const int* rowptr, const int* col, const float* val, int m, int n, int nnz, int mode,
                                  float* u_row_out, float* u_col_out, int* ok_host, hipStream_t st) {
  *ok_host = 0;
  if (m <= 0 || n <= 0 || nnz <= 0) return hipSuccess;
  if (mode == 1) {
    row_first_value_kernel<<<(m + 255) / 256, 256, 0, st>>>(rowptr, val, m, u_row_out);
    const_fill_kernel<<<(n + 255) / 256, 256, 0, st>>>(u_col_out, n, 1.f);
  } else {
    const_fill_kernel<<<(m + 255) / 256, 256, 0, st>>>(u_row_out, m, 1.f);
    const_fill_kernel<<<(n + 255) / 256, 256, 0, st>>>(u_col_out, n, 1.f);   // (columns without entries: any value)
    col_any_value_kernel<<<4096, 256, 0, st>>>(col, val, nnz, u_col_out);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return verify_value_factors(rowptr, col, val, u_row_out, u_col_out, m, ok_host, st);
}

// caller-supplied factors: val[r, c] == u_row[r] * u_col[c] (4 ulp) for every stored entry of the m-row matrix?
hipError_t verify_value_factors(const int* rowptr, const int* col, const float* val, const float* u_row,
                                const float* u_col, int m, int* ok_host, hipStream_t st) {
  *ok_host = 0;
  if (m <= 0) return hipSuccess;
  int* fail = nullptr;
  hipError_t e = hipMalloc((void**)&fail, sizeof(int));
  if (e != hipSuccess) return e;
  int h = 1;
  e = hipMemsetAsync(fail, 0, sizeof(int), st);
  if (e == hipSuccess) {
    int nb = (m + 3) / 4;
    if (nb > 16384) nb = 16384;
    rank1_check_kernel<<<nb, 256, 0, st>>>(rowptr, col, val, u_row, u_col, m, fail);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&h, fail, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(fail);
  if (e == hipSuccess) *ok_host = h ? 0 : 1;
  return e;
}

}  // namespace gcn
