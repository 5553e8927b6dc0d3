// spmm_panel.hip — LDS-staged feature tiles per row panel, for matrices whose non-zeros cluster in
// column ranges near their rows (graphs with community structure after Rabbit / RCM / Gorder
// renumbering).
//
// Plan time (device):
//   * every PANEL of R = 128 consecutive rows gets the column WINDOW — W = 512 consecutive columns,
//     start a multiple of 128 — that covers most of its non-zeros (panel_windows_kernel: one LDS
//     histogram per panel);
//   * the matrix is split  A = A_in + A_out :  A_in holds the entries inside their panel's window
//     (stored as byte offsets into the staged tile), A_out the rest (an ordinary CSR with its own
//     chunk plan).  Splitting keeps the two access patterns apart: a first version that mixed LDS
//     reads and L2/HBM gathers in one loop was latency-bound at 4 waves per SIMD and lost to the
//     chunk kernel.
// Run time, per 64-column tile:
//   1. spmm_panel_in_kernel: a 16-wave workgroup stages the window's feature rows (W x 64 floats =
//      128 KiB of the CU's 160 KiB LDS) with coalesced loads, then every wave sums whole rows of
//      A_in from LDS only — v_readlane + conflict-free ds_read_b32 + FMA, no global gather at all —
//      and writes C (raw sums, zeros for rows without staged entries);
//   2. spmm_chunk_kernel on A_out in ACCUMULATE mode adds the out-of-window part (and applies the
//      bias/ReLU epilogue).
// Rows of A_in are owned by one wave and summed in CSR order; hub rows (> LONG_ROW staged entries)
// are summed by all 16 waves (strided blocks, partials combined through LDS in wave order).
// Deterministic, no atomics.  Used only when asked for / when the measured window coverage says it
// pays; never a correctness question.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdlib.h>
#include "spmm_kernels.h"

namespace gcn {

constexpr int PANEL_WAVES = 16;           // 1024 threads: one workgroup per CU
constexpr int PANEL_W = 512;              // window: 512 feature rows x 64 columns x 4 B = 128 KiB
constexpr int PANEL_BIN = 128;            // window starts are multiples of this
constexpr int PANEL_MAX_BINS = 8192;      // histogram bins a plan-time workgroup can hold (n <= 1 M)
constexpr int PANEL_LONG_ROW = 2048;      // rows above this are summed by the whole workgroup

__device__ __forceinline__ int psgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __host__ __forceinline__ int panel_diag_window(int r0, int R, int m, int n) {
  long long centre = (long long)(r0 + R / 2) * n / (m > 0 ? m : 1);
  long long w0 = (centre - PANEL_W / 2) / PANEL_BIN * PANEL_BIN;
  const long long hi = (long long)n - PANEL_W;
  if (w0 > hi) w0 = hi;
  if (w0 < 0) w0 = 0;
  return (int)w0;
}

// plan time, one workgroup per panel: histogram of the panel's column indices in 128-column bins,
// best run of 4 bins -> w0[panel]; adds the covered non-zeros to *inside.
__global__ void __launch_bounds__(256)
panel_windows_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int m, int n, int R,
                     int* __restrict__ w0_out, unsigned long long* __restrict__ inside, int* __restrict__ cnt_out) {
  __shared__ unsigned int hist[PANEL_MAX_BINS];
  __shared__ int best_w0;
  __shared__ unsigned int best_cnt;
  const int p = blockIdx.x;
  const int r0 = p * R, r1 = min(m, r0 + R);
  const int nbins = (n + PANEL_BIN - 1) / PANEL_BIN;
  const int e0 = rowptr[r0], e1 = rowptr[r1];
  if (threadIdx.x == 0) best_cnt = 0;
  if (nbins > PANEL_MAX_BINS) {                       // too wide to histogram here: diagonal window
    if (threadIdx.x == 0) best_w0 = panel_diag_window(r0, R, m, n);
  } else {
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x) atomicAdd(&hist[col[e] / PANEL_BIN], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
      const int span = PANEL_W / PANEL_BIN;           // 4 bins
      unsigned int run = 0, best = 0;
      int best_bin = 0;
      for (int b = 0; b < nbins; ++b) {
        run += hist[b];
        if (b >= span) run -= hist[b - span];
        if (run > best) { best = run; best_bin = max(0, b - span + 1); }
      }
      long long w0 = (long long)best_bin * PANEL_BIN;
      if (w0 > (long long)n - PANEL_W) w0 = max(0LL, (long long)n - PANEL_W);
      best_w0 = (int)w0;
    }
  }
  __syncthreads();
  // exact count for the chosen window (clamping at the matrix edge can move it off bin alignment)
  const int w0 = best_w0;
  unsigned int c = 0;
  for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x) c += (unsigned)(col[e] - w0) < (unsigned)PANEL_W;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&best_cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) {
    w0_out[p] = w0;
    if (cnt_out) cnt_out[p] = (int)best_cnt;
    if (best_cnt) atomicAdd(inside, (unsigned long long)best_cnt);
  }
}

// plan time: per-row count of staged (in-window) entries; one wave per row
__global__ void __launch_bounds__(256)
panel_split_count_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                         const int* __restrict__ panel_w0, const int* __restrict__ dense_slot, int m, int R,
                         int* __restrict__ cnt_in, int* __restrict__ cnt_out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int r = wave; r < m; r += nw) {
    const int w0 = panel_w0[r / R];
    const int beg = rowptr[r], end = rowptr[r + 1];
    int c = 0;
    for (int e = beg + lane; e < end; e += 64) c += (unsigned)(col[e] - w0) < (unsigned)PANEL_W;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    // (a dense panel's in-window entries live in its dense tile: neither staged part nor rest)
    if (lane == 0) { cnt_in[r] = (dense_slot && dense_slot[r / R] >= 0) ? 0 : c; cnt_out[r] = end - beg - c; }
  }
}

// plan time: order-preserving split of every row into its staged part (byte offset into the LDS
// tile) and the rest (column index); one wave per row
__global__ void __launch_bounds__(256)
panel_split_scatter_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                           const float* __restrict__ val, const int* __restrict__ panel_w0,
                           const int* __restrict__ dense_slot,
                           const int* __restrict__ in_rowptr, const int* __restrict__ out_rowptr,
                           int m, int R, int* __restrict__ in_off, float* __restrict__ in_val,
                           int* __restrict__ out_col, float* __restrict__ out_val, float* __restrict__ adense) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int r = wave; r < m; r += nw) {
    const int w0 = panel_w0[r / R];
    const int slot = dense_slot ? dense_slot[r / R] : -1;
    const int beg = rowptr[r], end = rowptr[r + 1];
    int pin = in_rowptr[r], pout = out_rowptr[r];
    for (int base = beg; base < end; base += 64) {
      const bool valid = base + lane < end;
      const int c = valid ? col[base + lane] : 0;
      const float v = valid ? val[base + lane] : 0.f;
      const bool inwin = valid && (unsigned)(c - w0) < (unsigned)PANEL_W;
      const bool in = inwin && slot < 0;              // staged entry of an ordinary panel
      if (inwin && slot >= 0) {                       // dense panel: straight into its MFMA fragment image
        const int i = r - (r / R) * R, kk = c - w0;   // A[i][kk] of the panel -> row block i/32, k-step kk/2, lane (kk&1)*32 + i%32
        // (added, not assigned: a row that stores a column twice sums its entries in every other kernel — the tile is
        //  zero-filled before this pass, plan time only)
        atomicAdd(&adense[(((size_t)slot * 4 + (i >> 5)) * (PANEL_W / 2) + (kk >> 1)) * 64 + (kk & 1) * 32 + (i & 31)], v);
      }
      const unsigned long long mi = __ballot(in), mo = __ballot(valid && !inwin);
      const int ri = __builtin_amdgcn_mbcnt_hi((unsigned)(mi >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mi, 0));
      const int ro = __builtin_amdgcn_mbcnt_hi((unsigned)(mo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mo, 0));
      if (in) { in_off[pin + ri] = (c - w0) * 256; in_val[pin + ri] = v; }
      else if (valid && !inwin) { out_col[pout + ro] = c; out_val[pout + ro] = v; }
      pin += __builtin_popcountll(mi);
      pout += __builtin_popcountll(mo);
    }
  }
}

// run time: C[r, tile] = sum over the staged entries of row r, from LDS only
__global__ void __launch_bounds__(PANEL_WAVES * 64)
spmm_panel_in_kernel(const int* __restrict__ dense_slot, const int* __restrict__ in_rowptr, const int* __restrict__ in_off,
                     const float* __restrict__ in_val, const float* __restrict__ B,
                     float* __restrict__ C, const int* __restrict__ panel_w0,
                     int m, int n, int k, int R, int col_tile) {
  extern __shared__ float lds[];                    // [PANEL_W][64] tile + [PANEL_WAVES][64] scratch
  if (dense_slot && dense_slot[blockIdx.x] >= 0) return;       // a dense panel: spmm_panel_dense_mfma_kernel owns it
  const char* tile = reinterpret_cast<const char*>(lds);
  float* scratch = lds + PANEL_W * 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int lane4 = lane * 4;
  const int w = psgpr(tid >> 6);
  const int r0 = blockIdx.x * R;
  const int r1 = min(m, r0 + R);
  const int w0 = panel_w0[blockIdx.x];
  const int wn = min(PANEL_W, n - w0);
  const int fcol = col_tile * 64 + lane;
  const bool active = fcol < k;
  const size_t kk = (size_t)k;

  // ---- stage the window's feature tile: 8 independent row-segment loads per thread in flight ----
  for (int i0 = tid; i0 < wn * 64; i0 += PANEL_WAVES * 64 * 8) {
    float t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * PANEL_WAVES * 64;
      const int rr = i >> 6, cc = col_tile * 64 + (i & 63);
      t[q] = (i < wn * 64 && cc < k) ? B[(size_t)(w0 + rr) * kk + cc] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * PANEL_WAVES * 64;
      if (i < wn * 64) lds[i] = t[q];
    }
  }
  __syncthreads();

  // sum of the staged entries [beg, end) in 64-entry blocks `stride` blocks apart, first block `first`
  auto row_sum = [&](int beg, int end, int first, int stride) -> float {
    float acc = 0.f;
    int base = beg + first * 64;
    int oj_nx = 0;
    float vj_nx = 0.f;
    if (base + lane < end) { oj_nx = in_off[base + lane]; vj_nx = in_val[base + lane]; }
    for (; base < end; base += stride * 64) {
      const int cnt = min(64, end - base);
      const int oj = oj_nx;
      const int vji = __builtin_bit_cast(int, vj_nx);
      const int nb = base + stride * 64;              // next block, fetched one block ahead
      oj_nx = 0; vj_nx = 0.f;
      if (nb + lane < end) { oj_nx = in_off[nb + lane]; vj_nx = in_val[nb + lane]; }
      int j = 0;
      for (; j + 16 <= cnt; j += 16) {                // full batches: 16 LDS reads, then 16 FMAs
        float b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          b[u] = *reinterpret_cast<const float*>(tile + __builtin_amdgcn_readlane(oj, j + u) + lane4);
#pragma unroll
        for (int u = 0; u < 16; ++u)
          acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(vji, j + u)), b[u], acc);
      }
      for (; j < cnt; ++j)                            // tail of the row's last block
        acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(vji, j)),
                   *reinterpret_cast<const float*>(tile + __builtin_amdgcn_readlane(oj, j) + lane4), acc);
    }
    return acc;
  };

  // ---- ordinary rows: one wave per row ---------------------------------------------------------
  for (int r = r0 + w; r < r1; r += PANEL_WAVES) {
    const int beg = in_rowptr[r], end = in_rowptr[r + 1];
    if (end - beg > PANEL_LONG_ROW) continue;
    const float acc = row_sum(beg, end, 0, 1);
    if (active) C[(size_t)r * kk + fcol] = acc;
  }
  // ---- hub rows: all waves together (every wave walks the same list, so the barriers match) ----
  for (int r = r0; r < r1; ++r) {
    const int beg = in_rowptr[r], end = in_rowptr[r + 1];
    if (end - beg <= PANEL_LONG_ROW) continue;
    const float part = row_sum(beg, end, w, PANEL_WAVES);
    __syncthreads();                                 // scratch free again
    scratch[w * 64 + lane] = part;
    __syncthreads();
    if (w == 0) {
      float acc = 0.f;
      for (int i = 0; i < PANEL_WAVES; ++i) acc += scratch[i * 64 + lane];   // wave order: deterministic
      if (active) C[(size_t)r * kk + fcol] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same staged sum with FOUR entries per LDS instruction (k % 4 == 0): lane = sub*16 + f reads
// 16 bytes (ds_read_b128) of the staged row of entry (step, sub) — conflict-free whatever the four
// rows are, because a 256-byte row covers every bank exactly once and the b128 lane groups take
// disjoint quarters of it — after a DPP row broadcast of the entry's (offset, value) from the
// transposed 64-entry block (exactly the layout of spmm_quad.hip).  Per 4 entries: 2 DPP moves, one
// address add, one ds_read_b128, two packed FMAs; the first version above needs 4 x (2 v_readlane +
// add + ds_read_b32 + FMA) and ran at the same 62 G entries/s as the L2 gather path.
// ---------------------------------------------------------------------------------------------
template <int UU>
__device__ __forceinline__ int panel_bcast(int v) {
  return __builtin_amdgcn_mov_dpp(v, 0x150 + UU, 0xf, 0xf, true);              // row_newbcast:UU
}

__global__ void __launch_bounds__(PANEL_WAVES * 64)
spmm_panel_in_quad_kernel(const int* __restrict__ dense_slot, const int* __restrict__ in_rowptr, const int* __restrict__ in_off,
                          const float* __restrict__ in_val, const float* __restrict__ B,
                          float* __restrict__ C, const int* __restrict__ panel_w0,
                          int m, int n, int k, int R, int col_tile) {
  extern __shared__ float lds[];                    // [PANEL_W][64] tile + [PANEL_WAVES][16] float4 scratch
  if (dense_slot && dense_slot[blockIdx.x] >= 0) return;       // a dense panel: spmm_panel_dense_mfma_kernel owns it
  const char* tile = reinterpret_cast<const char*>(lds);
  float4* scratch = reinterpret_cast<float4*>(lds + PANEL_W * 64);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = psgpr(tid >> 6);
  const int sub = lane >> 4, f = lane & 15;
  const int tl = f * 4 + sub;                       // transposed position this lane loads
  const int r0 = blockIdx.x * R;
  const int r1 = min(m, r0 + R);
  const int w0 = panel_w0[blockIdx.x];
  const int wn = min(PANEL_W, n - w0);
  const int fcol = col_tile * 64 + f * 4;
  const bool writer = fcol < k && sub == 0;
  const size_t kk = (size_t)k;
  const int foff = f * 16;

  // ---- stage the window's feature tile (columns past k are zero-filled) ----
  for (int i0 = tid; i0 < wn * 64; i0 += PANEL_WAVES * 64 * 8) {
    float t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * PANEL_WAVES * 64;
      const int rr = i >> 6, cc = col_tile * 64 + (i & 63);
      t[q] = (i < wn * 64 && cc < k) ? B[(size_t)(w0 + rr) * kk + cc] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * PANEL_WAVES * 64;
      if (i < wn * 64) lds[i] = t[q];
    }
  }
  __syncthreads();

  // this lane's (sub, f) partial over the staged entries [beg, end), 64-entry blocks `stride` apart
  auto row_sum = [&](int beg, int end, int first, int stride) -> float4 {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int base = beg + first * 64;
    int oj_nx = 0;
    float vj_nx = 0.f;
    if (base + tl < end) { oj_nx = in_off[base + tl]; vj_nx = in_val[base + tl]; }
    for (; base < end; base += stride * 64) {
      const int cnt = min(64, end - base);
      const int oj = oj_nx;
      const int vji = __builtin_bit_cast(int, vj_nx);
      const int nb = base + stride * 64;
      oj_nx = 0; vj_nx = 0.f;
      if (nb + tl < end) { oj_nx = in_off[nb + tl]; vj_nx = in_val[nb + tl]; }
      float4 b[16];
#define GCN_P_READ(UU) b[UU] = *reinterpret_cast<const float4*>(tile + panel_bcast<UU>(oj) + foff);
#define GCN_P_ALL(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
      GCN_P_ALL(GCN_P_READ)                          // entries past cnt carry offset 0: row 0 of the tile, masked below
      if (cnt == 64) {
#define GCN_P_FMA(UU) { const float v = __builtin_bit_cast(float, panel_bcast<UU>(vji));           \
          acc.x = fmaf(v, b[UU].x, acc.x); acc.y = fmaf(v, b[UU].y, acc.y);                        \
          acc.z = fmaf(v, b[UU].z, acc.z); acc.w = fmaf(v, b[UU].w, acc.w); }
        GCN_P_ALL(GCN_P_FMA)
#undef GCN_P_FMA
      } else {                                       // the row's last block: mask on the product
#define GCN_P_TAIL(UU) { const bool in = UU * 4 + sub < cnt;                                       \
          const float v = __builtin_bit_cast(float, panel_bcast<UU>(vji));                         \
          acc.x = in ? fmaf(v, b[UU].x, acc.x) : acc.x; acc.y = in ? fmaf(v, b[UU].y, acc.y) : acc.y; \
          acc.z = in ? fmaf(v, b[UU].z, acc.z) : acc.z; acc.w = in ? fmaf(v, b[UU].w, acc.w) : acc.w; }
        GCN_P_ALL(GCN_P_TAIL)
#undef GCN_P_TAIL
      }
#undef GCN_P_ALL
#undef GCN_P_READ
    }
    // over the four subs
    acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16);
    acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32);
    acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    return acc;
  };

  // ---- ordinary rows: one wave per row ----
  for (int r = r0 + w; r < r1; r += PANEL_WAVES) {
    const int beg = in_rowptr[r], end = in_rowptr[r + 1];
    if (end - beg > PANEL_LONG_ROW) continue;
    const float4 acc = row_sum(beg, end, 0, 1);
    if (writer) *reinterpret_cast<float4*>(C + (size_t)r * kk + fcol) = acc;
  }
  // ---- hub rows: all waves together, partials combined through LDS in wave order ----
  for (int r = r0; r < r1; ++r) {
    const int beg = in_rowptr[r], end = in_rowptr[r + 1];
    if (end - beg <= PANEL_LONG_ROW) continue;
    const float4 part = row_sum(beg, end, w, PANEL_WAVES);
    __syncthreads();                                 // scratch free again
    if (sub == 0) scratch[w * 16 + f] = part;
    __syncthreads();
    if (w == 0) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = 0; i < PANEL_WAVES; ++i) {
        const float4 t = scratch[i * 16 + f];
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
      if (writer) *reinterpret_cast<float4*>(C + (size_t)r * kk + fcol) = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Dense panels on the matrix cores (BASELINE north_star: "MFMA used only on the dense row-panel x feature-tile
// inner product where nnz-per-panel actually forms a dense contraction").  A panel whose 128 x 512 window is
// dense enough is stored at plan time as a DENSE fp32 tile in MFMA fragment order (256 KiB: above ~50 % density
// smaller than its CSR), and its staged product  C[128 x 64] = A[128 x 512] * Bwin[512 x 64]  runs as 8 waves x
// 256 v_mfma_f32_32x32x2_f32 — exact fp32 (a k-ordered fma chain, cdna_hip_programming.md §3), so parity-safe:
// wave (ib, nb) owns the 32 x 32 block (rows 32 ib .., columns 32 nb ..), its A fragment of k-step j is one
// coalesced 256-byte load (lane l = A[32 ib + l%32][2 j + l/32]), its B fragment one conflict-free ds_read_b32
// of the staged window.  32.8 k cycles per panel and tile whatever the density, against one LDS read per ENTRY
// in the kernels above: break-even near 13 % density, 5x at 70 %.
// A dense contraction multiplies the zeros of A too, and 0 * Inf = NaN would leak a non-finite feature row into
// rows of A that do not reference it; the staging pass therefore looks for non-finite values and, if the window
// holds any, the workgroup sums its rows entry by entry from the dense tile (skipping zeros) instead.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MFMA_WAVES = 8;

__global__ void __launch_bounds__(MFMA_WAVES * 64)
spmm_panel_dense_mfma_kernel(const float* __restrict__ adense, const int* __restrict__ dense_panel,
                             const float* __restrict__ B, float* __restrict__ C, const int* __restrict__ panel_w0,
                             int m, int n, int k, int R, int col_tile) {
  extern __shared__ float lds[];                    // [PANEL_W][64] window tile
  __shared__ int nonfinite;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = psgpr(tid >> 6);
  const int pd = blockIdx.x;
  const int p = dense_panel[pd];
  const int r0 = p * R, r1 = min(m, r0 + R);
  const int w0 = panel_w0[p];
  const int wn = min(PANEL_W, n - w0);
  const size_t kk = (size_t)k;
  if (tid == 0) nonfinite = 0;
  __syncthreads();
  // ---- stage the window's feature tile; rows past the matrix edge and columns past k are zero ----
  bool bad = false;
  for (int i0 = tid; i0 < PANEL_W * 64; i0 += MFMA_WAVES * 64 * 8) {
    float t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * MFMA_WAVES * 64;
      const int rr = i >> 6, cc = col_tile * 64 + (i & 63);
      t[q] = (i < wn * 64 && cc < k) ? B[(size_t)(w0 + rr) * kk + cc] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * MFMA_WAVES * 64;
      if (i < PANEL_W * 64) { lds[i] = t[q]; bad |= !(fabsf(t[q]) <= 3.402823466e+38f); }
    }
  }
  if (bad) nonfinite = 1;
  __syncthreads();
  const float* __restrict__ Ap = adense + (size_t)pd * 4 * (PANEL_W / 2) * 64;      // this panel's fragment image
  if (!nonfinite) {
    const int ib = w >> 1, nb = w & 1;              // this wave's 32 x 32 block of the 128 x 64 result
    const float* __restrict__ Af = Ap + (size_t)ib * (PANEL_W / 2) * 64 + lane;
    const float* __restrict__ Bl = lds + (lane >> 5) * 64 + nb * 32 + (lane & 31);   // B[k = 2 j + lane/32][col]
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = Af[(size_t)u * 64];
    for (int j0 = 0; j0 < PANEL_W / 2; j0 += 8) {   // A fragments are fetched eight k-steps ahead
      float an[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) an[u] = j0 + 8 + u < PANEL_W / 2 ? Af[(size_t)(j0 + 8 + u) * 64] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], Bl[(j0 + u) * 128], acc, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = an[u];
    }
    // C/D layout of the 32x32 forms: column = lane % 32, row = (reg % 4) + 8 (reg / 4) + 4 (lane / 32)
    const int c = col_tile * 64 + nb * 32 + (lane & 31);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int r = r0 + ib * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
      if (r < r1 && c < k) C[(size_t)r * kk + c] = acc[v];
    }
    return;
  }
  // ---- a non-finite feature value inside the window: entry by entry, zeros of A skipped (rare; slow is fine) ----
  const int c = col_tile * 64 + lane;
  for (int i = w; i < R && r0 + i < r1; i += MFMA_WAVES) {
    const float* __restrict__ Ai = Ap + (size_t)(i >> 5) * (PANEL_W / 2) * 64 + (i & 31);
    float acc = 0.f;
    for (int q = 0; q < PANEL_W; ++q) {
      const float av = Ai[(size_t)(q >> 1) * 64 + (q & 1) * 32];                    // A[i][q] (wave-uniform)
      if (av != 0.f) acc = fmaf(av, lds[q * 64 + lane], acc);
    }
    if (c < k) C[(size_t)(r0 + i) * kk + c] = acc;
  }
}

// C = act(C + bias): the epilogue alone, for when the out-of-window part is empty
__global__ void panel_epilogue_kernel(float* __restrict__ C, const float* __restrict__ bias, int relu,
                                      long long total, int k) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    float s = C[i] + (bias ? bias[i % k] : 0.f);
    if (relu) s = fmaxf(s, 0.f);
    C[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// plan time: choose the windows (device array w0_dev[npanels], caller-allocated) and measure coverage
hipError_t panel_plan(const int* rowptr, const int* col, int m, int n, int R, int* w0_dev,
                      unsigned long long* inside_host, hipStream_t st, int* cnt_dev) {
  *inside_host = 0;
  if (m <= 0) return hipSuccess;
  unsigned long long* d_inside = nullptr;
  hipError_t e;
  if ((e = hipMalloc((void**)&d_inside, sizeof(unsigned long long))) != hipSuccess) return e;
  (void)hipMemsetAsync(d_inside, 0, sizeof(unsigned long long), st);
  const int panels = (m + R - 1) / R;
  panel_windows_kernel<<<panels, 256, 0, st>>>(rowptr, col, m, n, R, w0_dev, d_inside, cnt_dev);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(inside_host, d_inside, sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_inside);
  return e;
}

// plan time: A = A_in + A_out.  Outputs are caller-allocated device arrays: in_rowptr/out_rowptr
// [m+1], in_off/in_val [nnz_in], out_col/out_val [nnz - nnz_in].  Pass 1 (entry arrays null) fills
// the row pointers and *nnz_in_host; pass 2 scatters the entries.
hipError_t panel_split(const int* rowptr, const int* col, const float* val, const int* w0_dev, int m,
                       int R, int* in_rowptr, int* out_rowptr, int* in_off, float* in_val,
                       int* out_col, float* out_val, int* nnz_in_host, hipStream_t st,
                       const int* dense_slot, float* adense) {
  hipError_t e = hipSuccess;
  int nb = (m + 3) / 4;
  if (nb > 16384) nb = 16384;
  if (!in_off) {                                      // pass 1: counts -> row pointers
    int *ci = nullptr, *co = nullptr;
    void* tmp = nullptr;
    size_t tb = 0, tb2 = 0;
    auto done = [&](hipError_t x) {
      if (ci) (void)hipFree(ci);
      if (co) (void)hipFree(co);
      if (tmp) (void)hipFree(tmp);
      return x;
    };
    if ((e = hipMalloc((void**)&ci, sizeof(int) * (size_t)(m + 1))) != hipSuccess) return done(e);
    if ((e = hipMalloc((void**)&co, sizeof(int) * (size_t)(m + 1))) != hipSuccess) return done(e);
    (void)hipMemsetAsync(ci + m, 0, sizeof(int), st);
    (void)hipMemsetAsync(co + m, 0, sizeof(int), st);
    panel_split_count_kernel<<<nb, 256, 0, st>>>(rowptr, col, w0_dev, dense_slot, m, R, ci, co);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    if ((e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, ci, in_rowptr, m + 1, st)) != hipSuccess) return done(e);
    if ((e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, co, out_rowptr, m + 1, st)) != hipSuccess) return done(e);
    if (tb2 > tb) tb = tb2;
    if ((e = hipMalloc(&tmp, tb ? tb : 16)) != hipSuccess) return done(e);
    if ((e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, ci, in_rowptr, m + 1, st)) != hipSuccess) return done(e);
    if ((e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, co, out_rowptr, m + 1, st)) != hipSuccess) return done(e);
    if ((e = hipMemcpyAsync(nnz_in_host, in_rowptr + m, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return done(e);
    return done(hipStreamSynchronize(st));
  }
  panel_split_scatter_kernel<<<nb, 256, 0, st>>>(rowptr, col, val, w0_dev, dense_slot, in_rowptr, out_rowptr, m, R,
                                                 in_off, in_val, out_col, out_val, adense);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return hipStreamSynchronize(st);
}

// staged part: C[:, tile] = A_in * B[:, tile] for one 64-column tile (raw sums, every row written)
hipError_t launch_panel_in(const int* in_rowptr, const int* in_off, const float* in_val, const float* B,
                           float* C, const int* panel_w0, int m, int n, int k, int R, int tile,
                           hipStream_t s, const int* dense_slot) {
  const size_t lds_bytes = sizeof(float) * (size_t)(PANEL_W * 64 + PANEL_WAVES * 64);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_panel_in_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_panel_in_quad_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int panels = (m + R - 1) / R;
  // four entries per LDS instruction when rows of C can take 16-byte stores (3.70 -> 2.47 ms against one entry per read, r01)
  if (k % 4 == 0 && ((uintptr_t)C & 15) == 0)
    spmm_panel_in_quad_kernel<<<dim3(panels), dim3(PANEL_WAVES * 64), lds_bytes, s>>>(
        dense_slot, in_rowptr, in_off, in_val, B, C, panel_w0, m, n, k, R, tile);
  else
    spmm_panel_in_kernel<<<dim3(panels), dim3(PANEL_WAVES * 64), lds_bytes, s>>>(
        dense_slot, in_rowptr, in_off, in_val, B, C, panel_w0, m, n, k, R, tile);
  return hipGetLastError();
}

// dense panels: C[rows of the panel, tile] = Adense * Bwin on the matrix cores (every row of those panels written)
hipError_t launch_panel_dense(const float* adense, const int* dense_panel, int ndense, const float* B, float* C,
                              const int* panel_w0, int m, int n, int k, int R, int tile, hipStream_t s) {
  if (ndense <= 0) return hipSuccess;
  const size_t lds_bytes = sizeof(float) * (size_t)(PANEL_W * 64);
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_panel_dense_mfma_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  spmm_panel_dense_mfma_kernel<<<dim3(ndense), dim3(MFMA_WAVES * 64), lds_bytes, s>>>(adense, dense_panel, B, C, panel_w0,
                                                                                     m, n, k, R, tile);
  return hipGetLastError();
}

hipError_t launch_panel_epilogue(float* C, const float* bias, int relu, int m, int k, hipStream_t s) {
  const long long total = (long long)m * k;
  if (total <= 0 || (!bias && !relu)) return hipSuccess;
  int nb = (int)((total + 255) / 256);
  if (nb > 8192) nb = 8192;
  panel_epilogue_kernel<<<nb, 256, 0, s>>>(C, bias, relu, total, k);
  return hipGetLastError();
}

}  // namespace gcn
