// spmm_panel.hip — LDS-staged feature tiles per row panel, for matrices whose non-zeros cluster in
// column ranges near their rows (graphs with community structure after Rabbit / RCM / Gorder
// renumbering).
//
// A workgroup of 16 waves owns a PANEL of R consecutive rows.  At plan time every panel gets the
// column WINDOW (W = 512 consecutive columns, start a multiple of 128) that covers most of its
// non-zeros (panel_windows_kernel).  At run time the workgroup copies the window's feature rows for
// one 64-column tile — W x 64 floats = 128 KiB of the CU's 160 KiB LDS — with coalesced loads, then
// every wave sums whole rows.  Per 64-entry block of a row the wave splits the lanes into "in the
// window" / "outside" with one vector compare and compacts both groups in registers (ballot, mbcnt
// rank, ds_permute): outside entries are whole-row gathers from L2/HBM, issued first, inside entries
// are conflict-free ds_read_b32 from the staged tile, summed while the gathers are in flight.  On a renumbered
// community graph most non-zeros hit the window, and the L2->CU traffic that bounds
// spmm_chunk_kernel shrinks by the window hit rate.
//
// Rows are owned by one wave (no partial slab, no atomics); a row is summed as one chain over its
// outside entries plus one chain over its inside entries per 64-entry block, in a fixed order —
// deterministic.  Rows longer than LONG_ROW non-zeros (hubs) are summed by all 16 waves together
// (strided blocks, per-wave partials combined through LDS in wave order).  Whether the path is used
// at all is decided at plan time from the measured window coverage; it is never a correctness question.
#include <hip/hip_runtime.h>
#include <stdint.h>
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
                     int* __restrict__ w0_out, unsigned long long* __restrict__ inside) {
  __shared__ unsigned int hist[PANEL_MAX_BINS];
  __shared__ int best_w0;
  __shared__ unsigned int best_cnt;
  const int p = blockIdx.x;
  const int r0 = p * R, r1 = min(m, r0 + R);
  const int nbins = (n + PANEL_BIN - 1) / PANEL_BIN;
  const int e0 = rowptr[r0], e1 = rowptr[r1];
  if (nbins > PANEL_MAX_BINS) {                       // too wide to histogram here: diagonal window
    const int w0 = panel_diag_window(r0, R, m, n);
    unsigned int c = 0;
    for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x) c += (unsigned)(col[e] - w0) < (unsigned)PANEL_W;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(inside, (unsigned long long)c);
    if (threadIdx.x == 0) w0_out[p] = w0;
    return;
  }
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x) atomicAdd(&hist[col[e] / PANEL_BIN], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int span = PANEL_W / PANEL_BIN;             // 4 bins
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
    best_cnt = 0;
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
    if (best_cnt) atomicAdd(inside, (unsigned long long)best_cnt);
  }
}

template <bool EPI, bool BUF>
__global__ void __launch_bounds__(PANEL_WAVES * 64)
spmm_panel_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                  const float* __restrict__ val, const float* __restrict__ B, float* __restrict__ C,
                  const float* __restrict__ bias, const int* __restrict__ panel_w0,
                  int relu, int m, int n, int k, int R, int col_tile) {
  extern __shared__ float lds[];                    // [PANEL_W][64] tile + [PANEL_WAVES][64] scratch
  float* tile = lds;
  float* scratch = lds + PANEL_W * 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = psgpr(tid >> 6);
  const int r0 = blockIdx.x * R;
  const int r1 = min(m, r0 + R);
  const int w0 = panel_w0[blockIdx.x];
  const int wn = min(PANEL_W, n - w0);
  const int fcol = col_tile * 64 + lane;
  const bool active = fcol < k;
  const size_t kk = (size_t)k;

  // ---- stage the window's feature tile (each wave copies whole 256-B row segments) -------------
  for (int i = tid; i < wn * 64; i += PANEL_WAVES * 64) {
    const int rr = i >> 6, cc = col_tile * 64 + (i & 63);
    tile[i] = cc < k ? B[(size_t)(w0 + rr) * kk + cc] : 0.f;
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(B), 0, 0xFFFFFFFFu, 0x00020000);
  const int voff = (active ? fcol : 0) * 4;
  const unsigned row_bytes = (unsigned)k * 4u;
  const float* __restrict__ Bl = B + (active ? fcol : 0);
  const float bias_f = (EPI && bias && active) ? bias[fcol] : 0.f;

  // sum of the non-zeros [beg, end) taken in 64-entry blocks `stride` blocks apart, first block `first`.
  // Per block the 64 (col, val) pairs are COMPACTED in registers — staged (in-window) entries to
  // lanes [0, nin), the others to [nin, cnt), each group in its original order (one ballot + mbcnt
  // rank + ds_permute) — and their byte offsets are pre-multiplied with one vector op, so that the
  // two inner loops are plain counted loops of v_readlane + load + FMA like spmm_chunk_kernel's.
  const int lane4 = lane * 4;
  auto row_sum = [&](int beg, int end, int first, int stride) -> float {
    float acc = 0.f;
    int base = beg + first * 64;
    int cj_nx = 0;
    float vj_nx = 0.f;
    if (base + lane < end) { cj_nx = col[base + lane]; vj_nx = val[base + lane]; }
    for (; base < end; base += stride * 64) {
      const int cnt = min(64, end - base);
      const int cj = cj_nx;
      const int vji = __builtin_bit_cast(int, vj_nx);
      const int nb = base + stride * 64;              // (col, val) of the next block, one block ahead
      if (nb + lane < end) { cj_nx = col[nb + lane]; vj_nx = val[nb + lane]; }
      const bool valid = lane < cnt;
      const bool in = valid && (unsigned)(cj - w0) < (unsigned)wn;
      const unsigned long long m_in  = __ballot(in);
      const unsigned long long m_out = __ballot(valid && !in);
      const int nin = __builtin_popcountll(m_in);
      const int rin  = __builtin_amdgcn_mbcnt_hi((unsigned)(m_in >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_in, 0));
      const int rout = __builtin_amdgcn_mbcnt_hi((unsigned)(m_out >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_out, 0));
      const int dest = in ? rin : (valid ? nin + rout : lane);
      // byte offset of the entry: into the LDS tile (staged) or into B (elsewhere)
      // (flat addressing, B >= 4 GiB: the column index itself travels, the 64-bit product is formed later)
      const int off = in ? (cj - w0) * 256 : (BUF ? (int)((unsigned)cj * row_bytes) : cj);
      const int offp = __builtin_amdgcn_ds_permute(dest * 4, valid ? off : 0);
      const int valp = __builtin_amdgcn_ds_permute(dest * 4, valid ? vji : 0);

      // ---- first batch of outside gathers goes out before the staged entries are summed --------
      int jo = nin;
      float bo[16];
      int no = min(16, cnt - jo);
      if (no > 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int so = (u < no) ? __builtin_amdgcn_readlane(offp, (jo + u) & 63) : 0;
          if (BUF) bo[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, so, 0));
          else     bo[u] = Bl[(size_t)so * kk];
        }
      }
      // ---- staged entries: LDS reads, 16 at a time ----------------------------------------------
      for (int j = 0; j < nin; j += 16) {
        float bi[16];
        if (j + 16 <= nin) {
#pragma unroll
          for (int u = 0; u < 16; ++u)
            bi[u] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(tile) +
                        __builtin_amdgcn_readlane(offp, j + u) + lane4);
#pragma unroll
          for (int u = 0; u < 16; ++u)
            acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(valp, j + u)), bi[u], acc);
        } else {
          const int ni = nin - j;
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int so = (u < ni) ? __builtin_amdgcn_readlane(offp, (j + u) & 63) : 0;
            bi[u] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(tile) + so + lane4);
          }
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (u < ni) acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(valp, (j + u) & 63)), bi[u], acc);
        }
      }
      // ---- consume the gathers; further batches if the block has more than 16 outside entries ----
      while (true) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (u < no) acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(valp, (jo + u) & 63)), bo[u], acc);
        jo += 16;
        if (jo >= cnt) break;
        no = min(16, cnt - jo);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int so = (u < no) ? __builtin_amdgcn_readlane(offp, (jo + u) & 63) : 0;
          if (BUF) bo[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, so, 0));
          else     bo[u] = Bl[(size_t)so * kk];
        }
      }
    }
    return acc;
  };
  auto finish = [&](int r, float acc) {
    if (EPI) {
      acc += bias_f;
      if (relu) acc = fmaxf(acc, 0.f);
    }
    if (active) C[(size_t)r * kk + fcol] = acc;
  };

  // ---- ordinary rows: one wave per row ---------------------------------------------------------
  for (int r = r0 + w; r < r1; r += PANEL_WAVES) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    if (end - beg > PANEL_LONG_ROW) continue;
    finish(r, row_sum(beg, end, 0, 1));
  }
  // ---- hub rows: all waves together (every wave walks the same list, so the barriers match) ----
  for (int r = r0; r < r1; ++r) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    if (end - beg <= PANEL_LONG_ROW) continue;
    const float part = row_sum(beg, end, w, PANEL_WAVES);
    __syncthreads();                                 // scratch free again
    scratch[w * 64 + lane] = part;
    __syncthreads();
    if (w == 0) {
      float acc = 0.f;
      for (int i = 0; i < PANEL_WAVES; ++i) acc += scratch[i * 64 + lane];   // wave order: deterministic
      finish(r, acc);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// plan time: choose the windows (device array w0_dev[npanels], caller-allocated) and measure coverage
hipError_t panel_plan(const int* rowptr, const int* col, int m, int n, int R, int* w0_dev,
                      unsigned long long* inside_host, hipStream_t st) {
  *inside_host = 0;
  if (m <= 0) return hipSuccess;
  unsigned long long* d_inside = nullptr;
  hipError_t e;
  if ((e = hipMalloc((void**)&d_inside, sizeof(unsigned long long))) != hipSuccess) return e;
  (void)hipMemsetAsync(d_inside, 0, sizeof(unsigned long long), st);
  const int panels = (m + R - 1) / R;
  panel_windows_kernel<<<panels, 256, 0, st>>>(rowptr, col, m, n, R, w0_dev, d_inside);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(inside_host, d_inside, sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_inside);
  return e;
}

hipError_t launch_spmm_panel(const SpmmArgs& a, int R, const int* panel_w0, hipStream_t s) {
  if (a.m <= 0 || a.k <= 0) return hipSuccess;
  const size_t lds_bytes = sizeof(float) * (size_t)(PANEL_W * 64 + PANEL_WAVES * 64);
  const bool epi = (a.bias != nullptr) || a.relu;
  const bool buf = (unsigned long long)a.n * (unsigned long long)a.k * 4ull < 0xFFFFFFF0ull;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e;
#define GCN_PANEL_ATTR(K) \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&K), \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
    GCN_PANEL_ATTR((spmm_panel_kernel<false, false>)) GCN_PANEL_ATTR((spmm_panel_kernel<false, true>))
    GCN_PANEL_ATTR((spmm_panel_kernel<true, false>))  GCN_PANEL_ATTR((spmm_panel_kernel<true, true>))
#undef GCN_PANEL_ATTR
    attr_done = true;
  }
  const int panels = (a.m + R - 1) / R;
  const int tiles = (a.k + 63) / 64;
  hipError_t e;
  if (a.ev_start && (e = hipEventRecord(a.ev_start, s)) != hipSuccess) return e;
  for (int t = 0; t < tiles; ++t) {
#define GCN_PANEL_ARGS a.rowptr, a.col, a.val, a.B, a.C, a.bias, panel_w0, a.relu, a.m, a.n, a.k, R, t
    dim3 grid(panels), block(PANEL_WAVES * 64);
    if (epi) { if (buf) spmm_panel_kernel<true, true><<<grid, block, lds_bytes, s>>>(GCN_PANEL_ARGS);
               else     spmm_panel_kernel<true, false><<<grid, block, lds_bytes, s>>>(GCN_PANEL_ARGS); }
    else     { if (buf) spmm_panel_kernel<false, true><<<grid, block, lds_bytes, s>>>(GCN_PANEL_ARGS);
               else     spmm_panel_kernel<false, false><<<grid, block, lds_bytes, s>>>(GCN_PANEL_ARGS); }
#undef GCN_PANEL_ARGS
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (a.ev_stop && (e = hipEventRecord(a.ev_stop, s)) != hipSuccess) return e;
  return hipSuccess;
}

}  // namespace gcn
