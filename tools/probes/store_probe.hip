// store_probe.hip — what does a partial-row store cost beside L2-served gathers?
// Development probe (not product code).  Every wave gathers pseudo-random 256-byte rows of an L2-resident table
// (4 rows per global_load_dwordx4, 16 in flight, as the group kernel) and every `period` gather instructions
// issues ONE store to a streaming output:
//   lanes=16: 16 lanes x 16 B (one 256-byte row piece under an EXEC mask — what the group kernel does),
//   lanes=64: 64 lanes x 16 B (four row pieces, 1 KiB contiguous) — at a quarter of the rate for equal bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 plain, 1 sc1, 2 nt, 3 sc0 sc1
__device__ __forceinline__ void st(float* dst, const float4& v) {
  const f32x4 t = {v.x, v.y, v.z, v.w};
  if constexpr (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(dst), "v"(t) : "memory");
  if constexpr (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(t) : "memory");
  if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(dst), "v"(t) : "memory");
  if constexpr (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(dst), "v"(t) : "memory");
}

// period: gather instructions between two stores (0: never); lanes: 16 or 64 active lanes per store
template <int MODE>
__global__ void __launch_bounds__(256) probe(const char* __restrict__ tab, float* __restrict__ outbuf, int iters, unsigned rows_per_slice,
                                             int period, int lanes, size_t out_stride_wave) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, f = lane & 15;
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned s = wave * 977u + g * 131u + 12345u;
  const unsigned base_row = (blockIdx.x & 7) * rows_per_slice;
  float4 acc = make_float4(0, 0, 0, 0);
  float* op = outbuf + (size_t)wave * out_stride_wave + (lanes == 64 ? lane * 4 : f * 4);
  const bool active = lanes == 64 || g == 0;
  int since = 0;
  for (int it = 0; it < iters; ++it) {
    float4 b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      s = s * 1664525u + 1013904223u;
      const unsigned r = base_row + (s >> 8) % rows_per_slice;
      b[u] = *reinterpret_cast<const float4*>(tab + (size_t)r * 256 + f * 16);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc.x += b[u].x; acc.y += b[u].y; acc.z += b[u].z; acc.w += b[u].w;
      if (period > 0 && ++since == period) {           // (wave-uniform)
        since = 0;
        if (active) st<MODE>(op, acc);
        op += lanes == 64 ? 256 : 64;                   // floats: 1 KiB or 256 B further
      }
    }
  }
  if (acc.x == 12345.678f) outbuf[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
double run(const char* name, const char* tab, float* out, unsigned rps, int period, int lanes, double base_ms) {
  const int iters = 1024;
  const int nblocks = 256 * 4;
  const size_t nstores = period > 0 ? (size_t)iters * 16 / period : 0;
  const size_t stride = (nstores + 1) * (lanes == 64 ? 256 : 64);          // floats per wave
  if ((size_t)nblocks * 4 * stride * 4 > (6ull << 30)) { printf("output too large\n"); exit(1); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<MODE><<<nblocks, 256>>>(tab, out, iters, rps, period, lanes, stride);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) probe<MODE><<<nblocks, 256>>>(tab, out, iters, rps, period, lanes, stride);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  const double gathers = (double)nblocks * 4 * iters * 16;
  const double stores = (double)nblocks * 4 * nstores;
  const double cu_cycles = base_ms > 0 && stores > 0 ? (ms - base_ms) * 1e-3 * 2.4e9 * 256 / stores : 0.0;
  printf("%-10s period=%3d lanes=%2d: %7.3f ms  %6.1f G rows/s  stores %.2e (%.2f GB)  extra CU-cycles per store %.1f\n", name, period, lanes, ms,
         gathers * 4 / ms / 1e6, stores, stores * lanes * 16 / 1e9, cu_cycles);
  return ms;
}

int main() {
  const size_t bytes = 64ull << 20;
  char* tab; float* out;
  CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 6ull << 30));
  CK(hipMemset(tab, 0, bytes));
  for (double mb : {2.0, 7.45}) {                    // per XCD: all gathers hit L2 / as the Reddit-shaped slices (73 % hits)
  const unsigned rps = (unsigned)(mb * 1e6 / 256);
  printf("== table %.2f MB per XCD\n", mb);
  run<0>("no stores", tab, out, rps, 0, 16, 0);
  const double base = run<0>("no stores", tab, out, rps, 0, 16, 0);
  for (int period : {16, 8}) {
    run<0>("plain", tab, out, rps, period, 16, base);
    run<1>("sc1", tab, out, rps, period, 16, base);
    run<2>("nt", tab, out, rps, period, 16, base);
    run<3>("sc0 sc1", tab, out, rps, period, 16, base);
    run<0>("plain", tab, out, rps, period * 4, 64, base);
    run<1>("sc1", tab, out, rps, period * 4, 64, base);
    run<2>("nt", tab, out, rps, period * 4, 64, base);
  }
  }
  return 0;
}
