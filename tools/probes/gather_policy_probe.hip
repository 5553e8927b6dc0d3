// gather_policy_probe.hip — do the cache-policy bits of global_load_dwordx4 change the rate of L2-served
// random 256-byte row gathers?  Development probe (not product code): as gather_probe.hip (4 rows per
// instruction, 16 in flight), the loads issued from inline asm with the bits under test.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gather_policy_probe.hip -o /tmp/gpp && /tmp/gpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void ld(f32x4& d, const char* p) {
  if constexpr (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
  if constexpr (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(d) : "v"(p) : "memory");
  if constexpr (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(d) : "v"(p) : "memory");
  if constexpr (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(d) : "v"(p) : "memory");
  if constexpr (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(d) : "v"(p) : "memory");
  if constexpr (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(d) : "v"(p) : "memory");
}

template <int MODE>
__global__ void __launch_bounds__(256) probe(const char* __restrict__ tab, float* out, int iters, unsigned rows_per_slice, unsigned stride) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, f = lane & 15;
  unsigned s = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 977u + g * 131u + 12345u;
  const unsigned base_row = (blockIdx.x & 7) * rows_per_slice;   // XCD x gathers from its own part of the table
  f32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x4 b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      s = s * 1664525u + 1013904223u;
      const unsigned r = base_row + (s >> 8) % rows_per_slice;
      ld<MODE>(b[u], tab + (size_t)r * stride + f * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += b[u];
  }
  if (acc.x == 12345.678f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
void run(const char* name, const char* tab, float* out, double mb, unsigned stride = 256) {
  const int iters = 1024;
  const unsigned rps = (unsigned)(mb * 1e6 / 256);
  const int nblocks = 256 * 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<MODE><<<nblocks, 256>>>(tab, out, iters, rps, stride);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) probe<MODE><<<nblocks, 256>>>(tab, out, iters, rps, stride);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  const double rows = (double)nblocks * 4 * 4 * iters * 16;
  printf("%-10s row stride %4u B, table %5.2f MB per XCD: %7.3f ms  %6.1f G rows/s  %6.2f TB/s\n", name, stride, mb, ms, rows / ms / 1e6, rows * 256 / ms / 1e9);
}

int main() {
  char* tab; float* out;
  CK(hipMalloc(&tab, 512ull << 20)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(tab, 0, 512ull << 20));
  // the 64-column tile of a k = 128 / 256 table: 256 bytes used out of every 512 / 1024
  for (double mb : {1.0, 2.0, 3.7}) {
    run<0>("plain", tab, out, mb, 256);
    run<0>("plain", tab, out, mb, 512);
    run<0>("plain", tab, out, mb, 1024);
    run<0>("plain", tab, out, mb, 768);
  }
  for (double mb : {2.0, 3.7, 7.45}) {
    run<0>("plain", tab, out, mb);
    run<1>("sc0", tab, out, mb);
    run<2>("sc1", tab, out, mb);
    run<3>("nt", tab, out, mb);
    run<4>("sc0 sc1", tab, out, mb);
    run<5>("sc0 nt", tab, out, mb);
    run<0>("plain", tab, out, mb);
  }
  return 0;
}
