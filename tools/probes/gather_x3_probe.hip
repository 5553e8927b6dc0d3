// gather_x3_probe.hip — what does a 16-lane row engine pay for rows of 192 bytes (global_load_dwordx3, 12 B per lane)
// against rows of 256 (dwordx4) and 128 (dwordx2)?  Development probe (not product code); L2-resident tables, 4 rows per
// wave instruction, U loads in flight per wave, no index stream.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gather_x3_probe.hip -o /tmp/gather_x3_probe && /tmp/gather_x3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, int W>       // W floats per lane: row = 16 lanes x W x 4 bytes
__global__ void __launch_bounds__(256) probe(const char* __restrict__ tab, float* out, int iters, unsigned rows_per_slice) {
  typedef float vec_t __attribute__((ext_vector_type(W)));
  constexpr int ROWB = 16 * W * 4;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, f = lane & 15;
  unsigned s = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 977u + g * 131u + 12345u;
  const unsigned base_row = (blockIdx.x & 7) * rows_per_slice;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    vec_t b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s = s * 1664525u + 1013904223u;
      const unsigned r = base_row + (s >> 8) % rows_per_slice;
      b[u] = *reinterpret_cast<const vec_t*>(tab + (size_t)r * ROWB + f * (W * 4));
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < W; ++i) acc += b[u][i];
  }
  if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int U, int W>
void run(const char* tab, float* out, double mb_per_xcd) {
  constexpr int ROWB = 16 * W * 4;
  const int iters = 4096 / U * 4;
  const unsigned rps = (unsigned)(mb_per_xcd * 1e6 / ROWB);
  const int nblocks = 256 * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<U, W><<<nblocks, 256>>>(tab, out, iters, rps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) probe<U, W><<<nblocks, 256>>>(tab, out, iters, rps);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  const double rowsg = (double)nblocks * 4 * 4 * iters * U;
  printf("dwordx%d rows of %3d B, U=%2d, %4.1f MB per XCD: %7.3f ms  %6.1f G rows/s  %6.2f TB/s\n", W, ROWB, U, mb_per_xcd, ms,
         rowsg / ms / 1e6, rowsg * ROWB / ms / 1e9);
}

int main() {
  const size_t bytes = 256ull << 20;
  char* tab; float* out;
  CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(tab, 0, bytes));
  for (double mb : {2.0, 3.0, 3.7}) {
    run<16, 4>(tab, out, mb);
    run<16, 3>(tab, out, mb);
    run<16, 2>(tab, out, mb);
    run<16, 1>(tab, out, mb);
  }
  // the same ROW COUNT per XCD (what a slice of a given graph holds): 15.4 k rows = 3.94 MB at 256 B, 2.96 MB at 192 B
  run<16, 4>(tab, out, 15400 * 256 / 1e6);
  run<16, 3>(tab, out, 15400 * 192 / 1e6);
  run<16, 3>(tab, out, 20500 * 192 / 1e6);
  return 0;
}
