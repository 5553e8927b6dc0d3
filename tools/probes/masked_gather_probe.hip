// masked_gather_probe.hip — does a global_load_dwordx4 under a partial EXEC mask cost the vector-memory path
// less than a full one?  Development probe (not product code): gathers as gather_probe.hip (16 in flight, all
// L2 hits), only the first `ag` of the four 16-lane groups active in every load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) probe(const char* __restrict__ tab, float* out, int iters, unsigned rows_per_slice, int ag, int lds_groups) {
  extern __shared__ float4 hub[];                    // 16 KiB: rows for the groups served from LDS
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, f = lane & 15;
  for (int i = threadIdx.x; i < 1024; i += 256) hub[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  unsigned s = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 977u + g * 131u + 12345u;
  const unsigned base_row = (blockIdx.x & 7) * rows_per_slice;
  float4 acc = make_float4(0, 0, 0, 0);
  const bool glob = g < ag;
  const bool lds = !glob && g < ag + lds_groups;
  for (int it = 0; it < iters; ++it) {
    float4 b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      s = s * 1664525u + 1013904223u;
      const unsigned r = base_row + (s >> 8) % rows_per_slice;
      b[u] = make_float4(0, 0, 0, 0);
      if (glob) b[u] = *reinterpret_cast<const float4*>(tab + (size_t)r * 256 + f * 16);
      else if (lds) b[u] = hub[(r & 63) * 16 + f];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) { acc.x += b[u].x; acc.y += b[u].y; acc.z += b[u].z; acc.w += b[u].w; }
  }
  if (acc.x == 12345.678f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

void run(const char* tab, float* out, int ag, int lg) {
  const int iters = 1024;
  const unsigned rps = (unsigned)(2e6 / 256);
  const int nblocks = 256 * 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<<<nblocks, 256, 16384>>>(tab, out, iters, rps, ag, lg);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) probe<<<nblocks, 256, 16384>>>(tab, out, iters, rps, ag, lg);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  const double instr = (double)nblocks * 4 * iters * 16;
  printf("global groups %d, LDS groups %d: %7.3f ms  %6.2f ns per wave instruction per CU-slot  %6.1f G rows/s (global + LDS)\n", ag, lg, ms,
         ms * 1e6 / (instr / 256), instr * (ag + lg) / ms / 1e6);
}

int main() {
  char* tab; float* out;
  CK(hipMalloc(&tab, 64ull << 20)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(tab, 0, 64ull << 20));
  for (int ag = 4; ag >= 1; --ag) run(tab, out, ag, 0);
  run(tab, out, 3, 1);
  run(tab, out, 2, 2);
  run(tab, out, 0, 4);
  run(tab, out, 4, 0);
  return 0;
}
