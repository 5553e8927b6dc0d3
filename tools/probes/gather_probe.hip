// gather_probe.hip — what is the chip's ceiling for random 256-byte row gathers served by L2?
// Development probe (not product code): every 16-lane group sums pseudo-random rows of a table with
// global_load_dwordx4 (4 rows per wave instruction), U loads in flight per wave, no index stream.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int U, int ROWB>     // ROWB: bytes per gathered row (256: 16 lanes x 16 B; 512: 32 lanes; 1024: 64 lanes)
__global__ void __launch_bounds__(256) probe(const char* __restrict__ tab, unsigned rows, float* out, int iters, int slices,
                                             unsigned rows_per_slice) {
  const int lane = threadIdx.x & 63;
  constexpr int LPR = ROWB / 16;                   // lanes per row
  const int g = lane / LPR, f = lane % LPR;
  unsigned s = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 977u + g * 131u + 12345u;
  // XCD x walks slice (x % slices) of the table: every XCD its own resident part
  const unsigned base_row = (blockIdx.x & 7) % slices * rows_per_slice;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    float4 b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s = s * 1664525u + 1013904223u;
      const unsigned r = base_row + (s >> 8) % rows_per_slice;
      b[u] = *reinterpret_cast<const float4*>(tab + (size_t)r * ROWB + f * 16);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += b[u].x; acc.y += b[u].y; acc.z += b[u].z; acc.w += b[u].w; }
  }
  if (acc.x == 12345.678f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int U, int ROWB>
void run(const char* name, const char* tab, unsigned rows, float* out, int slices, int blocks_per_cu) {
  const int iters = 4096 / U * 4;
  const unsigned rps = rows / slices;
  const int nblocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<U, ROWB><<<nblocks, 256>>>(tab, rows, out, iters, slices, rps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) probe<U, ROWB><<<nblocks, 256>>>(tab, rows, out, iters, slices, rps);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  const double rowsg = (double)nblocks * 4 * (64 / (ROWB / 16)) * iters * U;
  printf("%-34s U=%2d row=%4d B table=%6.1f MB per XCD, %2d blocks/CU: %7.3f ms  %6.1f G rows/s  %6.2f TB/s\n", name, U, ROWB,
         (double)rps * ROWB / 1e6, blocks_per_cu, ms, rowsg / ms / 1e6, rowsg * ROWB / ms / 1e9);
}

int main() {
  const size_t bytes = 512ull << 20;
  char* tab; float* out;
  CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(tab, 0, bytes));
  // per-XCD resident tables of 1, 2, 3, 4, 8 MB (8 slices of a 256-B-row table), and a 60 MB table shared by all
  for (double mb : {1.0, 2.0, 3.0, 4.0, 7.45}) {
    const unsigned rows = (unsigned)(mb * 1e6 / 256) * 8;
    run<16, 256>("dwordx4, 4 rows/instr", tab, rows, out, 8, 4);
  }
  {
    const unsigned rows = (unsigned)(2e6 / 256) * 8;
    run<16, 256>("dwordx4, 4 rows/instr", tab, rows, out, 8, 8);
    run<8, 256>("dwordx4, 4 rows/instr", tab, rows, out, 8, 8);
    run<8, 256>("dwordx4, 4 rows/instr", tab, rows, out, 8, 4);
    run<4, 256>("dwordx4, 4 rows/instr", tab, rows, out, 8, 8);
    run<16, 512>("dwordx4, 2 rows/instr (512 B)", tab, rows / 2, out, 8, 4);
    run<16, 1024>("dwordx4, 1 row/instr (1 KiB)", tab, rows / 4, out, 8, 4);
    run<8, 1024>("dwordx4, 1 row/instr (1 KiB)", tab, rows / 4, out, 8, 8);
  }
  run<16, 256>("dwordx4, one 60 MB table", tab, (unsigned)(59.6e6 / 256), out, 1, 4);
  run<16, 256>("dwordx4, one 400 MB table", tab, (unsigned)(400e6 / 256), out, 1, 4);
  return 0;
}
