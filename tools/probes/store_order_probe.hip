// store_order_probe.hip — does a streaming store cost a gather loop more when it is issued IN FRONT of a block's gathers
// (the block's wait for its gathers then also waits for the store's acknowledgement: loads and stores share the in-order
// vmcnt counter) than BEHIND them (the wait can leave it outstanding: vmcnt(1))?  Development probe, all memory
// instructions and waits in inline assembly so that the compiler's own wait insertion is out of the picture.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_order_probe.hip -o /tmp/store_order_probe && /tmp/store_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: no stores; 1: store in front of the gathers, wait vmcnt(0); 2: store behind the gathers, wait vmcnt(1);
// 3: store behind the gathers but wait vmcnt(0) (what a compiler emits at a control-flow join)
// POL 0 plain, 1 nt, 2 sc1
template <int MODE, int POL, int EVERY>
__global__ void __launch_bounds__(256) probe(const char* __restrict__ tab, char* __restrict__ slab, float* out, int iters,
                                             unsigned rows_per_slice, unsigned long long slab_bytes, int scatter) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, f = lane & 15;
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned s = wave * 977u + g * 131u + 12345u;
  const unsigned base_row = (blockIdx.x & 7) * rows_per_slice;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned long long so = ((unsigned long long)wave * 1024ull * 64ull) % slab_bytes;      // this wave's streaming position
  for (int it = 0; it < iters; ++it) {
    unsigned off[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      s = s * 1664525u + 1013904223u;
      off[u] = (base_row + (s >> 8) % rows_per_slice) * 256u + f * 16u;
    }
    const bool st = MODE != 0 && (it % EVERY) == 0;              // wave-uniform
    char* sp = slab + so + lane * 16;
    if (st) {
      if (scatter == 1) {             // anywhere in the slab (1 KiB aligned): every store its own page, as far as the TLBs care
        unsigned h = (wave * 2654435761u) ^ ((unsigned)it * 40503u + 0x9E3779B9u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        so = ((unsigned long long)h % (slab_bytes >> 10)) << 10;
      } else if (scatter == 2) {      // four 256-byte pieces 512 bytes apart (a 64-column tile's rows in a 128-column slab), region per wave
        so += 2048; if (so + 2048 > slab_bytes) so = 0;
        sp = slab + so + (lane >> 4) * 512 + (lane & 15) * 16;
      } else { so += 1024; if (so + 1024 > slab_bytes) so = 0; }
    }
    if (scatter == 1) sp = slab + so + lane * 16;
#define ST() do { if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(sp), "v"(acc) : "memory"); \
                  else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(sp), "v"(acc) : "memory"); \
                  else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(sp), "v"(acc) : "memory"); \
                  else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(sp), "v"(acc) : "memory"); \
                  else if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(sp), "v"(acc) : "memory"); \
                  else if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(sp), "v"(acc) : "memory"); \
                  else if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(sp), "v"(acc) : "memory"); \
                  else asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(sp), "v"(acc) : "memory"); } while (0)
    if (MODE == 1 && st) ST();
    f32x4 b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(b[u]) : "v"(off[u]), "s"(tab) : "memory");
    if (MODE >= 2 && st) {
      ST();
      if (MODE == 2) asm volatile("s_waitcnt vmcnt(1)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]),
                                  "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15]) :: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]),
                        "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15]) :: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]),
                   "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15]) :: "memory");
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += b[u];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc.x == 12345.678f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE, int POL, int EVERY>
void run(const char* what, const char* tab, char* slab, float* out, unsigned long long slab_bytes, double table_mb = 3.7, int scatter = 0) {
  const int iters = 1024;
  const unsigned rps = (unsigned)(table_mb * 1e6 / 256);
  const int nblocks = 256 * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<MODE, POL, EVERY><<<nblocks, 256>>>(tab, slab, out, iters, rps, slab_bytes, scatter);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) probe<MODE, POL, EVERY><<<nblocks, 256>>>(tab, slab, out, iters, rps, slab_bytes, scatter);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  const double rowsg = (double)nblocks * 4 * 4 * iters * 16;
  printf("%-58s store every %2d blocks: %7.3f ms  %6.1f G rows/s\n", what, EVERY, ms, rowsg / ms / 1e6);
}

int main() {
  const unsigned long long slab_bytes = 2ull << 30;
  char *tab, *slab; float* out;
  CK(hipMalloc(&tab, 64ull << 20)); CK(hipMalloc(&slab, slab_bytes)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(tab, 0, 64ull << 20));
  run<0, 1, 8>("no stores", tab, slab, out, slab_bytes);
  run<1, 1, 8>("nt store IN FRONT of the gathers, vmcnt(0)", tab, slab, out, slab_bytes);
  run<2, 1, 8>("nt store BEHIND the gathers, vmcnt(1)", tab, slab, out, slab_bytes);
  run<3, 1, 8>("nt store behind the gathers, vmcnt(0)", tab, slab, out, slab_bytes);
  run<1, 0, 8>("plain store in front, vmcnt(0)", tab, slab, out, slab_bytes);
  run<2, 0, 8>("plain store behind, vmcnt(1)", tab, slab, out, slab_bytes);
  run<1, 2, 8>("sc1 store in front, vmcnt(0)", tab, slab, out, slab_bytes);
  run<2, 2, 8>("sc1 store behind, vmcnt(1)", tab, slab, out, slab_bytes);
  run<1, 1, 2>("nt store IN FRONT of the gathers, vmcnt(0)", tab, slab, out, slab_bytes);
  run<2, 1, 2>("nt store BEHIND the gathers, vmcnt(1)", tab, slab, out, slab_bytes);
  run<3, 1, 2>("nt store behind the gathers, vmcnt(0)", tab, slab, out, slab_bytes);
  // the same with tables that no longer fit an XCD's L2 (4 MiB) cleanly: some gathers miss and share the fabric with the stores
  for (double mb : {3.98, 4.2, 4.5}) {
    printf("table %.2f MB per XCD\n", mb);
    run<0, 1, 8>("  no stores", tab, slab, out, slab_bytes, mb);
    run<1, 1, 8>("  nt store in front, vmcnt(0)", tab, slab, out, slab_bytes, mb);
    run<2, 1, 8>("  nt store behind, vmcnt(1)", tab, slab, out, slab_bytes, mb);
    run<1, 0, 8>("  plain store in front, vmcnt(0)", tab, slab, out, slab_bytes, mb);
    run<1, 1, 2>("  nt store in front, vmcnt(0)", tab, slab, out, slab_bytes, mb);
    run<2, 1, 2>("  nt store behind, vmcnt(1)", tab, slab, out, slab_bytes, mb);
  }
  printf("where the stores go (table 3.3 MB per XCD, nt, in front)\n");
  run<0, 1, 8>("  no stores", tab, slab, out, slab_bytes, 3.3, 0);
  run<1, 1, 8>("  a region per wave, 1 KiB after 1 KiB", tab, slab, out, slab_bytes, 3.3, 0);
  run<1, 1, 8>("  anywhere in the 2 GB slab", tab, slab, out, slab_bytes, 3.3, 1);
  run<1, 1, 8>("  four 256-byte pieces 512 bytes apart", tab, slab, out, slab_bytes, 3.3, 2);
  run<1, 1, 2>("  a region per wave, 1 KiB after 1 KiB", tab, slab, out, slab_bytes, 3.3, 0);
  run<1, 1, 2>("  anywhere in the 2 GB slab", tab, slab, out, slab_bytes, 3.3, 1);
  run<1, 1, 2>("  four 256-byte pieces 512 bytes apart", tab, slab, out, slab_bytes, 3.3, 2);
  printf("store policies, table 3.98 MB per XCD, store in front, every 2 blocks\n");
  run<0, 0, 2>("  no stores", tab, slab, out, slab_bytes, 3.98);
  run<1, 0, 2>("  plain", tab, slab, out, slab_bytes, 3.98);
  run<1, 1, 2>("  nt", tab, slab, out, slab_bytes, 3.98);
  run<1, 2, 2>("  sc1", tab, slab, out, slab_bytes, 3.98);
  run<1, 3, 2>("  sc0", tab, slab, out, slab_bytes, 3.98);
  run<1, 4, 2>("  sc0 sc1", tab, slab, out, slab_bytes, 3.98);
  run<1, 5, 2>("  sc0 nt", tab, slab, out, slab_bytes, 3.98);
  run<1, 6, 2>("  sc1 nt", tab, slab, out, slab_bytes, 3.98);
  run<1, 7, 2>("  sc0 sc1 nt", tab, slab, out, slab_bytes, 3.98);
  return 0;
}
