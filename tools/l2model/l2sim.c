// crude L2 model: 4 MiB, 16-way, 128-B lines, LRU, hashed set index; one XCD at a time.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
typedef struct { uint64_t *tag; uint32_t *age; int sets, ways; uint32_t clock; } Cache;
static inline uint32_t hash64(uint64_t x){ x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return (uint32_t)x; }
static int access_line(Cache* c, uint64_t line, int allocate){
  uint32_t s = hash64(line) % (uint32_t)c->sets;
  uint64_t* t = c->tag + (size_t)s * c->ways; uint32_t* a = c->age + (size_t)s * c->ways;
  ++c->clock;
  int victim = 0; uint32_t oldest = 0xffffffffu;
  for (int w = 0; w < c->ways; ++w){
    if (t[w] == line + 1){ a[w] = c->clock; return 1; }
    if (a[w] < oldest){ oldest = a[w]; victim = w; }
  }
  if (allocate){ t[victim] = line + 1; a[victim] = c->clock; }
  return 0;
}
// vrowptr: [nv+1] virtual CSR; vcol: column (row of B') per entry; process entries [e_lo, e_hi) as chunks of T,
// `open` chunks interleaved in blocks of 64.  row_lines: lines per gathered row (2 for 256 B).
// store_alloc: do partial-row stores allocate in L2 (1) or not (0); col_bytes: bytes of index stream per entry.
// returns hits, misses via out[0], out[1]; out[2] = stream lines inserted
void simulate(const int64_t* vrowptr, int64_t nv, const int32_t* vcol, int64_t e_lo, int64_t e_hi, int T, int open,
              int row_lines, int store_alloc, int col_bytes, int cache_bytes, int ways, int64_t* out){
  Cache c; c.ways = ways; c.sets = cache_bytes / 128 / ways; c.clock = 0;
  c.tag = calloc((size_t)c.sets * ways, 8); c.age = calloc((size_t)c.sets * ways, 4);
  int64_t hits = 0, miss = 0, stream = 0;
  const uint64_t TABLE = 0, CV = 1ull << 40, COL = 2ull << 40;
  int64_t nchunks = (e_hi - e_lo + T - 1) / T;
  // row index cursor per open chunk
  int64_t* rcur = malloc(sizeof(int64_t) * open);
  for (int64_t w0 = 0; w0 < nchunks; w0 += open){
    int64_t nw = nchunks - w0 < open ? nchunks - w0 : open;
    for (int64_t i = 0; i < nw; ++i){
      int64_t start = e_lo + (w0 + i) * T;
      // binary search row containing start
      int64_t lo = 0, hi = nv; while (lo < hi){ int64_t mid = (lo + hi) >> 1; if (vrowptr[mid + 1] <= start) lo = mid + 1; else hi = mid; }
      rcur[i] = lo;
    }
    for (int b = 0; b < T; b += 64){
      for (int64_t i = 0; i < nw; ++i){
        int64_t start = e_lo + (w0 + i) * T + b;
        int64_t end = start + 64; int64_t cend = e_lo + (w0 + i + 1) * T; if (cend > e_hi) cend = e_hi; if (end > cend) end = cend;
        if (start >= end) continue;
        // index stream: col_bytes*64 bytes per block
        if (col_bytes){ uint64_t l0 = (COL + (uint64_t)start * col_bytes) >> 7, l1 = (COL + (uint64_t)(end) * col_bytes - 1) >> 7;
          for (uint64_t l = l0; l <= l1; ++l){ if (!access_line(&c, l, 1)) ++stream; } }
        for (int64_t e = start; e < end; ++e){
          uint64_t base = (TABLE + (uint64_t)vcol[e] * (uint64_t)(row_lines * 128)) >> 7;
          for (int l = 0; l < row_lines; ++l){ if (access_line(&c, base + l, 1)) ++hits; else ++miss; }
          while (rcur[i] < nv && vrowptr[rcur[i] + 1] <= e + 1){   // row(s) ending here: partial-row store
            uint64_t sb = (CV + (uint64_t)rcur[i] * (uint64_t)(row_lines * 128)) >> 7;
            for (int l = 0; l < row_lines; ++l){ access_line(&c, sb + l, store_alloc); ++stream; }
            ++rcur[i];
          }
        }
      }
    }
  }
  out[0] = hits; out[1] = miss; out[2] = stream;
  free(c.tag); free(c.age); free(rcur);
}
