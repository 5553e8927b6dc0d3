// spmm_group.hip — the sliced main pass as FOUR independent 16-lane row engines per wave.
//
// Why (profiles/r02c_pmc_quad_kernel_reddit_k128_S8.txt, Reddit-shaped k = 128, 8 slices): the four-per-gather kernel
// of spmm_quad.hip keeps one row per WAVE — its four 16-lane groups hold four interleaved partial sums of
// the same row — so every row end costs a cross-lane reduction, a scalar walk over the 64-entry block with
// per-lane masks, and scalar row-pointer loads.  With 8 column slices a virtual row is 62 entries long, so
// nearly every 64-entry block takes that slow path: 236 scalar and 189 vector instructions per block
// against 17 vector-memory instructions, the texture addresser busy 58 % of the time, and every further
// slice (shorter virtual rows) made it slower although the L2 misses halved.
//
// Here each 16-lane group (lane = g*16 + f, f = which float4 of the 64-column tile) walks its OWN chunk
// of the slice-major stream, one entry per step, and owns the complete sum of its current row:
//   * one global_load_dwordx4 still fetches four feature rows (4 x 256 B), one per group;
//   * a row end is a bit in the stream (bit 15 of the 16-bit entry) and costs the group ONE 256-byte write
//     under an EXEC mask — into its LDS ring, from where four consecutive rows leave with one 64-lane store
//     (spmm_group_ring_kernel), or straight to memory — no cross-lane reduction, no row pointers in the kernel;
//   * entries are 16 bits: the column's offset inside its slice (slices <= 32 767 columns); every virtual
//     row has at least one entry (empty ones get a padding entry that gathers the slice's all-zero row),
//     so "next row" is pointer arithmetic; runs of 64 entries are stored lane-major, so a lane fetches its
//     entries of four blocks with one 8-byte load;
//   * the byte offset of the gathered row is computed once per entry at load time (one lane = one entry
//     of its group's 16-entry block) and reaches the group by a DPP row broadcast fused into the address
//     add: per step one VALU op for the address, one load, two packed adds.
// Chunks are T entries of ONE group; a wave owns four consecutive chunks, a block sixteen, and the blocks
// of an XCD take consecutive chunks in dispatch order, so an XCD walks its part of the stream front to
// back and slices meet its L2 one after the other (with more chunks per wave, as before, the second
// chunk of an early wave ran beside the first chunk of a late one: two slices in one L2).
// A row cut by chunk ends leaves its first piece in Cv[row] and the pieces of the following chunks in the slab P (one
// head piece per chunk); the slice reduction adds them behind the row's slices, in chunk order (cut lists, slicing.hip;
// group_fixup_kernel does the same as a pass of its own where the reduction cannot): results are bitwise reproducible.
//
// Value-free (spmm_group_kernel: every stored entry counts 1; the caller gathers from a copy of B whose rows
// were scaled by u_col and scales finished rows by u_row, api_spmm.cpp) or, for values that do not factor,
// with one fp32 value per entry beside the stream (spmm_group_weighted_kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "spmm_kernels.h"

namespace gcn {

// value held by lane UU of this lane's 16-lane row (DPP row_newbcast)
template <int UU>
__device__ __forceinline__ int row_bcast(int v) {
  return __builtin_amdgcn_mov_dpp(v, 0x150 + UU, 0xf, 0xf, true);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Partial-row store, POLICY: 0 plain, 1 sc1 (write-through), 2 nt (streaming; the default).  The slab of
// partial rows is read back only by the reduction that follows; left in L2 its lines push out feature rows
// the gathers are about to reuse.  Measured (profiles/r02z2_row_end_store_ablation.log, r02z4_store_policy.log,
// Reddit-shaped k = 128): with the store instruction alone removed the main passes run 2.69 ms at 8 slices and
// 2.22 ms at 16 — the whole cost of a row end is its store; sc1 stores cost 2.86 / 2.76 ms (8 / 16 slices),
// plain ones 2.86 / 2.71, nt ones 2.82 / 2.53.  tools/probes/store_probe.hip shows why: beside L2-served
// gathers an sc1 store holds the vector-memory path ~30 cycles per instruction, a plain or nt one ~5, and
// only sc1 and nt keep the written lines from displacing the table.  (There is no builtin for a 16-byte sc1
// store; the trailing s_nop keeps the compiler's next instruction off the data registers until the store has
// read them, cdna_hip_programming.md §5.7.)
// GCN_ABLATE (development builds only, tools/ablate_group.sh: wrong results, exact costs): bit 0 no partial-row stores,
// bit 1 no row-end handling, bit 2 no stream loads after the first run, bit 3 partial rows at a stride of 64 floats;
// weighted walk (r04): bit 4 the value stream read from its first 4 KiB only (cache-resident: its bytes without its
// traffic), bit 5 no value broadcast (every lane multiplies by its OWN entry's value), bit 6 adds instead of FMAs,
// bit 7 the value-free walk WITHOUT the LDS ring, bit 8 the weighted walk WITH it (these two give right results)
#ifndef GCN_ABLATE
#define GCN_ABLATE 0
#endif
#ifndef GCN_STORE_POLICY
#define GCN_STORE_POLICY -1                          // development builds: 0 plain / 1 sc1 / 2 nt for EVERY partial-row store
#endif
template <int POLICY_>
__device__ __forceinline__ void store_row_piece(float* dst, const float4& v) {
  constexpr int POLICY = GCN_STORE_POLICY >= 0 ? GCN_STORE_POLICY : POLICY_;
  if constexpr ((GCN_ABLATE & 1) != 0) { asm volatile("" : : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); return; }
  const f32x4 t = {v.x, v.y, v.z, v.w};
  if constexpr (POLICY == 1) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(t) : "memory");
  } else if constexpr (POLICY == 2) {
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(dst));
  } else {
    *reinterpret_cast<f32x4*>(dst) = t;
  }
}

// stream  [nchunks*T] u16: bits 0..14 column offset inside the slice (== slice width: the all-zero row),
//                          bit 15 = last entry of its virtual row; every run of 64 entries stored lane-major (group_phys)
// chunk_meta [nchunks]: {2 * (virtual row holding entry c*T) + (that row began in an earlier chunk), first row of
// the chunk's slice in Bp}
// Bp: scaled copy of B, slice s at rows [s*(w+1), (s+1)*(w+1)), row w of every slice all zero
// nchunks % 32 == 0 (the stream is padded), so every XCD owns whole waves.
// vals (VALS only) [nchunks*T]: the matrix values in stream order, 0 at padding entries
// BIG: the sliced copy of B is 4 GiB or more (or has 2^24 rows or more): the slice's first row is added to the table
// pointer in 64 bits, per lane, and only the offset INSIDE the slice (< 32 768 rows x < 128 KiB) stays in 32 bits —
// one more vector instruction per gather (add + carry instead of one add).  Without it the 32-bit byte offset
// (entry + base) * row_bytes would wrap silently.
template <int POLICY, bool VALS, bool RING, bool BIG>
__device__ __forceinline__ void
group_walk(const unsigned short* __restrict__ stream, const float* __restrict__ vals, const int2* __restrict__ chunk_meta,
           const float* __restrict__ Bp, float* __restrict__ Cv, float* __restrict__ P,
           int nchunks, int T, int k, int seg_blocks, int ldb, int stream_nt, int blocks_per_tile, const int* __restrict__ dyn) {
  // T: entries per chunk of ONE group, a multiple of 64 (a chunk is whole runs of four blocks) — a run-time value: the
  // plan picks it so that the blocks fill whole rounds of the chip on small matrices (group_chunk, plan_policy.cpp)
  // dyn (drop-in flexspmm only): {buffers recognised, chunk count} written by dropin_guard_kernel — the grid was
  // sized from an upper bound of the chunk count, and buffers this library did not pack are not walked at all
  if (dyn) { if (dyn[0] == 0) return; nchunks = dyn[1]; }
  const int lane = threadIdx.x & 63;
  const int wib  = threadIdx.x >> 6;
  const int g    = lane >> 4;
  const int f    = lane & 15;
  const int per_xcd = nchunks >> 3;
  // One launch can cover several 64-column tiles: blocks [t*blocks_per_tile, (t+1)*blocks_per_tile) walk the whole
  // stream for tile col_tile + t.  Blocks are dispatched in index order, so the next tile starts on the CUs the
  // previous one's last blocks leave idle (blocks_per_tile % 8 == 0: a block's XCD is blockIdx % 8 either way).
  // Order of the (tile, block) pairs (launch_group_t says why).  seg_blocks == 0: tile-major.  seg_blocks = Q > 0: every
  // XCD's blocks in runs of Q — run 0 for tile 0, run 0 for tile 1, ..., then run 1.  Placement only: the result is the same.
  int tile_in_launch, bx, col_tile;
  if (seg_blocks > 0) {
    const int Q = seg_blocks, nbx = blocks_per_tile >> 3, tiles = (k + 63) >> 6;
    const int x = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
    const int nseg = (nbx + Q - 1) / Q, full = (nseg - 1) * tiles * Q;
    int j;
    if (i < full) { const int seg = i / (tiles * Q), r = i - seg * tiles * Q; tile_in_launch = r / Q; j = seg * Q + (r - tile_in_launch * Q); }
    else { const int last = nbx - (nseg - 1) * Q, r = i - full; tile_in_launch = r / last; j = (nseg - 1) * Q + (r - tile_in_launch * last); }
    bx = j * 8 + x;
    col_tile = tile_in_launch;
  } else {
    tile_in_launch = (int)blockIdx.x / blocks_per_tile;
    bx = (int)blockIdx.x - tile_in_launch * blocks_per_tile;
    col_tile = tile_in_launch;
  }
  const int c_in = ((bx >> 3) * 4 + wib) * 4;
  if (c_in >= per_xcd) return;                                  // (whole wave: per_xcd % 4 == 0)
  const int c = (bx & 7) * per_xcd + c_in + g;                  // this group's chunk

  const int fcol = col_tile * 64 + f * 4;
  const bool fok = fcol < k;                                    // (k % 4 == 0: a float4 is all in or all out)
  const unsigned row_bytes = (unsigned)ldb * 4u;
  const unsigned foff = (unsigned)(fok ? fcol : col_tile * 64) * 4u;
  const char* Bb = reinterpret_cast<const char*>(Bp);
  const size_t kk = (GCN_ABLATE & 8) ? (size_t)64 : (size_t)k;  // (bit 3: partial rows of a tile contiguous — a layout experiment)

  const int2 meta = chunk_meta[c];                              // one load: nothing else stands before the first gather
  const int vrow = meta.x >> 1;                                 // virtual row holding the chunk's first entry
  const bool head = meta.x & 1;                                 // ... which began in an earlier chunk
  const int base = BIG ? 0 : meta.y;                            // first row of this chunk's slice in Bp
  if constexpr (BIG) Bb += (size_t)meta.y * (size_t)row_bytes;  // (per lane: the groups of a wave can sit in different slices)
  float* ptr  = head ? P + (size_t)(2 * c) * kk + fcol : Cv + (size_t)vrow * kk + fcol;
  float* nptr = Cv + (size_t)(vrow + 1) * kk + fcol;
  bool first = true;                                            // no row of this chunk has ended yet
  // RING: finished rows wait in LDS, four slots per group, and leave four at a time — consecutive rows of ONE group,
  // written by the whole wave with one 64-lane store instead of four 16-lane ones (a store occupies the addressers
  // like a gather whatever its width)
  __shared__ f32x4 ring[RING ? 4 : 1][4][4][16];
  int ring_n = 0;                                               // rows of this lane's group waiting in the ring
  float* ring_base = nullptr;                                   // ... the first of them goes here (the next ones kk further each)
#define GCN_G_DRAIN(G2, ROWS)                                                                       \
  {                                                                                                 \
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(uintptr_t)ring_base, 16 * G2);    \
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((uintptr_t)ring_base >> 32), 16 * G2); \
    float* b0 = reinterpret_cast<float*>(((uintptr_t)hi << 32) | lo);                              \
    const f32x4 rv = ring[wib][G2][lane >> 4][f];                                                   \
    if (fok && (lane >> 4) < (ROWS))                                                                \
      store_row_piece<POLICY>(b0 + (size_t)(lane >> 4) * kk + f * 4, make_float4(rv.x, rv.y, rv.z, rv.w)); \
    if (g == G2) ring_n = 0;                                                                        \
  }

  // the stream is stored in runs of 64 entries, lane-major (slicing.hip, group_phys): lane f reads its entries of
  // four consecutive blocks with one 8-byte load (16 bytes for the values)
  typedef unsigned int u32x2_g __attribute__((ext_vector_type(2)));
  const u32x2_g* __restrict__ sp = reinterpret_cast<const u32x2_g*>(stream + (size_t)c * T) + f;
  const f32x4* __restrict__ vp = VALS ? reinterpret_cast<const f32x4*>(vals + (size_t)c * T) + f : nullptr;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // stream_nt (streams too large to stay cached from one SpMM to the next, launch_spmm_group): non-temporal loads
  // keep them from displacing the table — 2.94 -> 2.87 ms per SpMM on the 232 MB stream of the Reddit-shaped
  // graph (profiles/r02zn_*); a stream that fits the caches is better left there (profiles/r02zo_*)
  u32x2_g eq = stream_nt ? __builtin_nontemporal_load(sp) : sp[0], eq_nx = eq;
  f32x4 vq = {0.f, 0.f, 0.f, 0.f}, vq_nx = vq;
  if constexpr (VALS) { vq = (GCN_ABLATE & 16) ? (reinterpret_cast<const f32x4*>(vals) + f)[0] : (stream_nt ? __builtin_nontemporal_load(vp) : vp[0]); vq_nx = vq; }
  unsigned fl = 0;
#pragma unroll 1
  for (int blk = 0; blk < T / 16; ++blk) {
    const int j = blk & 3;
    if (j == 0 && blk + 4 < T / 16 && !(GCN_ABLATE & 4)) {      // the next run, a whole run ahead of its use
      const int nx = (blk / 4 + 1) * 16;
      eq_nx = stream_nt ? __builtin_nontemporal_load(sp + nx) : sp[nx];
      if constexpr (VALS) vq_nx = (GCN_ABLATE & 16) ? (reinterpret_cast<const f32x4*>(vals) + f)[nx & 63]   /* every chunk the same 1 KiB */
                                                    : (stream_nt ? __builtin_nontemporal_load(vp + nx) : vp[nx]);
    }
    const unsigned e = ((j & 2 ? eq.y : eq.x) >> (16 * (j & 1))) & 0xFFFFu;
    int vbits = 0;                                              // this lane's entry's value; step u takes lane u's
    if constexpr (VALS) vbits = __builtin_bit_cast(int, j == 0 ? vq.x : j == 1 ? vq.y : j == 2 ? vq.z : vq.w);
    if (j == 3) { eq = eq_nx; vq = vq_nx; }
    const int rowoff = (int)(__umul24((e & 0x7FFFu) + (unsigned)base, row_bytes));
    fl = e >> 15;
    float4 b[16];
#define GCN_G_ALL(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define GCN_G_GATHER(UU) \
    b[UU] = *reinterpret_cast<const float4*>(Bb + (size_t)((unsigned)row_bcast<UU>(rowoff) + foff));
    GCN_G_ALL(GCN_G_GATHER)
#undef GCN_G_GATHER
    const unsigned long long ends = (GCN_ABLATE & 2) ? 0ull : __ballot(fl != 0);   // bit g*16+u: entry u of group g ends a row
    if (ends == 0ull) {
#define GCN_G_ADD(UU)                                                                               \
      if constexpr (VALS && (GCN_ABLATE & 64) != 0) {                                               \
        acc.x += b[UU].x; acc.y += b[UU].y; acc.z += b[UU].z; acc.w += b[UU].w;                     \
        asm volatile("" : : "v"(vbits));                                                            \
      } else if constexpr (VALS) {                                                                  \
        const float vu = __builtin_bit_cast(float, (GCN_ABLATE & 32) ? vbits : row_bcast<UU>(vbits)); \
        acc.x = fmaf(vu, b[UU].x, acc.x); acc.y = fmaf(vu, b[UU].y, acc.y);                         \
        acc.z = fmaf(vu, b[UU].z, acc.z); acc.w = fmaf(vu, b[UU].w, acc.w);                         \
      } else { acc.x += b[UU].x; acc.y += b[UU].y; acc.z += b[UU].z; acc.w += b[UU].w; }
      GCN_G_ALL(GCN_G_ADD)
    } else {
#define GCN_G_STEP(UU)                                                                              \
      GCN_G_ADD(UU)                                                                                 \
      if (ends & (0x0001000100010001ull << UU)) {                /* some group ends a row here */    \
        if (row_bcast<UU>((int)fl)) {                                                               \
          if (RING && !(first && head) && ring_n < 4) {                                             \
            if (fok) ring[wib][g][ring_n][f] = f32x4{acc.x, acc.y, acc.z, acc.w};                   \
            if (ring_n == 0) ring_base = ptr;                                                       \
            ++ring_n;                                                                               \
          } else if (fok) store_row_piece<POLICY>(ptr, acc);                                        \
          acc = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
          ptr = nptr; nptr += kk; first = false;                                                    \
        }                                                                                           \
      }
      GCN_G_ALL(GCN_G_STEP)
#undef GCN_G_STEP
#undef GCN_G_ADD
      if constexpr (RING) {
        const unsigned long long full = __ballot(ring_n == 4);
        if (full) {
          if (full & 0x0000000000000001ull) GCN_G_DRAIN(0, 4)
          if (full & 0x0000000000010000ull) GCN_G_DRAIN(1, 4)
          if (full & 0x0000000100000000ull) GCN_G_DRAIN(2, 4)
          if (full & 0x0001000000000000ull) GCN_G_DRAIN(3, 4)
        }
      }
    }
#undef GCN_G_ALL
  }
  if constexpr (RING) {                                         // what is left in the rings
    const unsigned long long some = __ballot(ring_n > 0);
    if (some & 0x0000000000000001ull) GCN_G_DRAIN(0, __builtin_amdgcn_readlane(ring_n, 0))
    if (some & 0x0000000000010000ull) GCN_G_DRAIN(1, __builtin_amdgcn_readlane(ring_n, 16))
    if (some & 0x0000000100000000ull) GCN_G_DRAIN(2, __builtin_amdgcn_readlane(ring_n, 32))
    if (some & 0x0001000000000000ull) GCN_G_DRAIN(3, __builtin_amdgcn_readlane(ring_n, 48))
  }
#undef GCN_G_DRAIN
  // the row piece that sticks out of the chunk's end (the last entry did not end its row): it is the FIRST piece of its
  // row — unless the whole chunk lies inside one row, then it is this chunk's head piece — and goes where the row's
  // partial sum lives, Cv[row]; the pieces of the chunks the row runs on into (their head pieces, P[2c]) are added by
  // the slice reduction (cut lists) or by group_fixup_kernel
  if (!row_bcast<15>((int)fl)) {
    if (fok) store_row_piece<POLICY>(ptr, acc);
  }
}

// (the value-free walk WITHOUT the LDS ring — every finished row stored by its own group at once — was the r02 kernel; with
//  the ring 2.929 -> 2.874 ms, profiles/r02zt_*: only the ring variant is instantiated.  The weighted walk has none: it
//  sits at 126 VGPRs already and the ring bought nothing there, 3.184 -> 3.176 ms.)
template <int POLICY, bool BIG>
__global__ void __launch_bounds__(256)
spmm_group_ring_kernel(const unsigned short* __restrict__ stream, const int2* __restrict__ chunk_meta,
                       const float* __restrict__ Bp, float* __restrict__ Cv, float* __restrict__ P,
                       int nchunks, int T, int k, int seg_blocks, int ldb, int stream_nt, int blocks_per_tile, const int* __restrict__ dyn) {
  group_walk<POLICY, false, (GCN_ABLATE & 128) == 0, BIG>(stream, nullptr, chunk_meta, Bp, Cv, P, nchunks, T, k, seg_blocks, ldb, stream_nt, blocks_per_tile, dyn);
}

// the same walk for matrices whose values do not factor: one fp32 value per entry beside the 16-bit stream,
// handed from the lane that loaded it to its group by the same DPP broadcast as the address (one more vector
// instruction and four FMAs instead of two packed adds per step); Bp is then a plain (unscaled) sliced copy of B
template <int POLICY, bool BIG>
__global__ void __launch_bounds__(256)
spmm_group_weighted_kernel(const unsigned short* __restrict__ stream, const float* __restrict__ vals,
                           const int2* __restrict__ chunk_meta, const float* __restrict__ Bp, float* __restrict__ Cv,
                           float* __restrict__ P, int nchunks, int T, int k, int seg_blocks, int ldb, int stream_nt, int blocks_per_tile,
                           const int* __restrict__ dyn) {
  group_walk<POLICY, true, (GCN_ABLATE & 256) != 0, BIG>(stream, vals, chunk_meta, Bp, Cv, P, nchunks, T, k, seg_blocks, ldb, stream_nt, blocks_per_tile, dyn);
}

// ------------------------------------------------------------------------------------------------------------
// k <= 32: EIGHT independent 8-lane row engines per wave (lane = g*8 + f, f = which float4 of the 32-column tile).
// A gather instruction then fetches eight feature rows of 128 bytes instead of four of 256, so a non-zero costs half
// the addresser time of the 64-column pass it would otherwise ride in with half its lanes idle.  Same stream, same
// lane-major runs of 64 entries (a lane of an 8-lane group reads its entries of the run's eight 8-entry blocks as two
// 8-byte words: u16 4f.. and 32+4f..), same chunk_meta / partial slab / fix list; a wave owns eight consecutive
// chunks.  The DPP broadcasts still work on rows of 16 lanes = two groups: the upper group of a row takes its
// entry from a copy rotated by eight lanes.  Two 8-entry blocks are in flight together (sixteen gathers).
// Non-temporal stores; value-free (with the LDS ring) or weighted.
template <int UU>
__device__ __forceinline__ int row_ror8_bcast(int v, int vrot, bool upper) {   // entry UU of THIS lane's 8-lane group
  const int lo = row_bcast<UU>(v), hi = row_bcast<UU>(vrot);
  return upper ? hi : lo;
}

template <bool RING, bool VALS, bool BIG>
__device__ __forceinline__ void
group8_walk(const unsigned short* __restrict__ stream, const float* __restrict__ vals, const int2* __restrict__ chunk_meta,
            const float* __restrict__ Bp, float* __restrict__ Cv, float* __restrict__ P,
            int nchunks, int T, int k, int ldb, int stream_nt, const int* __restrict__ dyn) {
  if (dyn) { if (dyn[0] == 0) return; nchunks = dyn[1]; }      // (as group_walk)
  const int lane = threadIdx.x & 63;
  const int wib  = threadIdx.x >> 6;
  const int g    = lane >> 3;
  const int f    = lane & 7;
  const bool upper = (lane & 8) != 0;                           // the upper group of its 16-lane DPP row
  const int per_xcd = nchunks >> 3;
  const int c_in = ((int)(blockIdx.x >> 3) * 4 + wib) * 8;
  if (c_in >= per_xcd) return;                                  // (whole wave: per_xcd % 8 == 0)
  const int c = (int)(blockIdx.x & 7) * per_xcd + c_in + g;     // this group's chunk

  const int fcol = f * 4;
  const bool fok = fcol < k;
  const unsigned row_bytes = (unsigned)ldb * 4u;
  const unsigned foff = (unsigned)(fok ? fcol : 0) * 4u;
  const char* Bb = reinterpret_cast<const char*>(Bp);
  const size_t kk = (size_t)k;

  const int2 meta = chunk_meta[c];
  const int vrow = meta.x >> 1;
  const bool head = meta.x & 1;
  const int base = BIG ? 0 : meta.y;
  if constexpr (BIG) Bb += (size_t)meta.y * (size_t)row_bytes;  // (as group_walk)
  float* ptr  = head ? P + (size_t)(2 * c) * kk + fcol : Cv + (size_t)vrow * kk + fcol;
  float* nptr = Cv + (size_t)(vrow + 1) * kk + fcol;
  bool first = true;
  // RING: finished rows wait in LDS, eight slots per group, and leave eight at a time — consecutive rows of ONE group,
  // written by the whole wave with one 64-lane store (as spmm_group_ring_kernel; here a row is 128 bytes)
  __shared__ f32x4 ring[RING ? 4 : 1][8][8][8];
  int ring_n = 0;
  float* ring_base = nullptr;
#define GCN_G8_DRAIN(G2, ROWS)                                                                      \
  {                                                                                                 \
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(uintptr_t)ring_base, 8 * G2);     \
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((uintptr_t)ring_base >> 32), 8 * G2); \
    float* b0 = reinterpret_cast<float*>(((uintptr_t)hi << 32) | lo);                              \
    const f32x4 rv = ring[wib][G2][lane >> 3][f];                                                   \
    if (fok && (lane >> 3) < (ROWS))                                                                \
      store_row_piece<2>(b0 + (size_t)(lane >> 3) * kk + f * 4, make_float4(rv.x, rv.y, rv.z, rv.w)); \
    if (g == G2) ring_n = 0;                                                                        \
  }

  typedef unsigned int u32x2_g8 __attribute__((ext_vector_type(2)));
  const u32x2_g8* __restrict__ sp = reinterpret_cast<const u32x2_g8*>(stream + (size_t)c * T);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // words of the current run: w0 = entries of the even 8-entry blocks (0, 2, 4, 6), w1 = of the odd ones
  u32x2_g8 w0 = stream_nt ? __builtin_nontemporal_load(sp + f) : sp[f];
  u32x2_g8 w1 = stream_nt ? __builtin_nontemporal_load(sp + 8 + f) : sp[8 + f];
  u32x2_g8 w0_nx = w0, w1_nx = w1;
  // VALS: the values of the same entries (slicing.hip lays them out like the stream): two 16-byte words per run
  const f32x4* __restrict__ vp = VALS ? reinterpret_cast<const f32x4*>(vals + (size_t)c * T) : nullptr;
  f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
  if constexpr (VALS) {
    v0 = stream_nt ? __builtin_nontemporal_load(vp + f) : vp[f];
    v1 = stream_nt ? __builtin_nontemporal_load(vp + 8 + f) : vp[8 + f];
  }
  f32x4 v0_nx = v0, v1_nx = v1;
  unsigned fl1 = 0;
#pragma unroll 1
  for (int d = 0; d < T / 16; ++d) {                            // two 8-entry blocks (2d, 2d+1 of the chunk) per turn
    const int j = d & 3;                                        // ... the j-th pair of the current run
    if (j == 0 && d + 4 < T / 16) {                             // the next run, a whole run ahead of its use
      const u32x2_g8* nx = sp + (d / 4 + 1) * 16;
      w0_nx = stream_nt ? __builtin_nontemporal_load(nx + f) : nx[f];
      w1_nx = stream_nt ? __builtin_nontemporal_load(nx + 8 + f) : nx[8 + f];
      if constexpr (VALS) {
        const f32x4* vnx = vp + (d / 4 + 1) * 16;
        v0_nx = stream_nt ? __builtin_nontemporal_load(vnx + f) : vnx[f];
        v1_nx = stream_nt ? __builtin_nontemporal_load(vnx + 8 + f) : vnx[8 + f];
      }
    }
    const unsigned e0 = ((j & 2 ? w0.y : w0.x) >> (16 * (j & 1))) & 0xFFFFu;
    const unsigned e1 = ((j & 2 ? w1.y : w1.x) >> (16 * (j & 1))) & 0xFFFFu;
    int vb0 = 0, vb1 = 0, vb0r = 0, vb1r = 0;                   // this lane's entries' values (bit patterns) and their rotated copies
    if constexpr (VALS) {
      vb0 = __builtin_bit_cast(int, j == 0 ? v0.x : j == 1 ? v0.y : j == 2 ? v0.z : v0.w);
      vb1 = __builtin_bit_cast(int, j == 0 ? v1.x : j == 1 ? v1.y : j == 2 ? v1.z : v1.w);
      vb0r = __builtin_amdgcn_mov_dpp(vb0, 0x128, 0xf, 0xf, true);
      vb1r = __builtin_amdgcn_mov_dpp(vb1, 0x128, 0xf, 0xf, true);
    }
    if (j == 3) { w0 = w0_nx; w1 = w1_nx; v0 = v0_nx; v1 = v1_nx; }
    const int ro0 = (int)(__umul24((e0 & 0x7FFFu) + (unsigned)base, row_bytes));
    const int ro1 = (int)(__umul24((e1 & 0x7FFFu) + (unsigned)base, row_bytes));
    const int ro0r = __builtin_amdgcn_mov_dpp(ro0, 0x128, 0xf, 0xf, true);   // row_ror:8 — lane l <- lane (l + 8) % 16 of its row
    const int ro1r = __builtin_amdgcn_mov_dpp(ro1, 0x128, 0xf, 0xf, true);
    const unsigned fl0 = e0 >> 15;
    fl1 = e1 >> 15;
    const int fl0r = __builtin_amdgcn_mov_dpp((int)fl0, 0x128, 0xf, 0xf, true);
    const int fl1r = __builtin_amdgcn_mov_dpp((int)fl1, 0x128, 0xf, 0xf, true);
    float4 b[16];
#define GCN_G8_ALL(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define GCN_G8_GATHER0(UU) \
    b[UU] = *reinterpret_cast<const float4*>(Bb + (size_t)((unsigned)row_ror8_bcast<UU>(ro0, ro0r, upper) + foff));
#define GCN_G8_GATHER1(UU) \
    b[8 + UU] = *reinterpret_cast<const float4*>(Bb + (size_t)((unsigned)row_ror8_bcast<UU>(ro1, ro1r, upper) + foff));
    GCN_G8_ALL(GCN_G8_GATHER0)
    GCN_G8_ALL(GCN_G8_GATHER1)
#undef GCN_G8_GATHER0
#undef GCN_G8_GATHER1
    const unsigned long long ends0 = __ballot(fl0 != 0);        // bit g*8+u: entry u of group g (first block) ends a row
    const unsigned long long ends1 = __ballot(fl1 != 0);
#define GCN_G8_ADDV(UU, I, VB, VBR)                                                                 \
      if constexpr (VALS) {                                                                         \
        const float vu = __builtin_bit_cast(float, row_ror8_bcast<UU>(VB, VBR, upper));             \
        acc.x = fmaf(vu, b[I].x, acc.x); acc.y = fmaf(vu, b[I].y, acc.y);                           \
        acc.z = fmaf(vu, b[I].z, acc.z); acc.w = fmaf(vu, b[I].w, acc.w);                           \
      } else { acc.x += b[I].x; acc.y += b[I].y; acc.z += b[I].z; acc.w += b[I].w; }
#define GCN_G8_ADD0(UU) GCN_G8_ADDV(UU, UU, vb0, vb0r)
#define GCN_G8_ADD1(UU) GCN_G8_ADDV(UU, 8 + UU, vb1, vb1r)
    if ((ends0 | ends1) == 0ull) {
      GCN_G8_ALL(GCN_G8_ADD0)
      GCN_G8_ALL(GCN_G8_ADD1)
    } else {
#define GCN_G8_STEP(UU, I, ENDS, FL, FLR, VB, VBR)                                                  \
      GCN_G8_ADDV(UU, I, VB, VBR)                                                                   \
      if (ENDS & (0x0101010101010101ull << UU)) {                /* some group ends a row here */    \
        if (row_ror8_bcast<UU>((int)FL, FLR, upper)) {                                              \
          if (RING && !(first && head) && ring_n < 8) {                                             \
            if (fok) ring[wib][g][ring_n][f] = f32x4{acc.x, acc.y, acc.z, acc.w};                   \
            if (ring_n == 0) ring_base = ptr;                                                       \
            ++ring_n;                                                                               \
          } else if (fok) store_row_piece<2>(ptr, acc);                                             \
          acc = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
          ptr = nptr; nptr += kk; first = false;                                                    \
        }                                                                                           \
      }
#define GCN_G8_STEP0(UU) GCN_G8_STEP(UU, UU, ends0, fl0, fl0r, vb0, vb0r)
#define GCN_G8_STEP1(UU) GCN_G8_STEP(UU, 8 + UU, ends1, fl1, fl1r, vb1, vb1r)
      GCN_G8_ALL(GCN_G8_STEP0)
      GCN_G8_ALL(GCN_G8_STEP1)
#undef GCN_G8_STEP0
#undef GCN_G8_STEP1
#undef GCN_G8_STEP
      if constexpr (RING) {
        const unsigned long long full = __ballot(ring_n == 8);
        if (full) {
          if (full & (1ull << 0))  GCN_G8_DRAIN(0, 8)
          if (full & (1ull << 8))  GCN_G8_DRAIN(1, 8)
          if (full & (1ull << 16)) GCN_G8_DRAIN(2, 8)
          if (full & (1ull << 24)) GCN_G8_DRAIN(3, 8)
          if (full & (1ull << 32)) GCN_G8_DRAIN(4, 8)
          if (full & (1ull << 40)) GCN_G8_DRAIN(5, 8)
          if (full & (1ull << 48)) GCN_G8_DRAIN(6, 8)
          if (full & (1ull << 56)) GCN_G8_DRAIN(7, 8)
        }
      }
    }
#undef GCN_G8_ADD1
#undef GCN_G8_ADD0
#undef GCN_G8_ADDV
#undef GCN_G8_ALL
  }
  if constexpr (RING) {                                         // what is left in the rings
    const unsigned long long some = __ballot(ring_n > 0);
    if (some & (1ull << 0))  GCN_G8_DRAIN(0, __builtin_amdgcn_readlane(ring_n, 0))
    if (some & (1ull << 8))  GCN_G8_DRAIN(1, __builtin_amdgcn_readlane(ring_n, 8))
    if (some & (1ull << 16)) GCN_G8_DRAIN(2, __builtin_amdgcn_readlane(ring_n, 16))
    if (some & (1ull << 24)) GCN_G8_DRAIN(3, __builtin_amdgcn_readlane(ring_n, 24))
    if (some & (1ull << 32)) GCN_G8_DRAIN(4, __builtin_amdgcn_readlane(ring_n, 32))
    if (some & (1ull << 40)) GCN_G8_DRAIN(5, __builtin_amdgcn_readlane(ring_n, 40))
    if (some & (1ull << 48)) GCN_G8_DRAIN(6, __builtin_amdgcn_readlane(ring_n, 48))
    if (some & (1ull << 56)) GCN_G8_DRAIN(7, __builtin_amdgcn_readlane(ring_n, 56))
  }
#undef GCN_G8_DRAIN
  // the row piece that sticks out of the chunk's end (the chunk's last entry — entry 7 of its last block — did not end its row)
  if (!row_ror8_bcast<7>((int)fl1, __builtin_amdgcn_mov_dpp((int)fl1, 0x128, 0xf, 0xf, true), upper)) {
    if (fok) store_row_piece<2>(ptr, acc);                        // (to Cv[row], or the chunk's head piece: as in group_walk)
  }
}

template <bool RING, bool BIG>
__global__ void __launch_bounds__(256)
spmm_group8_kernel(const unsigned short* __restrict__ stream, const int2* __restrict__ chunk_meta,
                   const float* __restrict__ Bp, float* __restrict__ Cv, float* __restrict__ P,
                   int nchunks, int T, int k, int ldb, int stream_nt, const int* __restrict__ dyn) {
  group8_walk<RING, false, BIG>(stream, nullptr, chunk_meta, Bp, Cv, P, nchunks, T, k, ldb, stream_nt, dyn);
}

// ... and with the values beside the stream (matrices whose values do not factor); no ring: 138 VGPRs without
template <bool BIG>
__global__ void __launch_bounds__(256)
spmm_group8_weighted_kernel(const unsigned short* __restrict__ stream, const float* __restrict__ vals,
                            const int2* __restrict__ chunk_meta, const float* __restrict__ Bp, float* __restrict__ Cv,
                            float* __restrict__ P, int nchunks, int T, int k, int ldb, int stream_nt, const int* __restrict__ dyn) {
  group8_walk<false, true, BIG>(stream, vals, chunk_meta, Bp, Cv, P, nchunks, T, k, ldb, stream_nt, dyn);
}

// ------------------------------------------------------------------------------------------------------------
// 33 <= k <= 48: FIVE 12-lane row engines per wave (lane = g*12 + f, f = which float4 of the 48-column row; lanes 60..63
// idle).  The addressers charge a gather instruction for its 64 lane addresses whatever they fetch (gather_x3_probe), so
// a row of 192 bytes occupies 12 lanes here, not 16 with four of them idle: five rows per instruction instead of four.
// (Worth 4-7 % of the whole SpMM, not the 20 % the instruction count suggests: a 192-byte row still arrives as two
// whole 128-byte lines, and at 64 B/clk per CU five rows of two lines cost the L2 -> L1 path 20 clocks where four cost
// 16 — with 16 lanes x 16 bytes the address rate and the line rate bind together)
// SAME stream as the 16-lane kernel (blocks of 16 entries, runs of 64 stored lane-major): lane f < 12 holds entry f of
// its group's block, lanes f < 4 hold entries 12..15 as well (a second 8-byte word per run); an entry reaches the
// group's lanes through ds_bpermute (a DPP row is 16 lanes wide and would straddle the groups).  Row ends: two ballots
// (entries 0..11 and 12..15).  Same chunk_meta, partial slab, cut lists and pieces; a wave owns five consecutive
// chunks, a block twenty.  Value-free pass, 32-bit slice bases, LDS ring of five rows per group (one 60-lane store).
__device__ __forceinline__ int lane_bcast(int src_lane_bytes, int v) { return __builtin_amdgcn_ds_bpermute(src_lane_bytes, v); }

__device__ __forceinline__ void
group12_walk(const unsigned short* __restrict__ stream, const int2* __restrict__ chunk_meta,
             const float* __restrict__ Bp, float* __restrict__ Cv, float* __restrict__ P,
             int nchunks, int T, int k, int ldb, int stream_nt, const int* __restrict__ dyn) {
  if (dyn) { if (dyn[0] == 0) return; nchunks = dyn[1]; }
  const int lane = threadIdx.x & 63;
  const int wib  = threadIdx.x >> 6;
  const int g    = lane / 12;                                   // 0..4; 5: the four spare lanes
  const int f    = lane - g * 12;
  const int per_xcd = nchunks >> 3;
  const int bx = (int)blockIdx.x;
  const int c_w = ((bx >> 3) * 4 + wib) * 5;                    // first chunk of this wave inside its XCD's range
  if (c_w >= per_xcd) return;                                   // (whole wave)
  // groups past the XCD's range and the spare lanes walk the wave's first chunk along (loads only, nothing stored)
  const bool live = g < 5 && c_w + g < per_xcd;
  const int c = (bx & 7) * per_xcd + (live ? c_w + g : c_w);

  const int fcol = f * 4;
  const bool fok = live && fcol < k;
  const unsigned row_bytes = (unsigned)ldb * 4u;
  const unsigned foff = (unsigned)(fcol < k ? fcol : 0) * 4u;
  const char* Bb = reinterpret_cast<const char*>(Bp);
  const size_t kk = (size_t)k;

  const int2 meta = chunk_meta[c];
  const int vrow = meta.x >> 1;
  const bool head = meta.x & 1;
  const int base = meta.y;
  float* ptr  = head ? P + (size_t)(2 * c) * kk + fcol : Cv + (size_t)vrow * kk + fcol;
  float* nptr = Cv + (size_t)(vrow + 1) * kk + fcol;
  bool first = true;
  __shared__ f32x4 ring[4][5][5][12];                           // [wave][group][slot][float4 of the row]
  int ring_n = 0;
  float* ring_base = nullptr;
  const int row_l = lane / 12;                                  // which ring row this lane writes out in a drain (== g)
#define GCN_G12_DRAIN(G2, ROWS)                                                                     \
  {                                                                                                 \
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(uintptr_t)ring_base, 12 * G2);    \
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((uintptr_t)ring_base >> 32), 12 * G2); \
    float* b0 = reinterpret_cast<float*>(((uintptr_t)hi << 32) | lo);                              \
    if (row_l < (ROWS) && fcol < k) {                                                               \
      const f32x4 rv = ring[wib][G2][row_l][f];                                                     \
      store_row_piece<2>(b0 + (size_t)row_l * kk + fcol, make_float4(rv.x, rv.y, rv.z, rv.w));      \
    }                                                                                               \
    if (g == G2) ring_n = 0;                                                                        \
  }

  typedef unsigned int u32x2_g __attribute__((ext_vector_type(2)));
  const u32x2_g* __restrict__ sp = reinterpret_cast<const u32x2_g*>(stream + (size_t)c * T);
  const int f2 = f < 4 ? 12 + f : f;                            // the second word of lanes 0..3: entries 12..15 (others: a copy)
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  u32x2_g eq = stream_nt ? __builtin_nontemporal_load(sp + f) : sp[f], eq_nx = eq;
  u32x2_g eq2 = stream_nt ? __builtin_nontemporal_load(sp + f2) : sp[f2], eq2_nx = eq2;
  const int srcA = g * 48;                                      // byte index of the group's lane 0 (ds_bpermute counts bytes)
  unsigned long long endsA = 0ull, endsB = 0ull;
#pragma unroll 1
  for (int blk = 0; blk < T / 16; ++blk) {
    const int j = blk & 3;
    if (j == 0 && blk + 4 < T / 16) {
      const int nx = (blk / 4 + 1) * 16;
      eq_nx = stream_nt ? __builtin_nontemporal_load(sp + nx + f) : sp[nx + f];
      eq2_nx = stream_nt ? __builtin_nontemporal_load(sp + nx + f2) : sp[nx + f2];
    }
    const unsigned e  = ((j & 2 ? eq.y : eq.x) >> (16 * (j & 1))) & 0xFFFFu;
    const unsigned e2 = ((j & 2 ? eq2.y : eq2.x) >> (16 * (j & 1))) & 0xFFFFu;
    if (j == 3) { eq = eq_nx; eq2 = eq2_nx; }
    const int rowoff  = (int)(__umul24((e & 0x7FFFu) + (unsigned)base, row_bytes));
    const int rowoff2 = (int)(__umul24((e2 & 0x7FFFu) + (unsigned)base, row_bytes));
    float4 b[16];
#define GCN_G12_ALL(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define GCN_G12_GATHER(UU) \
    b[UU] = *reinterpret_cast<const float4*>(Bb + (size_t)((unsigned)lane_bcast(srcA + 4 * (UU < 12 ? UU : UU - 12), UU < 12 ? rowoff : rowoff2) + foff));
    GCN_G12_ALL(GCN_G12_GATHER)
#undef GCN_G12_GATHER
    endsA = __ballot((e >> 15) != 0);                           // bit g*12+u: entry u < 12 of group g ends a row
    endsB = __ballot((e2 >> 15) != 0 && f < 4);                 // bit g*12+u-12: entry u >= 12
    if ((endsA | endsB) == 0ull) {
#define GCN_G12_ADD(UU) acc.x += b[UU].x; acc.y += b[UU].y; acc.z += b[UU].z; acc.w += b[UU].w;
      GCN_G12_ALL(GCN_G12_ADD)
    } else {
      const unsigned long long mineA = endsA >> (g < 5 ? g * 12 : 60);    // this lane's group's bits at 0..11
      const unsigned long long mineB = endsB >> (g < 5 ? g * 12 : 60);    // ... entries 12..15 at 0..3
#define GCN_G12_STEP(UU)                                                                            \
      GCN_G12_ADD(UU)                                                                               \
      if ((UU < 12 ? endsA : endsB) & (0x0001001001001001ull << (UU < 12 ? UU : UU - 12))) {        \
        if (((UU < 12 ? mineA : mineB) >> (UU < 12 ? UU : UU - 12)) & 1ull) {                       \
          if (!(first && head) && ring_n < 5) {                                                     \
            if (fok) ring[wib][g < 5 ? g : 0][ring_n][f] = f32x4{acc.x, acc.y, acc.z, acc.w};       \
            if (ring_n == 0) ring_base = ptr;                                                       \
            ++ring_n;                                                                               \
          } else if (fok) store_row_piece<2>(ptr, acc);                                             \
          acc = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
          ptr = nptr; nptr += kk; first = false;                                                    \
        }                                                                                           \
      }
      GCN_G12_ALL(GCN_G12_STEP)
#undef GCN_G12_STEP
#undef GCN_G12_ADD
      const unsigned long long full = __ballot(ring_n == 5 && live);
      if (full) {
        if (full & (1ull << 0))  GCN_G12_DRAIN(0, 5)
        if (full & (1ull << 12)) GCN_G12_DRAIN(1, 5)
        if (full & (1ull << 24)) GCN_G12_DRAIN(2, 5)
        if (full & (1ull << 36)) GCN_G12_DRAIN(3, 5)
        if (full & (1ull << 48)) GCN_G12_DRAIN(4, 5)
      }
    }
#undef GCN_G12_ALL
  }
  {
    const unsigned long long some = __ballot(ring_n > 0 && live);
    if (some & (1ull << 0))  GCN_G12_DRAIN(0, __builtin_amdgcn_readlane(ring_n, 0))
    if (some & (1ull << 12)) GCN_G12_DRAIN(1, __builtin_amdgcn_readlane(ring_n, 12))
    if (some & (1ull << 24)) GCN_G12_DRAIN(2, __builtin_amdgcn_readlane(ring_n, 24))
    if (some & (1ull << 36)) GCN_G12_DRAIN(3, __builtin_amdgcn_readlane(ring_n, 36))
    if (some & (1ull << 48)) GCN_G12_DRAIN(4, __builtin_amdgcn_readlane(ring_n, 48))
  }
#undef GCN_G12_DRAIN
  // the row piece that sticks out of the chunk's end: the chunk's last entry is entry 15 of its last block (lane 3's second word)
  if (!((endsB >> (g < 5 ? g * 12 + 3 : 63)) & 1ull)) {
    if (fok) store_row_piece<2>(ptr, acc);
  }
}

__global__ void __launch_bounds__(256)
spmm_group12_kernel(const unsigned short* __restrict__ stream, const int2* __restrict__ chunk_meta,
                    const float* __restrict__ Bp, float* __restrict__ Cv, float* __restrict__ P,
                    int nchunks, int T, int k, int ldb, int stream_nt, const int* __restrict__ dyn) {
  group12_walk(stream, chunk_meta, Bp, Cv, P, nchunks, T, k, ldb, stream_nt, dyn);
}

// 32-bit byte offsets (entry + slice base) * row_bytes reach every row of the sliced copy?  (__umul24: both factors
// below 2^24, and the product below 2^32.)  Otherwise the BIG variants add the slice base in 64 bits.
bool spmm_group_needs_big(long long table_rows, int ldb) {
  // GCN_AMD_GROUP_BIG=1 (development / tests): the 64-bit variants whatever the size
  static const bool forced = [] { const char* e = getenv("GCN_AMD_GROUP_BIG"); return e && e[0] == '1'; }();
  return forced || table_rows >= (1LL << 24) || table_rows * (long long)ldb * 4 >= (1LL << 32);
}

// table_rows = rows of the sliced copy the kernel gathers from, S * (w + 1) (0: unknown / not checked)
bool spmm_group_eligible(int k, int ldb, long long table_rows, const void* B, const void* C, const void* P) {
  const uintptr_t al = (uintptr_t)B | (uintptr_t)C | (uintptr_t)P;
  if (ldb <= 0) ldb = k;
  if (!(k % 4 == 0 && ldb % 4 == 0 && (al & 15) == 0 && ldb * 4 < (1 << 24))) return false;
  // BIG: the offset inside a slice (< 32 768 rows) must still fit 32 bits
  return !spmm_group_needs_big(table_rows, ldb) || ldb * 4 < (1 << 17);
}

// 33 <= k <= 48, value-free, 32-bit slice bases: the five-engine kernel (GCN_AMD_GROUP12=0: the 64-column pass)
bool spmm_group12_applies(const GroupArgs& a) {
  static const bool on = [] { const char* e = getenv("GCN_AMD_GROUP12"); return !e || e[0] != '0'; }();
  const int ldb = a.ldb > 0 ? a.ldb : a.k;
  return on && a.narrow12 && !a.vals && a.k > 32 && a.k <= 48 && a.k % 4 == 0 && !spmm_group_needs_big(a.table_rows, ldb);
}

// k <= 32, whole waves of eight chunks per XCD: the eight-engine kernels take the launch
bool spmm_group8_applies(const GroupArgs& a) {
  return a.narrow8 && a.k <= 32 && a.k % 4 == 0 && a.nchunks % 64 == 0;
}

namespace {

template <bool BIG>
hipError_t launch_group8_t(const GroupArgs& a, int ldb, hipStream_t s) {
  const int per_xcd = a.nchunks / 8;
  const int stream_nt8 = (size_t)a.nchunks * (size_t)a.T * (a.vals ? 6u : 2u) > ((size_t)64 << 20) ? 1 : 0;
  const int nb8 = 8 * ((per_xcd + 31) / 32);
  const int2* meta = reinterpret_cast<const int2*>(a.chunk_meta);
  if (a.vals) spmm_group8_weighted_kernel<BIG><<<dim3(nb8), dim3(256), 0, s>>>(a.stream, a.vals, meta, a.Bp, a.Cv, a.P, a.nchunks, a.T, a.k, ldb, stream_nt8, a.dyn);
  else        spmm_group8_kernel<true, BIG><<<dim3(nb8), dim3(256), 0, s>>>(a.stream, meta, a.Bp, a.Cv, a.P, a.nchunks, a.T, a.k, ldb, stream_nt8, a.dyn);
  return hipGetLastError();
}

// partial rows leave with non-temporal stores (POLICY 2): sc1 2.86 / 2.74 / 2.76 ms, plain 2.86 / 2.75 / 2.71, nt 2.82 / 2.66 /
// 2.53 at 8 / 12 / 16 slices (profiles/r02z4_store_policy.log) — the only policy instantiated
template <bool BIG>
hipError_t launch_group_t(const GroupArgs& a, int ldb, hipStream_t s) {
  const int per_xcd = a.nchunks / 8;
  int nblocks = 8 * ((per_xcd + 15) / 16);
  const int tiles = (a.k + 63) / 64;
  // streams (2 or 6 bytes per entry) beyond what the L2s and a good part of the Infinity Cache hold are read non-temporally
  const int stream_nt = (size_t)a.nchunks * (size_t)a.T * (a.vals ? 6u : 2u) > ((size_t)64 << 20) ? 1 : 0;
  // all tiles in ONE launch: tile t+1 starts on the CUs that tile t's last blocks leave idle (k = 128 / 256: 2.89 / 5.70 ->
  // 2.87 / 5.66 ms, profiles/r02zzb_merged_tile_launch.log)
  if ((long long)nblocks * tiles >= (1LL << 31)) return hipErrorInvalidValue;
  const int blocks_per_tile = nblocks;
  nblocks *= tiles;
  const int2* meta = reinterpret_cast<const int2*>(a.chunk_meta);
  // Order of the (tile, block) pairs inside the launch (r04).  Tile-major — all of tile 0, then all of tile 1 — reads the
  // whole stream (2 bytes per entry, 6 with values) once per tile from HBM: by the time tile 1 starts, tile 0's stream has
  // long left the 256 MiB Infinity Cache.  In SEGMENTS — every XCD's blocks cut into nseg runs, run 0 for tile 0, run 0 for
  // tile 1, ..., then run 1 — a run's stream is read again while it still sits there.  The price is a switch of the XCD's
  // L2-resident table slice at every run, so as few runs as keep one run's stream (all XCDs together) near half the cache:
  // nseg = ceil(stream bytes / 120 MB); measured (profiles/r04k_*, r04l_*; Reddit-shaped, whole SpMM): value-free (230 MB)
  // k = 128: 2.845 -> 2.680 ms at 2 runs (3 / 5 / 8 / 12 runs: 2.73 / 2.75 / 2.82 / 2.92), k = 512: 11.52 -> 10.89;
  // weighted (689 MB) k = 128: 3.135 -> 3.043 at 6 runs (2 / 8: 3.19 / 3.08); half-size graph: value-free (57 MB) stays
  // tile-major, weighted (172 MB) 1.452 -> 1.382 at 2.  GCN_AMD_GROUP_SEGMENTS (development) overrides nseg; 1 = tile-major.
  static const int forced_seg = [] { const char* e = getenv("GCN_AMD_GROUP_SEGMENTS"); return e ? atoi(e) : 0; }();
  const size_t stream_bytes = (size_t)a.nchunks * (size_t)a.T * (a.vals ? 6u : 2u);
  const int nseg = forced_seg > 0 ? forced_seg : (int)((stream_bytes + ((size_t)120 << 20) - 1) / ((size_t)120 << 20));
  const int nbx = blocks_per_tile / 8;
  const int seg_blocks = (nseg > 1 && tiles > 1 && nbx > 1) ? (nbx + nseg - 1) / nseg : 0;      // 0: tile-major
  if (a.vals)
    spmm_group_weighted_kernel<2, BIG><<<dim3(nblocks), dim3(256), 0, s>>>(a.stream, a.vals, meta, a.Bp, a.Cv, a.P, a.nchunks, a.T, a.k, seg_blocks, ldb, stream_nt, blocks_per_tile, a.dyn);
  else
    spmm_group_ring_kernel<2, BIG><<<dim3(nblocks), dim3(256), 0, s>>>(a.stream, meta, a.Bp, a.Cv, a.P, a.nchunks, a.T, a.k, seg_blocks, ldb, stream_nt, blocks_per_tile, a.dyn);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_spmm_group(const GroupArgs& a, hipStream_t s) {
  if (a.nchunks <= 0 || a.k <= 0) return hipSuccess;
  if (a.nchunks % 32 != 0 || a.k % 4 != 0 || a.T < 64 || a.T % 64 != 0) return hipErrorInvalidValue;
  const int ldb = a.ldb > 0 ? a.ldb : a.k;
  if (a.table_rows <= 0) return hipErrorInvalidValue;                 // (the addressing mode depends on it)
  const bool big = spmm_group_needs_big(a.table_rows, ldb);
  if (big && ldb * 4 >= (1 << 17)) return hipErrorInvalidValue;
  if (spmm_group8_applies(a)) return big ? launch_group8_t<true>(a, ldb, s) : launch_group8_t<false>(a, ldb, s);
  if (spmm_group12_applies(a)) {
    const int per_xcd = a.nchunks / 8;
    const int nb12 = 8 * ((per_xcd + 19) / 20);
    const int stream_nt12 = (size_t)a.nchunks * (size_t)a.T * 2u > ((size_t)64 << 20) ? 1 : 0;
    spmm_group12_kernel<<<dim3(nb12), dim3(256), 0, s>>>(a.stream, reinterpret_cast<const int2*>(a.chunk_meta), a.Bp, a.Cv, a.P,
                                                         a.nchunks, a.T, a.k, ldb, stream_nt12, a.dyn);
    return hipGetLastError();
  }
  return big ? launch_group_t<true>(a, ldb, s) : launch_group_t<false>(a, ldb, s);
}

}  // namespace gcn
