// philox.h — counter-based random bits for the dropout mask of the fused epilogue (the
// reference applies torch's F.dropout after the layer, pygcn/gcn6.py:245-246).  Philox4x32-10 (Salmon et
// al., SC'11): the mask of output element i is a pure function of (seed, offset, i), so the forward pass,
// the backward pass and every kernel family that applies it agree without storing the mask.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gcn {

struct DropoutSpec {
  float p = 0.f;                       // drop probability in [0, 1); 0 = no dropout
  unsigned long long seed = 0;         // key
  unsigned long long offset = 0;       // stream position (so that successive layers / iterations differ)
  __host__ __device__ bool on() const { return p > 0.f; }
};

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0; k.y += W1;
  }
  return c;
}

// the four 32-bit words that decide elements 4j .. 4j+3
__device__ __forceinline__ uint4 dropout_words(const DropoutSpec& d, unsigned long long j) {
  return philox4x32_10(make_uint4((uint32_t)j, (uint32_t)(j >> 32), (uint32_t)d.offset, (uint32_t)(d.offset >> 32)),
                       make_uint2((uint32_t)d.seed, (uint32_t)(d.seed >> 32)));
}

__device__ __forceinline__ uint32_t dropout_threshold(float p) {        // keep iff word >= threshold
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}

// v scaled by 1/(1-p) if element `idx` is kept, else 0
__device__ __forceinline__ float dropout_apply(const DropoutSpec& d, unsigned long long idx, float v) {
  const uint4 w = dropout_words(d, idx >> 2);
  const uint32_t words[4] = {w.x, w.y, w.z, w.w};
  return words[idx & 3] >= dropout_threshold(d.p) ? v * (1.f / (1.f - d.p)) : 0.f;
}

// four consecutive elements starting at idx (idx % 4 == 0): one Philox call
__device__ __forceinline__ float4 dropout_apply4(const DropoutSpec& d, unsigned long long idx, float4 v) {
  const uint4 w = dropout_words(d, idx >> 2);
  const uint32_t t = dropout_threshold(d.p);
  const float s = 1.f / (1.f - d.p);
  return make_float4(w.x >= t ? v.x * s : 0.f, w.y >= t ? v.y * s : 0.f, w.z >= t ? v.z * s : 0.f, w.w >= t ? v.w * s : 0.f);
}

}  // namespace gcn
