// Counter-based random numbers for the training-time kernels (dither, SpecAugment / SpecCutout geometry, dropout):
// Philox4x32-10 (Salmon et al., SC'11).  A value is a pure function of (seed, stream, counter), so a kernel needs no
// generator state, a backward pass can RE-DRAW the forward's mask instead of storing it, and a hipGraph replay only needs
// a new seed word in device memory.  oracle/philox.py restates exactly this arithmetic; tests compare the raw 32-bit
// words bit for bit.
#pragma once
#include <stdint.h>

namespace ts {

struct Philox4 { uint32_t v[4]; };

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  return Philox4{{c0, c1, c2, c3}};
}

// streams: which consumer a counter belongs to (third counter word), so that one seed never hands two kernels the same words
enum : uint32_t { PHILOX_DITHER = 1, PHILOX_SPEC = 2, PHILOX_DROPOUT = 3 };

__host__ __device__ inline Philox4 philox(uint64_t seed, uint32_t stream, uint64_t counter) {
  return philox4x32_10((uint32_t)counter, (uint32_t)(counter >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// uniform in [0, 1) with 24 random bits (every value is exactly representable in f32)
__host__ __device__ inline float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

// standard normal pair by Box-Muller from two words: u1 in (0, 1], u2 in [0, 1)
__device__ inline void normal2(uint32_t x0, uint32_t x1, float& n0, float& n1) {
  const float u1 = (float)((x0 >> 8) + 1u) * 5.9604644775390625e-08f;
  const float r = sqrtf(-2.0f * logf(u1));
  float s, c;
  sincospif(2.0f * u01(x1), &s, &c);
  n0 = r * c;
  n1 = r * s;
}

}  // namespace ts
