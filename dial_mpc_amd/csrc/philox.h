// philox.h -- counter-based RNG for the in-kernel noise of K1 ("keyed by (seed, anneal
// iteration, sample, node, dim) so any rank can regenerate any sample's noise").
// Philox4x32-10 (Salmon et al., SC'11) + Box-Muller.  One call yields 4 standard normals for the element quad
// q of sample n:  counter = (n, q, iteration, 0), key = (seed_lo, seed_hi).
// The reference draws jax.random.normal (threefry, version dependent): bit parity with JAX is neither possible
// nor attempted -- for parity runs the noise crosses the boundary as data (`eps`), and dial_rng_fill()
// materialises exactly the noise the kernels generate.
#pragma once
#include <stdint.h>
#include "wave.h"

namespace dial {

DIAL_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                            uint32_t* out) {
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 4 standard normals for (sample n, quad q, iteration it)
DIAL_DEV void normal_quad(uint32_t n, uint32_t q, uint32_t it, uint32_t seed_lo, uint32_t seed_hi, float* z) {
  uint32_t u[4];
  philox4x32_10(n, q, it, 0u, seed_lo, seed_hi, u);
  for (int h = 0; h < 2; h++) {
    const float u1 = ((float)(u[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1)
    const float u2 = ((float)(u[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
#ifdef DIAL_EMU
    const float r = std::sqrt(-2.0f * std::log(u1));
    z[2 * h] = r * std::cos(6.283185307179586f * u2);
    z[2 * h + 1] = r * std::sin(6.283185307179586f * u2);
#else
    const float r = sqrtf(-2.0f * logf(u1));
    z[2 * h] = r * cosf(6.283185307179586f * u2);
    z[2 * h + 1] = r * sinf(6.283185307179586f * u2);
#endif
  }
}

}  // namespace dial
