/* to_simd.c — SSE2 BitPacker4x decode for the CPU BASELINE leg of bench.py (test infrastructure:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use anything under oracle/).
 *
 * The reference decodes posting blocks with the `bitpacking` crate's SSE3/SSE2 BitPacker4x
 * (Cargo.toml:42-44; call sites src/postings/compression/mod.rs:105-150): the 4-lane vertical
 * layout exists so that one 128-bit register holds one value of each of the four bit streams.
 * The scalar restatement in to_codec.c unpacks one value at a time through byte loads; timing
 * tantivy's algorithms with it under-states the reference's CPU path.  This file is the same
 * decode written the way the layout asks for — one __m128i per 4 values, variable shifts, the
 * strict-delta prefix sum as two shifted adds per register plus a carried broadcast — and is
 * switched in by to_set_simd(1) for the baseline runs (bench.py reports both: qps_scalar,
 * qps_simd).  tests/test_oracle_kat.py checks it value for value against the scalar decode for
 * every bit width, both delta flavours and random payloads. */
#include <emmintrin.h>
#include <string.h>

#include "tantivy_oracle.h"

static inline __m128i unpack_reg(const uint8_t *in, uint32_t k, uint32_t b, __m128i mask) {
  const uint32_t p = k * b, w = p >> 5, s = p & 31u;
  __m128i x = _mm_srl_epi32(_mm_loadu_si128((const __m128i *)(in + 16u * w)), _mm_cvtsi32_si128((int)s));
  if (s + b > 32u) {
    const __m128i hi = _mm_loadu_si128((const __m128i *)(in + 16u * (w + 1u)));
    x = _mm_or_si128(x, _mm_sll_epi32(hi, _mm_cvtsi32_si128((int)(32u - s))));
  }
  return _mm_and_si128(x, mask);
}

size_t to_simd_bp4_decompress(const uint8_t *in, uint32_t *out, uint8_t b) {
  if (b == 0) {
    memset(out, 0, TO_BLOCK_LEN * sizeof(uint32_t));
    return 0;
  }
  const __m128i mask = _mm_set1_epi32((int)(b >= 32 ? 0xFFFFFFFFu : ((1u << b) - 1u)));
  for (uint32_t k = 0; k < 32u; k++)
    _mm_storeu_si128((__m128i *)(out + 4u * k), unpack_reg(in, k, b, mask));
  return (size_t)b * 16u;
}

/* out[i] = prev = prev + d[i] + add (wrapping), add = 1 for strict deltas, 0 for plain ones */
size_t to_simd_bp4_decompress_delta(uint32_t seed, uint32_t add, const uint8_t *in, uint32_t *out,
                                    uint8_t b) {
  const __m128i mask = _mm_set1_epi32((int)(b >= 32 ? 0xFFFFFFFFu : ((1u << b) - 1u)));
  const __m128i inc = _mm_set1_epi32((int)add);
  __m128i carry = _mm_set1_epi32((int)seed);
  for (uint32_t k = 0; k < 32u; k++) {
    __m128i x = b ? unpack_reg(in, k, b, mask) : _mm_setzero_si128();
    x = _mm_add_epi32(x, inc);
    x = _mm_add_epi32(x, _mm_slli_si128(x, 4));
    x = _mm_add_epi32(x, _mm_slli_si128(x, 8));
    x = _mm_add_epi32(x, carry);
    _mm_storeu_si128((__m128i *)(out + 4u * k), x);
    carry = _mm_shuffle_epi32(x, 0xFF);
  }
  return (size_t)b * 16u;
}
