/* to_codec.c — CPU ORACLE (test infrastructure): integer codecs, fieldnorm table, BM25.
 * See tantivy_oracle.h for the scope / pinning statement. */
#include "tantivy_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ to_buf */
void to_buf_init(to_buf *b) {
  b->data = NULL;
  b->len = b->cap = 0;
}
void to_buf_free(to_buf *b) {
  free(b->data);
  b->data = NULL;
  b->len = b->cap = 0;
}
static void to_buf_reserve(to_buf *b, size_t extra) {
  if (b->len + extra <= b->cap) return;
  size_t cap = b->cap ? b->cap : 256;
  while (cap < b->len + extra) cap *= 2;
  b->data = (uint8_t *)realloc(b->data, cap);
  b->cap = cap;
}
void to_buf_push(to_buf *b, const void *src, size_t n) {
  to_buf_reserve(b, n);
  if (n) memcpy(b->data + b->len, src, n);
  b->len += n;
}
void to_buf_push_u8(to_buf *b, uint8_t v) { to_buf_push(b, &v, 1); }
void to_buf_push_u32(to_buf *b, uint32_t v) {
  uint8_t le[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
  to_buf_push(b, le, 4);
}
void to_buf_clear(to_buf *b) { b->len = 0; }

/* ------------------------------------------------------------------ fieldnorm
 * src/fieldnorm/code.rs:13-270 holds the table as a literal; its own unit test
 * (code.rs:297-328, decode_fieldnorm_byte) gives the generating rule used here. */
static uint32_t g_fieldnorm_table[256];
static int g_fieldnorm_init = 0;

static uint32_t decode_field_norm_exp_part(uint8_t b) {
  uint32_t bits = (uint32_t)(b & 7u);
  uint8_t shift = b >> 3;
  if (shift == 0) return bits;
  return (bits | 8u) << (uint32_t)(shift - 1u);
}
static void fieldnorm_init(void) {
  if (g_fieldnorm_init) return;
  for (int i = 0; i < 256; i++) {
    uint8_t b = (uint8_t)i;
    g_fieldnorm_table[i] = (b < 24u) ? (uint32_t)b : 24u + decode_field_norm_exp_part(b - 24u);
  }
  g_fieldnorm_init = 1;
}
const uint32_t *to_fieldnorm_table(void) {
  fieldnorm_init();
  return g_fieldnorm_table;
}
uint32_t to_id_to_fieldnorm(uint8_t id) {
  fieldnorm_init();
  return g_fieldnorm_table[id];
}
/* code.rs:7-11: binary_search(...).unwrap_or_else(|idx| idx - 1) */
uint8_t to_fieldnorm_to_id(uint32_t fieldnorm) {
  fieldnorm_init();
  int lo = 0, hi = 256; /* first index with table[idx] > fieldnorm */
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    if (g_fieldnorm_table[mid] <= fieldnorm)
      lo = mid + 1;
    else
      hi = mid;
  }
  return (uint8_t)(lo - 1);
}

/* ------------------------------------------------------------------ common VInt
 * common/src/vint.rs:61-112 — 7-bit little-endian groups, STOP bit (0x80) on the last byte. */
size_t to_vint_serialize(uint64_t v, uint8_t *out) {
  size_t n = 0;
  for (;;) {
    uint8_t byte = (uint8_t)(v % 128u);
    v /= 128u;
    if (v == 0) {
      out[n++] = byte | 128u;
      return n;
    }
    out[n++] = byte;
  }
}
size_t to_vint_deserialize(const uint8_t *data, size_t len, uint64_t *out) {
  uint64_t result = 0;
  unsigned shift = 0;
  for (size_t i = 0; i < len; i++) {
    uint8_t b = data[i];
    result |= (uint64_t)(b % 128u) << shift;
    if (b >= 128u) {
      *out = result;
      return i + 1;
    }
    shift += 7;
  }
  return 0;
}

/* ------------------------------------------------------------------ postings vint blocks
 * src/postings/compression/vint.rs:1-108 */
size_t to_vint_compress_sorted(const uint32_t *in, size_t n, uint8_t *out, uint32_t offset) {
  size_t w = 0;
  for (size_t i = 0; i < n; i++) {
    uint32_t to_encode = in[i] - offset;
    offset = in[i];
    for (;;) {
      uint8_t next = (uint8_t)(to_encode % 128u);
      to_encode /= 128u;
      if (to_encode == 0) {
        out[w++] = next | 128u;
        break;
      }
      out[w++] = next;
    }
  }
  return w;
}
size_t to_vint_compress_unsorted(const uint32_t *in, size_t n, uint8_t *out) {
  size_t w = 0;
  for (size_t i = 0; i < n; i++) {
    uint32_t to_encode = in[i];
    for (;;) {
      uint8_t next = (uint8_t)(to_encode % 128u);
      to_encode /= 128u;
      if (to_encode == 0) {
        out[w++] = next | 128u;
        break;
      }
      out[w++] = next;
    }
  }
  return w;
}
size_t to_vint_uncompress_sorted(const uint8_t *data, uint32_t *out, size_t n, uint32_t offset) {
  size_t r = 0;
  uint32_t result = offset;
  for (size_t i = 0; i < n; i++) {
    uint32_t shift = 0;
    for (;;) {
      uint8_t cur = data[r++];
      result += (uint32_t)(cur % 128u) << shift;
      if (cur & 128u) break;
      shift += 7;
    }
    out[i] = result;
  }
  return r;
}
size_t to_vint_uncompress_unsorted(const uint8_t *data, uint32_t *out, size_t n) {
  size_t r = 0;
  for (size_t i = 0; i < n; i++) {
    uint32_t result = 0, shift = 0;
    for (;;) {
      uint8_t cur = data[r++];
      result += (uint32_t)(cur % 128u) << shift;
      if (cur & 128u) break;
      shift += 7;
    }
    out[i] = result;
  }
  return r;
}
size_t to_vint_uncompress_unsorted_until_end(const uint8_t *data, size_t len, uint32_t *out,
                                             size_t out_cap) {
  size_t r = 0;
  for (size_t i = 0; i < out_cap; i++) {
    if (r == len) return i;
    uint32_t result = 0, shift = 0;
    for (;;) {
      uint8_t cur = data[r++];
      result += (uint32_t)(cur % 128u) << shift;
      if (cur & 128u) break;
      shift += 7;
    }
    out[i] = result;
  }
  return out_cap;
}

/* ------------------------------------------------------------------ BitPacker4x
 * Third-party crate bitpacking 0.9.3 (not in the reference tree).  Published layout:
 * 128 values = 32 "registers" of 4 u32 lanes; lane l is an independent little-endian bit stream
 * of v[l], v[4+l], ...; output 128-bit word w = word w of lanes 0..3. */
static uint32_t rd32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static void wr32(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)v;
  p[1] = (uint8_t)(v >> 8);
  p[2] = (uint8_t)(v >> 16);
  p[3] = (uint8_t)(v >> 24);
}
static uint8_t bits_of(uint32_t x) {
  uint8_t n = 0;
  while (x) {
    n++;
    x >>= 1;
  }
  return n;
}
uint8_t to_bp4_num_bits(const uint32_t *v) {
  uint32_t acc = 0;
  for (int i = 0; i < TO_BLOCK_LEN; i++) acc |= v[i];
  return bits_of(acc);
}
uint8_t to_bp4_num_bits_strictly_sorted(int has_initial, uint32_t initial, const uint32_t *v) {
  uint32_t prev = has_initial ? initial : 0xFFFFFFFFu, acc = 0;
  for (int i = 0; i < TO_BLOCK_LEN; i++) {
    acc |= v[i] - prev - 1u; /* wrapping */
    prev = v[i];
  }
  return bits_of(acc);
}
uint8_t to_bp4_num_bits_sorted(uint32_t initial, const uint32_t *v) {
  uint32_t prev = initial, acc = 0;
  for (int i = 0; i < TO_BLOCK_LEN; i++) {
    acc |= v[i] - prev;
    prev = v[i];
  }
  return bits_of(acc);
}
size_t to_bp4_compress(const uint32_t *v, uint8_t *out, uint8_t b) {
  size_t nbytes = (size_t)b * 16u;
  if (b == 0) return 0;
  uint32_t W[32][4];
  memset(W, 0, sizeof W);
  uint32_t mask = (b >= 32) ? 0xFFFFFFFFu : ((1u << b) - 1u);
  for (int i = 0; i < TO_BLOCK_LEN; i++) {
    int l = i & 3, k = i >> 2;
    uint32_t p = (uint32_t)k * b, w = p >> 5, s = p & 31u;
    uint32_t val = v[i] & mask;
    W[w][l] |= val << s;
    if (s + b > 32u) W[w + 1][l] |= val >> (32u - s);
  }
  for (uint32_t w = 0; w < b; w++)
    for (int l = 0; l < 4; l++) wr32(out + 16u * w + 4u * (uint32_t)l, W[w][l]);
  return nbytes;
}
/* baseline runs only: the SSE2 decode of to_simd.c instead of the scalar loops below */
static int g_simd = 0;
void to_set_simd(int on) { g_simd = on != 0; }
int to_get_simd(void) { return g_simd; }

size_t to_bp4_decompress(const uint8_t *in, uint32_t *out, uint8_t b) {
  if (g_simd) return to_simd_bp4_decompress(in, out, b);
  if (b == 0) {
    memset(out, 0, TO_BLOCK_LEN * sizeof(uint32_t));
    return 0;
  }
  uint32_t mask = (b >= 32) ? 0xFFFFFFFFu : ((1u << b) - 1u);
  for (int i = 0; i < TO_BLOCK_LEN; i++) {
    int l = i & 3, k = i >> 2;
    uint32_t p = (uint32_t)k * b, w = p >> 5, s = p & 31u;
    uint32_t x = rd32(in + 16u * w + 4u * (uint32_t)l) >> s;
    if (s + b > 32u) x |= rd32(in + 16u * (w + 1u) + 4u * (uint32_t)l) << (32u - s);
    out[i] = x & mask;
  }
  return (size_t)b * 16u;
}
size_t to_bp4_compress_strictly_sorted(int has_initial, uint32_t initial, const uint32_t *v,
                                       uint8_t *out, uint8_t b) {
  uint32_t d[TO_BLOCK_LEN];
  uint32_t prev = has_initial ? initial : 0xFFFFFFFFu;
  for (int i = 0; i < TO_BLOCK_LEN; i++) {
    d[i] = v[i] - prev - 1u;
    prev = v[i];
  }
  return to_bp4_compress(d, out, b);
}
size_t to_bp4_decompress_strictly_sorted(int has_initial, uint32_t initial, const uint8_t *in,
                                         uint32_t *out, uint8_t b) {
  if (g_simd) return to_simd_bp4_decompress_delta(has_initial ? initial : 0xFFFFFFFFu, 1u, in, out, b);
  size_t n = to_bp4_decompress(in, out, b);
  uint32_t prev = has_initial ? initial : 0xFFFFFFFFu;
  for (int i = 0; i < TO_BLOCK_LEN; i++) {
    prev = prev + out[i] + 1u;
    out[i] = prev;
  }
  return n;
}
size_t to_bp4_compress_sorted(uint32_t initial, const uint32_t *v, uint8_t *out, uint8_t b) {
  uint32_t d[TO_BLOCK_LEN];
  uint32_t prev = initial;
  for (int i = 0; i < TO_BLOCK_LEN; i++) {
    d[i] = v[i] - prev;
    prev = v[i];
  }
  return to_bp4_compress(d, out, b);
}
size_t to_bp4_decompress_sorted(uint32_t initial, const uint8_t *in, uint32_t *out, uint8_t b) {
  if (g_simd) return to_simd_bp4_decompress_delta(initial, 0u, in, out, b);
  size_t n = to_bp4_decompress(in, out, b);
  uint32_t prev = initial;
  for (int i = 0; i < TO_BLOCK_LEN; i++) {
    prev += out[i];
    out[i] = prev;
  }
  return n;
}

/* ------------------------------------------------------------------ BlockEncoder/Decoder
 * src/postings/compression/mod.rs */
uint8_t to_compress_block_sorted(const uint32_t *block, uint32_t offset, uint8_t *out,
                                 size_t *out_len) {
  /* mod.rs:36-45: offset 0 <-> None */
  int has = offset != 0;
  uint8_t nb = to_bp4_num_bits_strictly_sorted(has, offset, block);
  *out_len = to_bp4_compress_strictly_sorted(has, offset, block, out, nb);
  return nb;
}
uint8_t to_compress_block_unsorted(const uint32_t *block, int minus_one_encoded, uint8_t *out,
                                   size_t *out_len) {
  /* mod.rs:54-75 */
  uint32_t tmp[TO_BLOCK_LEN];
  const uint32_t *src = block;
  if (minus_one_encoded) {
    for (int i = 0; i < TO_BLOCK_LEN; i++) tmp[i] = block[i] - 1u;
    src = tmp;
  }
  uint8_t nb = to_bp4_num_bits(src);
  *out_len = to_bp4_compress(src, out, nb);
  return nb;
}
size_t to_uncompress_block_sorted(const uint8_t *data, uint32_t offset, uint8_t num_bits,
                                  int strict_delta, uint32_t *out) {
  /* mod.rs:105-127 */
  if (strict_delta) return to_bp4_decompress_strictly_sorted(offset != 0, offset, data, out, num_bits);
  return to_bp4_decompress_sorted(offset, data, out, num_bits);
}
size_t to_uncompress_block_unsorted(const uint8_t *data, uint8_t num_bits, int minus_one_encoded,
                                    uint32_t *out) {
  /* mod.rs:134-150 */
  size_t n = to_bp4_decompress(data, out, num_bits);
  if (minus_one_encoded)
    for (int i = 0; i < TO_BLOCK_LEN; i++) out[i] += 1u;
  return n;
}
/* src/postings/block_search.rs:38-76 — result contract only (first idx with arr[idx] >= target);
 * the reference's branchless 8-ary layout is a CPU micro-optimisation. */
size_t to_search_block(const uint32_t *arr, uint32_t target) {
  size_t lo = 0, hi = TO_BLOCK_LEN;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (arr[mid] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

/* ------------------------------------------------------------------ BM25
 * src/query/bm25.rs:8-9,52-69,158-193.  All arithmetic in f32, no contraction. */
static const float K1 = 1.2f;
static const float Bp = 0.75f;

float to_idf(uint64_t doc_freq, uint64_t doc_count) {
  float x = ((float)(doc_count - doc_freq) + 0.5f) / ((float)doc_freq + 0.5f);
  return logf(1.0f + x);
}
static float cached_tf_component(uint32_t fieldnorm, float average_fieldnorm) {
  return K1 * (1.0f - Bp + Bp * (float)fieldnorm / average_fieldnorm);
}
void to_bm25_new(to_bm25 *w, float idf, float average_fieldnorm) {
  w->weight = idf * (1.0f + K1);
  w->average_fieldnorm = average_fieldnorm;
  for (int id = 0; id < 256; id++)
    w->cache[id] = cached_tf_component(to_id_to_fieldnorm((uint8_t)id), average_fieldnorm);
}
void to_bm25_for_one_term(to_bm25 *w, uint64_t term_doc_freq, uint64_t total_num_docs,
                          float avg_fieldnorm) {
  to_bm25_new(w, to_idf(term_doc_freq, total_num_docs), avg_fieldnorm);
}
void to_bm25_boost_by(to_bm25 *w, float boost) {
  if (boost == 1.0f) return;
  w->weight = w->weight * boost;
}
float to_bm25_tf_factor(const to_bm25 *w, uint8_t fieldnorm_id, uint32_t term_freq) {
  float tf = (float)term_freq;
  float norm = w->cache[fieldnorm_id];
  return tf / (tf + norm);
}
float to_bm25_score(const to_bm25 *w, uint8_t fieldnorm_id, uint32_t term_freq) {
  return w->weight * to_bm25_tf_factor(w, fieldnorm_id, term_freq);
}
float to_bm25_max_score(const to_bm25 *w) { return to_bm25_score(w, 255u, 2013265944u); }

/* ------------------------------------------------------------------ skip codes
 * src/postings/skip.rs:16-43 */
uint8_t to_encode_bitwidth(uint8_t bitwidth, int delta_1) {
  return (uint8_t)(bitwidth | ((delta_1 ? 1u : 0u) << 6));
}
void to_decode_bitwidth(uint8_t raw, uint8_t *bitwidth, int *delta_1) {
  *delta_1 = ((raw >> 6) & 1u) != 0;
  *bitwidth = raw & 0x1Fu;
}
uint8_t to_encode_block_wand_max_tf(uint32_t max_tf) {
  return (uint8_t)(max_tf < 255u ? max_tf : 255u);
}
uint32_t to_decode_block_wand_max_tf(uint8_t code) {
  return code == 255u ? 0xFFFFFFFFu : (uint32_t)code;
}
