/* to_postings.c — CPU ORACLE (test infrastructure): skip list, postings/positions
 * serializers and readers, block + in-block cursors.  See tantivy_oracle.h. */
#include "tantivy_oracle.h"

#include <assert.h>
#include <stdlib.h>
#include <string.h>

static uint32_t rd32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static size_t compressed_block_size(unsigned num_bits) { /* compression/mod.rs:13-15 */
  return (size_t)num_bits * TO_BLOCK_LEN / 8;
}

/* ------------------------------------------------------------------ SkipReader
 * src/postings/skip.rs:93-303 */
static void skip_read_block_info(to_skip_reader *r) {
  const uint8_t *b = r->data;
  size_t adv;
  r->last_doc_in_block = rd32(b);
  uint8_t doc_bits;
  int strict;
  to_decode_bitwidth(b[4], &doc_bits, &strict);
  to_block_info *bi = &r->block_info;
  memset(bi, 0, sizeof *bi);
  bi->is_vint = 0;
  bi->doc_num_bits = doc_bits;
  bi->strict_delta_encoded = strict;
  switch (r->skip_info) {
    case TO_BASIC:
      adv = 5;
      break;
    case TO_WITH_FREQS:
      bi->tf_num_bits = b[5];
      bi->block_wand_fieldnorm_id = b[6];
      bi->block_wand_term_freq = to_decode_block_wand_max_tf(b[7]);
      adv = 8;
      break;
    default:
      bi->tf_num_bits = b[5];
      bi->tf_sum = rd32(b + 6);
      bi->block_wand_fieldnorm_id = b[10];
      bi->block_wand_term_freq = to_decode_block_wand_max_tf(b[11]);
      adv = 12;
      break;
  }
  r->data += adv;
  r->data_len -= adv;
}
void to_skip_reader_new(to_skip_reader *r, const uint8_t *data, size_t len, uint32_t doc_freq,
                        int skip_info) {
  /* skip.rs:131-151 */
  memset(r, 0, sizeof *r);
  r->last_doc_in_block = doc_freq >= TO_BLOCK_LEN ? 0u : TO_TERMINATED;
  r->last_doc_in_previous_block = 0;
  r->data = data;
  r->data_len = len;
  r->skip_info = skip_info;
  r->block_info.is_vint = 1;
  r->block_info.num_docs = doc_freq;
  r->byte_offset = 0;
  r->remaining_docs = doc_freq;
  r->position_offset = 0;
  if (doc_freq >= TO_BLOCK_LEN) skip_read_block_info(r);
}
void to_skip_reader_advance(to_skip_reader *r) {
  /* skip.rs:275-302 */
  if (!r->block_info.is_vint) {
    r->remaining_docs -= TO_BLOCK_LEN;
    r->byte_offset +=
        compressed_block_size((unsigned)r->block_info.doc_num_bits + r->block_info.tf_num_bits);
    r->position_offset += (uint64_t)r->block_info.tf_sum;
  } else {
    r->remaining_docs = 0;
    r->byte_offset = (size_t)-1;
  }
  r->last_doc_in_previous_block = r->last_doc_in_block;
  if (r->remaining_docs >= TO_BLOCK_LEN) {
    skip_read_block_info(r);
  } else {
    r->last_doc_in_block = TO_TERMINATED;
    memset(&r->block_info, 0, sizeof r->block_info);
    r->block_info.is_vint = 1;
    r->block_info.num_docs = r->remaining_docs;
  }
}
int to_skip_reader_seek(to_skip_reader *r, uint32_t target) {
  /* skip.rs:263-273 */
  if (r->last_doc_in_block >= target) return 0;
  for (;;) {
    to_skip_reader_advance(r);
    if (r->last_doc_in_block >= target) return 1;
  }
}

/* ------------------------------------------------------------------ PostingsSerializer
 * src/postings/serializer.rs:258-487 (+ SkipSerializer skip.rs:55-91) */
struct to_postings_serializer {
  uint32_t last_doc_id_encoded;
  uint32_t doc_ids[TO_BLOCK_LEN], term_freqs[TO_BLOCK_LEN];
  size_t block_len;
  to_buf postings_write, skip_write;
  int mode;
  const uint8_t *fieldnorm_ids;
  uint32_t num_docs;
  int has_fieldnorm;
  int has_bm25;
  to_bm25 bm25;
  float avg_fieldnorm;
  int term_has_freq;
};
to_postings_serializer *to_postings_serializer_new(float avg_fieldnorm, int mode,
                                                   const uint8_t *fieldnorm_ids,
                                                   uint32_t num_docs) {
  to_postings_serializer *s = (to_postings_serializer *)calloc(1, sizeof *s);
  to_buf_init(&s->postings_write);
  to_buf_init(&s->skip_write);
  s->mode = mode;
  s->fieldnorm_ids = fieldnorm_ids;
  s->has_fieldnorm = fieldnorm_ids != NULL;
  s->num_docs = num_docs;
  s->avg_fieldnorm = avg_fieldnorm;
  return s;
}
void to_postings_serializer_free(to_postings_serializer *s) {
  if (!s) return;
  to_buf_free(&s->postings_write);
  to_buf_free(&s->skip_write);
  free(s);
}
void to_postings_serializer_new_term(to_postings_serializer *s, uint32_t term_doc_freq,
                                     int record_term_freq) {
  /* FieldSerializer::new_term calls clear() first: serializer.rs:190-192,483-486 */
  s->block_len = 0;
  s->last_doc_id_encoded = 0;
  /* serializer.rs:353-377 */
  s->has_bm25 = 0;
  s->term_has_freq = (s->mode != TO_BASIC) && record_term_freq;
  if (!s->term_has_freq) return;
  if (!s->has_fieldnorm) return;
  if (s->num_docs == 0) return;
  to_bm25_for_one_term(&s->bm25, term_doc_freq, s->num_docs, s->avg_fieldnorm);
  s->has_bm25 = 1;
}
static void ser_write_block(to_postings_serializer *s) {
  /* serializer.rs:379-431 */
  uint8_t tmp[TO_BLOCK_LEN * 5];
  size_t n;
  {
    uint8_t nb = to_compress_block_sorted(s->doc_ids, s->last_doc_id_encoded, tmp, &n);
    s->last_doc_id_encoded = s->doc_ids[TO_BLOCK_LEN - 1];
    /* SkipSerializer::write_doc skip.rs:64-67: strict-delta flag always set */
    to_buf_push_u32(&s->skip_write, s->last_doc_id_encoded);
    to_buf_push_u8(&s->skip_write, to_encode_bitwidth(nb, 1));
    to_buf_push(&s->postings_write, tmp, n);
  }
  if (s->term_has_freq) {
    uint8_t nb = to_compress_block_unsorted(s->term_freqs, 1, tmp, &n);
    to_buf_push(&s->postings_write, tmp, n);
    to_buf_push_u8(&s->skip_write, nb);
    if (s->mode == TO_WITH_FREQS_AND_POSITIONS) {
      uint32_t sum = 0;
      for (int i = 0; i < TO_BLOCK_LEN; i++) sum += s->term_freqs[i];
      to_buf_push_u32(&s->skip_write, sum);
    }
    uint8_t bw_fn = 0;
    uint32_t bw_tf = 0;
    if (s->has_bm25 && s->has_fieldnorm) {
      /* Iterator::max_by keeps the LAST maximal element (serializer.rs:404-428) */
      float best = 0.0f;
      for (int i = 0; i < TO_BLOCK_LEN; i++) {
        uint8_t fid = s->fieldnorm_ids[s->doc_ids[i]];
        float f = to_bm25_tf_factor(&s->bm25, fid, s->term_freqs[i]);
        if (i == 0 || !(f < best)) { /* partial_cmp != Less => replace (ties -> later wins) */
          best = f;
          bw_fn = fid;
          bw_tf = s->term_freqs[i];
        }
      }
    }
    to_buf_push_u8(&s->skip_write, bw_fn);
    to_buf_push_u8(&s->skip_write, to_encode_block_wand_max_tf(bw_tf));
  }
  s->block_len = 0;
}
void to_postings_serializer_write_doc(to_postings_serializer *s, uint32_t doc, uint32_t tf) {
  s->doc_ids[s->block_len] = doc;
  s->term_freqs[s->block_len] = tf;
  s->block_len++;
  if (s->block_len == TO_BLOCK_LEN) ser_write_block(s);
}
void to_postings_serializer_close_term(to_postings_serializer *s, uint32_t doc_freq, to_buf *out) {
  /* serializer.rs:443-481 */
  if (s->block_len) {
    uint8_t tmp[TO_BLOCK_LEN * 5];
    size_t n = to_vint_compress_sorted(s->doc_ids, s->block_len, tmp, s->last_doc_id_encoded);
    to_buf_push(&s->postings_write, tmp, n);
    if (s->term_has_freq) {
      n = to_vint_compress_unsorted(s->term_freqs, s->block_len, tmp);
      to_buf_push(&s->postings_write, tmp, n);
    }
    s->block_len = 0;
  }
  if (doc_freq >= TO_BLOCK_LEN) {
    uint8_t v[10];
    size_t n = to_vint_serialize(s->skip_write.len, v);
    to_buf_push(out, v, n);
    to_buf_push(out, s->skip_write.data, s->skip_write.len);
  }
  to_buf_push(out, s->postings_write.data, s->postings_write.len);
  to_buf_clear(&s->skip_write);
  to_buf_clear(&s->postings_write);
  s->has_bm25 = 0;
}

/* FieldSerializer's loop over the terms of a field (serializer.rs:186-260: new_term, write_doc
 * per posting, close_term) for lists given as flat arrays; out_term_starts = postings_range. */
void to_serialize_postings_batch(float avg_fieldnorm, int mode, const uint8_t *fieldnorm_ids,
                                 uint32_t num_docs, uint32_t n_terms, const uint64_t *term_starts,
                                 const uint32_t *docs, const uint32_t *tfs, to_buf *out,
                                 uint64_t *out_term_starts) {
  to_postings_serializer *s = to_postings_serializer_new(avg_fieldnorm, mode, fieldnorm_ids, num_docs);
  for (uint32_t t = 0; t < n_terms; t++) {
    const uint64_t lo = term_starts[t], hi = term_starts[t + 1];
    out_term_starts[t] = out->len;
    to_postings_serializer_new_term(s, (uint32_t)(hi - lo), 1);
    for (uint64_t i = lo; i < hi; i++)
      to_postings_serializer_write_doc(s, docs[i], tfs ? tfs[i] : 1u);
    to_postings_serializer_close_term(s, (uint32_t)(hi - lo), out);
  }
  out_term_starts[n_terms] = out->len;
  to_postings_serializer_free(s);
}

/* ------------------------------------------------------------------ PositionSerializer
 * src/positions/serializer.rs:12-93 */
struct to_position_serializer {
  to_buf *out;
  to_buf positions_buffer;
  uint32_t block[TO_BLOCK_LEN];
  size_t block_len;
  to_buf bit_widths;
};
to_position_serializer *to_position_serializer_new(to_buf *out) {
  to_position_serializer *s = (to_position_serializer *)calloc(1, sizeof *s);
  s->out = out;
  to_buf_init(&s->positions_buffer);
  to_buf_init(&s->bit_widths);
  return s;
}
void to_position_serializer_free(to_position_serializer *s) {
  if (!s) return;
  to_buf_free(&s->positions_buffer);
  to_buf_free(&s->bit_widths);
  free(s);
}
static void pos_flush_block(to_position_serializer *s) {
  if (s->block_len == 0) return;
  uint8_t tmp[TO_BLOCK_LEN * 5];
  size_t n;
  if (s->block_len == TO_BLOCK_LEN) {
    uint8_t bw = to_compress_block_unsorted(s->block, 0, tmp, &n);
    to_buf_push_u8(&s->bit_widths, bw);
    to_buf_push(&s->positions_buffer, tmp, n);
  } else {
    n = to_vint_compress_unsorted(s->block, s->block_len, tmp);
    to_buf_push(&s->positions_buffer, tmp, n);
  }
  s->block_len = 0;
}
void to_position_serializer_write_positions_delta(to_position_serializer *s, const uint32_t *d,
                                                  size_t n) {
  while (n) {
    size_t room = TO_BLOCK_LEN - s->block_len;
    size_t m = room < n ? room : n;
    memcpy(s->block + s->block_len, d, m * sizeof(uint32_t));
    s->block_len += m;
    d += m;
    n -= m;
    if (s->block_len == TO_BLOCK_LEN) pos_flush_block(s);
  }
}
void to_position_serializer_close_term(to_position_serializer *s) {
  pos_flush_block(s);
  uint8_t v[10];
  size_t n = to_vint_serialize(s->bit_widths.len, v);
  to_buf_push(s->out, v, n);
  to_buf_push(s->out, s->bit_widths.data, s->bit_widths.len);
  to_buf_push(s->out, s->positions_buffer.data, s->positions_buffer.len);
  to_buf_clear(&s->bit_widths);
  to_buf_clear(&s->positions_buffer);
}

/* ------------------------------------------------------------------ PositionReader
 * src/positions/reader.rs:43-148 */
int to_position_reader_open(to_position_reader *r, const uint8_t *data, size_t len) {
  uint64_t nblocks;
  size_t c = to_vint_deserialize(data, len, &nblocks);
  if (c == 0 || c + nblocks > len) return -1;
  memset(r, 0, sizeof *r);
  r->bit_widths = r->orig_bit_widths = data + c;
  r->n_bit_widths = r->orig_n_bit_widths = (size_t)nblocks;
  r->positions = r->orig_positions = data + c + nblocks;
  r->positions_len = r->orig_positions_len = len - c - (size_t)nblocks;
  r->block_offset = (uint64_t)INT64_MAX;
  r->anchor_offset = 0;
  return 0;
}
static void pos_reset(to_position_reader *r) {
  r->positions = r->orig_positions;
  r->positions_len = r->orig_positions_len;
  r->bit_widths = r->orig_bit_widths;
  r->n_bit_widths = r->orig_n_bit_widths;
  r->block_offset = (uint64_t)INT64_MAX;
  r->anchor_offset = 0;
}
static void pos_advance_num_blocks(to_position_reader *r, size_t num_blocks) {
  size_t num_bits = 0;
  for (size_t i = 0; i < num_blocks; i++) num_bits += r->bit_widths[i];
  size_t skip = num_bits * TO_BLOCK_LEN / 8;
  r->bit_widths += num_blocks;
  r->n_bit_widths -= num_blocks;
  r->positions += skip;
  r->positions_len -= skip;
  r->anchor_offset += (uint64_t)num_blocks * TO_BLOCK_LEN;
}
static void pos_load_block(to_position_reader *r, size_t block_rel_id) {
  size_t bits = 0;
  for (size_t i = 0; i < block_rel_id && i < r->n_bit_widths; i++) bits += r->bit_widths[i];
  size_t byte_offset = bits * TO_BLOCK_LEN / 8;
  const uint8_t *data = r->positions + byte_offset;
  if (r->n_bit_widths > block_rel_id) {
    to_uncompress_block_unsorted(data, r->bit_widths[block_rel_id], 0, r->block);
  } else {
    /* BlockDecoder::default() prefill is 0 and output_len shrinks; values past the end are
     * never read by a well-formed caller. */
    memset(r->block, 0, sizeof r->block);
    to_vint_uncompress_unsorted_until_end(data, r->positions_len - byte_offset, r->block,
                                          TO_BLOCK_LEN);
  }
  r->block_offset = r->anchor_offset + (uint64_t)block_rel_id * TO_BLOCK_LEN;
}
void to_position_reader_read(to_position_reader *r, uint64_t offset, uint32_t *out, size_t n) {
  if (offset < r->anchor_offset) pos_reset(r);
  int64_t delta = (int64_t)offset - (int64_t)r->block_offset;
  if (!(delta >= 0 && delta < 128)) {
    uint64_t d = offset - r->anchor_offset;
    pos_advance_num_blocks(r, (size_t)(d / TO_BLOCK_LEN));
    pos_load_block(r, 0);
  } else {
    size_t nb = (size_t)((r->block_offset - r->anchor_offset) / TO_BLOCK_LEN);
    pos_advance_num_blocks(r, nb);
  }
  for (size_t i = 1;; i++) {
    size_t off_in_block = (size_t)(offset % TO_BLOCK_LEN);
    size_t remaining = TO_BLOCK_LEN - off_in_block;
    if (remaining >= n) {
      memcpy(out, r->block + off_in_block, n * sizeof(uint32_t));
      break;
    }
    memcpy(out, r->block + off_in_block, remaining * sizeof(uint32_t));
    out += remaining;
    n -= remaining;
    offset += remaining;
    pos_load_block(r, i);
  }
}

/* ------------------------------------------------------------------ BlockSegmentPostings
 * src/postings/block_segment_postings.rs */
int to_block_postings_open(to_block_postings *p, uint32_t doc_freq, const uint8_t *data,
                           size_t len, int record_option, int requested_option) {
  memset(p, 0, sizeof *p);
  const uint8_t *skip = NULL;
  size_t skip_len = 0;
  if (doc_freq >= TO_BLOCK_LEN) { /* split_into_skips_and_postings :78-88 */
    uint64_t sl;
    size_t c = to_vint_deserialize(data, len, &sl);
    if (c == 0 || c + sl > len) return -1;
    skip = data + c;
    skip_len = (size_t)sl;
    data += c + sl;
    len -= c + sl;
    size_t block_count = doc_freq / TO_BLOCK_LEN;
    if (skip_len < 8 * block_count) record_option = TO_BASIC; /* :107-116 JSON quirk */
  }
  to_skip_reader_new(&p->skip, skip, skip_len, doc_freq, record_option);
  if (record_option == TO_BASIC)
    p->freq_reading = 0;
  else if (requested_option == TO_BASIC)
    p->freq_reading = 1;
  else
    p->freq_reading = 2;
  for (int i = 0; i < TO_BLOCK_LEN; i++) {
    p->docs[i] = TO_TERMINATED;
    p->freqs[i] = 1;
  }
  p->doc_freq = doc_freq;
  p->data = data;
  p->data_len = len;
  to_block_postings_load_block(p);
  return 0;
}
void to_block_postings_load_block(to_block_postings *p) {
  /* :343-391 */
  if (p->block_loaded) return;
  size_t offset = p->skip.byte_offset;
  const to_block_info *bi = &p->skip.block_info;
  if (!bi->is_vint) {
    size_t c = to_uncompress_block_sorted(p->data + offset, p->skip.last_doc_in_previous_block,
                                          bi->doc_num_bits, bi->strict_delta_encoded, p->docs);
    p->block_len = TO_BLOCK_LEN;
    if (p->freq_reading == 2)
      to_uncompress_block_unsorted(p->data + offset + c, bi->tf_num_bits,
                                   bi->strict_delta_encoded, p->freqs);
  } else {
    size_t n = bi->num_docs;
    for (int i = 0; i < TO_BLOCK_LEN; i++) p->docs[i] = TO_TERMINATED;
    size_t consumed = 0;
    if (n) consumed = to_vint_uncompress_sorted(p->data + offset, p->docs, n,
                                                p->skip.last_doc_in_previous_block);
    p->block_len = n;
    if (p->freq_reading == 2) {
      size_t avail = n ? p->data_len - offset : 0;
      if (avail > consumed) {
        for (int i = 0; i < TO_BLOCK_LEN; i++) p->freqs[i] = TO_TERMINATED;
        to_vint_uncompress_unsorted(p->data + offset + consumed, p->freqs, n);
      }
    }
  }
  p->block_loaded = 1;
}
void to_block_postings_advance(to_block_postings *p) {
  to_skip_reader_advance(&p->skip);
  p->block_loaded = 0;
  p->has_block_max_cache = 0;
  to_block_postings_load_block(p);
}
void to_block_postings_seek_block(to_block_postings *p, uint32_t target) {
  if (to_skip_reader_seek(&p->skip, target)) {
    p->has_block_max_cache = 0;
    p->block_loaded = 0;
  }
}
size_t to_block_postings_seek(to_block_postings *p, uint32_t target) {
  to_block_postings_seek_block(p, target);
  to_block_postings_load_block(p);
  return to_search_block(p->docs, target);
}
float to_block_postings_block_max_score(to_block_postings *p, const uint8_t *fieldnorm_ids,
                                        uint8_t const_fieldnorm_id, const to_bm25 *w) {
  /* :147-179 */
  if (p->has_block_max_cache) return p->block_max_cache;
  if (!p->skip.block_info.is_vint) {
    float s = to_bm25_score(w, p->skip.block_info.block_wand_fieldnorm_id,
                            p->skip.block_info.block_wand_term_freq);
    p->has_block_max_cache = 1;
    p->block_max_cache = s;
    return s;
  }
  if (p->block_loaded) {
    float best = 0.0f;
    int any = 0;
    for (size_t i = 0; i < p->block_len; i++) {
      uint8_t fid = fieldnorm_ids ? fieldnorm_ids[p->docs[i]] : const_fieldnorm_id;
      float s = to_bm25_score(w, fid, p->freqs[i]);
      if (!any || s > best) best = s;
      any = 1;
    }
    p->has_block_max_cache = 1;
    p->block_max_cache = any ? best : 0.0f;
    return p->block_max_cache;
  }
  return to_bm25_max_score(w);
}

/* ------------------------------------------------------------------ SegmentPostings
 * src/postings/segment_postings.rs:156-255 */
int to_segment_postings_open(to_segment_postings *sp, uint32_t doc_freq, const uint8_t *postings,
                             size_t postings_len, const uint8_t *positions, size_t positions_len,
                             int record_option, int requested_option) {
  /* inverted_index_reader.rs:226-247: option.downgrade(record_option) */
  if (requested_option > record_option) requested_option = record_option;
  sp->cur = 0;
  sp->has_positions = 0;
  if (to_block_postings_open(&sp->bp, doc_freq, postings, postings_len, record_option,
                             requested_option))
    return -1;
  if (requested_option == TO_WITH_FREQS_AND_POSITIONS) {
    if (to_position_reader_open(&sp->pos, positions, positions_len)) return -1;
    sp->has_positions = 1;
  }
  return 0;
}
uint32_t to_sp_doc(const to_segment_postings *sp) { return sp->bp.docs[sp->cur]; }
uint32_t to_sp_advance(to_segment_postings *sp) {
  if (sp->cur == TO_BLOCK_LEN - 1) {
    sp->cur = 0;
    to_block_postings_advance(&sp->bp);
  } else {
    sp->cur++;
  }
  return to_sp_doc(sp);
}
uint32_t to_sp_seek(to_segment_postings *sp, uint32_t target) {
  if (to_sp_doc(sp) >= target) return to_sp_doc(sp);
  sp->cur = sp->cur + 1 < TO_BLOCK_LEN - 1 ? sp->cur + 1 : TO_BLOCK_LEN - 1;
  if (to_sp_doc(sp) >= target) return to_sp_doc(sp);
  sp->cur = to_block_postings_seek(&sp->bp, target);
  return to_sp_doc(sp);
}
uint32_t to_sp_term_freq(const to_segment_postings *sp) { return sp->bp.freqs[sp->cur]; }
size_t to_sp_positions_with_offset(to_segment_postings *sp, uint32_t offset, uint32_t *out) {
  /* :232-254 */
  uint32_t tf = to_sp_term_freq(sp);
  if (!sp->has_positions) return 0;
  uint64_t read_offset = sp->bp.skip.position_offset;
  for (size_t i = 0; i < sp->cur; i++) read_offset += sp->bp.freqs[i];
  to_position_reader_read(&sp->pos, read_offset, out, tf);
  uint32_t cum = offset;
  for (uint32_t i = 0; i < tf; i++) {
    cum += out[i];
    out[i] = cum;
  }
  return tf;
}

size_t to_skip_walk(const uint8_t *data, size_t len, uint32_t doc_freq, int skip_info,
                    size_t n_advances, to_skip_state *out) {
  to_skip_reader r;
  to_skip_reader_new(&r, data, len, doc_freq, skip_info);
  for (size_t i = 0; i <= n_advances; i++) {
    to_skip_state *s = &out[i];
    s->last_doc_in_block = r.last_doc_in_block;
    s->is_vint = r.block_info.is_vint;
    s->doc_num_bits = r.block_info.doc_num_bits;
    s->strict = r.block_info.strict_delta_encoded;
    s->tf_num_bits = r.block_info.tf_num_bits;
    s->tf_sum = r.block_info.tf_sum;
    s->bw_fieldnorm_id = r.block_info.block_wand_fieldnorm_id;
    s->bw_term_freq = r.block_info.block_wand_term_freq;
    s->num_docs = r.block_info.num_docs;
    s->byte_offset = (uint64_t)r.byte_offset;
    s->position_offset = r.position_offset;
    if (i < n_advances) to_skip_reader_advance(&r);
  }
  return n_advances + 1;
}

/* the same for the positions of a field: write_positions_delta + close_term per term */
void to_serialize_positions_batch(uint32_t n_terms, const uint64_t *term_starts,
                                  const uint32_t *deltas, to_buf *out, uint64_t *out_term_starts) {
  to_position_serializer *s = to_position_serializer_new(out);
  for (uint32_t t = 0; t < n_terms; t++) {
    out_term_starts[t] = out->len;
    to_position_serializer_write_positions_delta(s, deltas + term_starts[t],
                                                 (size_t)(term_starts[t + 1] - term_starts[t]));
    to_position_serializer_close_term(s);
  }
  out_term_starts[n_terms] = out->len;
  to_position_serializer_free(s);
}
