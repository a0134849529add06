/*
 * tantivy_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's query-execution hot path
 * (postings codec -> AND / OR / phrase -> BM25 -> top-k), written from the
 * reference's published behaviour.  Every function cites the reference
 * file:line it follows (paths are relative to the tantivy checkout).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library.  The product path (tantivy_amd/) never does.
 *
 * Also restated: the generic scorer tree behind boolean queries (Intersection,
 * BufferedUnionScorer, Disjunction, RequiredOptionalScorer, Exclude; to_query.c),
 * PostingsSerializer / PositionSerializer (to_postings.c: the checker of the device-side
 * codec writers) and, in oracle.py, the file framing and the TermInfoStore.
 *
 * PARITY PINNING:
 *   - everything above the 128-int codec (decoded ints, skip entries, doc ids,
 *     BM25 scores, top-k ordering) is pinned by the reference's own known-answer
 *     tests (see tests/test_oracle_kat.py and SURVEY.md §8c);
 *   - file framing, TermInfoStore and vint posting / position bytes are pinned by the
 *     reference's compat fixtures (tests/golden/compat_index.json: files written by
 *     released tantivy versions);
 *   - the *byte layout* of a BitPacker4x block comes from the third-party crate
 *     `bitpacking = 0.9.3` (Cargo.toml:42), whose source is not in the reference
 *     tree and for which the tree holds no golden bytes  =>  "parity unpinned"
 *     at that one boundary (round-trip, size and junk-byte properties only).
 */
#ifndef TANTIVY_ORACLE_H
#define TANTIVY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TO_TERMINATED 0x7FFFFFFFu /* src/docset.rs:12 (i32::MAX as u32) */
#define TO_BLOCK_LEN 128          /* src/postings/compression/mod.rs:3 */

/* record options: src/schema/index_record_option.rs */
enum { TO_BASIC = 0, TO_WITH_FREQS = 1, TO_WITH_FREQS_AND_POSITIONS = 2 };

/* ---------- growable byte buffer ---------- */
typedef struct {
  uint8_t *data;
  size_t len, cap;
} to_buf;
void to_buf_init(to_buf *b);
void to_buf_free(to_buf *b);
void to_buf_push(to_buf *b, const void *src, size_t n);
void to_buf_push_u8(to_buf *b, uint8_t v);
void to_buf_push_u32(to_buf *b, uint32_t v);
void to_buf_clear(to_buf *b);

/* ---------- fieldnorm: src/fieldnorm/code.rs ---------- */
uint32_t to_id_to_fieldnorm(uint8_t id);
uint8_t to_fieldnorm_to_id(uint32_t fieldnorm);
const uint32_t *to_fieldnorm_table(void);

/* ---------- common VInt: common/src/vint.rs:61-143 ---------- */
size_t to_vint_serialize(uint64_t v, uint8_t *out /* >=10 bytes */);
/* returns bytes consumed, 0 on error */
size_t to_vint_deserialize(const uint8_t *data, size_t len, uint64_t *out);

/* ---------- postings vint blocks: src/postings/compression/vint.rs ---------- */
size_t to_vint_compress_sorted(const uint32_t *in, size_t n, uint8_t *out, uint32_t offset);
size_t to_vint_compress_unsorted(const uint32_t *in, size_t n, uint8_t *out);
size_t to_vint_uncompress_sorted(const uint8_t *data, uint32_t *out, size_t n, uint32_t offset);
size_t to_vint_uncompress_unsorted(const uint8_t *data, uint32_t *out, size_t n);
size_t to_vint_uncompress_unsorted_until_end(const uint8_t *data, size_t len, uint32_t *out,
                                             size_t out_cap);

/* ---------- BitPacker4x (third-party bitpacking 0.9.3, restated) ---------- */
uint8_t to_bp4_num_bits(const uint32_t *v /*128*/);
uint8_t to_bp4_num_bits_strictly_sorted(int has_initial, uint32_t initial, const uint32_t *v);
uint8_t to_bp4_num_bits_sorted(uint32_t initial, const uint32_t *v);
size_t to_bp4_compress(const uint32_t *v, uint8_t *out, uint8_t num_bits);
size_t to_bp4_decompress(const uint8_t *in, uint32_t *out, uint8_t num_bits);
/* to_simd.c: the same decode with SSE2 registers (the CPU baseline's decoder; to_set_simd(1) routes
 * to_bp4_decompress* through it; tests cross-check it against the scalar loops) */
size_t to_simd_bp4_decompress(const uint8_t *in, uint32_t *out, uint8_t num_bits);
size_t to_simd_bp4_decompress_delta(uint32_t seed, uint32_t add, const uint8_t *in, uint32_t *out,
                                    uint8_t num_bits);
void to_set_simd(int on);
int to_get_simd(void);
size_t to_bp4_compress_strictly_sorted(int has_initial, uint32_t initial, const uint32_t *v,
                                       uint8_t *out, uint8_t num_bits);
size_t to_bp4_decompress_strictly_sorted(int has_initial, uint32_t initial, const uint8_t *in,
                                         uint32_t *out, uint8_t num_bits);
size_t to_bp4_compress_sorted(uint32_t initial, const uint32_t *v, uint8_t *out, uint8_t num_bits);
size_t to_bp4_decompress_sorted(uint32_t initial, const uint8_t *in, uint32_t *out,
                                uint8_t num_bits);

/* ---------- BlockEncoder / BlockDecoder: src/postings/compression/mod.rs:17-169 ---------- */
/* returns num_bits, writes bytes to out (>= 512 B), *out_len = bytes */
uint8_t to_compress_block_sorted(const uint32_t *block, uint32_t offset, uint8_t *out,
                                 size_t *out_len);
uint8_t to_compress_block_unsorted(const uint32_t *block, int minus_one_encoded, uint8_t *out,
                                   size_t *out_len);
size_t to_uncompress_block_sorted(const uint8_t *data, uint32_t offset, uint8_t num_bits,
                                  int strict_delta, uint32_t *out);
size_t to_uncompress_block_unsorted(const uint8_t *data, uint8_t num_bits, int minus_one_encoded,
                                    uint32_t *out);
/* src/postings/block_search.rs:38-76: first idx with arr[idx] >= target, arr padded to 128 */
size_t to_search_block(const uint32_t *arr /*128*/, uint32_t target);

/* ---------- BM25: src/query/bm25.rs ---------- */
typedef struct {
  float weight;
  float cache[256];
  float average_fieldnorm;
} to_bm25;
float to_idf(uint64_t doc_freq, uint64_t doc_count);
void to_bm25_new(to_bm25 *w, float idf, float average_fieldnorm);
void to_bm25_for_one_term(to_bm25 *w, uint64_t term_doc_freq, uint64_t total_num_docs,
                          float avg_fieldnorm);
void to_bm25_boost_by(to_bm25 *w, float boost);
float to_bm25_tf_factor(const to_bm25 *w, uint8_t fieldnorm_id, uint32_t term_freq);
float to_bm25_score(const to_bm25 *w, uint8_t fieldnorm_id, uint32_t term_freq);
float to_bm25_max_score(const to_bm25 *w);

/* ---------- skip list: src/postings/skip.rs ---------- */
uint8_t to_encode_bitwidth(uint8_t bitwidth, int delta_1);
void to_decode_bitwidth(uint8_t raw, uint8_t *bitwidth, int *delta_1);
uint8_t to_encode_block_wand_max_tf(uint32_t max_tf);
uint32_t to_decode_block_wand_max_tf(uint8_t code);

typedef struct {
  int is_vint; /* BlockInfo::VInt vs BitPacked */
  uint8_t doc_num_bits;
  int strict_delta_encoded;
  uint8_t tf_num_bits;
  uint32_t tf_sum;
  uint8_t block_wand_fieldnorm_id;
  uint32_t block_wand_term_freq;
  uint32_t num_docs; /* VInt */
} to_block_info;

typedef struct {
  uint32_t last_doc_in_block;
  uint32_t last_doc_in_previous_block;
  const uint8_t *data; /* remaining skip bytes */
  size_t data_len;
  int skip_info; /* record option of the *indexed* field */
  size_t byte_offset;
  uint32_t remaining_docs;
  to_block_info block_info;
  uint64_t position_offset;
} to_skip_reader;
void to_skip_reader_new(to_skip_reader *r, const uint8_t *data, size_t len, uint32_t doc_freq,
                        int skip_info);
void to_skip_reader_advance(to_skip_reader *r);
int to_skip_reader_seek(to_skip_reader *r, uint32_t target);

/* ---------- postings serializer: src/postings/serializer.rs:307-487 ---------- */
typedef struct to_postings_serializer to_postings_serializer;
/* fieldnorm_ids may be NULL (no fieldnorm reader). num_docs = fieldnorm reader's num_docs */
to_postings_serializer *to_postings_serializer_new(float avg_fieldnorm, int mode,
                                                   const uint8_t *fieldnorm_ids,
                                                   uint32_t num_docs);
void to_postings_serializer_free(to_postings_serializer *s);
void to_postings_serializer_new_term(to_postings_serializer *s, uint32_t term_doc_freq,
                                     int record_term_freq);
void to_postings_serializer_write_doc(to_postings_serializer *s, uint32_t doc, uint32_t tf);
void to_postings_serializer_close_term(to_postings_serializer *s, uint32_t doc_freq, to_buf *out);

/* ---------- positions: src/positions/{serializer,reader}.rs ---------- */
typedef struct to_position_serializer to_position_serializer;
to_position_serializer *to_position_serializer_new(to_buf *out);
void to_position_serializer_free(to_position_serializer *s);
void to_position_serializer_write_positions_delta(to_position_serializer *s, const uint32_t *d,
                                                  size_t n);
void to_position_serializer_close_term(to_position_serializer *s);
void to_serialize_postings_batch(float avg_fieldnorm, int mode, const uint8_t *fieldnorm_ids,
                                 uint32_t num_docs, uint32_t n_terms, const uint64_t *term_starts,
                                 const uint32_t *docs, const uint32_t *tfs, to_buf *out,
                                 uint64_t *out_term_starts);
void to_serialize_positions_batch(uint32_t n_terms, const uint64_t *term_starts,
                                  const uint32_t *deltas, to_buf *out, uint64_t *out_term_starts);

typedef struct {
  const uint8_t *bit_widths;
  size_t n_bit_widths;
  const uint8_t *positions;
  size_t positions_len;
  uint32_t block[TO_BLOCK_LEN];
  uint64_t block_offset;
  uint64_t anchor_offset;
  const uint8_t *orig_bit_widths;
  size_t orig_n_bit_widths;
  const uint8_t *orig_positions;
  size_t orig_positions_len;
} to_position_reader;
int to_position_reader_open(to_position_reader *r, const uint8_t *data, size_t len);
void to_position_reader_read(to_position_reader *r, uint64_t offset, uint32_t *out, size_t n);

/* ---------- block cursor + in-block cursor ----------
 * src/postings/block_segment_postings.rs, src/postings/segment_postings.rs */
typedef struct {
  uint32_t docs[TO_BLOCK_LEN];
  uint32_t freqs[TO_BLOCK_LEN];
  size_t block_len;
  int block_loaded;
  int freq_reading; /* 0 NoFreq, 1 SkipFreq, 2 ReadFreq */
  int has_block_max_cache;
  float block_max_cache;
  uint32_t doc_freq;
  const uint8_t *data; /* postings payload (after the skip data) */
  size_t data_len;
  to_skip_reader skip;
} to_block_postings;
/* data = the term's postings_range bytes. Returns 0 on success. */
int to_block_postings_open(to_block_postings *p, uint32_t doc_freq, const uint8_t *data,
                           size_t len, int record_option, int requested_option);
void to_block_postings_load_block(to_block_postings *p);
void to_block_postings_advance(to_block_postings *p);
void to_block_postings_seek_block(to_block_postings *p, uint32_t target);
size_t to_block_postings_seek(to_block_postings *p, uint32_t target);
float to_block_postings_block_max_score(to_block_postings *p, const uint8_t *fieldnorm_ids,
                                        uint8_t const_fieldnorm_id, const to_bm25 *w);

typedef struct {
  to_block_postings bp;
  size_t cur;
  int has_positions;
  to_position_reader pos;
} to_segment_postings;
int to_segment_postings_open(to_segment_postings *sp, uint32_t doc_freq, const uint8_t *postings,
                             size_t postings_len, const uint8_t *positions, size_t positions_len,
                             int record_option, int requested_option);
uint32_t to_sp_doc(const to_segment_postings *sp);
uint32_t to_sp_advance(to_segment_postings *sp);
uint32_t to_sp_seek(to_segment_postings *sp, uint32_t target);
uint32_t to_sp_term_freq(const to_segment_postings *sp);
/* appends term_freq positions (offset added, prefix-summed) to out; returns count */
size_t to_sp_positions_with_offset(to_segment_postings *sp, uint32_t offset, uint32_t *out);

/* ---------- top-k: src/collector/sort_key/sort_by_score.rs:86-161 ---------- */
typedef struct {
  float score;
  uint32_t doc;
} to_hit;
typedef struct {
  to_hit *heap; /* min-heap on (score asc, doc desc) */
  size_t len, top_n;
  int has_threshold;
  float threshold;
} to_topn_heap;
void to_topn_init(to_topn_heap *h, size_t top_n);
void to_topn_free(to_topn_heap *h);
void to_topn_push(to_topn_heap *h, float score, uint32_t doc);
/* heap-order dump (into_vec) */
size_t to_topn_into_vec(const to_topn_heap *h, to_hit *out);

typedef struct {
  float score;
  uint32_t segment_ord;
  uint32_t doc;
} to_global_hit;
/* src/collector/sort_key_top_collector.rs:76-95 + top_score_collector.rs:590-600:
 * top-(offset+limit) by (score desc, (segment_ord, doc) asc), then skip offset.
 * returns number written to out (<= limit). */
size_t to_merge_top_k(const to_global_hit *hits, size_t n, size_t offset, size_t limit,
                      to_global_hit *out);

/* ---------- segment + query execution ---------- */
typedef struct {
  uint32_t doc_freq;
  uint64_t postings_start, postings_end;   /* relative to idx body (after 8-byte header) */
  uint64_t positions_start, positions_end; /* relative to pos body */
} to_term_info;

typedef struct {
  uint32_t max_doc;
  int record_option;       /* how the field was indexed */
  const uint8_t *idx;      /* field's .idx sub-file, INCLUDING the 8-byte total_num_tokens */
  size_t idx_len;
  const uint8_t *pos;      /* field's .pos sub-file or NULL */
  size_t pos_len;
  const uint8_t *fieldnorm; /* max_doc bytes or NULL => constant id 1 (term_weight.rs:209-219) */
  uint64_t total_num_tokens;
} to_segment_view;

enum { TO_MODE_AND = 0, TO_MODE_OR = 1, TO_MODE_PHRASE = 2, TO_MODE_BOOL = 3 };

/* A query against one segment.  weights[i]/tf cache follow Bm25Weight semantics; for a phrase
 * there is a single weight (weights[0]) built with idf summed over the terms. */
typedef struct {
  uint32_t n_terms;
  const to_term_info *terms; /* per-term info in this segment; doc_freq==0 => absent */
  const to_bm25 *weights;    /* n_terms entries (AND/OR) or 1 (phrase) */
  const uint32_t *phrase_offsets; /* phrase only */
  int mode;
  uint32_t k;
  /* TO_MODE_BOOL: Occur per term (0 Should, 1 Must, 2 MustNot), clause index per term (terms
   * sharing one form a nested union; NULL = one clause per term), minimum_number_should_match */
  const uint8_t *occurs;
  const uint8_t *clause_of;
  uint32_t min_should_match;
} to_query;

/* A nested boolean query of any depth, prefix order (a node, then its children's subtrees): the scorer tree
 * BooleanWeight::complex_scorer builds recursively (boolean_weight.rs:225-233,236-431). */
enum { TO_TREE_TERM = 0, TO_TREE_BOOL = 1, TO_TREE_PHRASE = 2 };
typedef struct {
  uint8_t kind;    /* TO_TREE_* */
  uint8_t occur;   /* the node's occur inside its parent (src/query/occur.rs: 0 Should, 1 Must, 2 MustNot) */
  uint16_t n_kids; /* BOOL: children; PHRASE: terms (consecutive in terms[], from `first`) */
  uint32_t first;  /* TERM: index into terms[]; PHRASE: index of its first term (its Bm25Weight: weights[first]) */
  uint32_t msm;    /* BOOL: minimum_number_should_match */
} to_tree_node;
/* 1 (default): a union re-seeks a member that a failed Intersection::seek_danger left half-seeked (the scorer tree's
 * intended semantics); 0: BufferedUnionScorer::seek exactly as written (buffered_union.rs:254-259) — see to_query.c */
void to_set_union_reseek_invalid(int on);
/* every match of the tree, docs ascending: returns their number ((size_t)-1: malformed tree) */
size_t to_tree_match_all(const to_segment_view *seg, const to_tree_node *nodes, size_t n_nodes,
                         const to_term_info *terms, const to_bm25 *weights, const uint32_t *phrase_offsets,
                         size_t n_terms, uint32_t *docs, float *scores, size_t cap);

/* Faithful executors (what the reference runs for TopDocs order_by_score):
 *   AND    -> block_wand_intersection (boolean_query/block_wand_intersection.rs:19-179)
 *   OR     -> block_wand / block_wand_single_scorer (boolean_query/block_wand_union.rs)
 *   PHRASE -> for_each_pruning_scorer over PhraseScorer (query/weight.rs:47-60)
 * feeding TopNHeap.  Output: heap-order hits; returns count. */
size_t to_search_pruned(const to_segment_view *seg, const to_query *q, to_hit *out);
/* Exhaustive executors: Intersection / BufferedUnionScorer / PhraseScorer advance()+score()
 * over every match, (score desc, doc asc) top-k via the same TopNHeap. */
size_t to_search_exhaustive(const to_segment_view *seg, const to_query *q, to_hit *out);
/* All matches (doc asc) with scores; caller supplies capacity; returns total match count
 * (may exceed cap; only cap are written). */
size_t to_match_all(const to_segment_view *seg, const to_query *q, uint32_t *docs, float *scores,
                    size_t cap);
/* sort hits by (score desc, doc asc) in place */
void to_sort_hits(to_hit *hits, size_t n);

/* decode a whole posting list (docs, tfs) -- for codec parity tests */
size_t to_decode_postings(const to_segment_view *seg, const to_term_info *ti, uint32_t *docs,
                          uint32_t *tfs);
/* decode all positions of doc index i.. : returns positions of every doc concatenated (raw,
 * prefix-summed per doc) */
size_t to_decode_positions(const to_segment_view *seg, const to_term_info *ti, uint32_t *out,
                           size_t cap);

/* ---------- synthetic index ---------- */
typedef struct {
  uint32_t max_doc;
  uint32_t n_terms;
  int record_option;
  to_buf idx, pos, fieldnorm;
  to_term_info *terms; /* n_terms, rank r = index+1 */
  uint64_t total_num_tokens;
} to_synth_segment;
/* with_positions: 0 => WithFreqs (8-B skip entries), 1 => WithFreqsAndPositions (12-B). */
to_synth_segment *to_synth_build(uint32_t max_doc, uint32_t n_terms, uint32_t segment_ord,
                                 int with_positions, uint32_t phrase_terms);
void to_synth_free(to_synth_segment *s);
void to_synth_view(const to_synth_segment *s, to_segment_view *v);

/* deterministic rng (splitmix64 seeded xoshiro256**) exposed for the query-stream generator */
typedef struct {
  uint64_t s[4];
} to_rng;
void to_rng_seed(to_rng *r, uint64_t seed);
uint64_t to_rng_next(to_rng *r);
double to_rng_uniform(to_rng *r); /* [0,1) */

/* Zipf(s=1) rank sampler over 1..n (inverse-CDF on the harmonic table) */
uint32_t to_zipf_rank(to_rng *r, uint32_t n);

/* ---------- CPU baseline driver (bench.py cpu_baseline leg) ----------
 * Runs `n_queries` queries with `n_threads` pthreads (query-level parallelism, one query per
 * thread at a time, like tantivy's one-task-per-segment executor) through to_search_pruned.
 * Fills per-query latency in seconds; returns wall seconds for the whole batch. */
double to_baseline_run(const to_segment_view *seg, const to_query *queries, size_t n_queries,
                       int n_threads, double *latency_s, to_hit *out_hits /* n*k */,
                       uint32_t *out_counts);

/* ---------- TermScorer handle (tests replay term_scorer.rs unit tests) ---------- */
void *to_ts_new(const to_segment_view *seg, const to_term_info *ti, const to_bm25 *w);
void to_ts_free(void *t);
uint32_t to_ts_doc(void *t);
uint32_t to_ts_advance(void *t);
uint32_t to_ts_seek(void *t, uint32_t target);
void to_ts_seek_block(void *t, uint32_t target);
uint32_t to_ts_term_freq(void *t);
float to_ts_score(void *t);
float to_ts_block_max_score(void *t);
float to_ts_max_score(void *t);
uint32_t to_ts_last_doc_in_block(void *t);

/* skip reader state dump for tests (skip.rs:333-448) */
typedef struct {
  uint32_t last_doc_in_block;
  int is_vint;
  uint8_t doc_num_bits;
  int strict;
  uint8_t tf_num_bits;
  uint32_t tf_sum;
  uint8_t bw_fieldnorm_id;
  uint32_t bw_term_freq;
  uint32_t num_docs;
  uint64_t byte_offset;
  uint64_t position_offset;
} to_skip_state;
/* walks the skip data, writing up to cap states (initial state first); returns count */
size_t to_skip_walk(const uint8_t *data, size_t len, uint32_t doc_freq, int skip_info,
                    size_t n_advances, to_skip_state *out);

#ifdef __cplusplus
}
#endif
#endif
