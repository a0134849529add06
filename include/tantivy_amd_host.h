/*
 * tantivy_amd_host.h — flat C entry points over the C++ host mirror (tantivy_amd/host/): the
 * Searcher / Query::weight / TopDocs surface of the reference (src/core/searcher.rs:180-238,
 * src/query/query.rs:128-160, src/collector/top_score_collector.rs:61-96) driving the device
 * through tantivy_amd.h.  Used by the Python tests and bench; a C++ application includes
 * tantivy_amd/host/searcher.hpp directly.
 */
#ifndef TANTIVY_AMD_HOST_H
#define TANTIVY_AMD_HOST_H
#include "tantivy_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tqh_searcher tqh_searcher;

/* TermInfo (src/postings/term_info.rs:10-17) keyed by a caller-chosen term id; stands in for the
 * reference's TermDictionary, which stays with the caller (termdict untouched). */
typedef struct tqh_term_info {
  uint32_t term_id;
  uint32_t doc_freq;
  uint64_t postings_start, postings_end, positions_start, positions_end;
} tqh_term_info;

/* mode: the values of enum tq_mode (tantivy_amd.h) —
 *       TQ_MODE_AND (0) = BooleanQuery of Must term clauses, TQ_MODE_OR (1) = BooleanQuery of
 *       Should term clauses, TQ_MODE_PHRASE (2) = PhraseQuery (offsets 0..n unless phrase_offsets
 *       given), TQ_MODE_BOOL (3) = BooleanQuery with per-term occurs (0 Should, 1 Must,
 *       2 MustNot); terms sharing a clause_of value form one nested query: a union of its terms
 *       (`+a +(b OR c)`) or, with nested_occurs, a BooleanQuery with an occur per term
 *       (`+a +(+b -c)`: clause_of {0,1,1}, occurs {1,1,1}, nested_occurs {255,1,2}; 255 / NULL =
 *       Should).  A Must clause holding a nested query with a Must term is hoisted into its
 *       parent (boolean_weight.rs:308-431: same docs, same score terms); the other nestings of
 *       depth 2 — an intersection inside a union or under MustNot, nested MustNot / optional
 *       terms, a nested minimum_number_should_match (clause_min_should) — run on the device over
 *       the lists' bitmaps (TQ_KERNEL_TREE); deeper trees and phrases inside boolean queries are
 *       TQ_ERR_UNSUPPORTED and stay on tantivy's CPU scorer;
 *       min_should_match as BooleanQuery::set_minimum_number_should_match —
 *       plus TQH_MODE_TERM (4) = TermQuery.  Anything else: TQ_ERR_INVALID. */
#define TQH_MODE_TERM 4
typedef struct tqh_query {
  uint8_t mode;
  uint32_t n_terms;
  const uint32_t *terms;
  const uint32_t *phrase_offsets;
  const uint8_t *occurs;
  const uint8_t *clause_of;
  uint32_t min_should_match;
  const float *boosts; /* per term: the TermQuery is wrapped in BoostQuery(boost) (boost_query.rs);
                          PHRASE: boosts[0] wraps the PhraseQuery; NULL = 1 */
  const uint8_t *nested_occurs; /* TQ_MODE_BOOL: occur of a term inside its clause_of group (read for
                                   groups of >= 2 terms); NULL = unions */
  const uint8_t *clause_min_should; /* TQ_MODE_BOOL: minimum_number_should_match of the nested query of a
                                       clause_of group, 16 entries indexed by the clause_of value; NULL = 0 */
  const uint8_t *atom_of; /* TQ_MODE_BOOL: terms of one clause_of group sharing an atom_of value are ONE member of the
                             nested query — an intersection of terms one level further down (`+a +((+b +c) d)`:
                             clause_of {0,1,1,1}, atom_of {0,1,1,2}, nested_occurs {-,0,0,0}); NULL = every term its own */
} tqh_query;

const char *tqh_last_error(void);
int tqh_searcher_new(tq_ctx *ctx, tqh_searcher **out);
void tqh_searcher_free(tqh_searcher *s);
/* SegmentReader for one field of one segment; segment_ord = order of addition. */
int tqh_searcher_add_segment(tqh_searcher *s, int device, uint32_t max_doc, uint8_t record_option,
                             const uint8_t *idx, size_t idx_len, const uint8_t *pos,
                             size_t pos_len, const uint8_t *fieldnorm, size_t fn_len,
                             const tqh_term_info *terms, uint32_t n_terms);
/* Bm25StatisticsProvider input for segments held by other ranks (src/query/bm25.rs:11-50):
 * their max_doc, total_num_tokens and per-term doc_freq are added to the local sums. */
int tqh_searcher_add_remote_stats(tqh_searcher *s, uint64_t max_doc, uint64_t total_num_tokens,
                                  const uint32_t *term_ids, const uint32_t *doc_freqs,
                                  uint32_t n_terms);
/* Query::weight for every query of a batch (global BM25 statistics + executor choice). */
int tqh_prepare_batch(tqh_searcher *s, const tqh_query *queries, uint32_t n);
/* Query::weight of the NEXT batch while the current one executes on another thread (a second slot; thread-safe next to
 * the tqh_collect_* / tqh_search_* calls of the current batch), and the switch to it: tqh_commit_next is called by the
 * executing thread between two batches, after tqh_prepare_batch_next has returned. */
int tqh_prepare_batch_next(tqh_searcher *s, const tqh_query *queries, uint32_t n);
int tqh_commit_next(tqh_searcher *s);
/* Searcher::search of the prepared batch with TopDocs::with_limit(limit).and_offset(offset):
 * outputs [n][limit], (score desc, segment_ord asc, doc asc). */
int tqh_search_prepared(tqh_searcher *s, uint32_t offset, uint32_t limit, float *scores,
                        uint32_t *segment_ords, uint32_t *docs, uint32_t *counts);
/* Searcher::search called from n_threads host threads at once, one query per call — the reference's
 * own call pattern (src/core/searcher.rs:180-238): thread t takes queries t, t + n_threads, ...;
 * every call is Query::weight + one tq_search_one per segment (the calls of concurrent threads ride
 * in shared launches) + merge_fruits.  Outputs [n][limit] as tqh_search_prepared; latency_ms[q] =
 * wall time of query q's call (may be NULL); *wall_ms = the whole run (may be NULL). */
int tqh_search_concurrent(tqh_searcher *s, const tqh_query *queries, uint32_t n, uint32_t offset,
                          uint32_t limit, uint32_t n_threads, float *scores, uint32_t *segment_ords,
                          uint32_t *docs, uint32_t *counts, float *latency_ms, double *wall_ms);
/* Searcher::search(&query, &Count) of the prepared batch (src/collector/count_collector.rs:39-80):
 * counts[q] = alive matching docs summed over the segments. */
int tqh_count_prepared(tqh_searcher *s, uint64_t *counts);
/* Collector::collect_segment of the prepared batch on one segment: [n][k] sorted. */
int tqh_collect_segment_prepared(tqh_searcher *s, uint32_t segment_ord, uint32_t k, float *scores,
                                 uint32_t *docs, uint32_t *counts);
/* Same with DEVICE output pointers, enqueued on hip_stream without a host sync. */
int tqh_collect_segment_prepared_device(tqh_searcher *s, uint32_t segment_ord, uint32_t k,
                                        float *d_scores, uint32_t *d_docs, uint32_t *d_counts,
                                        void *hip_stream);
/* Bm25Weight::for_terms(...).boost_by(boost): weight + 256-entry tf cache. */
int tqh_bm25_for_terms(const uint64_t *term_doc_freqs, uint32_t n_terms, uint64_t total_num_docs,
                       uint64_t total_num_tokens, float boost, float *weight_out,
                       float *cache_out);
tq_segment *tqh_segment_raw(tqh_searcher *s, uint32_t segment_ord);
uint32_t tqh_term_handle(tqh_searcher *s, uint32_t segment_ord, uint32_t term_id);

/* ---- term dictionary values ----
 * replaces: TermInfoStore::{open, get} and TermInfoStoreWriter (src/termdict/fst_termdict/
 * term_info_store.rs:130-163, :166-290).  The FST (term bytes -> ordinal) stays with tantivy-fst;
 * everything after the ordinal is here, so a query batch crosses the boundary as ordinals.
 * tqh_term_dictionary_values: where the store sits inside a field's term dictionary file
 * (TermDictionary::open, termdict.rs:123-140 inside the wrapper of src/termdict/mod.rs:82-98:
 * fst | store | u64 store_len | u32 fst version | u32 dictionary type). */
typedef struct tqh_term_info_store tqh_term_info_store;
int tqh_term_dictionary_values(const uint8_t *file, size_t len, uint64_t *store_off,
                               uint64_t *store_len);
int tqh_term_info_store_open(const uint8_t *bytes, size_t len, tqh_term_info_store **out);
void tqh_term_info_store_free(tqh_term_info_store *s);
uint64_t tqh_term_info_store_num_terms(const tqh_term_info_store *s);
/* out[i] = TermInfo of term_ords[i] (term_id = the ordinal) */
int tqh_term_info_store_get(const tqh_term_info_store *s, const uint64_t *term_ords, uint32_t n,
                            tqh_term_info *out);
/* TermInfoStoreWriter::{write_term_info*, serialize} over infos in ordinal order; *out_len =
 * bytes needed (also when the buffer was too small). */
int tqh_term_info_store_write(const tqh_term_info *infos, uint32_t n, uint8_t *out,
                              uint64_t out_cap, uint64_t *out_len);
/* tqh_searcher_add_segment with the segment's TermInfoStore instead of a TermInfo array: term
 * ids are term ordinals and TermInfos are decoded on first use. */
int tqh_searcher_add_segment_with_store(tqh_searcher *s, int device, uint32_t max_doc,
                                        uint8_t record_option, const uint8_t *idx, size_t idx_len,
                                        const uint8_t *pos, size_t pos_len,
                                        const uint8_t *fieldnorm, size_t fn_len,
                                        const uint8_t *store, size_t store_len);

/* The same for sub-files that are resident on `device` (tq_segment_upload_device): device
 * pointers; total_num_tokens = the value of the .idx sub-file's 8-byte header. */
int tqh_searcher_add_segment_device_with_store(tqh_searcher *s, int device, uint32_t max_doc,
                                               uint8_t record_option, const uint8_t *d_idx,
                                               size_t idx_len, const uint8_t *d_pos, size_t pos_len,
                                               const uint8_t *d_fieldnorm, size_t fn_len,
                                               uint64_t total_num_tokens, const uint8_t *store,
                                               size_t store_len);

#ifdef __cplusplus
}
#endif
#endif
