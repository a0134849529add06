/*
 * tantivy_amd.h — C ABI of the MI355X-native query-execution path for tantivy.
 *
 * This is the drop-in boundary: everything below
 * `Weight::for_each_pruning` / `Collector::collect_segment` — posting-list decode, AND / OR /
 * exact-phrase evaluation, BM25, per-segment top-k and the cross-segment merge — runs on the
 * GPU behind these entry points.  Plain pointers and sizes only; callable from Rust through a
 * thin `extern "C"` crate (binding shown in INTEGRATION.md), from C++ (tantivy_amd/host/) and
 * from Python ctypes (tantivy_amd/binding.py).
 *
 * Ownership: inputs are borrowed for the duration of the call (segment bytes are copied to HBM
 * by tq_segment_upload); outputs are caller-allocated; handles are opaque and freed explicitly.
 * Every function returns TQ_OK (0) or an error code; tq_last_error() gives the message of the
 * last failure on the calling thread.  Nothing here panics or throws across the boundary.
 *
 * Thread-safety: every entry point that takes a tq_segment locks it for the duration of the call
 * (term table, planner scratch and staging buffers are per segment), so concurrent callers on one
 * segment are serialised, never undefined; different segments run concurrently — tantivy's own
 * "one task per segment" executor model (src/core/executor.rs:44-106).  Many threads each issuing
 * single queries should use tq_search_one / tq_submit + tq_wait below: their queries are coalesced
 * into batched launches instead of queueing up one launch each.
 * Planning runs on the calling thread.  TQ_PLAN_THREADS=N (N > 1) lets a process-wide pool of N - 1
 * helper threads take slabs of a large batch; they touch only the call's own planner scratch and
 * the call returns after the last slab (off by default: no bench workload planned faster with it,
 * and a descheduled helper stalls the batch).
 * Streams: consecutive batches on one segment share its scratch buffers.  A batch enqueued on
 * another stream than the previous one first waits (stream-side, no host block) for that batch;
 * the host paths (tq_search_batch, tq_count_batch, tq_decode_*, tq_segment_set_alive_bitset,
 * tq_last_batch_*) wait for whatever the segment still has in flight on any stream.
 */
#ifndef TANTIVY_AMD_H
#define TANTIVY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TQ_TERMINATED 0x7FFFFFFFu /* src/docset.rs:12 */
#define TQ_MAX_TERMS 16u          /* terms per device query */
#define TQ_MAX_K 1024u            /* largest per-segment k (offset+limit) the device heap holds */

enum tq_status {
  TQ_OK = 0,
  TQ_ERR_INVALID = 1,     /* bad argument */
  TQ_ERR_HIP = 2,         /* HIP runtime failure (maps to TantivyError::SystemError) */
  TQ_ERR_FORMAT = 3,      /* malformed index bytes (maps to TantivyError::DataCorruption) */
  TQ_ERR_UNSUPPORTED = 4, /* query shape the device path does not take (caller falls back) */
  TQ_ERR_NO_DEVICE = 5
};

/* IndexRecordOption of the *indexed* field (src/schema/index_record_option.rs); it fixes the
 * skip-entry size (5/8/12 B, src/postings/skip.rs:205-253). */
enum tq_record_option { TQ_BASIC = 0, TQ_WITH_FREQS = 1, TQ_WITH_FREQS_AND_POSITIONS = 2 };

/* Which executor the reference would pick (src/query/boolean_query/boolean_weight.rs:236-431):
 * AND    = all-Must term clauses   -> block_wand_intersection
 * OR     = all-Should term clauses -> block_wand / BufferedUnionScorer
 * PHRASE = PhraseQuery (slop 0)    -> PhraseScorer */
enum tq_mode { TQ_MODE_AND = 0, TQ_MODE_OR = 1, TQ_MODE_PHRASE = 2, TQ_MODE_BOOL = 3 };
/* TQ_MODE_BOOL = BooleanQuery whose clauses are terms or BooleanQuerys of terms (unions of terms; since round 5
 * also nested intersections / exclusions / minimums: tq_query.nested_occurs), with mixed occurs
 * (`+a b -c`, `+a +(b OR c)`, `+(a OR b) +(c OR d)`) and minimum_number_should_match: the
 * Intersection / RequiredOptionalScorer / Exclude / Disjunction part of
 * BooleanWeight::complex_scorer (boolean_weight.rs:236-431, intersection.rs:20-56,
 * reqopt_scorer.rs:85-98, exclude.rs, disjunction.rs): docs = AND of the Must clauses (or OR of
 * the Should clauses when there is no Must), with at least min_should_match Should clauses
 * matching, minus the MustNot clauses; score = Must clauses summed cheapest first
 * (left + right + sum(others)) + the matching Should terms.  Values of tq_query.occurs follow
 * src/query/occur.rs. */
enum tq_occur { TQ_SHOULD = 0, TQ_MUST = 1, TQ_MUST_NOT = 2 };
/* tq_query.nested_occurs[i] | TQ_NESTED_PHRASE: the terms of the clause that share atom_of[i] are a PhraseQuery */
#define TQ_NESTED_PHRASE 0x10
/* tq_query.nested_occurs[i] | TQ_NESTED_ANY: the terms of the clause that share atom_of[i] are a UNION — a BooleanQuery
 * of Should terms one level further down, `(+(b c) +d) e`, `+a +(+(b c) +d)`: present where ANY of its terms is, scoring
 * the sum of the present ones (BufferedUnionScorer's SumCombiner) */
#define TQ_NESTED_ANY 0x20

typedef struct tq_ctx tq_ctx;
typedef struct tq_segment tq_segment;
typedef uint32_t tq_term_handle;
#define TQ_TERM_ABSENT 0xFFFFFFFFu /* term not in this segment (TermInfo lookup returned None) */

/* One query against one segment.  Replaces the per-segment scorer tree the reference builds in
 * BooleanWeight::complex_scorer / TermWeight::specialized_scorer / PhraseWeight::phrase_scorer.
 * The BM25 statistics are computed by the caller from *global* (cross-segment) counts exactly as
 * Bm25Weight::for_terms does (src/query/bm25.rs:95-129): */
typedef struct tq_query {
  uint32_t n_terms;              /* 1..TQ_MAX_TERMS */
  const tq_term_handle *terms;   /* from tq_term_prepare, or TQ_TERM_ABSENT */
  const float *weights;          /* AND/OR: n_terms x (idf*(1+K1)*boost); PHRASE: weights[0] */
  const float *tf_cache;         /* Bm25Weight.cache: 256 f32, K1*(1-B+B*fieldnorm/avg) */
  uint8_t mode;                  /* enum tq_mode */
  const uint32_t *phrase_offsets; /* PHRASE: term offsets inside the phrase; BOOL with phrase atoms: per term; else NULL */
  uint32_t k;                    /* TopDocs offset+limit, 1..TQ_MAX_K */
  const uint8_t *occurs;         /* TQ_MODE_BOOL: n_terms x enum tq_occur; else NULL */
  const uint8_t *clause_of;      /* TQ_MODE_BOOL: clause index per term (terms sharing a value form
                                    one nested all-Should BooleanQuery, `+a +(b OR c)`); NULL =
                                    every term is its own clause */
  uint32_t min_should_match;     /* TQ_MODE_BOOL: BooleanQuery::minimum_number_should_match */
  /* TQ_MODE_BOOL, one level of nesting beyond unions of terms (round 5): a clause_of group is a nested
   * BooleanQuery of terms, every term with its own occur INSIDE it — `+a +(+b +c)`, `(+b +c) d`, `+a -(+b +c)`,
   * `+a +(+b c -d)` — and its own minimum_number_should_match.  Both NULL (or every nested occur TQ_SHOULD and every
   * nested minimum <= 1) = unions of terms as before.  Evaluated over the lists' bitmaps (TQ_KERNEL_TREE). */
  const uint8_t *nested_occurs;      /* n_terms x enum tq_occur: the term's occur inside its clause_of group; or NULL */
  const uint8_t *clause_min_should;  /* TQ_MAX_TERMS entries indexed by clause_of value: the nested query's
                                        minimum_number_should_match; or NULL */
  const uint8_t *atom_of;            /* n_terms values, or NULL: terms of one clause_of group that share an atom_of
                                        value form a CONJUNCTION — a BooleanQuery of Must terms one level further down,
                                        `+a +((+b +c) d)`: present where all its terms are, scoring their sum; its occur
                                        inside the group is nested_occurs of its terms (equal for all of them).  NULL =
                                        every term is its own member of its group.
                                        A PHRASE inside the boolean query (`+"a b" +c`, `"a b" c`, `+c -"a b"`,
                                        `+a +("b c" d)`; PhraseQuery as a clause of BooleanQuery: PhraseScorer under
                                        Intersection / union / Exclude, phrase_scorer.rs:347-587): the terms of the
                                        phrase share an atom_of value, carry nested_occurs | TQ_NESTED_PHRASE, their
                                        phrase_offsets[i] (0 on the other terms) and — each of them — the PHRASE's
                                        weight ((1 + K1) * boost * sum of idfs) in weights[i].  2..8 terms per phrase; a
                                        phrase that is a clause of its own is a clause_of group of one atom.
                                        A UNION one level further down (nested_occurs | TQ_NESTED_ANY, round 6): see above. */
} tq_query;

/* ---- lifecycle ---- */
/* replaces: nothing in the reference (device bring-up).  device_ids==NULL => {0}. */
int tq_init(const int *device_ids, int n_devices, tq_ctx **out);
void tq_shutdown(tq_ctx *ctx);
const char *tq_last_error(void);

/* ---- segment residency ----
 * replaces: SegmentReader::inverted_index(field) + InvertedIndexReader::new
 * (src/index/segment_reader.rs:221-283, src/index/inverted_index_reader.rs:66-81) and
 * FieldNormReader::open (src/fieldnorm/reader.rs:99-105): the field's raw sub-files are copied
 * to HBM unchanged.  idx = the field's `.idx` sub-file including its 8-byte total_num_tokens
 * header; pos = the field's `.pos` sub-file (NULL if no positions); fieldnorm = max_doc bytes
 * (NULL => FieldNormReader::constant(max_doc, 1), src/query/term_query/term_weight.rs:209-219).
 * The natural caller is a Warmer (src/reader/warming.rs:14-20). */
int tq_segment_upload(tq_ctx *ctx, int device, uint32_t max_doc, const uint8_t *idx,
                      size_t idx_len, const uint8_t *pos, size_t pos_len,
                      const uint8_t *fieldnorm, size_t fn_len, uint8_t record_option,
                      tq_segment **out);
/* The same for sub-files that are ALREADY in device memory on `device` (written by
 * tq_encode_postings_device / tq_encode_positions_device, or copied there by the caller): the
 * bytes are copied device to device and the library keeps no host copy — tq_term_prepare then
 * walks skip lists and positions headers in HBM (tq_prepare.hip) and builds the dense-list
 * tables with device scans; only a few dozen bytes of facts per term cross back to the host.
 * d_idx includes the 8-byte total_num_tokens header like idx above. */
int tq_segment_upload_device(tq_ctx *ctx, int device, uint32_t max_doc, const uint8_t *d_idx,
                             size_t idx_len, const uint8_t *d_pos, size_t pos_len,
                             const uint8_t *d_fieldnorm, size_t fn_len, uint8_t record_option,
                             tq_segment **out);
void tq_segment_free(tq_segment *seg);

/* replaces: InvertedIndexReader::read_postings_from_terminfo + BlockSegmentPostings::open +
 * SkipReader::new + PositionReader::open (inverted_index_reader.rs:204-247,
 * block_segment_postings.rs:97-140, skip.rs:131-151, positions/reader.rs:43-56).
 * Arguments are the fields of TermInfo (src/postings/term_info.rs:10-17): ranges are relative to
 * the sub-file body (after the 8-byte header for .idx).  Walks the sequential skip list once and
 * keeps it on the device as a random-access table; idempotent per (postings_off). */
int tq_term_prepare(tq_segment *seg, uint64_t postings_off, uint32_t postings_len,
                    uint64_t positions_off, uint32_t positions_len, uint32_t doc_freq,
                    tq_term_handle *out);
/* The same for many terms at once (the new terms of a query batch): one staged upload, one wait, one launch for the
 * doc signatures of the lists without a column — tq_term_prepare costs a blocking copy and two launches per term.
 * infos[i] = the fields of the term's TermInfo (postings/term_info.rs:10-17); out[i] = its handle (terms already
 * prepared: the handle they have).  Replaces n calls of InvertedIndexReader::read_block_postings_from_terminfo
 * (src/index/inverted_index_reader.rs:204-247) + BlockSegmentPostings::open (block_segment_postings.rs:97-140). */
typedef struct tq_term_info {
  uint64_t postings_off;
  uint64_t positions_off;
  uint32_t postings_len, positions_len;
  uint32_t doc_freq;
  uint32_t pad_;
} tq_term_info;
int tq_term_prepare_batch(tq_segment *seg, const tq_term_info *infos, uint32_t n, tq_term_handle *out);
/* Optional, before the first tq_term_prepare: names the posting lists (by postings_off) that get
 * the 40 columns of the segment's doc matrix — the caller knows every term's doc_freq from the
 * term dictionary and passes the densest lists.  Without it the columns go to the first 40 dense
 * lists that happen to be prepared, i.e. the layout (and 5..15 % of the union kernels' time)
 * depends on the order of the first queries.  Lists beyond the 40 named are ignored. */
int tq_segment_reserve_columns(tq_segment *seg, const uint64_t *postings_offs, uint32_t n);

/* ---- search ----
 * replaces, for each query: TopBySortKeyCollector::collect_segment ->
 * SortBySimilarityScore::collect_segment_top_k -> Weight::for_each_pruning -> {block_wand,
 * block_wand_intersection, for_each_pruning_scorer(PhraseScorer)} -> TopNHeap
 * (src/collector/sort_key_top_collector.rs:62-73, sort_key/sort_by_score.rs:35-161,
 *  src/query/boolean_query/{block_wand_union,block_wand_intersection}.rs,
 *  src/query/phrase_query/phrase_scorer.rs).
 * Output per query q (row stride = out_stride >= max k): the segment's top-k sorted by
 * (score desc, doc asc) — i.e. TopNHeap::into_vec after the sort merge_top_k applies — padded
 * with (0.0, TQ_TERMINATED); out_counts[q] = number of real hits.  Host buffers. */
int tq_search_batch(tq_segment *seg, const tq_query *queries, uint32_t n_queries,
                    uint32_t out_stride, float *out_scores, uint32_t *out_docs,
                    uint32_t *out_counts);

/* Same, but the outputs are DEVICE pointers on the segment's device and the call only enqueues
 * work on `hip_stream` (a hipStream_t; NULL = the segment's own stream) without a final host
 * synchronisation — for callers that feed an RCCL all-gather (tq_merge_topk_device) next. */
int tq_search_batch_device(tq_segment *seg, const tq_query *queries, uint32_t n_queries,
                           uint32_t out_stride, float *d_out_scores, uint32_t *d_out_docs,
                           uint32_t *d_out_counts, void *hip_stream);

/* Per-call execution options: everything a call needs that is not part of the queries.  Carried
 * per call (not as segment state) so that two Searchers / threads sharing the statistics of
 * different indexes never race on a segment's options.
 *   exhaustive       -1 = the segment's "exhaustive" option (tq_set_option), 0 = block-max pruning
 *                    as block_wand / block_wand_intersection do, 1 = score every match
 *   bound_slack_ppm  TQ_OPT_DEFAULT = the segment's option; else block-max bounds are widened by
 *                    (1 + ppm * 1e-6) for THIS call.  Callers whose Bm25Weights come from global
 *                    (cross-segment) statistics MUST pass ((1 + d)^2 - 1) * 1e6 with d = the
 *                    relative difference between the global and this segment's average
 *                    fieldnorm, or pruning is only as exact as the reference's own
 *                    (term_scorer.rs:58-70); the host mirror (Searcher) does. */
#define TQ_OPT_DEFAULT 0xFFFFFFFFu
typedef struct tq_search_opts {
  int32_t exhaustive;
  uint32_t bound_slack_ppm;
} tq_search_opts;
/* tq_search_batch / tq_search_batch_device with per-call options (opts == NULL: segment defaults). */
int tq_search_batch_opts(tq_segment *seg, const tq_query *queries, uint32_t n_queries,
                         uint32_t out_stride, float *out_scores, uint32_t *out_docs,
                         uint32_t *out_counts, const tq_search_opts *opts);
int tq_search_batch_device_opts(tq_segment *seg, const tq_query *queries, uint32_t n_queries,
                                uint32_t out_stride, float *d_out_scores, uint32_t *d_out_docs,
                                uint32_t *d_out_counts, const tq_search_opts *opts,
                                void *hip_stream);

/* ---- concurrent single-query entry (tantivy's own call pattern) ----
 * replaces: Collector::collect_segment(&dyn Weight, segment_ord, &SegmentReader) as
 * Searcher::search_with_executor calls it — from any number of threads at once, one query per call
 * (src/core/searcher.rs:180-238, src/collector/mod.rs:173-183; `Weight: Send + Sync`,
 * src/query/weight.rs:66).  One query per launch is the ~0.3 ms regime of the device; these entry
 * points coalesce the single queries of concurrent callers into the batched launch:
 *   tq_submit      puts one query on the segment's pending list and returns a ticket.  The query's
 *                  arrays (terms, weights, tf_cache, ...) and the output buffers (k scores / docs,
 *                  one count) are borrowed until tq_wait returns.
 *   tq_wait        blocks until the launch that carried the query has finished and frees the
 *                  ticket.  Whoever waits while no batch is running LEADS the next one: it takes
 *                  everything pending that runs under the same options, issues ONE tq_search_batch
 *                  for it and hands every caller its rows (callers keep arriving while the previous
 *                  batch runs — nobody waits for a batch to fill).  Work only happens inside
 *                  tq_wait: a ticket that is never waited for is never run (and leaks).
 *   tq_search_one  = tq_submit + tq_wait: the blocking, thread-safe single-query call.
 * A query the device does not take fails alone (TQ_ERR_*; tq_last_error on the waiting thread), its
 * batch mates are re-run without it. */
typedef struct tq_ticket tq_ticket;
int tq_submit(tq_segment *seg, const tq_query *query, const tq_search_opts *opts, float *out_scores,
              uint32_t *out_docs, uint32_t *out_count, tq_ticket **out);
int tq_wait(tq_ticket *ticket);
int tq_search_one(tq_segment *seg, const tq_query *query, const tq_search_opts *opts,
                  float *out_scores, uint32_t *out_docs, uint32_t *out_count);
typedef struct tq_submit_stats {
  uint64_t batches;   /* launches issued for submitted queries */
  uint64_t queries;   /* queries they carried */
  uint64_t max_batch; /* the largest of them */
} tq_submit_stats;
int tq_get_submit_stats(tq_segment *seg, tq_submit_stats *out, int reset);

/* replaces: TopBySortKeyCollector::merge_fruits -> merge_top_k
 * (src/collector/sort_key_top_collector.rs:54-95; ordering top_score_collector.rs:590-600):
 * merges n_segments per-segment results (layout [segment][query][stride], as gathered by an
 * all-gather over ranks) into the global top-(offset+limit) by (score desc, segment_ord asc,
 * doc asc), then skips `offset`.  Host version. */
int tq_merge_topk(const float *scores, const uint32_t *docs, const uint32_t *counts,
                  uint32_t n_segments, uint32_t n_queries, uint32_t stride, uint32_t offset,
                  uint32_t limit, float *out_scores, uint32_t *out_segment_ords,
                  uint32_t *out_docs, uint32_t *out_counts);
/* Device version of the same merge (inputs/outputs are device pointers on `device`; enqueued on
 * hip_stream).  segment_ords[s] gives the segment ordinal of slab s (NULL => s). */
int tq_merge_topk_device(tq_ctx *ctx, int device, const float *d_scores, const uint32_t *d_docs,
                         const uint32_t *d_counts, const uint32_t *segment_ords,
                         uint32_t n_segments, uint32_t n_queries, uint32_t stride,
                         uint32_t offset, uint32_t limit, float *d_out_scores,
                         uint32_t *d_out_segment_ords, uint32_t *d_out_docs,
                         uint32_t *d_out_counts, void *hip_stream);
/* The merged rows' way back to the host: an asynchronous copy of `bytes` from device memory on `device` to PINNED host
 * memory, enqueued on hip_stream behind the merge.  replaces: nothing in the reference (its fruits are host vectors);
 * what a Rust host does with hipMemcpyAsync after tq_merge_topk_device. */
int tq_copy_to_host_async(tq_ctx *ctx, int device, void *dst_pinned_host, const void *src_device, size_t bytes,
                          void *hip_stream);

/* ---- cross-GPU exchange (one segment set per GPU) ----
 * replaces: the fan-in of Searcher::search_with_executor (src/core/searcher.rs:230-235: the
 * executor maps collect_segment over the segments, src/core/executor.rs:61-104, and hands the
 * fruits to merge_fruits, src/collector/sort_key_top_collector.rs:54-95) when the segments live
 * on different GPUs, one process per GPU.  The only collective of the path: an RCCL all-gather of
 * the per-segment top-k lists over xGMI.  BM25 statistics need none (sums known to the host
 * before dispatch, src/query/bm25.rs:27-50).
 *
 * tq_comm_unique_id : ncclGetUniqueId on ONE rank; the host sends the 128 bytes to the others
 *                     (any channel: the Rust host's own control plane).
 * tq_comm_init      : ncclCommInitRank(world, id, rank) on `device`; collective over all ranks.
 * tq_allgather_topk : every rank passes the [n_rows][stride] result slabs of its local segments
 *                     (n_rows = local segments x queries; scores / docs, and [n_rows] counts) as
 *                     written by tq_search_batch_device, and receives all ranks' slabs as
 *                     [world][n_rows][stride] / [world][n_rows] — with rank r holding segments
 *                     r*S..r*S+S-1 that is the [segment][query][stride] input of
 *                     tq_merge_topk_device.  Device pointers; enqueued on hip_stream (stream
 *                     ordered after the searches, no host synchronisation); one grouped RCCL
 *                     launch.  librccl is opened at run time (TQ_RCCL_LIB overrides the path);
 *                     TQ_ERR_UNSUPPORTED if it cannot be loaded. */
#define TQ_COMM_ID_BYTES 128
typedef struct tq_comm tq_comm;
int tq_comm_unique_id(uint8_t *id_out /* TQ_COMM_ID_BYTES */);
int tq_comm_init(tq_ctx *ctx, int device, const uint8_t *id, int rank, int world, tq_comm **out);
void tq_comm_free(tq_comm *comm);
int tq_comm_info(const tq_comm *comm, int *rank, int *world, const char **library);
int tq_allgather_topk(tq_comm *comm, const float *d_scores, const uint32_t *d_docs,
                      const uint32_t *d_counts, uint32_t n_rows, uint32_t stride,
                      float *d_all_scores, uint32_t *d_all_docs, uint32_t *d_all_counts,
                      void *hip_stream);

/* ---- codec access (parity tests / tooling) ----
 * replaces: BlockSegmentPostings::load_block over a whole list (block_segment_postings.rs:343-391;
 * BitPacker4x decode + strict-delta prefix sum + vint tail).  docs/tfs: doc_freq u32 each (host). */
int tq_decode_postings(tq_segment *seg, tq_term_handle term, uint32_t *docs, uint32_t *tfs);
/* replaces: PositionReader::read over the whole term (positions/reader.rs:104-148): raw position
 * deltas, n_positions values (host); *n_out = number available. */
int tq_decode_position_deltas(tq_segment *seg, tq_term_handle term, uint32_t *out, uint64_t cap,
                              uint64_t *n_out);

/* ---- codec writers (segment finalisation / merges on the device) ----
 * replaces: PostingsSerializer::{new_term, write_doc, close_term} (src/postings/serializer.rs:
 * 342-481; skip entries src/postings/skip.rs:55-88; block-max selection serializer.rs:404-428)
 * for a batch of terms.  Term t owns docs/tfs[term_starts[t] .. term_starts[t+1]) (doc ids
 * strictly increasing, tfs >= 1; tfs ignored / may be NULL with TQ_BASIC).  fieldnorm_ids =
 * num_docs bytes of the field's fieldnorm sub-file or NULL (then no block-max metadata, as when
 * the serializer has no FieldNormReader); avg_fieldnorm = total_num_tokens / num_docs
 * (serializer.rs:130-135).  Writes the posting lists back to back, exactly the bytes the
 * reference writes after the 8-byte total_num_tokens header of the field's .idx sub-file;
 * out_term_starts[t] .. [t+1] is TermInfo::postings_range relative to that point;
 * *out_len = bytes needed (valid also when TQ_ERR_INVALID reports out_cap too small). */
typedef struct tq_encoder tq_encoder;
int tq_encoder_create(tq_ctx *ctx, int device, tq_encoder **out);
void tq_encoder_free(tq_encoder *enc);
int tq_encode_postings(tq_encoder *enc, uint32_t n_terms, const uint64_t *term_starts,
                       const uint32_t *docs, const uint32_t *tfs, const uint8_t *fieldnorm_ids,
                       uint32_t num_docs, float avg_fieldnorm, uint8_t record_option, uint8_t *out,
                       uint64_t out_cap, uint64_t *out_term_starts, uint64_t *out_len);
/* replaces: PositionSerializer::{write_positions_delta, close_term} (src/positions/serializer.rs:
 * 46-91) for a batch of terms: term t owns position_deltas[term_starts[t] .. term_starts[t+1])
 * (the within-document deltas of all its docs, concatenated); output = the terms' .pos bytes back
 * to back, out_term_starts = TermInfo::positions_range. */
int tq_encode_positions(tq_encoder *enc, uint32_t n_terms, const uint64_t *term_starts,
                        const uint32_t *position_deltas, uint8_t *out, uint64_t out_cap,
                        uint64_t *out_term_starts, uint64_t *out_len);
/* Same with inputs and outputs resident in device memory (d_*); term_starts is needed on both
 * sides (host copy: block planning).  Only the 8-byte total crosses back to the host. */
int tq_encode_postings_device(tq_encoder *enc, uint32_t n_terms, const uint64_t *term_starts,
                              const uint64_t *d_term_starts, const uint32_t *d_docs,
                              const uint32_t *d_tfs, const uint8_t *d_fieldnorm_ids,
                              uint32_t num_docs, float avg_fieldnorm, uint8_t record_option,
                              uint8_t *d_out, uint64_t out_cap, uint64_t *d_out_term_starts,
                              uint64_t *out_len, void *hip_stream);
int tq_encode_positions_device(tq_encoder *enc, uint32_t n_terms, const uint64_t *term_starts,
                               const uint64_t *d_term_starts, const uint32_t *d_position_deltas,
                               uint8_t *d_out, uint64_t out_cap, uint64_t *d_out_term_starts,
                               uint64_t *out_len, void *hip_stream);
/* HIP-event time of the last call's kernels (measure + scan + write), milliseconds. */
int tq_encoder_last_kernel_ms(tq_encoder *enc, float *ms);

/* ---- deletes and counting ----
 * replaces: SegmentReader::alive_bitset + AliveBitSet::is_alive in the collector callback
 * (src/fastfield/alive_bitset.rs:52-61, src/collector/sort_key/sort_by_score.rs:44-53).  bytes =
 * the segment's `.del` file body as BitSet::serialize writes it (common/src/bitset.rs:215-223:
 * u32 LE max_value == max_doc, then 64-bit words); NULL = no deletes.  Deleted docs are skipped
 * by every search and count on this segment; BM25 statistics still count them, like the
 * reference (bm25.rs:38-45). */
int tq_segment_set_alive_bitset(tq_segment *seg, const uint8_t *bytes, size_t len);
/* replaces: Count collector (src/collector/count_collector.rs:39-80) over the same query shapes:
 * out_counts[q] = number of alive docs matching query q on this segment (the top-k fields of the
 * queries are ignored).  Host buffer. */
int tq_count_batch(tq_segment *seg, const tq_query *queries, uint32_t n_queries,
                   uint32_t *out_counts);
/* Number of docs scored per query by the last tq_search_batch* call on this segment (with
 * "exhaustive" = 1 that is the number of matches, which makes (Count, TopDocs) one pass). */
int tq_last_batch_match_counts(tq_segment *seg, uint32_t *out, uint32_t n_queries);

/* ---- introspection ---- */
typedef struct tq_batch_stats {
  uint64_t algorithmic_bytes; /* sum len(postings_range) [+positions] + matches + 8k */
  uint64_t matches;           /* docs whose BM25 was evaluated (AND/phrase matches, OR union) */
  float kernel_ms;            /* HIP-event time of the scan kernel(s), mean over the batches */
  float total_ms;             /* ... of the whole batch on the stream: launched since the last call */
  uint32_t tiles;
  uint32_t chunks;
  uint32_t batches_averaged;  /* how many batches kernel_ms / total_ms average (<= 16) */
  float host_plan_ms;         /* host CPU time inside tq_search_batch_device (validate + plan + stage +
                                 enqueue; waiting for the previous batch's staging copy excluded),
                                 mean over the calls since the last tq_last_batch_stats */
  uint32_t kernel_mask;       /* TQ_KERNEL_*: the scan-kernel families the last batch was run by */
  uint64_t unique_bytes;      /* each DISTINCT posting list (+ positions, phrases) of the last batch once:
                                 what the batch needs from the index when no byte is read twice; the
                                 per-query sum above counts a list once per query that names it */
} tq_batch_stats;
/* scan-kernel families (tq_batch_stats.kernel_mask) */
#define TQ_KERNEL_AND_DENSE 0x001u    /* and_kernel, every non-leader list with a bitmap */
#define TQ_KERNEL_AND 0x002u          /* and_kernel, general */
#define TQ_KERNEL_UNION 0x004u        /* union_kernel (candidate-driven, per query) */
#define TQ_KERNEL_OR_WINDOWS 0x008u   /* or_kernel (4096-doc windows) */
#define TQ_KERNEL_PHRASE 0x010u       /* phrase_kernel */
#define TQ_KERNEL_PHRASE_SWEEP 0x020u /* phrase_sweep_kernel */
#define TQ_KERNEL_BOOL 0x040u         /* union_kernel, boolean instantiation */
#define TQ_KERNEL_USHARE 0x080u       /* ushare_kernel (unions, term-major for the batch) */
#define TQ_KERNEL_XUNION 0x100u       /* xunion_kernel (unpruned unions, doc-major for the batch) */
#define TQ_KERNEL_ASHARE 0x200u       /* ashare_kernel (intersections, leader-major for the batch) */
#define TQ_KERNEL_BSHARE 0x400u       /* ashare_kernel, boolean leads (TQ_MODE_BOOL, leader-major) */
#define TQ_KERNEL_COUNT_BITMAPS 0x800u /* count_bitmap_kernel (tq_count_batch over bitmap words) */
#define TQ_KERNEL_TREE 0x1000u         /* tree_kernel (nested boolean queries over bitmap words) */
int tq_last_batch_stats(tq_segment *seg, tq_batch_stats *out);
/* Which scan-kernel family (one TQ_KERNEL_* bit) evaluated every query of the last tq_search_batch* call on this
 * segment; needs the option "record_query_kernels" set before that call (diagnosis / parity tooling: bench.py
 * draws its oracle sample from every family a batch ran on). */
int tq_last_batch_query_kernels(tq_segment *seg, uint32_t *out, uint32_t n_queries);
/* Bytes the segment keeps resident in HBM, by kind: the reference's own sub-files (copied
 * verbatim) and the derived side tables of DESIGN.md section 2 — term tables (unrolled skip
 * records, coarse seek tables, decoded vint tails, position-block tables: what SkipReader /
 * PositionReader hold per open term in the reference), bitmaps + rank directories of the dense
 * lists, the doc matrix, position directories — plus the per-batch scratch. */
typedef struct tq_segment_stats {
  uint64_t index_bytes, positions_bytes, fieldnorm_bytes, alive_bytes; /* tantivy's bytes */
  uint64_t term_table_bytes, bitmap_bytes /* + byte-wide tfs */, docmat_bytes, posdir_bytes; /* derived */
  uint64_t scratch_bytes;       /* the segment's own batch scratch: staging, threshold slots, result slabs */
  uint64_t dense_budget_bytes;  /* cap on bitmap (+ byte-wide tf) + docmat + posdir bytes ("dense_budget_x") */
  uint32_t n_terms, n_dense_lists, n_docmat_columns;
  uint32_t probe_evictions;     /* lists whose probe tables ("probe_budget_x") gave their slot to another list */
  uint64_t device_scratch_bytes; /* partial / result lists + staging lists of the term-major launches: ONE set
                                    per device, shared by all its segments (count it once per device) */
} tq_segment_stats;
int tq_segment_get_stats(tq_segment *seg, tq_segment_stats *out);
/* knobs: "exhaustive" (0/1, default 0: block-max pruning as block_wand / block_wand_intersection
 *        do — the reference's execution; 1: score every match — same top-k either way, and
 *        tq_last_batch_match_counts then returns match counts),
 *        "timing" (0/1: record HIP events per batch),
 *        "dense" (0/1, default 1: tq_term_prepare also builds a bitmap + rank directory for lists
 *        with doc_freq >= max_doc/dense_ratio, while bitmaps + doc matrix + position directories
 *        stay below "dense_budget_x" times the segment's bytes), "dense_ratio" (default 128), "use_dense" (0/1, default 1: the AND kernel may use
 *        them),
 *        "or_windows" (-1/0/1, default -1 = auto: unions run window-parallel when exhaustive and
 *        candidate-driven when pruning; 0 / 1 force one kernel),
 *        "bound_slack_ppm" (default 0): block-max bounds are widened by (1 + ppm * 1e-6).  The
 *        block-max pairs were selected under the segment's own average fieldnorm
 *        (serializer.rs:130-135); a caller whose Bm25Weights use global statistics passes
 *        ((1 + d)^2 - 1) * 1e6 with d = relative difference of the two averages and pruning
 *        stays exact (the reference accepts the approximation, term_scorer.rs:58-70),
 *        "dense_budget_x" (default 8: bitmaps + byte-wide term freqs of the dense lists + doc matrix
 *        + doc signatures + position directories together stay below this multiple of the
 *        segment), "docmat" (0/1, default 1: the first 40 dense lists also get a column in a
 *        doc-major matrix: one 8-byte word per doc = fieldnorm id + the doc's membership in those
 *        lists), "docsig" (0/1, default 1: the top 16 bits of those words are a signature of the
 *        prepared lists WITHOUT a column: a clear bit proves the doc is not in such a list), "device_prepare" (0/1, default 0: tq_term_prepare
 *        works on the device copy even when a host copy exists; always so for segments from
 *        tq_segment_upload_device),
 *        "xunion_ratio" (default 64, 0 = never) / "xunion_min_queries" (default 64): with
 *        "exhaustive", pure unions (<= 8 lists, k <= 128, positive weights) whose lists together
 *        hold >= max_doc / xunion_ratio postings are evaluated doc-major for the whole batch —
 *        every list's tf/(tf+norm) built once per 128-doc tile, every query reading its lists' rows —
 *        when the batch has that many of them (up to 255 distinct lists and 8192 queries; lists
 *        without a bitmap are then also kept as plain doc / tf arrays, inside "dense_budget_x"),
 *        "probe_budget_x" (default 16, 0 = never): a boolean query rides in the shared leader-major launch
 *        (TQ_KERNEL_BSHARE) only if every list it probes has a bitmap + byte-wide tfs; for lists below
 *        "dense_ratio" they are built the first time a boolean query (or a nested one: TQ_KERNEL_TREE reaches EVERY
 *        list through a bitmap) names the list, into one of the equal slots of the segment's probe pool — this
 *        multiple of the segment's bytes worth of slots, at least TQ_MAX_TERMS; a list that needs a slot when all are
 *        taken gets the slot of the list used longest ago (tq_segment_stats.probe_evictions), never one the batch
 *        being planned uses: a batch that names more such lists than the budget holds grows the pool (the other
 *        kernels keep treating these lists as sparse),
 *        "rdir_budget_x" (default 4, 0 = never): tq_term_prepare / tq_term_prepare_batch also give a list below
 *        "dense_ratio" with at least 256 postings a RANGE DIRECTORY — one u32 per posting in posting order (16 bits of
 *        the doc id, min(tf, 0xFFFF)) and the number of postings before every 2^S-doc range, S chosen per list so
 *        that a range holds one to two postings: 6-8 bytes per posting, counted in
 *        tq_segment_stats.term_table_bytes — while the directories stay below this multiple of the segment's bytes.
 *        The shared intersection launch (TQ_KERNEL_ASHARE) probes the second list of a 2-term intersection through
 *        it ("is doc d in the list, with which tf": two directory slots and the range's entries) and drops the
 *        leader blocks in whose doc span the directory counts no posting; before, such a list needed a
 *        max_doc / 4-byte slot of the probe pool (19 x the segment's bytes resident after a 65 536-term stream,
 *        5 x with directories),
 *        "count_bitmap_ratio" (default 128, 0 = never): tq_count_batch evaluates a query as a bitwise
 *        expression over bitmap words (4-8 bytes per list per 32 docs, no postings decoded; a list
 *        without a bitmap is scattered into a scratch bitmap once per batch) when the clause a scan
 *        would walk holds at least max_doc / ratio postings per list of the query,
 *        "ashare_min_batch" (default 16; 512 until round 6): intersections take the shared leader-major launch
 *        (TQ_KERNEL_ASHARE) when at least this many queries of the batch qualify for it — below, its
 *        two launches and per-task set-up cost more than sharing the leader blocks saves (round 6, synchronous
 *        batches, shared against per-query: 8 queries 0.23 ms against 0.26; 16: 0.25 against 0.37; 64: 0.34
 *        against 0.44; 256: 0.64 against 0.69 — the batches concurrent single-query callers coalesce into),
 *        "submit_window_us" (default 100): tq_submit / tq_search_one — how long the leader of a batch
 *        holds it open for the callers of the previous batch to come back with their next query
 *        (0 = launch with whatever is pending),
 *        "use_dpp" (0/1: DPP or ds_bpermute prefix sums),
 *        "record_query_kernels" (0/1, default 0: see tq_last_batch_query_kernels), "debug" (-1 = the TQ_DEBUG
 *        environment word, else the kernels' diagnosis word for this segment: work counters / ablations) */
int tq_set_option(tq_segment *seg, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif
