// tq_launch.h — kernel parameter blocks and launch entry points (tq_*.hip <-> tq_api.cpp)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tq_device.h"

// Output locations of a scan launch.  Kept in device memory behind ONE kernel-argument pointer:
// they are touched once per (chunk, query), and as kernel arguments they would pin 8 SGPRs of a
// kernel that is already spilling scalar registers.
struct TqkSinks {
  uint64_t *partials;                 // partial top-k lists, KPL*64 keys each
  unsigned long long *match_counter;  // docs scored by the whole launch
  uint32_t *query_matches;            // per query of the BATCH (through out_index): docs scored
  const uint32_t *out_index;          // launch-group query -> batch query
};

struct TqkScanParams {
  TqdSegment seg;
  const TqdTerm *terms;
  const TqdQuery *queries;      // the launch group's queries, contiguous
  const uint32_t *tile_starts;  // n_queries + 1, non-decreasing
  // one record per chunk, IN LAUNCH ORDER: {first tile, end tile, query of the first tile, chunk
  // id} — one 16-byte scalar load where a permutation lookup, two neighbouring starts and a
  // binary search over the queries' tile ranges (~14 dependent loads) used to be
  const uint4 *chunk_recs;
  const float *caches;          // n_caches x 256
  const TqkSinks *sinks;        // where results go (read at flush time only)
  uint32_t *thr_slots;          // [n_thr_rows][TQD_THR_SLOTS] shared thresholds (pruned mode)
  uint32_t n_queries;
  uint32_t total_tiles;
  uint32_t n_chunks;
  uint32_t exhaustive;  // 1: score every match; 0: block-max pruning allowed
  uint32_t use_dense;   // 0: ignore the dense-list bitmaps (always seek + decode)
  uint32_t max_terms;   // largest n_terms among the launch's queries
  uint32_t or_windows;  // OR: 1 = window-parallel kernel, 0 = candidate-driven kernel
  uint32_t debug;       // TQ_DEBUG ablation bits (profiling only; results are wrong when set)
  uint32_t all_dense;   // AND: every non-leader list of every query of the launch has a bitmap
  uint32_t boolean;     // union kernel: queries carry roles / clauses / min_should (TQ_MODE_BOOL)
  uint32_t small_k;     // every query of the launch has k <= 16
  float bound_slack;    // >= 1: widens block-max bounds when the BM25 statistics are not the segment's own
};

// shared-union launch: a persistent grid of single-wave workgroups pulling tasks
struct TqkShareParams {
  TqdSegment seg;
  const TqdTerm *terms;
  const TqdQuery *queries;      // the launch group's queries (part_start / n_parts count list ENTRIES)
  const float *caches;
  const TqdLead *leads;
  const uint4 *tasks;           // {term handle, first block, n_blocks | n_leads << 16 | cache << 24, first lead}
  const TqkSinks *sinks;
  uint32_t *thr_slots;          // hashed score slots per query (as the other pruned kernels use)
  uint32_t *thr_val;            // [n_queries] current lower bound of each query's k-th best score
  uint32_t *task_counter;       // next task to hand out (zeroed per batch)
  const uint8_t *table_base;    // TqdLead::dense_off / tf8_off count 8-byte units from here
  uint64_t *stage;              // [grid][TQD_US_GROUP][capl] per-wave staging lists
  uint64_t *lists;              // per-query result lists (query q: entries part_start .. + n_parts)
  uint32_t *list_count;         // [n_queries] entries written so far
  uint32_t task_begin, n_tasks; // this launch hands out tasks [task_begin, n_tasks)
  uint32_t n_queries;
  uint32_t grid;
  uint32_t debug;
  float bound_slack;
};

// shared-intersection launch (tq_ashare.hip): a persistent grid of single-wave workgroups pulling tasks
struct TqkAShareParams {
  TqdSegment seg;
  const TqdTerm *terms;
  const TqdQuery *queries;      // the launch group's queries (part_start / n_parts count list ENTRIES)
  const float *caches;
  const TqdALead *leads;
  const uint4 *tasks;           // {leader term handle, first block, n_blocks | n_leads << 16 | cache << 24, first lead}
  const uint2 *qlists;          // boolean leads: [query][TQD_AS_MAX_TERMS] = {bitmap, tf bytes} of its lists (as dense_off / tf8_off)
  const TqkSinks *sinks;
  uint32_t *thr_slots;          // hashed score slots per query (as the other pruned kernels use)
  uint32_t *thr_val;            // [n_queries] current lower bound of each query's k-th best score
  uint32_t *task_counter;       // [n_queues] next task of each queue to hand out (zeroed per batch)
  const uint8_t *table_base;    // TqdALead::dense_off / tf8_off count 8-byte units from here
  uint64_t *stage;              // [grid][TQD_AS_GROUP][capl] per-wave staging lists
  uint64_t *lists;              // per-query result lists (query q: entries part_start .. + n_parts)
  uint32_t *list_count;         // [n_queries] entries written so far
  uint32_t task_begin, n_tasks; // this launch hands out tasks [task_begin, n_tasks)
  uint32_t n_queries;
  uint32_t grid;
  uint32_t debug;
  uint32_t boolean;             // the leads are (TQ_MODE_BOOL query, leading list) pairs
  uint32_t rdir_lists;          // ... and some entry of qlists is a range directory (the instantiation that knows them)
  float bound_slack;
  uint32_t n_queues;            // task queues of this launch (1, or 8 = one per XCD)
  uint32_t bound_mode;          // TQ_AS_BOUND bits: where list 1's range maxima replace its weight as the bound
};

// exhaustive pure unions, doc-major (tq_xunion.hip): a persistent grid of 16-wave workgroups; a
// workgroup builds the BM25 term scores of EVERY (list, weight) pair of the group for a tile of 128
// docs in LDS, then each of its waves evaluates its share of the group's queries against that tile
constexpr uint32_t TQK_XU_TILE = 128;         // docs per tile (two per lane)
constexpr uint32_t TQK_XU_MAX_ROWS = 256;     // rows of the tile: up to 255 distinct (list, weight) pairs + the all-zero padding row
constexpr uint32_t TQK_XU_MAX_QUERIES = 8192; // queries of a group
#ifndef TQK_XU_WAVES_N
#define TQK_XU_WAVES_N 16
#endif
constexpr uint32_t TQK_XU_WAVES = TQK_XU_WAVES_N;
struct TqkDenseRow {  // one posting list of the group.  Rows with a bitmap come first.
  const uint2 *dense;         // bitmap + rank directory (TqdTermHead::dense), or null
  const uint8_t *tf8;         // byte-wide tfs by posting index (with `dense`, or with `flat_docs`); null = every tf is 1
  const uint32_t *flat_docs;  // lists without a bitmap: the decoded doc ids (doc_freq entries)
  uint32_t handle;            // the list's term record (saturated tf bytes are read from the packed stream)
  uint32_t doc_freq;
  float w;                    // the row holds w * tf/(tf+norm): a list used at two weights is two rows
  uint32_t pad;
};
struct TqkDenseQuery {  // 16 bytes, loaded one query per lane
  uint32_t rows_lo, rows_hi;  // row of list t in byte t (lists in score-sum order); beyond n_terms: row n_rows (all zero)
  uint32_t nt_k;              // n_terms | k << 8
  uint32_t thr_row;           // first row of the query's threshold slots (64 for k <= 16, else 256)
};
struct TqkDenseParams {
  TqdSegment seg;
  const TqdTerm *terms;
  const TqkDenseRow *rows;
  const TqkDenseQuery *queries;
  const float *cache;           // Bm25Weight.cache of the group (256 floats)
  const TqkSinks *sinks;
  uint32_t *thr_slots;
  uint32_t *thr_val;            // [n_queries]
  uint32_t *list_count;         // [n_queries]
  uint32_t *task_counter;
  uint64_t *stage;              // [grid][n_queries][capl] staging lists private to a workgroup
  uint64_t *lists;              // [n_queries][list_stride] result lists
  uint32_t n_rows, n_bitmap_rows;
  uint32_t n_queries;
  uint32_t max_terms;           // the most lists a query of the launch has
  uint32_t n_tasks, tiles_per_task;
  uint32_t list_stride;
  uint32_t grid;
  uint32_t debug;
};

struct TqkMergeParams {
  const TqdQuery *queries;
  const uint64_t *partials;
  const uint32_t *out_index;  // query -> output row (null = identity)
  float *out_scores;
  uint32_t *out_docs;
  uint32_t *out_counts;
  uint32_t n_queries;
  uint32_t out_stride;
  // two-level merge (queries with thousands of partial lists: a small batch cut into short tiles): `pre_slices` > 0 =
  // a first launch of n_queries x pre_slices wavefronts reduces slice s of every query's lists — lists [s * per,
  // (s + 1) * per), per = tqk_merge_slice_lists(n_parts, pre_slices) — into the slice's FIRST list, in place; the
  // final launch then reads one list per slice.  Queries with fewer than TQK_MERGE_PRE_MIN lists are left alone.
  uint32_t pre_slices;
};
constexpr uint32_t TQK_MERGE_PRE_MIN = 96;
inline __host__ __device__ uint32_t tqk_merge_slice_lists(uint32_t n_parts, uint32_t pre_slices) {
  return (pre_slices == 0u || n_parts < TQK_MERGE_PRE_MIN) ? 1u : (n_parts + pre_slices - 1u) / pre_slices;
}

struct TqkSegMergeParams {
  const float *scores;          // [segment][query][stride]
  const uint32_t *docs;
  const uint32_t *counts;       // [segment][query]
  const uint32_t *segment_ords; // [segment] or null
  float *out_scores;            // [query][limit]
  uint32_t *out_segment_ords;
  uint32_t *out_docs;
  uint32_t *out_counts;
  uint32_t n_segments, n_queries, stride, offset, limit;
};

// ---- nested boolean queries over bitmaps (tq_tree.hip): a BooleanQuery whose clauses are terms or BooleanQuerys of
// terms.  Terms of a clause are contiguous; clauses in score-sum order (Must clauses cheapest first, then Should,
// then MustNot); inside a clause Must terms first.  Occurs: TQD_ROLE_* (= enum tq_occur).
constexpr uint32_t TQK_TREE_TILE_WORDS = 4096;  // bitmap words (131 072 docs) per (query, tile) wavefront
constexpr uint32_t TQK_TREE_MAX_TERMS = 16;
constexpr uint32_t TQK_TREE_PHRASE_TERMS = 8;  // terms of a phrase inside a boolean query (one position cursor each, in registers)
struct TqdTreeQuery {
  uint32_t n_terms, n_clauses;   // 0 / 0: the planner found the query empty
  uint32_t k, cache_idx;
  uint32_t part_start;           // first partial top-k list (one per tile)
  uint32_t top_has_must;         // the query has Must clauses (else: the union of its Should clauses)
  uint32_t top_need;             // Should clauses that have to match: minimum_number_should_match, at least 1 without a Must clause
  uint32_t has_phrase;           // some atom is a PhraseQuery (atom_end bit 1): the bitmap expression is a superset, every doc is re-checked
  uint32_t dense_off[TQK_TREE_MAX_TERMS];    // bitmap + rank directory / byte-wide tfs of the term's list, as offsets
  uint32_t tf8_off[TQK_TREE_MAX_TERMS];      // from TqkTreeParams::table_base in 8-byte units
  uint32_t weight_bits[TQK_TREE_MAX_TERMS];  // (float) idf * (1 + k1) * boost
  uint32_t handle[TQK_TREE_MAX_TERMS];       // term handle (saturated tf bytes read the packed value)
  uint32_t inner[TQK_TREE_MAX_TERMS];        // occur of the term's ATOM inside its clause (an atom = a run of terms that
                                             // must all be present: one term, or a nested intersection of terms)
  uint32_t atom_end[TQK_TREE_MAX_TERMS];     // bit 0: the term is the last of its atom; bit 1 (on every term of the atom): the atom is
                                             // a PhraseQuery of <= TQK_TREE_PHRASE_TERMS terms (weight_bits = the phrase's weight);
                                             // bit 2 (on every term of the atom): the atom is a UNION of its terms (present where
                                             // any of them is, scoring the present ones) instead of a conjunction
  uint32_t dir_off[TQK_TREE_MAX_TERMS];      // phrase terms: position directory of the list (TqdTerm::pos_dir layout), 8-byte units
  uint32_t phrase_off[TQK_TREE_MAX_TERMS];   // phrase terms: max_offset - term_offset (phrase_scorer.rs:372-385)
  uint32_t outer[TQK_TREE_MAX_TERMS];        // per clause: its occur in the query
  uint32_t inner_need[TQK_TREE_MAX_TERMS];   // per clause: Should terms that have to be present
  uint32_t first_term[TQK_TREE_MAX_TERMS + 1];  // per clause: its terms are [first_term[c], first_term[c + 1])
};
struct TqkTreeParams {
  TqdSegment seg;
  const TqdTerm *terms;
  const TqdTreeQuery *queries;
  const float *caches;
  const TqkSinks *sinks;
  const uint8_t *table_base;
  uint32_t n_queries, n_words;
  uint32_t any_phrase;  // some query of the launch has a phrase atom (selects the kernel instantiation)
};
hipError_t tqk_launch_tree(const TqkTreeParams &p, int kpl, hipStream_t st);
uint32_t tqk_tree_tiles(uint32_t n_words);

hipError_t tqk_launch_and(const TqkScanParams &p, int kpl, bool use_dpp, hipStream_t st);
hipError_t tqk_launch_or(const TqkScanParams &p, int kpl, bool use_dpp, hipStream_t st);
hipError_t tqk_launch_phrase(const TqkScanParams &p, int kpl, bool use_dpp, hipStream_t st);
hipError_t tqk_launch_merge(const TqkMergeParams &p, int kpl, hipStream_t st);
hipError_t tqk_launch_share(const TqkShareParams &p, int kpl, hipStream_t st);
// merge of the shared-union launch's per-query lists (m.partials = lists, list_count per query)
// the words a batch zeroes before its kernels (match counters, threshold slots, the shared launches' per-query words):
// ONE launch instead of a fill per buffer (five fills and their gaps were 30 us of a 1.1 ms headline step, 15 of a
// 140 us one-query call)
constexpr int TQK_ZERO_MAX = 8;
struct TqkZeroParams {
  uint32_t *ptr[TQK_ZERO_MAX];  // 4-byte aligned
  uint32_t words[TQK_ZERO_MAX];
  uint32_t n;
};
// tf classes of a column list into TqdSegment::doccls (slot < TQD_CLS_SLOTS; the words zeroed at allocation)
hipError_t tqk_launch_doccls_set(uint64_t *cls, const uint32_t *docs, const uint32_t *tfs, uint32_t n, uint32_t slot,
                                 uint32_t max_doc, hipStream_t st);
hipError_t tqk_launch_zero(const TqkZeroParams &p, hipStream_t st);
// signature bits (TQD_SIG_SHIFT + bits[i]) of n lists, given by their own records, into the doc matrix: one launch
// ... and, where tabs[i] is not null, the list's range directory (rdir_lookup, tq_common.hpp: tabs[i] = (max_doc >>
// shifts[i]) + 2 directory slots padded to a multiple of four, then dfs[i] entries; nothing needs zeroing); bits[i] =
// 0xFFFFFFFF: no signature bit for list i; mat may be null.  cache (256 floats: the segment's own Bm25 cache) not null:
// lmax_out[i] (zeroed) gets the list's largest tf/(tf + norm) as build_rmax rounds it
hipError_t tqk_launch_docsig_batch(const TqdSegment &seg, const TqdTerm *const *selfs, const uint32_t *bits, uint32_t n,
                                   uint64_t *mat, uint32_t *const *tabs, const uint32_t *shifts, const uint32_t *dfs,
                                   const float *cache, uint32_t *lmax_out, bool use_dpp, hipStream_t st);
// ... from a decoded list (docs / tfs: n postings, ascending)
hipError_t tqk_launch_rdir_fill(const uint32_t *docs, const uint32_t *tfs, uint32_t n, uint32_t max_doc, uint32_t *dir,
                                uint32_t S, hipStream_t st);
hipError_t tqk_launch_merge_lists(const TqkMergeParams &m, const uint32_t *list_count, int kpl,
                                  hipStream_t st);
uint32_t tqk_share_capl(int kpl);  // staging entries per lead slot
// ---- Count collector over bitmaps (tq_count.hip)
#define TQK_COUNT_MUST 0u
#define TQK_COUNT_NOT 1u
#define TQK_COUNT_SHOULD 2u
#define TQK_COUNT_HAS_MUST 1u     // flags: the doc set is the intersection of the Must clauses (else: the union of the Should lists)
#define TQK_COUNT_NEED_SHOULD 2u  // flags: ... and at least one Should list (minimum_number_should_match = 1)
struct TqkCountQuery {            // 152 bytes
  uint32_t n_terms;               // lists: Must clauses first (a clause = a union of lists), then MustNot, then Should
  uint32_t kinds;                 // 2 bits per list: TQK_COUNT_*
  uint32_t clause_end;            // bit m: list m is the last of its Must clause
  uint32_t flags;
  uint32_t narrow;                // bit m: list m's bitmap is a plain array of 32-bit words (a list without a
  uint32_t pad_;                  // bitmap of its own, scattered into the batch's scratch: count_scatter_kernel)
  const uint2 *dense[TQD_MAX_TERMS];  // the lists' bitmaps: {32 doc bits, postings before the word}
};
struct TqkCountParams {
  const TqkCountQuery *queries;
  const uint8_t *alive;  // AliveBitSet bits or null
  uint32_t *out_counts;  // [n_queries], zeroed by the caller
  uint32_t n_queries, n_words;
};
hipError_t tqk_launch_count_bitmaps(const TqkCountParams &p, hipStream_t st);
// the docs of lists without a bitmap as bits: wgs[i] = {term handle, first block, slot, -}: 4 blocks per workgroup
hipError_t tqk_launch_count_scatter(const TqdSegment &seg, const TqdTerm *terms, const uint4 *wgs, uint32_t n_wgs,
                                    uint32_t *bits, uint32_t words_per_list, hipStream_t st);
uint32_t tqk_count_tile_words();
hipError_t tqk_launch_ashare(const TqkAShareParams &p, int kpl, hipStream_t st);
uint32_t tqk_ashare_waves_per_cu();  // resident wavefronts per CU the kernel is built for
uint32_t tqk_bshare_waves_per_cu();  // ... its boolean instantiation
hipError_t tqk_launch_xunion(const TqkDenseParams &p, int kpl, hipStream_t st);
// a list without a bitmap as plain arrays: doc ids and min(tf, 255) per posting
hipError_t tqk_launch_flat_list(const TqdSegment &seg, const TqdTerm *terms, uint32_t handle,
                                uint32_t n_blocks, uint32_t *docs, uint8_t *tf8, hipStream_t st);
hipError_t tqk_launch_decode_list(const TqdSegment &seg, const TqdTerm *terms, uint32_t handle,
                                  uint32_t n_blocks, uint32_t *docs, uint32_t *tfs, bool use_dpp,
                                  hipStream_t st);
hipError_t tqk_launch_decode_positions(const TqdSegment &seg, const TqdTerm *terms,
                                       uint32_t handle, uint32_t *out, uint64_t n, hipStream_t st);
hipError_t tqk_launch_merge_segments(const TqkSegMergeParams &p, hipStream_t st);
// doc matrix (TqdSegment::docmat): fill with the fieldnorm ids / set one list's column
hipError_t tqk_launch_docmat_init(uint64_t *mat, const uint8_t *fieldnorm, uint32_t const_id,
                                  uint32_t max_doc, hipStream_t st);
hipError_t tqk_launch_docmat_set(uint64_t *mat, const uint32_t *docs, uint32_t n, uint32_t slot,
                                 uint32_t max_doc, hipStream_t st);
// term freqs of a decoded list as bytes (TqdTerm::tf8)
hipError_t tqk_launch_tf8_pack(const uint32_t *tfs, uint32_t n, uint8_t *out, hipStream_t st);


// ---- shared with tq_encode.hip: the C ABI's error slot and context checks live in tq_api.cpp
struct tq_ctx;
int tq_internal_fail(int code, const char *where, const char *what);
bool tq_internal_ctx_has_device(const tq_ctx *ctx, int device);
