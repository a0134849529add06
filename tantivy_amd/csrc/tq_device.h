// tq_device.h — structures shared by the host planner (tq_api.cpp) and the gfx950 kernels
// (tq_common.hpp and the kernel files).  Layouts are fixed (uploaded as raw bytes).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define TQD_TERMINATED 0x7FFFFFFFu
#define TQD_MAX_TERMS 16

// Tunables of the scan kernels (see DESIGN.md "AND kernel").
#define TQD_WAVES_PER_WG 4    // independent wavefronts per workgroup
#define TQD_AND_TILE 64       // leader-list blocks per AND tile (one lane each in the pre-filter)
#define TQD_DENSE_RATIO 128   // default: lists with doc_freq >= max_doc/128 also get a bitmap + rank directory
#define TQD_THR_SLOTS 64      // shared threshold slots per query (pruned mode)
#define TQD_OR_WINDOW 4096    // docs per OR tile (one workgroup)
#define TQD_MAT_SLOTS 40      // dense lists per segment with a column in the doc matrix (bits 8..47 of a word)
#define TQD_CLS_SLOTS 32     // column lists with a 2-bit tf class per doc in TqdSegment::doccls
#define TQD_SIG_SHIFT 48      // bits 48..63 of a doc-matrix word: the signature of the lists WITHOUT a column
#define TQD_SIG_BITS 16
// Range maxima of a list with a bitmap (tq_terms.cpp build_rmax, tq_ashare.hip): level l holds one byte per
// (1 << (TQD_RM_SHIFT + 2 l)) docs — the list's largest tf/(tf+norm) in that doc range, rounded up to 1/255 —
// level l + 1 is the maximum over four entries of level l.  A leader block picks the finest level at which its doc
// span touches at most two entries; a single doc reads level 0.
#define TQD_RM_SHIFT 10
#define TQD_RM_LEVELS 5
static inline __host__ __device__ uint32_t tqd_rm_level_entries(uint32_t max_doc, uint32_t level) {
  return (((max_doc >> (TQD_RM_SHIFT + 2u * level)) + 2u) + 3u) & ~3u;
}
static inline __host__ __device__ uint32_t tqd_rm_level_off(uint32_t max_doc, uint32_t level) {  // byte offset of a level
  uint32_t off = 0;
  for (uint32_t l = 0; l < level; ++l) off += tqd_rm_level_entries(max_doc, l);
  return off;
}

// A posting list on the device.  The skip list of src/postings/skip.rs:205-253 is unrolled from
// its sequential form (running byte / position offsets made absolute) into structure-of-arrays
// tables, so that any block can be located and decoded independently of the others:
//   rec[j].x     last doc id in block j (skip.rs `last_doc_in_block`)
//   rec[j].y     meta: doc_bits | strict<<6 | tf_bits<<8 | block-max fieldnorm_id<<16 | block-max
//                tf code<<24 (skip.rs:16-43,205-253); 0xFFFFFFFF = the vint tail (pre-decoded)
//   rec[j].z     offset of the bitpacked doc payload, relative to payload_base
//   rec[j].w     index (in the positions stream) of the block's first position
//                (one 16-byte record: a seek's last probe brings the whole entry)
//   coarse[b]    first j with last_doc[j] >= (b << coarse_shift): O(1) `seek_block`
struct TqdTermHead {  // what every kernel needs: fetched with scalar loads
  const uint4 *rec;           // n_blocks + 1 (the extra record carries the total position count)
  const uint32_t *coarse;     // ((max_doc-1) >> coarse_shift) + 2 entries
  // dense lists only (doc_freq >= max_doc / dense_ratio, within the memory budget), else null: membership bitmap with a
  // rank directory.  dense[d >> 5] = {bits of docs 32*(d>>5)..+31, number of postings before them}
  const uint2 *dense;
  const uint32_t *tail_docs;  // n_tail
  const uint32_t *tail_tfs;   // n_tail
  uint64_t payload_base;      // absolute offset (inside the .idx sub-file) of block 0's payload
  uint32_t n_blocks, n_tail;
  uint32_t has_freq;          // bit 0: 0 => every tf reads as 1; bits 8..15: doc-matrix slot + 1 (0 = none);
                              // bits 16..23: signature bit (0..15) + 1 of a list without a column (0 = none);
                              // bit 24: the list's tf classes are in TqdSegment::doccls (its slot < TQD_CLS_SLOTS)
  uint32_t coarse_shift;
};
struct TqdTerm : TqdTermHead {
  // positions stream (src/positions/reader.rs): per position-block absolute byte offset / width
  const uint64_t *pos_blk;        // n_pos_blocks: absolute byte offset | bit width << 56
  const uint32_t *pos_tail;       // vint tail, pre-decoded deltas
  uint32_t n_full, doc_freq;
  uint32_t n_pos_blocks, n_pos_tail;
  // dense lists with positions: pos_dir[j] = number of positions before posting 4*j.  With the
  // posting index from the bitmap's rank, a posting's first position index is pos_dir[i >> 2] +
  // the term freqs of the <= 3 postings before it in its group of four — which share one
  // 16-byte row of the bitpacked tf stream — instead of a prefix sum over its whole block.
  const uint32_t *pos_dir;
  // dense lists: min(tf, 255) per posting, indexed by the posting index (bitmap rank), or null
  const uint8_t *tf8;
  // dense lists with positions: the bitmap's doc bits alone, bits[w] = dense[w].x, zero-padded to a multiple of 256
  // words — what the phrase sweep streams (16-byte loads, 128 docs per lane and list; the rank half of `dense` is
  // gathered for the docs that survive the AND only), or null
  const uint32_t *bits;
};

struct TqdQuery {
  uint32_t term[TQD_MAX_TERMS];  // term handles in execution order
  float weight[TQD_MAX_TERMS];
  uint32_t phrase_off[TQD_MAX_TERMS];  // max_offset - term_offset (phrase_scorer.rs:372-385)
  uint32_t n_terms;
  uint32_t mode;
  uint32_t k;
  uint32_t cache_idx;   // which 256-float tf cache
  uint32_t tile_start;  // first global tile of this query
  uint32_t n_tiles;
  uint32_t part_start;  // first partial top-k list of this query
  uint32_t n_parts;
  uint32_t flags;       // TQD_QF_*
  uint32_t thr_index;   // row of the shared-threshold table (pruned mode), or 0xFFFFFFFF
  uint32_t chunk_first; // first chunk that holds a tile of this query
  uint32_t tile_blocks; // AND: leader blocks per tile (1..64, sized so tiles cost about the same)
  uint32_t lead_tile_start[TQD_MAX_TERMS + 1];  // candidate-driven OR: first tile led by list i
  // union kernel, boolean queries: terms are laid out [leader clause | other Must clauses |
  // MustNot terms | optional Should terms]
  uint32_t roles;       // 2 bits per term, TQD_ROLE_*
  uint32_t clause_end;  // bit m: term m is the last term of its Must clause (a union of terms)
  uint32_t n_lead;      // terms of the leader set: each leads its own run of tiles
  uint32_t n_opt_lead;  // the first n_opt_lead leaders are optional Should lists; the rest of
                        // the leader set is the cheapest Must clause (0: the set is one clause)
  uint32_t min_should;  // Should terms that have to match (minimum_number_should_match)
};

// ---- shared-union launch (tq_ushare.hip): the pure unions of a batch, driven term by term.
// A LEAD is one (query, list) pair: list i of query q leading the docs it is the first list to
// hold.  The leads of one term are cut into groups of <= TQD_US_GROUP; a TASK is a run of blocks of
// that term for one group: the blocks are decoded and their doc-matrix words gathered ONCE, every
// lead of the group tests them on registers.
#define TQD_US_GROUP 32      // leads per group (one lane each at task setup)
#define TQD_US_MAX_TERMS 8   // unions with more terms keep the per-query kernel
#define TQD_US_TILE 64       // blocks per pre-filter step (one lane each)
struct TqdLead {             // 128 bytes, written by the host planner
  uint32_t query;            // launch-group query index
  uint32_t info;             // i (bits 0-3) | lists after i with a membership bit (4-7) | n_terms
                             // (8-11) | bit 12: some list uses the signature word | bit 16+m: list m
                             // has NO doc-matrix column | bit 24+m: ... and no signature bit either
  float w;                   // weight of the leading term in this query
  float suffix;              // weights of lists i.. : the most a doc first seen in list i can score
  float sparse_after;        // weights of the lists after i with neither column nor signature bit
  uint32_t cols_lo, cols_hi; // membership bits of the after-lists that have one, in list order (0-3 / 4-6):
                             // bit positions in the doc-matrix word: 8 + slot = the list's column (exact),
                             // 48 + b = signature bit b (a clear bit proves absence, a set one means maybe)
  float aw[7];               // their weights, in list order
  uint64_t before_mask;      // doc-matrix bits of the lists before i that have a column
  uint8_t sig[8];            // per list of the query: signature bit + 1 (0 = none)
  // the lists after i (a = 1.. <-> list i + a): bitmap + rank directory and byte-wide term freqs
  // as offsets from TqkShareParams::table_base in 8-byte units (0 = the list has no bitmap)
  uint32_t dense_off[7];
  uint32_t tf8_off[7];
};
static_assert(sizeof(TqdLead) == 128, "TqdLead is uploaded as raw bytes");

// ---- shared-intersection launch (tq_ashare.hip): the AND queries of a batch, driven LEADER BY LEADER.
// block_wand_intersection walks the rarest list of a query (its leader) block by block; in a batch
// many queries lead with the same list.  A lead here is one query; the leads of one (leader list,
// Bm25 cache) are cut into groups of <= TQD_AS_GROUP, a TASK is a run of blocks of that list for one
// group: the blocks are decoded and their doc-matrix words gathered ONCE, every lead tests the 128
// docs on registers against its own other lists (column / signature bits) and its own threshold.
#define TQD_AS_GROUP 32      // leads per group (one lane each at task setup)
#define TQD_AS_MAX_TERMS 8   // intersections with more lists keep the per-query kernel
#define TQD_AS_TILE 64       // blocks per pre-filter step (one lane each)
#define TQD_AL_RDIR 0x10000000u  // TqdALead::info bit 28 (AND leads): list 1 is probed through its range directory
                                 // (rdir_lookup, tq_common.hpp), bits 16-20 = the directory's shift
struct TqdALead {            // 64 bytes, written by the host planner
  uint32_t query;            // launch-group query index
  uint32_t info;             // n_terms (bits 0-4) | bit 8: list 1's membership bit is its column (exact) |
                             // bit 9: the same query (lists, weights, k) as the lead before it in its group: a
                             // TWIN.  A group is a sequence of families (a head + its twins): a family is
                             // evaluated once, its results go to every member's result list |
                             // bits 16-19 (boolean leads): which list of the query leads here |
                             // bits 20-27 (boolean leads): bit m clear = the lead's exclusion mask has ruled
                             // list m out for every doc that gets to the scoring stage (never probed)
  float w;                   // weight of the leader in this query
  float rest;                // what the other lists can add at most (AND: their weights together = the weight
                             // of list 1 in a 2-term query; boolean: the weights of the lists after the leader)
  uint32_t dense_off;        // AND, list 1: bitmap + rank directory, byte-wide tfs, as offsets from
  uint32_t tf8_off;          // TqkAShareParams::table_base in 8-byte units (TQD_AL_RDIR: the list's directory
                             // and its entries).  Boolean leads: byte m of the two
                             // words = the doc-matrix bit of the query's list m (0: it has none): a doc whose
                             // word has the bit clear is not in the list
  uint32_t mask_lo, mask_hi; // doc-matrix bits every match has: the other lists' columns (exact) and
                             // signature bits (a clear bit proves absence, a set one means maybe).  Boolean
                             // leads: mask_lo = (float) the weights of the lists after the leader that have no
                             // doc-matrix bit, mask_hi unused
  uint32_t k;                // the query's k (<= 128)
  uint32_t thr_row;          // its row of threshold slots (identical queries of a batch share one)
  // intersections: excl_lo = list 1's range-maxima table (TermHost::rmax_blob) as an offset from table_base in
  // 8-byte units (0 = none: list 1 is bounded by its weight), excl_hi = (float) list 1's weight, any1_lo =
  // (float) the weights of lists 2.. together, any1_hi = the largest entry of the table (255 = none).
  // boolean leads (TQ_MODE_BOOL through the shared launch): a match has NONE of the excl bits (columns of
  // MustNot lists and of the lead-set lists before the leader: found there = excluded / another lead's doc)
  // and AT LEAST ONE bit of any1 and of any2 (the columns / signature bits of a Must clause each; 0 = no
  // such clause).  Further clauses are verified by the scoring stage only.
  uint32_t excl_lo, excl_hi, any1_lo, any1_hi, any2_lo, any2_hi;
};
struct TqdALeadLds {  // what the scoring stage keeps of a lead in LDS
  uint32_t query, info;
  float w, rest;
  uint32_t dense_off, tf8_off;
};

#define TQD_ROLE_SHOULD 0u
#define TQD_ROLE_MUST 1u
#define TQD_ROLE_MUST_NOT 2u

#define TQD_QF_PRUNE 1u  // block-max pruning allowed (all weights >= 0, caller asked for it)

struct TqdSegment {
  const uint8_t *idx;        // .idx sub-file (8-byte header included), padded
  const uint8_t *pos;        // .pos sub-file or null
  const uint8_t *fieldnorm;  // max_doc bytes or null
  const uint8_t *alive;      // AliveBitSet bits (bit d of byte d>>3) or null = no deletes
  // doc-major matrix of the segment's dense lists, or null: docmat[d] = fieldnorm id of doc d
  // (bits 0..7) | bit (8 + slot) set iff d is in the dense list that owns matrix slot `slot`
  // (TQD_MAT_SLOTS lists per segment).  ONE 8-byte gather gives a candidate's BM25 norm and its
  // membership in every dense list of the query — the scan kernels are bound by the number of
  // divergent gathers, not by bytes.  Derived data, built at tq_term_prepare like the bitmaps.
  const uint64_t *docmat;
  // tf classes of the first TQD_CLS_SLOTS column lists, or null: doccls[d] holds 2 bits per column slot — 0 the doc is
  // not in the list, 1 / 2 its term freq, 3 = three or more (read the tf byte).  ONE 8-byte gather bounds a phrase
  // candidate by min tf before any of its lists is touched (tq_phrase.hip); the exact tf of ~90 % of the postings.
  const uint64_t *doccls;
  uint32_t max_doc;
  uint32_t const_fieldnorm_id;
  uint32_t min_fieldnorm_id;  // smallest fieldnorm id present (lower bound of every doc's norm)
};
