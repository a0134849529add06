// tq_kernels.hip — gfx950 (CDNA4, wave64) kernels of the tantivy query-execution path.
//
// One wavefront decodes one 128-doc posting block: 64 lanes x 2 values, BitPacker4x funnel-shift
// unpack, DPP prefix sum for the strict-delta doc ids.  AND = leader list hashed into a per-wave
// LDS table, denser lists stream through and probe.  OR = 4096-doc window of f32 accumulators in
// LDS per workgroup.  Top-k = per-wave sorted key registers, flushed as partial lists and merged
// by a second kernel.  No MFMA: this is integer / byte work bound by HBM + LDS + VALU issue.
//
// Reference behaviour restated (file:line under the tantivy checkout):
//   decode      src/postings/compression/mod.rs:105-150, block_segment_postings.rs:343-391
//   AND         src/query/intersection.rs:120-179 (doc set),
//               src/query/boolean_query/block_wand_intersection.rs:144-165 (score order)
//   OR          src/query/union/buffered_union.rs:63-158 + score_combiner.rs:39-56
//   phrase      src/query/phrase_query/phrase_scorer.rs:82-136,463-507,578-586
//   BM25        src/query/bm25.rs:179-193
//   top-k       src/collector/sort_key/sort_by_score.rs:86-161 (score desc, doc asc)
//   merge       src/collector/sort_key_top_collector.rs:76-95, top_score_collector.rs:590-600
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tq_device.h"
#include "tq_launch.h"

namespace {

constexpr uint32_t EMPTY_SLOT = 0xFFFFFFFFu;
constexpr int WAVE = 64;

// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ uint32_t uni(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
}
__device__ __forceinline__ uint64_t uni64(uint64_t x) {
  uint32_t lo = uni((uint32_t)x), hi = uni((uint32_t)(x >> 32));
  return ((uint64_t)hi << 32) | lo;
}
template <typename T>
__device__ __forceinline__ const T *uni_ptr(const T *p) {
  return (const T *)uni64((uint64_t)p);
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t src_lane) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)src_lane);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)src_lane);
  return ((uint64_t)hi << 32) | lo;
}

struct __attribute__((packed, aligned(1))) U2Unaligned {
  uint32_t x, y;
};
__device__ __forceinline__ uint2 ld_u2(const uint8_t *p) {
  U2Unaligned v = *reinterpret_cast<const U2Unaligned *>(p);
  return make_uint2(v.x, v.y);
}
struct __attribute__((packed, aligned(1))) U1Unaligned {
  uint32_t x;
};
__device__ __forceinline__ uint32_t ld_u1(const uint8_t *p) {
  return reinterpret_cast<const U1Unaligned *>(p)->x;
}

// ------------------------------------------------------------------ wave64 inclusive scan (DPP)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_get(uint32_t x) {
  // lanes whose source is invalid / masked off keep `old` = 0
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, false);
}
template <bool USE_DPP>
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x, int lane) {
  if (USE_DPP) {
    x += dpp_get<0x111, 0xF>(x);  // row_shr:1
    x += dpp_get<0x112, 0xF>(x);  // row_shr:2
    x += dpp_get<0x114, 0xF>(x);  // row_shr:4
    x += dpp_get<0x118, 0xF>(x);  // row_shr:8
    x += dpp_get<0x142, 0xA>(x);  // row_bcast:15 -> rows 1,3
    x += dpp_get<0x143, 0xC>(x);  // row_bcast:31 -> rows 2,3
    return x;
  } else {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      uint32_t y = __shfl_up(x, d, WAVE);
      if (lane >= d) x += y;
    }
    return x;
  }
}

// ------------------------------------------------------------------ BitPacker4x unpack
// Lane t owns values 2t and 2t+1: register k = t>>1, SIMD lanes l = 2*(t&1), 2*(t&1)+1 of the
// 4-lane vertical layout.  Both values share the word index and the shift, and
// their 32-bit words are adjacent in memory => two 8-byte loads + two v_alignbit per pair.
__device__ __forceinline__ void unpack2(const uint8_t *p, uint32_t b, int lane, uint32_t &v0,
                                        uint32_t &v1) {
  if (b == 0) {  // wave-uniform
    v0 = 0;
    v1 = 0;
    return;
  }
  const uint32_t k = (uint32_t)lane >> 1;
  const uint32_t bitpos = k * b;
  const uint32_t w = bitpos >> 5, s = bitpos & 31u;
  const uint8_t *q = p + 16u * w + 8u * ((uint32_t)lane & 1u);
  const uint2 lo = ld_u2(q);
  const uint2 hi = ld_u2(q + 16);  // may over-read 16 B past the block: buffers are padded
  const uint32_t mask = (b >= 32u) ? 0xFFFFFFFFu : ((1u << b) - 1u);
  v0 = __funnelshift_r(lo.x, hi.x, s) & mask;
  v1 = __funnelshift_r(lo.y, hi.y, s) & mask;
}
// Single value at index i of a bitpacked block (positions random access).
__device__ __forceinline__ uint32_t unpack_one(const uint8_t *p, uint32_t b, uint32_t i) {
  if (b == 0) return 0;
  const uint32_t k = i >> 2, l = i & 3u;
  const uint32_t bitpos = k * b;
  const uint32_t w = bitpos >> 5, s = bitpos & 31u;
  const uint8_t *q = p + 16u * w + 4u * l;
  const uint32_t lo = ld_u1(q);
  const uint32_t hi = ld_u1(q + 16);
  const uint32_t mask = (b >= 32u) ? 0xFFFFFFFFu : ((1u << b) - 1u);
  return __funnelshift_r(lo, hi, s) & mask;
}

// ------------------------------------------------------------------ block decode
struct TermRef {
  const uint32_t *last_doc;
  const uint32_t *meta;
  const uint32_t *byte_off;
  const uint32_t *coarse;
  const uint32_t *tail_docs;
  const uint32_t *tail_tfs;
  uint64_t payload_base;
  uint32_t n_blocks;
  uint32_t n_tail;
  uint32_t has_freq;
  uint32_t shift;
};
__device__ __forceinline__ TermRef load_term(const TqdTerm *terms, uint32_t handle) {
  const TqdTerm *t = uni_ptr(terms + handle);
  TermRef r;
  r.last_doc = uni_ptr(t->last_doc);
  r.meta = uni_ptr(t->meta);
  r.byte_off = uni_ptr(t->byte_off);
  r.coarse = uni_ptr(t->coarse);
  r.tail_docs = uni_ptr(t->tail_docs);
  r.tail_tfs = uni_ptr(t->tail_tfs);
  r.payload_base = uni64(t->payload_base);
  r.n_blocks = uni(t->n_blocks);
  r.n_tail = uni(t->n_tail);
  r.has_freq = uni(t->has_freq);
  r.shift = uni(t->coarse_shift);
  return r;
}
__device__ __forceinline__ uint32_t block_first_possible(const TermRef &t, uint32_t j) {
  return j ? uni(t.last_doc[j - 1]) + 1u : 0u;
}

struct Dec {
  uint32_t d0, d1;  // doc ids (TQD_TERMINATED padded)
  uint32_t t0, t1;  // term freqs
};
constexpr uint32_t META_TAIL = 0xFFFFFFFFu;

// doc ids of block j (wave-uniform j): lane t gets docs 2t, 2t+1
template <bool USE_DPP>
__device__ __forceinline__ void decode_docs(const uint8_t *idx, const TermRef &t, uint32_t j,
                                            uint32_t meta, int lane, uint32_t &d0, uint32_t &d1) {
  if (meta == META_TAIL) {  // vint tail, pre-decoded at term_prepare
    const uint32_t i0 = 2u * (uint32_t)lane, i1 = i0 + 1u;
    d0 = i0 < t.n_tail ? t.tail_docs[i0] : TQD_TERMINATED;
    d1 = i1 < t.n_tail ? t.tail_docs[i1] : TQD_TERMINATED;
    return;
  }
  const uint8_t *p = idx + t.payload_base + uni(t.byte_off[j]);
  const uint32_t prev = j ? uni(t.last_doc[j - 1]) : 0u;
  const uint32_t doc_bits = meta & 31u;
  const uint32_t strict = (meta >> 6) & 1u;
  uint32_t x0, x1;
  unpack2(p, doc_bits, lane, x0, x1);
  const uint32_t a0 = x0 + strict;
  const uint32_t a1 = a0 + x1 + strict;
  const uint32_t incl = wave_inclusive_scan<USE_DPP>(a1, lane);
  // compression/mod.rs:36-39,112-121: offset 0 <=> None <=> seed u32::MAX (wrapping)
  const uint32_t base = (strict && prev == 0u) ? 0xFFFFFFFFu : prev;
  const uint32_t excl = base + (incl - a1);
  d0 = excl + a0;
  d1 = excl + a1;
}
// term freqs of block j for the same lane layout (padding of the tail reads as tf 0)
__device__ __forceinline__ void decode_tfs(const uint8_t *idx, const TermRef &t, uint32_t j,
                                           uint32_t meta, int lane, uint32_t &t0, uint32_t &t1) {
  if (meta == META_TAIL) {
    const uint32_t i0 = 2u * (uint32_t)lane, i1 = i0 + 1u;
    t0 = i0 < t.n_tail ? (t.has_freq ? t.tail_tfs[i0] : 1u) : 0u;
    t1 = i1 < t.n_tail ? (t.has_freq ? t.tail_tfs[i1] : 1u) : 0u;
    return;
  }
  if (!t.has_freq) {
    t0 = 1u;
    t1 = 1u;
    return;
  }
  const uint32_t doc_bits = meta & 31u;
  const uint32_t strict = (meta >> 6) & 1u;
  const uint32_t tf_bits = (meta >> 8) & 0xFFu;
  const uint8_t *p = idx + t.payload_base + uni(t.byte_off[j]) + 16u * doc_bits;
  unpack2(p, tf_bits, lane, t0, t1);
  t0 += strict;  // minus-one encoding is tied to the strict flag
  t1 += strict;  // (block_segment_postings.rs:45-57)
}
// term freq of the posting at index i (0..127) of block j; j and meta may differ per lane
__device__ __forceinline__ uint32_t block_tf_at(const uint8_t *idx, const TermRef &t, uint32_t j,
                                                uint32_t meta, uint32_t i) {
  if (!t.has_freq) return 1u;
  if (meta == META_TAIL) return t.tail_tfs[i];
  const uint32_t doc_bits = meta & 31u;
  const uint32_t strict = (meta >> 6) & 1u;
  const uint32_t tf_bits = (meta >> 8) & 0xFFu;
  const uint8_t *p = idx + t.payload_base + t.byte_off[j] + 16u * doc_bits;
  return unpack_one(p, tf_bits, i) + strict;
}

// WANT_TF_SCAN: also return the exclusive prefix sum of the tfs (position index inside the block)
template <bool USE_DPP, bool WANT_TF_SCAN>
__device__ __forceinline__ Dec decode_block(const uint8_t *idx, const TermRef &t, uint32_t j,
                                            int lane, uint32_t *tf_excl0 = nullptr,
                                            uint32_t *tf_excl1 = nullptr) {
  Dec r;
  const uint32_t meta = uni(t.meta[j]);
  decode_docs<USE_DPP>(idx, t, j, meta, lane, r.d0, r.d1);
  decode_tfs(idx, t, j, meta, lane, r.t0, r.t1);
  if (WANT_TF_SCAN) {
    const uint32_t s = r.t0 + r.t1;
    const uint32_t incl = wave_inclusive_scan<USE_DPP>(s, lane);
    *tf_excl0 = incl - s;
    *tf_excl1 = incl - s + r.t0;
  }
  return r;
}

// first block index j in [0, n_blocks) with last_doc(j) >= target, else n_blocks.
// Wave-uniform target: 64-ary cooperative search, 3 rounds for 40k blocks.
__device__ __forceinline__ uint32_t lower_bound_block(const TermRef &t, uint32_t target,
                                                      int lane) {
  uint32_t lo = 0, n = t.n_blocks;  // invariant: answer in [lo, lo+n]
  while (n > 0) {
    const uint32_t step = (n + 63u) >> 6;
    const uint32_t idx = lo + ((uint32_t)lane + 1u) * step - 1u;
    bool ge = true;
    if (idx < lo + n) ge = t.last_doc[idx] >= target;
    const uint64_t m = __ballot(ge);
    if (m == 0ull) {  // every probe (the last one sits on lo+n-1) is below the target
      lo += n;
      break;
    }
    const uint32_t first = (uint32_t)__builtin_ctzll(m);
    const uint32_t new_lo = lo + first * step;
    const uint32_t rem = n - first * step;
    n = (step - 1u < rem) ? step - 1u : rem;
    lo = new_lo;
    lo = uni(lo);
    n = uni(n);
  }
  return lo;
}
// The same for a per-lane target (BlockSegmentPostings::seek_block, skip.rs:263-273, made O(1)):
// the coarse table brackets the answer, a short binary search finishes.  doc < max_doc.
__device__ __forceinline__ uint32_t seek_block(const TermRef &t, uint32_t doc) {
  const uint32_t b = doc >> t.shift;
  uint32_t lo = t.coarse[b], hi = t.coarse[b + 1u];
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (t.last_doc[mid] >= doc)
      hi = mid;
    else
      lo = mid + 1u;
  }
  return lo;
}

// ------------------------------------------------------------------ BM25
__device__ __forceinline__ uint32_t fieldnorm_id(const TqdSegment &seg, uint32_t doc) {
  return seg.fieldnorm ? (uint32_t)seg.fieldnorm[doc] : seg.const_fieldnorm_id;
}
__device__ __forceinline__ float bm25(float weight, float norm, uint32_t tf) {
  const float f = (float)tf;
  return weight * (f / (f + norm));  // bm25.rs:179-193; compiled with -ffp-contract=off
}

// ------------------------------------------------------------------ top-k keys
// key = sortable(score) << 32 | ~doc : larger key == (higher score, then lower doc)
__device__ __forceinline__ uint64_t make_key(float score, uint32_t doc) {
  uint32_t fb = __float_as_uint(score);
  fb ^= (uint32_t)((int32_t)fb >> 31) | 0x80000000u;
  return ((uint64_t)fb << 32) | (uint64_t)(~doc);
}
__device__ __forceinline__ float key_score(uint64_t key) {
  uint32_t u = (uint32_t)(key >> 32);
  u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
  return __uint_as_float(u);
}
__device__ __forceinline__ uint32_t key_doc(uint64_t key) { return ~(uint32_t)key; }

template <int KPL>
struct TopK {
  uint64_t v[KPL];  // rank r*64+lane, descending
  uint64_t thr;     // k-th key (0 while not full)
  uint32_t k;
  __device__ __forceinline__ void reset(uint32_t kk) {
#pragma unroll
    for (int r = 0; r < KPL; ++r) v[r] = 0;
    thr = 0;
    k = kk;
  }
  __device__ __forceinline__ void refresh_thr() {
    const uint32_t kr = (k - 1u) >> 6, kl = (k - 1u) & 63u;
    uint64_t t = 0;
#pragma unroll
    for (int r = 0; r < KPL; ++r)
      if ((uint32_t)r == kr) t = readlane64(v[r], kl);
    thr = t;
  }
  // every lane may offer one candidate
  __device__ __forceinline__ void offer(bool has, uint64_t key, int lane) {
    uint64_t m = __ballot(has && key > thr);
    while (m) {
      const uint32_t src = (uint32_t)__builtin_ctzll(m);
      m &= m - 1;
      const uint64_t nk = readlane64(key, src);
      if (nk <= thr) continue;
      uint32_t pos = 0;
#pragma unroll
      for (int r = 0; r < KPL; ++r) pos += (uint32_t)__popcll(__ballot(v[r] > nk));
#pragma unroll
      for (int r = KPL - 1; r >= 0; --r) {
        uint64_t up = __shfl_up(v[r], 1, WAVE);
        if (r > 0) {
          const uint64_t carry = readlane64(v[r - 1], 63);
          if (lane == 0) up = carry;
        }
        const uint32_t rank = (uint32_t)r * 64u + (uint32_t)lane;
        if (rank > pos)
          v[r] = up;
        else if (rank == pos)
          v[r] = nk;
      }
      refresh_thr();
    }
  }
};

// ------------------------------------------------------------------ per-wave LDS hash table
struct WaveTable {
  uint32_t doc[TQD_PH_SLOTS];
  uint32_t val[TQD_PH_SLOTS];
};
__device__ __forceinline__ uint32_t ht_insert(WaveTable &tb, uint32_t doc, uint32_t val) {
  uint32_t h = doc & (TQD_PH_SLOTS - 1u);
  for (;;) {
    const uint32_t prev = atomicCAS(&tb.doc[h], EMPTY_SLOT, doc);
    if (prev == EMPTY_SLOT) break;
    h = (h + 1u) & (TQD_PH_SLOTS - 1u);
  }
  tb.val[h] = val;
  return h;
}
__device__ __forceinline__ bool ht_find(const WaveTable &tb, uint32_t doc, uint32_t &slot) {
  uint32_t h = doc & (TQD_PH_SLOTS - 1u);
  for (;;) {
    const uint32_t d = tb.doc[h];
    if (d == doc) {
      slot = h;
      return true;
    }
    if (d == EMPTY_SLOT) return false;
    h = (h + 1u) & (TQD_PH_SLOTS - 1u);
  }
}
__device__ __forceinline__ void wave_mem_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------ chunk -> query bookkeeping
// largest q with tile_starts[q] <= t  (tile_starts has n_queries+1 entries, non-decreasing)
__device__ __forceinline__ uint32_t find_query(const uint32_t *tile_starts, uint32_t n_queries,
                                               uint32_t t) {
  uint32_t lo = 0, hi = n_queries;  // answer in [lo, hi)
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (uni(tile_starts[mid]) <= t)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

template <int KPL>
__device__ __forceinline__ void flush_partial(const TopK<KPL> &tk, uint64_t *partials,
                                              uint32_t part, int lane) {
  uint64_t *dst = partials + (uint64_t)part * (uint64_t)(KPL * 64);
#pragma unroll
  for (int r = 0; r < KPL; ++r) {
    const uint32_t rank = (uint32_t)r * 64u + (uint32_t)lane;
    dst[rank] = rank < tk.k ? tk.v[r] : 0ull;
  }
}

// =================================================================== AND kernel
// block_wand_intersection (src/query/boolean_query/block_wand_intersection.rs:19-179) restated for
// wavefronts.  Terms are ordered by doc freq ascending; term 0 is the leader.  Tile = 64
// consecutive leader blocks; one wavefront = one chunk of consecutive tiles.
//   1. pre-filter, one LANE per leader block: O(1) seek_block of the block's doc range in every
//      other list; drop blocks past the end of a list and (pruned mode) blocks whose block-max sum
//      cannot reach the threshold (:81-85);
//   2. per surviving leader block, the whole wave: decode 128 docs + tfs (2 per lane), gather the
//      fieldnorm bytes, score the leader term (:107-125);
//   3. per other term, ascending doc freq: per-lane seek_block of every live candidate; pruned
//      mode drops candidates whose partial score + block-max of that block cannot reach the
//      threshold (:144-165); the distinct blocks that still hold candidates are decoded once each
//      (doc ids only) and searched; tfs are fetched individually for the docs found;
//   4. matches are offered to the wave's register top-k; pruned mode also publishes their score
//      into the query's 64 threshold slots (atomic max, fire and forget).  The k-th largest slot
//      is a lower bound of the final k-th best score (every slot holds a distinct real match),
//      monotone like the callback's threshold in the reference (:141-143,168-174).
// Candidates equal to the threshold are kept (>=, not >), so ties on the k-th score still resolve
// by doc id exactly as TopNHeap does; results are identical with and without pruning.
__device__ __forceinline__ uint32_t sortable(float x) {
  uint32_t fb = __float_as_uint(x);
  return fb ^ ((uint32_t)((int32_t)fb >> 31) | 0x80000000u);
}
// upper bound of a term's score inside one block (TermScorer::block_max_score,
// term_scorer.rs:58-75; skip.rs:175-184).  tail / no-freq / unknown block-max => weight itself
// (tf/(tf+norm) < 1).  Requires weight >= 0.
__device__ __forceinline__ float block_max_score(uint32_t meta, float w, const float *cache,
                                                 uint32_t has_freq) {
  const uint32_t tfc = meta >> 24;
  if (meta == META_TAIL || !has_freq || tfc == 0u) return w;
  const uint32_t tf = tfc == 255u ? 0xFFFFFFFFu : tfc;  // skip.rs:31-43
  return bm25(w, cache[(meta >> 16) & 0xFFu], tf);
}
// k-th largest of the 64 per-lane values (0 = empty slot); 0 if fewer than k are set
__device__ __forceinline__ uint32_t kth_largest64(uint32_t v, uint32_t k) {
  uint32_t best = 0;
  for (int i = 0; i < 64; ++i) {
    const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v, i);
    const uint32_t c = (uint32_t)__popcll(__ballot(v >= x));
    if (c >= k && x > best) best = x;
  }
  return best;
}

template <int KPL, bool USE_DPP>
__global__ __launch_bounds__(TQD_WAVES_PER_WG * 64) void and_kernel(TqkScanParams p) {
  __shared__ uint32_t lds_docs[TQD_WAVES_PER_WG][128];
  const int lane = (int)__lane_id();
  const uint32_t wave = uni(threadIdx.x >> 6);
  uint32_t *const blk = lds_docs[wave];

  const uint32_t chunk = blockIdx.x * TQD_WAVES_PER_WG + wave;
  if (chunk >= p.n_chunks) return;
  const uint32_t t_begin = chunk * p.tiles_per_chunk;
  uint32_t t_end = t_begin + p.tiles_per_chunk;
  if (t_end > p.total_tiles) t_end = p.total_tiles;

  const TqdSegment seg = p.seg;
  const uint8_t *idx = uni_ptr(seg.idx);

  uint32_t q = find_query(p.tile_starts, p.n_queries, t_begin);
  uint32_t q_tile_start = uni(p.tile_starts[q]);
  uint32_t q_tile_end = uni(p.tile_starts[q + 1]);
  const TqdQuery *Q = uni_ptr(p.queries + q);
  TopK<KPL> tk;
  tk.reset(uni(Q->k));
  uint32_t n_matches = 0;

  for (uint32_t t = t_begin; t < t_end; ++t) {
    while (t >= q_tile_end) {  // next query (queries with zero tiles are skipped)
      if (q_tile_end > q_tile_start && q_tile_end > t_begin) {  // this chunk touched query q
        const uint32_t part =
            uni(Q->part_start) + (chunk - q_tile_start / p.tiles_per_chunk);
        flush_partial<KPL>(tk, p.partials, part, lane);
      }
      ++q;
      q_tile_start = q_tile_end;
      q_tile_end = uni(p.tile_starts[q + 1]);
      Q = uni_ptr(p.queries + q);
      tk.reset(uni(Q->k));
    }
    const uint32_t nt = uni(Q->n_terms);
    const float *cache = uni_ptr(p.caches + (size_t)uni(Q->cache_idx) * 256u);
    const TermRef lead = load_term(p.terms, uni(Q->term[0]));
    const float w_lead = __uint_as_float(uni(__float_as_uint(Q->weight[0])));
    const bool prune = (uni(Q->flags) & TQD_QF_PRUNE) != 0u;
    const uint32_t thr_index = uni(Q->thr_index);
    uint32_t *slots = (prune && thr_index != 0xFFFFFFFFu)
                          ? p.thr_slots + (size_t)thr_index * TQD_THR_SLOTS
                          : nullptr;

    // threshold (sortable score bits): own k-th key and the k-th largest shared slot
    uint32_t thr = 0, thr_g = 0;
    if (prune) {
      if (slots) {
        const uint32_t sv = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        thr_g = kth_largest64(sv, tk.k);
      }
      thr = (uint32_t)(tk.thr >> 32);
      if (thr_g > thr) thr = thr_g;
    }

    // ---- 1. pre-filter: lane <-> leader block
    const uint32_t i_base = (t - q_tile_start) * TQD_AND_TILE;
    const uint32_t i_mine = i_base + (uint32_t)lane;
    bool surv = i_mine < lead.n_blocks;
    {
      uint32_t first = 0, last = 0;
      float ub = 0.0f;
      if (surv) {
        first = i_mine ? lead.last_doc[i_mine - 1u] + 1u : 0u;
        last = lead.last_doc[i_mine];
        if (prune) ub = block_max_score(lead.meta[i_mine], w_lead, cache, lead.has_freq);
      }
      for (uint32_t m = 1; m < nt; ++m) {
        const TermRef tr = load_term(p.terms, uni(Q->term[m]));
        const float w = __uint_as_float(uni(__float_as_uint(Q->weight[m])));
        if (surv) {
          const uint32_t j0 = seek_block(tr, first);
          if (j0 >= tr.n_blocks) {
            surv = false;  // the list ends before this leader block starts
          } else if (prune) {
            uint32_t j1 = seek_block(tr, last);
            if (j1 >= tr.n_blocks) j1 = tr.n_blocks - 1u;
            float bound = w;
            if (j1 - j0 <= 3u) {
              bound = block_max_score(tr.meta[j0], w, cache, tr.has_freq);
              for (uint32_t j = j0 + 1u; j <= j1; ++j) {
                const float b2 = block_max_score(tr.meta[j], w, cache, tr.has_freq);
                bound = b2 > bound ? b2 : bound;
              }
            }
            ub = ub + bound;
          }
        }
      }
      if (prune && nt > 2u) ub *= 1.000001f;  // the bound is summed in another order than scores
      if (prune && surv) surv = sortable(ub) >= thr;
    }
    uint64_t todo = __ballot(surv);

    // ---- 2..4: surviving leader blocks, one at a time, whole wave
    while (todo) {
      const uint32_t i = i_base + (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      const Dec d = decode_block<USE_DPP, false>(idx, lead, i, lane);
      const uint32_t c0 = d.d0, c1 = d.d1;
      bool alive0 = c0 != TQD_TERMINATED, alive1 = c1 != TQD_TERMINATED;
      float norm0 = 0.0f, norm1 = 0.0f, s0 = 0.0f, s1 = 0.0f;
      if (alive0) {
        norm0 = cache[fieldnorm_id(seg, c0)];
        s0 = bm25(w_lead, norm0, d.t0);
      }
      if (alive1) {
        norm1 = cache[fieldnorm_id(seg, c1)];
        s1 = bm25(w_lead, norm1, d.t1);
      }
      // bound of the terms after the current one (3+ terms, pruned mode)
      float rest = 0.0f;
      if (prune)
        for (uint32_t m = 2; m < nt; ++m) rest += __uint_as_float(uni(__float_as_uint(Q->weight[m])));

      for (uint32_t m = 1; m < nt; ++m) {
        const TermRef tr = load_term(p.terms, uni(Q->term[m]));
        const float w = __uint_as_float(uni(__float_as_uint(Q->weight[m])));
        uint32_t jb0 = tr.n_blocks, jb1 = tr.n_blocks;
        if (alive0) jb0 = seek_block(tr, c0);
        if (alive1) jb1 = seek_block(tr, c1);
        alive0 = alive0 && jb0 < tr.n_blocks;
        alive1 = alive1 && jb1 < tr.n_blocks;
        uint32_t meta0 = 0, meta1 = 0;
        if (alive0) meta0 = tr.meta[jb0];
        if (alive1) meta1 = tr.meta[jb1];
        if (prune) {
          if (alive0) {
            float ub = s0 + block_max_score(meta0, w, cache, tr.has_freq);
            if (nt > 2u) ub = (ub + rest) * 1.000001f;
            alive0 = sortable(ub) >= thr;
          }
          if (alive1) {
            float ub = s1 + block_max_score(meta1, w, cache, tr.has_freq);
            if (nt > 2u) ub = (ub + rest) * 1.000001f;
            alive1 = sortable(ub) >= thr;
          }
          if (m + 1u < nt) rest -= __uint_as_float(uni(__float_as_uint(Q->weight[m + 1u])));
          if (rest < 0.0f) rest = 0.0f;
        }
        // distinct blocks of list m that still hold candidates, ascending
        uint64_t pend0 = __ballot(alive0), pend1 = __ballot(alive1);
        uint32_t at0 = 0xFFFFFFFFu, at1 = 0xFFFFFFFFu;  // index of the doc inside its block
        while (pend0 | pend1) {
          const uint32_t l0 = pend0 ? (uint32_t)__builtin_ctzll(pend0) : 64u;
          const uint32_t l1 = pend1 ? (uint32_t)__builtin_ctzll(pend1) : 64u;
          const uint32_t j = l0 <= l1 ? (uint32_t)__builtin_amdgcn_readlane((int)jb0, (int)l0)
                                      : (uint32_t)__builtin_amdgcn_readlane((int)jb1, (int)l1);
          const bool in0 = alive0 && jb0 == j, in1 = alive1 && jb1 == j;
          uint64_t m0 = __ballot(in0), m1 = __ballot(in1);
          pend0 &= ~m0;
          pend1 &= ~m1;
          const uint32_t meta_j = uni(tr.meta[j]);
          uint32_t x0, x1;
          decode_docs<USE_DPP>(idx, tr, j, meta_j, lane, x0, x1);
          const uint32_t n_in = (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
          if (n_in <= 4u) {  // few candidates: broadcast each, compare against the 128 docs
            while (m0 | m1) {
              const bool from0 = m0 != 0ull && (m1 == 0ull || __builtin_ctzll(m0) <= __builtin_ctzll(m1));
              const uint32_t l = (uint32_t)__builtin_ctzll(from0 ? m0 : m1);
              if (from0)
                m0 &= m0 - 1ull;
              else
                m1 &= m1 - 1ull;
              const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)(from0 ? c0 : c1), (int)l);
              const uint64_t h0 = __ballot(x0 == c), h1 = __ballot(x1 == c);
              uint32_t at = 0xFFFFFFFFu;
              if (h0)
                at = 2u * (uint32_t)__builtin_ctzll(h0);
              else if (h1)
                at = 2u * (uint32_t)__builtin_ctzll(h1) + 1u;
              if ((uint32_t)lane == l) {
                if (from0)
                  at0 = at;
                else
                  at1 = at;
              }
            }
          } else {  // many candidates: the block goes to LDS, every candidate binary-searches it
            *reinterpret_cast<uint2 *>(blk + 2 * lane) = make_uint2(x0, x1);
            wave_mem_fence();
            if (in0) {
              uint32_t pos = 0;
#pragma unroll
              for (uint32_t step = 64u; step > 0u; step >>= 1)
                if (blk[pos + step - 1u] < c0) pos += step;
              at0 = blk[pos] == c0 ? pos : 0xFFFFFFFFu;
            }
            if (in1) {
              uint32_t pos = 0;
#pragma unroll
              for (uint32_t step = 64u; step > 0u; step >>= 1)
                if (blk[pos + step - 1u] < c1) pos += step;
              at1 = blk[pos] == c1 ? pos : 0xFFFFFFFFu;
            }
            wave_mem_fence();
          }
        }
        // score the docs found in list m (leader first, then ascending doc freq: :144-165)
        if (alive0) {
          if (at0 != 0xFFFFFFFFu)
            s0 = s0 + bm25(w, norm0, block_tf_at(idx, tr, jb0, meta0, at0));
          else
            alive0 = false;
        }
        if (alive1) {
          if (at1 != 0xFFFFFFFFu)
            s1 = s1 + bm25(w, norm1, block_tf_at(idx, tr, jb1, meta1, at1));
          else
            alive1 = false;
        }
      }
      // ---- 4. collect
      const uint64_t hit0 = __ballot(alive0), hit1 = __ballot(alive1);
      if (hit0 | hit1) {
        n_matches += (uint32_t)__popcll(hit0) + (uint32_t)__popcll(hit1);
        const uint64_t key0 = alive0 ? make_key(s0, c0) : 0ull;
        const uint64_t key1 = alive1 ? make_key(s1, c1) : 0ull;
        if (slots) {
          const uint32_t b0 = (uint32_t)(key0 >> 32), b1 = (uint32_t)(key1 >> 32);
          if (alive0 && b0 > thr_g) atomicMax(slots + ((c0 * 0x9E3779B1u) >> 26), b0);
          if (alive1 && b1 > thr_g) atomicMax(slots + ((c1 * 0x9E3779B1u) >> 26), b1);
        }
        tk.offer(alive0, key0, lane);
        tk.offer(alive1, key1, lane);
        if (prune) {
          const uint32_t own = (uint32_t)(tk.thr >> 32);
          if (own > thr) thr = own;
        }
      }
    }
  }
  // final flush
  if (q_tile_end > q_tile_start) {
    const uint32_t part = uni(Q->part_start) + (chunk - q_tile_start / p.tiles_per_chunk);
    flush_partial<KPL>(tk, p.partials, part, lane);
  }
  if (lane == 0 && n_matches) atomicAdd(p.match_counter, (unsigned long long)n_matches);
}


// =================================================================== OR kernel (exhaustive union)
// One workgroup = one chunk of consecutive 4096-doc windows.  Per window: f32 accumulators in
// LDS, terms applied one after the other (SumCombiner order = query term order), the 4 waves
// splitting each term's blocks; then every present doc is offered to the waves' top-k.
template <int KPL, bool USE_DPP>
__global__ __launch_bounds__(TQD_WAVES_PER_WG * 64) void or_kernel(TqkScanParams p) {
  __shared__ float acc[TQD_OR_WINDOW];
  __shared__ uint32_t present[TQD_OR_WINDOW / 32];
  const int lane = (int)__lane_id();
  const uint32_t wave = uni(threadIdx.x >> 6);
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = blockIdx.x;
  if (chunk >= p.n_chunks) return;
  const uint32_t t_begin = chunk * p.tiles_per_chunk;
  uint32_t t_end = t_begin + p.tiles_per_chunk;
  if (t_end > p.total_tiles) t_end = p.total_tiles;

  const TqdSegment seg = p.seg;
  const uint8_t *idx = uni_ptr(seg.idx);
  uint32_t q = find_query(p.tile_starts, p.n_queries, t_begin);
  uint32_t q_tile_start = uni(p.tile_starts[q]);
  uint32_t q_tile_end = uni(p.tile_starts[q + 1]);
  const TqdQuery *Q = uni_ptr(p.queries + q);
  TopK<KPL> tk;
  tk.reset(uni(Q->k));
  uint32_t n_matches = 0;

  for (uint32_t t = t_begin; t < t_end; ++t) {
    while (t >= q_tile_end) {
      if (q_tile_end > q_tile_start && q_tile_end > t_begin) {
        const uint32_t part = uni(Q->part_start) +
                              (chunk - q_tile_start / p.tiles_per_chunk) * TQD_WAVES_PER_WG + wave;
        flush_partial<KPL>(tk, p.partials, part, lane);
      }
      ++q;
      q_tile_start = q_tile_end;
      q_tile_end = uni(p.tile_starts[q + 1]);
      Q = uni_ptr(p.queries + q);
      tk.reset(uni(Q->k));
    }
    const uint32_t nt = uni(Q->n_terms);
    const float *cache = uni_ptr(p.caches + (size_t)uni(Q->cache_idx) * 256u);
    const uint32_t base = (t - q_tile_start) * TQD_OR_WINDOW;
    const uint32_t win_hi = base + (TQD_OR_WINDOW - 1u);

    for (uint32_t i = tid; i < TQD_OR_WINDOW; i += TQD_WAVES_PER_WG * 64) acc[i] = 0.0f;
    if (tid < TQD_OR_WINDOW / 32) present[tid] = 0u;
    __syncthreads();

    for (uint32_t m = 0; m < nt; ++m) {
      const TermRef tr = load_term(p.terms, uni(Q->term[m]));
      const float w = __uint_as_float(uni(__float_as_uint(Q->weight[m])));
      const uint32_t jb = lower_bound_block(tr, base, lane);
      for (uint32_t j = jb + wave; j < tr.n_blocks; j += TQD_WAVES_PER_WG) {
        if (block_first_possible(tr, j) > win_hi) break;
        const Dec d = decode_block<USE_DPP, false>(idx, tr, j, lane);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint32_t doc = e ? d.d1 : d.d0;
          const uint32_t tf = e ? d.t1 : d.t0;
          if (doc >= base && doc <= win_hi) {
            const uint32_t o = doc - base;
            const float s = bm25(w, cache[fieldnorm_id(seg, doc)], tf);
            acc[o] = acc[o] + s;  // one posting per (term, doc): no intra-phase conflict
            atomicOr(&present[o >> 5], 1u << (o & 31u));
          }
        }
      }
      __syncthreads();
    }
    // harvest
    for (uint32_t i = tid; i < TQD_OR_WINDOW; i += TQD_WAVES_PER_WG * 64) {
      const bool has = (present[i >> 5] >> (i & 31u)) & 1u;
      const uint64_t key = has ? make_key(acc[i], base + i) : 0ull;
      n_matches += (uint32_t)__popcll(__ballot(has));
      tk.offer(has, key, lane);
    }
    __syncthreads();
  }
  if (q_tile_end > q_tile_start) {
    const uint32_t part = uni(Q->part_start) +
                          (chunk - q_tile_start / p.tiles_per_chunk) * TQD_WAVES_PER_WG + wave;
    flush_partial<KPL>(tk, p.partials, part, lane);
  }
  if (lane == 0 && n_matches) atomicAdd(p.match_counter, (unsigned long long)n_matches);
}

// positions: raw deltas of a whole term (PositionReader::read over everything)
__device__ __forceinline__ uint32_t position_delta(const uint8_t *pos, const TqdTerm *t,
                                                   uint64_t i) {
  const uint64_t pb = i >> 7;
  if (pb < t->n_pos_blocks)
    return unpack_one(pos + t->pos_block_off[pb], t->pos_widths[pb], (uint32_t)(i & 127u));
  return t->pos_tail[i - ((uint64_t)t->n_pos_blocks << 7)];
}

// =================================================================== phrase kernel
// Exact phrase (slop 0).  One wavefront per workgroup.  Doc candidates come from the same
// leader-hash / stream-probe intersection as the AND kernel; every (term, candidate) records the
// index of the doc's first position in the term's position stream and its tf.  Complete
// candidates are then checked lane-per-candidate with an n-way merge over adjusted positions
// (position + max_offset - term_offset, phrase_scorer.rs:372-385), fetching single bitpacked
// deltas by index (positions/reader.rs:84-101).  count = intersection_count (:437-461).
#define TQD_PH_CH 2
#define TQD_PH_CAP (TQD_PH_CH * 128)
#define TQD_PH_MAX_TERMS 8
struct PhraseLds {
  WaveTable tb;
  uint32_t cand_doc[TQD_PH_CAP];
  uint32_t cand_cnt[TQD_PH_CAP];
  uint32_t pidx[TQD_PH_MAX_TERMS][TQD_PH_CAP];
  uint32_t ptf[TQD_PH_MAX_TERMS][TQD_PH_CAP];
};

struct PosCursor {
  uint32_t idx, end, cur;
  bool valid;
};
__device__ __forceinline__ void pos_advance(PosCursor &c, const uint8_t *pos, const TqdTerm *t) {
  if (c.idx < c.end) {
    c.cur += position_delta(pos, t, c.idx);
    c.idx++;
  } else {
    c.valid = false;
  }
}

template <int KPL, bool USE_DPP>
__global__ __launch_bounds__(64) void phrase_kernel(TqkScanParams p) {
  __shared__ PhraseLds L;
  const int lane = (int)__lane_id();
  WaveTable &tb = L.tb;
  for (int i = lane; i < TQD_PH_SLOTS; i += WAVE) tb.doc[i] = EMPTY_SLOT;
  wave_mem_fence();
  const uint32_t chunk = blockIdx.x;
  if (chunk >= p.n_chunks) return;
  const uint32_t t_begin = chunk * p.tiles_per_chunk;
  uint32_t t_end = t_begin + p.tiles_per_chunk;
  if (t_end > p.total_tiles) t_end = p.total_tiles;
  const TqdSegment seg = p.seg;
  const uint8_t *idx = uni_ptr(seg.idx);
  const uint8_t *pos = uni_ptr(seg.pos);

  uint32_t q = find_query(p.tile_starts, p.n_queries, t_begin);
  uint32_t q_tile_start = uni(p.tile_starts[q]);
  uint32_t q_tile_end = uni(p.tile_starts[q + 1]);
  const TqdQuery *Q = uni_ptr(p.queries + q);
  TopK<KPL> tk;
  tk.reset(uni(Q->k));
  uint32_t n_matches = 0;

  for (uint32_t t = t_begin; t < t_end; ++t) {
    while (t >= q_tile_end) {
      if (q_tile_end > q_tile_start && q_tile_end > t_begin) {
        const uint32_t part = uni(Q->part_start) + (chunk - q_tile_start / p.tiles_per_chunk);
        flush_partial<KPL>(tk, p.partials, part, lane);
      }
      ++q;
      q_tile_start = q_tile_end;
      q_tile_end = uni(p.tile_starts[q + 1]);
      Q = uni_ptr(p.queries + q);
      tk.reset(uni(Q->k));
    }
    const uint32_t nt = uni(Q->n_terms);
    const float *cache = uni_ptr(p.caches + (size_t)uni(Q->cache_idx) * 256u);
    const float weight = __uint_as_float(uni(__float_as_uint(Q->weight[0])));
    const uint32_t h_lead = uni(Q->term[0]), h_drv = uni(Q->term[nt - 1]);
    const TermRef lead = load_term(p.terms, h_lead);
    const TermRef drv = load_term(p.terms, h_drv);
    const uint32_t *lead_bpos = uni_ptr(p.terms[h_lead].block_pos);

    const uint32_t tl = t - q_tile_start;
    const uint32_t j0 = tl * TQD_PH_M;
    uint32_t j1 = j0 + TQD_PH_M;
    if (j1 > drv.n_blocks) j1 = drv.n_blocks;
    const uint32_t lo1 = block_first_possible(drv, j0);
    const uint32_t hi = uni(drv.last_doc[j1 - 1]);
    const uint32_t i0 = lower_bound_block(lead, lo1, lane);
    if (i0 >= lead.n_blocks) continue;
    uint32_t iL = lower_bound_block(lead, hi, lane);
    if (iL >= lead.n_blocks) iL = lead.n_blocks - 1u;

    for (uint32_t ia = i0; ia <= iL; ia += TQD_PH_CH) {
      uint32_t ib = ia + TQD_PH_CH - 1u;
      if (ib > iL) ib = iL;
      uint32_t sub_lo1 = block_first_possible(lead, ia);
      if (sub_lo1 < lo1) sub_lo1 = lo1;
      uint32_t sub_hi = hi;
      if (ib != iL) {
        const uint32_t l = uni(lead.last_doc[ib]);
        if (l < sub_hi) sub_hi = l;
      }
      // ---- fill from the leader
      uint32_t my_slot[TQD_PH_CH * 2];
#pragma unroll
      for (int c = 0; c < TQD_PH_CH; ++c) {
        my_slot[2 * c] = EMPTY_SLOT;
        my_slot[2 * c + 1] = EMPTY_SLOT;
        const uint32_t ca = (uint32_t)c * 128u + 2u * (uint32_t)lane;
        L.cand_cnt[ca] = 0x80000000u;
        L.cand_cnt[ca + 1u] = 0x80000000u;
        if (ia + (uint32_t)c <= ib) {
          uint32_t e0, e1;
          const Dec d = decode_block<USE_DPP, true>(idx, lead, ia + (uint32_t)c, lane, &e0, &e1);
          const uint32_t bp = uni(lead_bpos[ia + (uint32_t)c]);
          if (d.d0 >= sub_lo1 && d.d0 <= sub_hi) {
            L.cand_doc[ca] = d.d0;
            L.cand_cnt[ca] = 0u;
            L.pidx[0][ca] = bp + e0;
            L.ptf[0][ca] = d.t0;
            my_slot[2 * c] = ht_insert(tb, d.d0, ca);
          }
          if (d.d1 >= sub_lo1 && d.d1 <= sub_hi) {
            L.cand_doc[ca + 1u] = d.d1;
            L.cand_cnt[ca + 1u] = 0u;
            L.pidx[0][ca + 1u] = bp + e1;
            L.ptf[0][ca + 1u] = d.t1;
            my_slot[2 * c + 1] = ht_insert(tb, d.d1, ca + 1u);
          }
        }
      }
      wave_mem_fence();
      // ---- every other term (middle lists searched, the driver restricted to the tile)
      for (uint32_t m = 1; m < nt; ++m) {
        const uint32_t h_m = uni(Q->term[m]);
        const TermRef tr = load_term(p.terms, h_m);
        const uint32_t *bpos = uni_ptr(p.terms[h_m].block_pos);
        uint32_t jb, jend;
        if (m + 1 == nt) {
          jb = j0;
          jend = j1;
        } else {
          jb = lower_bound_block(tr, sub_lo1, lane);
          jend = tr.n_blocks;
        }
        for (uint32_t j = jb; j < jend; ++j) {
          if (uni(tr.last_doc[j]) < sub_lo1) continue;
          if (block_first_possible(tr, j) > sub_hi) break;
          uint32_t e0, e1;
          const Dec d = decode_block<USE_DPP, true>(idx, tr, j, lane, &e0, &e1);
          const uint32_t bp = uni(bpos[j]);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const uint32_t doc = e ? d.d1 : d.d0;
            uint32_t slot;
            if (doc >= sub_lo1 && doc <= sub_hi && ht_find(tb, doc, slot)) {
              const uint32_t c = tb.val[slot];
              L.pidx[m][c] = bp + (e ? e1 : e0);
              L.ptf[m][c] = e ? d.t1 : d.t0;
              L.cand_cnt[c] = L.cand_cnt[c] + 1u;
            }
          }
          wave_mem_fence();
        }
      }
      // ---- position check, lane per candidate
      const uint32_t n_cand = (ib - ia + 1u) * 128u;
      for (uint32_t c0 = 0; c0 < n_cand; c0 += 64u) {
        const uint32_t c = c0 + (uint32_t)lane;
        bool has = false;
        uint64_t key = 0;
        if (L.cand_cnt[c] == nt - 1u) {
          PosCursor cur[TQD_PH_MAX_TERMS];
#pragma unroll
          for (int m = 0; m < TQD_PH_MAX_TERMS; ++m) {
            cur[m].valid = false;
            cur[m].idx = cur[m].end = cur[m].cur = 0;
            if ((uint32_t)m < nt) {
              const uint32_t pi = L.pidx[m][c];
              cur[m].idx = pi + 1u;
              cur[m].end = pi + L.ptf[m][c];
              cur[m].cur = Q->phrase_off[m] + position_delta(pos, p.terms + Q->term[m], pi);
              cur[m].valid = true;
            }
          }
          uint32_t count = 0;
          bool done = false;
          while (cur[0].valid && !done) {
            const uint32_t a = cur[0].cur;
            bool ok = true;
#pragma unroll
            for (int m = 1; m < TQD_PH_MAX_TERMS; ++m) {
              if ((uint32_t)m < nt && !done) {
                while (cur[m].valid && cur[m].cur < a) pos_advance(cur[m], pos, p.terms + Q->term[m]);
                if (!cur[m].valid)
                  done = true;
                else if (cur[m].cur != a)
                  ok = false;
              }
            }
            if (done) break;
            if (ok) {
              ++count;
#pragma unroll
              for (int m = 1; m < TQD_PH_MAX_TERMS; ++m)
                if ((uint32_t)m < nt) pos_advance(cur[m], pos, p.terms + Q->term[m]);
            }
            pos_advance(cur[0], pos, p.terms + Q->term[0]);
          }
          if (count > 0) {
            const uint32_t doc = L.cand_doc[c];
            has = true;
            key = make_key(bm25(weight, cache[fieldnorm_id(seg, doc)], count), doc);
          }
        }
        n_matches += (uint32_t)__popcll(__ballot(has));
        tk.offer(has, key, lane);
      }
      // ---- clear
      wave_mem_fence();
#pragma unroll
      for (int c = 0; c < TQD_PH_CH * 2; ++c)
        if (my_slot[c] != EMPTY_SLOT) tb.doc[my_slot[c]] = EMPTY_SLOT;
      wave_mem_fence();
    }
  }
  if (q_tile_end > q_tile_start) {
    const uint32_t part = uni(Q->part_start) + (chunk - q_tile_start / p.tiles_per_chunk);
    flush_partial<KPL>(tk, p.partials, part, lane);
  }
  if (lane == 0 && n_matches) atomicAdd(p.match_counter, (unsigned long long)n_matches);
}

// =================================================================== merge kernel
// One wavefront per query: reduce its partial lists to the final top-k, sorted.
template <int KPL>
__global__ __launch_bounds__(64) void merge_kernel(TqkMergeParams p) {
  const int lane = (int)__lane_id();
  const uint32_t q = blockIdx.x;
  if (q >= p.n_queries) return;
  const TqdQuery *Q = uni_ptr(p.queries + q);
  const uint32_t k = uni(Q->k);
  const uint32_t part_start = uni(Q->part_start), n_parts = uni(Q->n_parts);
  TopK<KPL> tk;
  tk.reset(k);
  for (uint32_t pi = 0; pi < n_parts; ++pi) {
    const uint64_t *src = p.partials + (uint64_t)(part_start + pi) * (uint64_t)(KPL * 64);
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
      const uint64_t key = src[(uint32_t)r * 64u + (uint32_t)lane];
      tk.offer(key != 0ull, key, lane);
    }
  }
  const uint32_t out_q = p.out_index ? p.out_index[q] : q;
  uint32_t count = 0;
#pragma unroll
  for (int r = 0; r < KPL; ++r) {
    const uint32_t rank = (uint32_t)r * 64u + (uint32_t)lane;
    const bool real = rank < k && tk.v[r] != 0ull;
    count += (uint32_t)__popcll(__ballot(real));
    if (rank < p.out_stride) {
      p.out_scores[(uint64_t)out_q * p.out_stride + rank] = real ? key_score(tk.v[r]) : 0.0f;
      p.out_docs[(uint64_t)out_q * p.out_stride + rank] = real ? key_doc(tk.v[r]) : TQD_TERMINATED;
    }
  }
  for (uint32_t rank = (uint32_t)(KPL * 64) + (uint32_t)lane; rank < p.out_stride; rank += 64u) {
    p.out_scores[(uint64_t)out_q * p.out_stride + rank] = 0.0f;
    p.out_docs[(uint64_t)out_q * p.out_stride + rank] = TQD_TERMINATED;
  }
  if (lane == 0) p.out_counts[out_q] = count;
}

// =================================================================== whole-list decode (codec parity)
template <bool USE_DPP>
__global__ __launch_bounds__(256) void decode_list_kernel(TqdSegment seg, const TqdTerm *terms,
                                                          uint32_t handle, uint32_t *docs,
                                                          uint32_t *tfs) {
  const int lane = (int)__lane_id();
  const uint32_t wave = uni(threadIdx.x >> 6);
  const TermRef t = load_term(terms, handle);
  const uint32_t j = blockIdx.x * 4u + wave;
  if (j >= t.n_blocks) return;
  const Dec d = decode_block<USE_DPP, false>(uni_ptr(seg.idx), t, j, lane);
  const uint32_t i0 = j * 128u + 2u * (uint32_t)lane;
  const uint32_t df = uni(terms[handle].doc_freq);
  if (i0 < df) {
    docs[i0] = d.d0;
    tfs[i0] = d.t0;
  }
  if (i0 + 1u < df) {
    docs[i0 + 1u] = d.d1;
    tfs[i0 + 1u] = d.t1;
  }
}

__global__ void decode_positions_kernel(TqdSegment seg, const TqdTerm *terms, uint32_t handle,
                                        uint32_t *out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = position_delta(seg.pos, terms + handle, i);
}

// =================================================================== cross-segment merge
// merge_top_k (sort_key_top_collector.rs:76-95): key = (score desc, segment_ord asc, doc asc).
// One wavefront per query; S*stride candidates; selection by repeated max (k <= 1024, tiny).
__global__ __launch_bounds__(64) void merge_segments_kernel(TqkSegMergeParams p) {
  const int lane = (int)__lane_id();
  const uint32_t q = blockIdx.x;
  if (q >= p.n_queries) return;
  const uint32_t total = p.n_segments * p.stride;
  const uint32_t want = p.offset + p.limit;
  // Each output rank r: the candidate with exactly r candidates ordered before it.
  // O(total^2 / 64) compares per query; total is S*k (e.g. 8*10).
  uint32_t n_valid = 0;
  for (uint32_t s = 0; s < p.n_segments; ++s) n_valid += p.counts[(uint64_t)s * p.n_queries + q];
  for (uint32_t c = (uint32_t)lane; c < total; c += 64u) {
    const uint32_t s = c / p.stride, i = c % p.stride;
    const uint32_t cnt = p.counts[(uint64_t)s * p.n_queries + q];
    if (i >= cnt) continue;
    const uint64_t at = ((uint64_t)s * p.n_queries + q) * p.stride + i;
    const float sc = p.scores[at];
    const uint32_t dc = p.docs[at];
    const uint32_t so = p.segment_ords ? p.segment_ords[s] : s;
    uint32_t before = 0;
    for (uint32_t s2 = 0; s2 < p.n_segments; ++s2) {
      const uint32_t cnt2 = p.counts[(uint64_t)s2 * p.n_queries + q];
      const uint32_t so2 = p.segment_ords ? p.segment_ords[s2] : s2;
      const uint64_t b2 = ((uint64_t)s2 * p.n_queries + q) * p.stride;
      for (uint32_t i2 = 0; i2 < cnt2; ++i2) {
        const float sc2 = p.scores[b2 + i2];
        const uint32_t dc2 = p.docs[b2 + i2];
        const bool lt = (sc2 > sc) || (sc2 == sc && (so2 < so || (so2 == so && dc2 < dc)));
        before += lt ? 1u : 0u;
      }
    }
    if (before >= p.offset && before < want) {
      const uint64_t o = (uint64_t)q * p.limit + (before - p.offset);
      p.out_scores[o] = sc;
      p.out_segment_ords[o] = so;
      p.out_docs[o] = dc;
    }
  }
  const uint32_t got = n_valid > p.offset ? (n_valid - p.offset < p.limit ? n_valid - p.offset : p.limit) : 0u;
  for (uint32_t r = got + (uint32_t)lane; r < p.limit; r += 64u) {
    const uint64_t o = (uint64_t)q * p.limit + r;
    p.out_scores[o] = 0.0f;
    p.out_segment_ords[o] = 0xFFFFFFFFu;
    p.out_docs[o] = TQD_TERMINATED;
  }
  if (lane == 0) p.out_counts[q] = got;
}

}  // namespace

// =================================================================== launch wrappers
template <int KPL>
static void launch_and_t(const TqkScanParams &p, bool dpp, dim3 grid, dim3 block, hipStream_t st) {
  if (dpp)
    and_kernel<KPL, true><<<grid, block, 0, st>>>(p);
  else
    and_kernel<KPL, false><<<grid, block, 0, st>>>(p);
}
template <int KPL>
static void launch_or_t(const TqkScanParams &p, bool dpp, dim3 grid, dim3 block, hipStream_t st) {
  if (dpp)
    or_kernel<KPL, true><<<grid, block, 0, st>>>(p);
  else
    or_kernel<KPL, false><<<grid, block, 0, st>>>(p);
}

hipError_t tqk_launch_and(const TqkScanParams &p, int kpl, bool use_dpp, hipStream_t st) {
  if (p.n_chunks == 0) return hipSuccess;
  const dim3 grid((p.n_chunks + TQD_WAVES_PER_WG - 1) / TQD_WAVES_PER_WG);
  const dim3 block(TQD_WAVES_PER_WG * 64);
  switch (kpl) {
    case 1: launch_and_t<1>(p, use_dpp, grid, block, st); break;
    case 2: launch_and_t<2>(p, use_dpp, grid, block, st); break;
    case 4: launch_and_t<4>(p, use_dpp, grid, block, st); break;
    default: launch_and_t<16>(p, use_dpp, grid, block, st); break;
  }
  return hipGetLastError();
}
hipError_t tqk_launch_or(const TqkScanParams &p, int kpl, bool use_dpp, hipStream_t st) {
  if (p.n_chunks == 0) return hipSuccess;
  const dim3 grid(p.n_chunks), block(TQD_WAVES_PER_WG * 64);
  switch (kpl) {
    case 1: launch_or_t<1>(p, use_dpp, grid, block, st); break;
    case 2: launch_or_t<2>(p, use_dpp, grid, block, st); break;
    case 4: launch_or_t<4>(p, use_dpp, grid, block, st); break;
    default: launch_or_t<16>(p, use_dpp, grid, block, st); break;
  }
  return hipGetLastError();
}
template <int KPL>
static void launch_phrase_t(const TqkScanParams &p, bool dpp, dim3 grid, dim3 block, hipStream_t st) {
  if (dpp)
    phrase_kernel<KPL, true><<<grid, block, 0, st>>>(p);
  else
    phrase_kernel<KPL, false><<<grid, block, 0, st>>>(p);
}
hipError_t tqk_launch_phrase(const TqkScanParams &p, int kpl, bool use_dpp, hipStream_t st) {
  if (p.n_chunks == 0) return hipSuccess;
  const dim3 grid(p.n_chunks), block(64);
  switch (kpl) {
    case 1: launch_phrase_t<1>(p, use_dpp, grid, block, st); break;
    case 2: launch_phrase_t<2>(p, use_dpp, grid, block, st); break;
    case 4: launch_phrase_t<4>(p, use_dpp, grid, block, st); break;
    default: launch_phrase_t<16>(p, use_dpp, grid, block, st); break;
  }
  return hipGetLastError();
}
hipError_t tqk_launch_merge(const TqkMergeParams &p, int kpl, hipStream_t st) {
  if (p.n_queries == 0) return hipSuccess;
  const dim3 grid(p.n_queries), block(64);
  switch (kpl) {
    case 1: merge_kernel<1><<<grid, block, 0, st>>>(p); break;
    case 2: merge_kernel<2><<<grid, block, 0, st>>>(p); break;
    case 4: merge_kernel<4><<<grid, block, 0, st>>>(p); break;
    default: merge_kernel<16><<<grid, block, 0, st>>>(p); break;
  }
  return hipGetLastError();
}
hipError_t tqk_launch_decode_list(const TqdSegment &seg, const TqdTerm *terms, uint32_t handle,
                                  uint32_t n_blocks, uint32_t *docs, uint32_t *tfs, bool use_dpp,
                                  hipStream_t st) {
  if (n_blocks == 0) return hipSuccess;
  const dim3 grid((n_blocks + 3) / 4), block(256);
  if (use_dpp)
    hipLaunchKernelGGL((decode_list_kernel<true>), grid, block, 0, st, seg, terms, handle, docs, tfs);
  else
    hipLaunchKernelGGL((decode_list_kernel<false>), grid, block, 0, st, seg, terms, handle, docs, tfs);
  return hipGetLastError();
}
hipError_t tqk_launch_decode_positions(const TqdSegment &seg, const TqdTerm *terms,
                                       uint32_t handle, uint32_t *out, uint64_t n,
                                       hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(decode_positions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     seg, terms, handle, out, n);
  return hipGetLastError();
}
hipError_t tqk_launch_merge_segments(const TqkSegMergeParams &p, hipStream_t st) {
  if (p.n_queries == 0) return hipSuccess;
  hipLaunchKernelGGL(merge_segments_kernel, dim3(p.n_queries), dim3(64), 0, st, p);
  return hipGetLastError();
}
