// tq_kernels.hip — gfx950 (CDNA4, wave64) kernels of the tantivy query-execution path.
//
// One wavefront decodes one 128-doc posting block: 64 lanes x 2 values, the bitpacked payload
// staged in LDS by one 16-byte load per lane, BitPacker4x funnel-shift unpack, DPP prefix sum for
// the strict-delta doc ids.  AND = leader-block tiles, candidates flowing through per-wave LDS
// queues (decode -> locate in the other list -> verify + score), dense lists probed through a
// bitmap + rank directory, block-max pruning against a threshold shared through atomic-max slots.
// OR = 4096-doc window of f32 accumulators in LDS per workgroup.  Top-k = per-wave sorted key
// registers, flushed as partial lists and merged by a second kernel.  No MFMA: this is integer /
// byte work bound by vector-memory issue, LDS and VALU issue.
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
// ------------------------------------------------------------------ block decode
// Wave-uniform data (term tables, query descriptors, skip entries of a uniform block index) is
// read through the constant address space: with a uniform address the compiler emits scalar
// loads (s_load_*), which leave the vector-memory pipeline — the bottleneck of these kernels —
// to the per-lane gathers.  The data is written by the host before the launch and never by a
// kernel.
#define TQ_AS4 __attribute__((address_space(4)))
template <typename X>
__device__ __forceinline__ X sload(const X *p) {
  static_assert(sizeof(X) % 4 == 0, "dword-sized objects only");
  const TQ_AS4 uint32_t *q = (const TQ_AS4 uint32_t *)(uintptr_t)p;
  uint32_t w[sizeof(X) / 4];
#pragma unroll
  for (size_t i = 0; i < sizeof(X) / 4; ++i) w[i] = q[i];
  X x;
  __builtin_memcpy(&x, w, sizeof(X));
  return x;
}

struct TermRef {
  const uint4 *rec;  // {last_doc, meta, byte_off, first position index}
  const uint32_t *coarse;
  const uint2 *dense;
  const uint32_t *tail_docs;
  const uint32_t *tail_tfs;
  uint64_t payload_base;
  uint32_t n_blocks;
  uint32_t n_tail;
  uint32_t has_freq;
  uint32_t shift;
};
__device__ __forceinline__ TermRef load_term(const TqdTerm *terms, uint32_t handle) {
  const TqdTermHead h = sload(reinterpret_cast<const TqdTermHead *>(terms + handle));
  TermRef r;
  r.rec = h.rec;
  r.coarse = h.coarse;
  r.dense = h.dense;
  r.tail_docs = h.tail_docs;
  r.tail_tfs = h.tail_tfs;
  r.payload_base = h.payload_base;
  r.n_blocks = h.n_blocks;
  r.n_tail = h.n_tail;
  r.has_freq = h.has_freq;
  r.shift = h.coarse_shift;
  return r;
}
// j wave-uniform
__device__ __forceinline__ uint32_t block_prev_last(const TermRef &t, uint32_t j) {
  return j ? sload(&t.rec[j - 1u].x) : 0u;
}
__device__ __forceinline__ uint2 uni_mo(const TermRef &t, uint32_t j) {
  const uint4 r = sload(t.rec + j);
  return make_uint2(r.y, r.z);
}
__device__ __forceinline__ uint2 rec_mo(const uint4 &r) { return make_uint2(r.y, r.z); }

struct Dec {
  uint32_t d0, d1;  // doc ids (TQD_TERMINATED padded)
  uint32_t t0, t1;  // term freqs
};
constexpr uint32_t META_TAIL = 0xFFFFFFFFu;

// doc ids of one block (wave-uniform mo = {meta, byte_off}; prev = last doc of the previous
// block, 0 for block 0): lane t gets docs 2t, 2t+1
template <bool USE_DPP>
__device__ __forceinline__ void finish_docs(uint32_t x0, uint32_t x1, uint32_t strict,
                                            uint32_t prev, int lane, uint32_t &d0, uint32_t &d1) {
  const uint32_t a0 = x0 + strict;
  const uint32_t a1 = a0 + x1 + strict;
  const uint32_t incl = wave_inclusive_scan<USE_DPP>(a1, lane);
  // compression/mod.rs:36-39,112-121: offset 0 <=> None <=> seed u32::MAX (wrapping)
  const uint32_t base = (strict && prev == 0u) ? 0xFFFFFFFFu : prev;
  const uint32_t excl = base + (incl - a1);
  d0 = excl + a0;
  d1 = excl + a1;
}
template <bool USE_DPP>
__device__ __forceinline__ void decode_docs(const uint8_t *idx, const TermRef &t, uint2 mo,
                                            uint32_t prev, int lane, uint32_t &d0, uint32_t &d1) {
  if (mo.x == META_TAIL) {  // vint tail, pre-decoded at term_prepare
    const uint32_t i0 = 2u * (uint32_t)lane, i1 = i0 + 1u;
    d0 = i0 < t.n_tail ? t.tail_docs[i0] : TQD_TERMINATED;
    d1 = i1 < t.n_tail ? t.tail_docs[i1] : TQD_TERMINATED;
    return;
  }
  const uint8_t *p = idx + t.payload_base + mo.y;
  uint32_t x0, x1;
  unpack2(p, mo.x & 31u, lane, x0, x1);
  finish_docs<USE_DPP>(x0, x1, (mo.x >> 6) & 1u, prev, lane, d0, d1);
}
// term freqs of the same block and lane layout (padding of the tail reads as tf 0)
__device__ __forceinline__ void decode_tfs(const uint8_t *idx, const TermRef &t, uint2 mo,
                                           int lane, uint32_t &t0, uint32_t &t1) {
  if (mo.x == META_TAIL) {
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
  const uint32_t doc_bits = mo.x & 31u;
  const uint32_t strict = (mo.x >> 6) & 1u;
  const uint32_t tf_bits = (mo.x >> 8) & 0xFFu;
  const uint8_t *p = idx + t.payload_base + mo.y + 16u * doc_bits;
  unpack2(p, tf_bits, lane, t0, t1);
  t0 += strict;  // minus-one encoding is tied to the strict flag
  t1 += strict;  // (block_segment_postings.rs:45-57)
}

// ---- LDS-staged variant: the block's payload (16*(doc_bits+tf_bits) <= 1008 bytes) is fetched
// with ONE 16-byte load per lane, parked in LDS, and the 4 interleaved bit streams are unpacked
// from there (two ds_read_b64 per stream pair instead of two global loads).
struct __attribute__((packed, aligned(1))) U4Unaligned {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ void stage_payload(uint32_t *pay, const uint8_t *p, uint32_t nbytes,
                                              int lane) {
  const uint32_t o = 16u * (uint32_t)lane;
  if (o < nbytes) {
    const U4Unaligned v = *reinterpret_cast<const U4Unaligned *>(p + o);
    *reinterpret_cast<uint4 *>(pay + 4 * lane) = make_uint4(v.x, v.y, v.z, v.w);
  }
}
// pay4 = first 16-byte row of the stream group inside the LDS copy
__device__ __forceinline__ void unpack2_lds(const uint32_t *pay4, uint32_t b, int lane,
                                            uint32_t &v0, uint32_t &v1) {
  if (b == 0) {  // wave-uniform
    v0 = 0;
    v1 = 0;
    return;
  }
  const uint32_t k = (uint32_t)lane >> 1;
  const uint32_t bitpos = k * b;
  const uint32_t w = bitpos >> 5, s = bitpos & 31u;
  const uint32_t *q = pay4 + 4u * w + 2u * ((uint32_t)lane & 1u);
  const uint2 lo = *reinterpret_cast<const uint2 *>(q);
  const uint2 hi = *reinterpret_cast<const uint2 *>(q + 4);  // row w+1 (stale bytes are masked)
  const uint32_t mask = (b >= 32u) ? 0xFFFFFFFFu : ((1u << b) - 1u);
  v0 = __funnelshift_r(lo.x, hi.x, s) & mask;
  v1 = __funnelshift_r(lo.y, hi.y, s) & mask;
}

// term freq of the posting at slot i (0..127) of a block; mo and i may differ per lane
__device__ __forceinline__ uint32_t block_tf_at(const uint8_t *idx, const TermRef &t, uint2 mo,
                                                uint32_t i) {
  if (!t.has_freq) return 1u;
  if (mo.x == META_TAIL) return t.tail_tfs[i];
  const uint32_t doc_bits = mo.x & 31u;
  const uint32_t strict = (mo.x >> 6) & 1u;
  const uint32_t tf_bits = (mo.x >> 8) & 0xFFu;
  if (tf_bits == 0u) return strict;
  const uint8_t *p = idx + t.payload_base + mo.y + 16u * doc_bits;
  const uint32_t k = i >> 2, l = i & 3u;
  const uint32_t bitpos = k * tf_bits;
  const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
  const uint8_t *q = p + 16u * w + 4u * l;
  const uint32_t lo = ld_u1(q);
  uint32_t hi = 0;
  if (sh + tf_bits > 32u) hi = ld_u1(q + 16);  // the value straddles two words of its stream
  const uint32_t mask = (tf_bits >= 32u) ? 0xFFFFFFFFu : ((1u << tf_bits) - 1u);
  return (__funnelshift_r(lo, hi, sh) & mask) + strict;
}

// WANT_TF_SCAN: also return the exclusive prefix sum of the tfs (position index inside the block)
template <bool USE_DPP, bool WANT_TF_SCAN>
__device__ __forceinline__ Dec decode_block(const uint8_t *idx, const TermRef &t, uint32_t j,
                                            int lane, uint32_t *tf_excl0 = nullptr,
                                            uint32_t *tf_excl1 = nullptr) {
  Dec r;
  const uint2 mo = uni_mo(t, j);
  const uint32_t prev = block_prev_last(t, j);
  decode_docs<USE_DPP>(idx, t, mo, prev, lane, r.d0, r.d1);
  decode_tfs(idx, t, mo, lane, r.t0, r.t1);
  if (WANT_TF_SCAN) {
    const uint32_t s = r.t0 + r.t1;
    const uint32_t incl = wave_inclusive_scan<USE_DPP>(s, lane);
    *tf_excl0 = incl - s;
    *tf_excl1 = incl - s + r.t0;
  }
  return r;
}

// first block index j in [0, n_blocks) with last_doc(j) >= doc, else n_blocks, for a per-lane
// target (BlockSegmentPostings::seek_block, skip.rs:263-273, made O(1)):
// the coarse table brackets the answer, a short binary search finishes.  doc < max_doc.
__device__ __forceinline__ uint32_t seek_block(const TermRef &t, uint32_t doc) {
  const uint32_t b = doc >> t.shift;
  uint32_t lo = t.coarse[b], hi = t.coarse[b + 1u];
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (t.rec[mid].x >= doc)
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
// AliveBitSet::is_alive (src/fastfield/alive_bitset.rs:58-61; ReadOnlyBitSet::contains,
// common/src/bitset.rs:305-311): deleted docs never reach the collector
// (sort_by_score.rs:44-53); statistics and block-max metadata still include them.
__device__ __forceinline__ bool doc_is_alive(const TqdSegment &seg, uint32_t doc) {
  return !seg.alive || ((seg.alive[doc >> 3] >> (doc & 7u)) & 1u);
}
__device__ __forceinline__ float bm25(float weight, float norm, uint32_t tf) {
  const float f = (float)tf;
  return weight * (f / (f + norm));  // bm25.rs:179-193; compiled with -ffp-contract=off
}

// The same quantity for threshold tests only: v_rcp_f32 (1 ulp) instead of the IEEE division
// sequence.  Three roundings of <= 1 ulp each: callers widen the bound by 1.000002 (> 16 ulp).
__device__ __forceinline__ float bm25_bound(float weight, float norm, uint32_t tf) {
  const float f = (float)tf;
  return weight * (f * __builtin_amdgcn_rcpf(f + norm));
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
    if (sload(tile_starts + mid) <= t)
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
// consecutive leader blocks; one wavefront = one chunk of consecutive tiles.  The work of one
// leader block is cut into three stages joined by per-wave LDS queues, so that every gather runs
// with (nearly) all 64 lanes carrying a live candidate — vector-memory instructions, not bytes,
// are what this kernel is short of:
//   pre-filter (one LANE per leader block): O(1) seek_block of the block's doc range in the other
//      lists; drop blocks past the end of a list and, pruned mode, blocks whose block-max sum
//      cannot reach the threshold (:81-85);
//   A  (whole wave, one leader block): ONE 16-byte load per lane stages the bitpacked doc+tf
//      payload in LDS; unpack + DPP prefix sum; pruned mode keeps the candidates whose tf-only
//      score bound can reach the threshold; survivors -> queue 1;
//   B  (64 candidates, one per lane): pruned mode scores the leader term exactly (fieldnorm gather)
//      and filters (:107-125); locates the candidate in list 1 — dense lists: one bitmap/rank
//      load gives membership and the posting index; others: O(1) seek_block — and, pruned mode,
//      filters on the block-max of that block (:144-165); survivors -> queue 2;
//   C  (64 candidates): verifies membership (non-dense lists: the distinct blocks are decoded once
//      each and searched), fetches the tfs of the docs found, scores in the reference's order
//      (leader, then ascending doc freq), runs the remaining lists of a 3+ term query, and offers
//      the matches to the wave's register top-k.  Pruned mode also publishes each match's score
//      into the query's 64 threshold slots (atomic max, fire and forget): the k-th largest slot
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
// block_max_score for threshold tests (see bm25_bound)
__device__ __forceinline__ float block_max_bound(uint32_t meta, float w, const float *cache,
                                                 uint32_t has_freq) {
  const uint32_t tfc = meta >> 24;
  if (meta == META_TAIL || !has_freq || tfc == 0u) return w;
  const uint32_t tf = tfc == 255u ? 0xFFFFFFFFu : tfc;
  return bm25_bound(w, cache[(meta >> 16) & 0xFFu], tf);
}
// k-th largest of the 64*S per-lane values (0 = empty slot); 0 if fewer than k are set.
// Radix select: the largest x with |{v >= x}| >= k, one bit per step.
template <int S>
__device__ __forceinline__ uint32_t kth_largest_multi(const uint32_t (&v)[S], uint32_t k) {
  uint32_t ans = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t trial = ans | (1u << bit);
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < S; ++r) c += (uint32_t)__popcll(__ballot(v[r] >= trial));
    if (c >= k) ans = trial;
  }
  return ans;
}
__device__ __forceinline__ uint32_t kth_largest64(uint32_t v, uint32_t k) {
  const uint32_t a[1] = {v};
  return kth_largest_multi<1>(a, k);
}
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
constexpr uint32_t NOT_FOUND = 0xFFFFFFFFu;

template <bool DENSE>
struct AndLdsT {  // per wavefront
  // staged bitpacked payload: one leader block (stage A, <= 1008 B) or four 512-byte regions
  // (find_in_blocks, not needed when every other list has a bitmap), + one spare 16-byte row
  uint32_t pay[DENSE ? 260 : 516];
  // queue 1 holds < 64 leftovers + one block (128), queue 2 < 64 leftovers + one batch (64):
  // sized to the entry so that the lean instantiation fits 32 wavefronts per CU (5120 B each)
  uint32_t q1_doc[191], q1_tf[191];
  uint32_t q2_doc[127], q2_tf[127], q2_loc[127];
  float cache[256];    // Bm25Weight.cache of the current query
};

// Where is `doc` inside block jb of list tr?  (lane-private jb/doc; lanes with !alive idle.)
// Up to FOUR distinct blocks are decoded per step, one per 16-lane row: a lane unpacks 8
// consecutive values (registers 2r, 2r+1 of all four bit streams: two ds_read_b128 each), sums
// them locally and the row finishes the prefix sum with 4 DPP row shifts.  The doc ids replace the
// payload in LDS and every candidate binary-searches its block (search_block,
// block_search.rs:38-76).  Seeks into sparse lists land in many different blocks; decoding them
// one per wave-step (as stage A does for the leader, where all 128 docs are wanted) would leave
// this path with a quarter of the throughput.
// TFS = false: `key` is a doc id, the result is its slot in block jb (or NOT_FOUND).
// TFS = true (phrase queries): `key` is a slot; the block's term freqs are decoded instead and
// the result is the posting's tf, with *excl = sum of the tfs before it (its first position's
// index inside the block, segment_postings.rs:232-254).
template <bool TFS>
__device__ __forceinline__ uint32_t lookup_in_blocks(const uint8_t *idx, const TermRef &tr,
                                                     uint32_t jb, uint32_t key, bool alive,
                                                     uint32_t *P, int lane, uint32_t *excl) {
  uint32_t result = NOT_FOUND;
  uint64_t pend = __ballot(alive);
  const uint32_t row = (uint32_t)lane >> 4, l16 = (uint32_t)lane & 15u;
  // P: 4 regions of 128 words — the payload, then the decoded values
  while (pend) {
    // ---- up to four distinct blocks among the pending candidates
    uint32_t js[4] = {0u, 0u, 0u, 0u};
    uint32_t gid = 4u, n_groups = 0;
#pragma unroll
    for (uint32_t g = 0; g < 4u; ++g) {
      if (pend) {
        const uint32_t l = (uint32_t)__builtin_ctzll(pend);
        js[g] = (uint32_t)__builtin_amdgcn_readlane((int)jb, (int)l);
        const bool in = alive && gid == 4u && jb == js[g];
        if (in) gid = g;
        pend &= ~__ballot(in);
        n_groups = g + 1u;
      }
    }
    // ---- row r decodes block js[r]
    const uint32_t my_j = row == 0u ? js[0] : (row == 1u ? js[1] : (row == 2u ? js[2] : js[3]));
    const bool row_on = row < n_groups;
    uint4 rec = make_uint4(0u, META_TAIL, 0u, 0u);
    uint32_t prev = 0;
    if (row_on) {
      rec = tr.rec[my_j];
      if (!TFS && my_j) prev = tr.rec[my_j - 1u].x;
    }
    const bool is_tail = rec.y == META_TAIL;
    const uint32_t doc_bits = rec.y & 31u;
    // width of the stream to unpack: doc deltas (<= 31 bits, skip.rs:16-22) or tfs (<= 32)
    const uint32_t b = is_tail ? 0u : (TFS ? (tr.has_freq ? (rec.y >> 8) & 0xFFu : 0u) : doc_bits);
    const uint32_t strict = is_tail ? 0u : (rec.y >> 6) & 1u;
    wave_mem_fence();
    if (row_on && !is_tail) {
      const uint8_t *src = idx + tr.payload_base + rec.z + (TFS ? 16u * doc_bits : 0u) + 16u * l16;
      if (l16 < b) {
        const U4Unaligned v = *reinterpret_cast<const U4Unaligned *>(src);
        *reinterpret_cast<uint4 *>(P + row * 128u + 4u * l16) = make_uint4(v.x, v.y, v.z, v.w);
      }
      if (16u + l16 < b) {
        const U4Unaligned v = *reinterpret_cast<const U4Unaligned *>(src + 256);
        *reinterpret_cast<uint4 *>(P + row * 128u + 64u + 4u * l16) =
            make_uint4(v.x, v.y, v.z, v.w);
      }
    }
    wave_mem_fence();
    uint32_t d[8];
    {
      const uint32_t mask = b >= 32u ? 0xFFFFFFFFu : (1u << b) - 1u;
      // a tf stream without stored bits (or without freqs) reads as tf = strict ? 1 : 0 -> 1
      const uint32_t add = TFS ? ((tr.has_freq && !is_tail) ? strict : 1u) : strict;
#pragma unroll
      for (uint32_t kk = 0; kk < 2u; ++kk) {
        const uint32_t bitpos = (2u * l16 + kk) * b;
        const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
        const uint4 lo = *reinterpret_cast<const uint4 *>(P + row * 128u + 4u * w);
        const uint4 hi = *reinterpret_cast<const uint4 *>(P + row * 128u + 4u * w + 4u);
        d[4 * kk + 0] = (__funnelshift_r(lo.x, hi.x, sh) & mask) + add;
        d[4 * kk + 1] = (__funnelshift_r(lo.y, hi.y, sh) & mask) + add;
        d[4 * kk + 2] = (__funnelshift_r(lo.z, hi.z, sh) & mask) + add;
        d[4 * kk + 3] = (__funnelshift_r(lo.w, hi.w, sh) & mask) + add;
      }
    }
    if (__ballot(row_on && is_tail)) {  // the pre-decoded vint tail of the list
      if (row_on && is_tail) {
#pragma unroll
        for (uint32_t e = 0; e < 8u; ++e) {
          const uint32_t i = 8u * l16 + e;
          if (TFS)
            d[e] = i < tr.n_tail ? (tr.has_freq ? tr.tail_tfs[i] : 1u) : 0u;
          else
            d[e] = i < tr.n_tail ? tr.tail_docs[i] : TQD_TERMINATED;
        }
      }
    }
    if (TFS || !is_tail) {  // (uniform per 16-lane row)
      // prefix sum: local, then 4 DPP row shifts inside the 16-lane row
      uint32_t loc[8];
      loc[0] = d[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) loc[e] = loc[e - 1] + d[e];
      uint32_t incl = loc[7];
      incl += dpp_get<0x111, 0xF>(incl);
      incl += dpp_get<0x112, 0xF>(incl);
      incl += dpp_get<0x114, 0xF>(incl);
      incl += dpp_get<0x118, 0xF>(incl);
      // docs: compression/mod.rs:36-39,112-121: offset 0 <=> None <=> seed u32::MAX (wrapping)
      const uint32_t base =
          (TFS ? 0u : ((strict && prev == 0u) ? 0xFFFFFFFFu : prev)) + (incl - loc[7]);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = loc[e] + base;
    }
    wave_mem_fence();  // every lane has read its payload words: the values may overwrite them
    if (row_on) {
      *reinterpret_cast<uint4 *>(P + row * 128u + 8u * l16) = make_uint4(d[0], d[1], d[2], d[3]);
      *reinterpret_cast<uint4 *>(P + row * 128u + 8u * l16 + 4u) =
          make_uint4(d[4], d[5], d[6], d[7]);
    }
    wave_mem_fence();
    if (gid < 4u) {
      const uint32_t *blk = P + gid * 128u;
      if (TFS) {  // inclusive tf prefix sums: tf = I[at] - I[at-1]
        const uint32_t hi_v = blk[key];
        const uint32_t lo_v = key ? blk[key - 1u] : 0u;
        result = hi_v - lo_v;
        *excl = lo_v;
      } else {
        uint32_t pos = 0;
#pragma unroll
        for (uint32_t step = 64u; step > 0u; step >>= 1)
          if (blk[pos + step - 1u] < key) pos += step;
        result = blk[pos] == key ? pos : NOT_FOUND;
      }
    }
  }
  return result;
}
template <bool USE_DPP>
__device__ __forceinline__ uint32_t find_in_blocks(const uint8_t *idx, const TermRef &tr,
                                                   uint32_t jb, uint32_t doc, bool alive,
                                                   AndLdsT<false> &L, int lane) {
  uint32_t unused;
  return lookup_in_blocks<false>(idx, tr, jb, doc, alive, L.pay, lane, &unused);
}

template <int KPL, bool PRUNE, bool DENSE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(80))) void and_kernel(TqkScanParams p) {
  constexpr bool USE_DPP = true;
  __shared__ AndLdsT<DENSE> L;  // one wavefront per workgroup: finished chunks free their slot at once
  const int lane = (int)__lane_id();
  if (blockIdx.x >= p.n_chunks) return;
  const uint32_t chunk = sload(p.chunk_perm + blockIdx.x);
  const uint32_t t_begin = sload(p.chunk_starts + chunk);
  const uint32_t t_end = sload(p.chunk_starts + chunk + 1u);

  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;

  // ---- per-query state (wave-uniform)
  uint32_t q = uni(find_query(p.tile_starts, p.n_queries, t_begin));
  uint32_t q_tile_start = 0, q_tile_end = 0;
  const TqdQuery *Q = nullptr;
  uint32_t nt = 0, tile_blocks = TQD_AND_TILE;
  TermRef lead{}, t1{};
  float w_lead = 0.0f, w1 = 0.0f, rest_after1 = 0.0f, min_norm = 0.0f;
  bool prune = false;
  uint32_t *slots = nullptr;
  uint32_t thr = 0, thr_g = 0;
  uint32_t cache_loaded = 0xFFFFFFFFu;
  TopK<KPL> tk;
  uint32_t n_matches = 0, n_q = 0;  // docs scored: whole chunk / current query
  uint32_t q1n = 0, q2n = 0;  // queue fill

  auto setup_query = [&]() __attribute__((always_inline)) {
    q_tile_start = sload(p.tile_starts + q);
    q_tile_end = sload(p.tile_starts + q + 1u);
    Q = p.queries + q;
    nt = sload(&Q->n_terms);
    tile_blocks = sload(&Q->tile_blocks);
    lead = load_term(p.terms, sload(&Q->term[0]));
    t1 = load_term(p.terms, sload(&Q->term[1]));
    if (!p.use_dense) t1.dense = nullptr;
    w_lead = sload(&Q->weight[0]);
    w1 = sload(&Q->weight[1]);
    rest_after1 = 0.0f;
    for (uint32_t m = 2; m < nt; ++m) rest_after1 += sload(&Q->weight[m]);
    prune = PRUNE && (sload(&Q->flags) & TQD_QF_PRUNE) != 0u;
    const uint32_t thr_index = sload(&Q->thr_index);
    slots = (prune && thr_index != 0xFFFFFFFFu) ? p.thr_slots + (size_t)thr_index * TQD_THR_SLOTS
                                                : nullptr;
    const uint32_t ci = sload(&Q->cache_idx);
    if (ci != cache_loaded) {
      const float *cg = p.caches + (size_t)ci * 256u;
      wave_mem_fence();
      for (int i = lane; i < 256; i += WAVE) L.cache[i] = cg[i];
      wave_mem_fence();
      cache_loaded = ci;
    }
    // every doc's norm is >= the norm of the smallest fieldnorm id present (cache is monotone)
    min_norm = sload(p.caches + (size_t)ci * 256u +
                     (seg.fieldnorm ? seg.min_fieldnorm_id : seg.const_fieldnorm_id));
    thr = 0;
    thr_g = 0;
    tk.reset(sload(&Q->k));
  };

  // ---- stage C: verify in list 1, score, remaining lists, collect
  auto stageC = [&](uint32_t n) __attribute__((always_inline)) {
    const uint32_t base = q2n - n;
    q2n = base;
    if (p.debug & 128u) n_matches += n;  // COUNTERS
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, tf = 0, loc = 0;
    float norm = 0.0f;
    if (alive) {
      doc = L.q2_doc[base + lane];
      tf = L.q2_tf[base + lane];
      loc = L.q2_loc[base + lane];
      norm = L.cache[fieldnorm_id(seg, doc)];
    }
    float s = 0.0f;
    {
      uint32_t jb = loc, at = NOT_FOUND;
      uint2 mo = make_uint2(0u, 0u);
      if (DENSE || t1.dense) {
        jb = loc >> 7;
        at = loc & 127u;
      }
      if (alive) mo = rec_mo(t1.rec[jb]);
      if (prune && alive) {  // block_wand_intersection.rs:144-165
        // 96 % of the candidates end here: the test runs on reciprocal-based bounds, the exact
        // (IEEE-divided) leader score is only computed for the survivors
        float ub = bm25_bound(w_lead, norm, tf) + block_max_bound(mo.x, w1, L.cache, t1.has_freq);
        if (nt > 2u) ub = ub + rest_after1;
        alive = sortable(ub * 1.000002f) >= thr;
      }
      s = bm25(w_lead, norm, tf);
      if constexpr (!DENSE) {
        if (!t1.dense) {
          at = find_in_blocks<USE_DPP>(idx, t1, jb, doc, alive, L, lane);
          alive = alive && at != NOT_FOUND;
        }
      }
      // leader first, then ascending doc freq (block_wand_intersection.rs:144-165)
      if (alive) s = s + bm25(w1, norm, block_tf_at(idx, t1, mo, at));
    }
    float rest = rest_after1;
    for (uint32_t m = 2; m < nt; ++m) {
      TermRef tr = load_term(p.terms, sload(&Q->term[m]));
      if (!p.use_dense) tr.dense = nullptr;
      const float w = sload(&Q->weight[m]);
      rest -= w;
      if (rest < 0.0f) rest = 0.0f;
      uint32_t jb = 0, at = NOT_FOUND;
      if (DENSE || tr.dense) {
        if (alive) {
          const uint2 wd = tr.dense[doc >> 5];
          const uint32_t bit = doc & 31u;
          alive = (wd.x >> bit) & 1u;
          const uint32_t pi = wd.y + (uint32_t)__popc(wd.x & ((1u << bit) - 1u));
          jb = pi >> 7;
          at = pi & 127u;
        }
      } else if (alive) {
        jb = seek_block(tr, doc);
        alive = jb < tr.n_blocks;
      }
      uint2 mo = make_uint2(0u, 0u);
      if (alive) mo = rec_mo(tr.rec[jb]);
      if (prune && alive) {
        const float ub = (s + block_max_score(mo.x, w, L.cache, tr.has_freq) + rest) * 1.000001f;
        alive = sortable(ub) >= thr;
      }
      if constexpr (!DENSE) {
        if (!tr.dense) {
          at = find_in_blocks<USE_DPP>(idx, tr, jb, doc, alive, L, lane);
          alive = alive && at != NOT_FOUND;
        }
      }
      if (alive) s = s + bm25(w, norm, block_tf_at(idx, tr, mo, at));
    }
    if (alive) alive = doc_is_alive(seg, doc);
    const uint64_t hit = __ballot(alive);
    if (hit) {
      if (!(p.debug & 224u)) n_matches += (uint32_t)__popcll(hit);  // COUNTERS
      n_q += (uint32_t)__popcll(hit);
      const uint64_t key = alive ? make_key(s, doc) : 0ull;
      if (slots) {
        const uint32_t sb = (uint32_t)(key >> 32);
        if (alive && sb > thr_g) atomicMax(slots + ((doc * 0x9E3779B1u) >> 26), sb);
      }
      tk.offer(alive, key, lane);
      if (prune) {
        const uint32_t own = (uint32_t)(tk.thr >> 32);
        if (own > thr) thr = own;
      }
    }
  };

  // ---- stage B: locate in list 1.  Dense list: the bitmap answers membership, which is the
  // strongest filter there is, so nothing else is looked at first.  Other lists: pruned mode
  // scores the leader exactly (one fieldnorm gather) before paying for the seek.
  auto stageB = [&](uint32_t n) __attribute__((always_inline)) {
    const uint32_t base = q1n - n;
    q1n = base;
    if (p.debug & 64u) n_matches += n;  // COUNTERS
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, tf = 0, loc = 0;
    if (alive) {
      doc = L.q1_doc[base + lane];
      tf = L.q1_tf[base + lane];
    }
    if (DENSE || t1.dense) {
      if (alive) {
        const uint2 wd = t1.dense[doc >> 5];
        const uint32_t bit = doc & 31u;
        alive = (wd.x >> bit) & 1u;
        loc = wd.y + (uint32_t)__popc(wd.x & ((1u << bit) - 1u));
      }
    } else {
      if (prune && alive) {  // the other lists can add at most their weights
        const float s = bm25(w_lead, L.cache[fieldnorm_id(seg, doc)], tf);
        alive = sortable((s + (w1 + rest_after1)) * 1.000001f) >= thr;
      }
      if (alive) {
        loc = seek_block(t1, doc);
        alive = loc < t1.n_blocks;
      }
    }
    const uint64_t m = __ballot(alive);
    if (m) {
      const uint32_t pos = q2n + mbcnt64(m);
      wave_mem_fence();
      if (alive) {
        L.q2_doc[pos] = doc;
        L.q2_tf[pos] = tf;
        L.q2_loc[pos] = loc;
      }
      wave_mem_fence();
      q2n += (uint32_t)__popcll(m);
    }
  };

  auto drain = [&]() __attribute__((always_inline)) {
    while (q1n) {
      stageB(q1n < 64u ? q1n : 64u);
      while (q2n >= 64u) stageC(64u);
    }
    while (q2n) stageC(q2n < 64u ? q2n : 64u);
  };

  setup_query();
  for (uint32_t t = t_begin; t < t_end; ++t) {
    while (t >= q_tile_end) {  // next query (queries with zero tiles are skipped)
      if (q_tile_end > q_tile_start && q_tile_end > t_begin) {  // this chunk touched query q
        drain();
        const uint32_t part = sload(&Q->part_start) + (chunk - sload(&Q->chunk_first));
        flush_partial<KPL>(tk, sload(&p.sinks->partials), part, lane);
        if (lane == 0 && n_q) atomicAdd(sload(&p.sinks->query_matches) + sload(sload(&p.sinks->out_index) + q), n_q);
        n_q = 0;
      }
      ++q;
      setup_query();
    }

    // threshold (sortable score bits): own k-th key and the k-th largest shared slot
    if (slots) {
      const uint32_t sv =
          __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      thr_g = kth_largest64(sv, tk.k);
      if (thr_g > thr) thr = thr_g;
    }

    // ---- pre-filter: lane <-> leader block
    const uint32_t i_base = (t - q_tile_start) * tile_blocks;
    const uint32_t i_mine = i_base + (uint32_t)lane;
    bool surv = (uint32_t)lane < tile_blocks && i_mine < lead.n_blocks;
    uint2 mo_mine = make_uint2(0u, 0u);
    uint32_t prev_mine = 0;
    float rest_mine = 0.0f;  // bound of the other terms inside this leader block's doc range
    uint32_t tfmin_mine = 1u;  // pruned mode: smallest tf that can still reach the threshold
    {
      uint32_t first = 0, last = 0;
      float ub = 0.0f;
      if (surv) {
        const uint4 r = lead.rec[i_mine];
        last = r.x;
        mo_mine = make_uint2(r.y, r.z);
      }
      prev_mine = __shfl_up(last, 1, WAVE);
      if (lane == 0) prev_mine = block_prev_last(lead, i_base);
      first = i_mine ? prev_mine + 1u : 0u;
      if (prune && surv) {
        ub = block_max_score(mo_mine.x, w_lead, L.cache, lead.has_freq);
        // cheapest test first: not even with the other lists at their full weights?  (a rare
        // leader next to a stop word: most blocks end here, before any seek)
        surv = sortable((ub + (w1 + rest_after1)) * 1.000001f) >= thr;
      }
      for (uint32_t m = 1; m < nt; ++m) {
        const TermRef tr = m == 1u ? t1 : load_term(p.terms, sload(&Q->term[m]));
        const float w = m == 1u ? w1 : sload(&Q->weight[m]);
        if (surv) {
          const uint32_t j0 = seek_block(tr, first);
          if (j0 >= tr.n_blocks) {
            surv = false;  // the list ends before this leader block starts
          } else if (prune) {
            // block-max of list m over this leader block's doc range, if it spans <= 4 blocks
            // (records j0..j0+3 share a cache line with the seek's last probe)
            float bound = 0.0f;
            bool closed = false;
            for (uint32_t k = 0; k < 4u && !closed; ++k) {
              const uint32_t j = j0 + k;
              const uint4 r = tr.rec[j];
              const float b2 = block_max_score(r.y, w, L.cache, tr.has_freq);
              bound = b2 > bound ? b2 : bound;
              closed = r.x >= last || j + 1u >= tr.n_blocks;
            }
            if (!closed) bound = w;
            rest_mine = rest_mine + bound;
          }
        }
      }
      if (prune && nt > 2u) rest_mine *= 1.000001f;  // summed in another order than the scores
      if (prune && surv) surv = sortable(ub + rest_mine) >= thr;
      if (prune && surv) {
        // smallest tf whose tf-only score bound (norm replaced by its lower bound) can reach the
        // threshold inside this block: stage A then compares integers instead of scoring 128 docs
        auto pass = [&](uint32_t tfv) __attribute__((always_inline)) {
          return sortable(bm25(w_lead, min_norm, tfv) + rest_mine) >= thr;
        };
        if (!pass(0xFFFFFFFFu)) {
          surv = false;
        } else {
          uint32_t u = thr ^ ((thr >> 31) ? 0x80000000u : 0xFFFFFFFFu);  // sortable^-1
          const float x = __uint_as_float(u) - rest_mine;
          float est = 1.0f;
          if (x > 0.0f) est = x < w_lead ? x * min_norm / (w_lead - x) : 4.0e9f;
          uint32_t tfm = est >= 4.0e9f ? 0xFFFFFFF0u : (uint32_t)est;
          if (tfm < 1u) tfm = 1u;
          for (int it = 0; it < 4 && tfm > 1u && pass(tfm - 1u); ++it) --tfm;
          if (tfm > 1u && pass(tfm - 1u)) tfm = 1u;  // estimate way off: keep everything
          for (int it = 0; it < 4 && !pass(tfm); ++it) ++tfm;
          tfmin_mine = tfm;
        }
      }
    }
    uint64_t todo = __ballot(surv);
    if (p.debug & 32u) n_matches += (uint32_t)__popcll(todo);  // COUNTERS

    // ---- stage A per surviving leader block
    auto stageA = [&](uint32_t b) __attribute__((always_inline)) {
      const uint2 mo_l = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)mo_mine.x, (int)b),
                                    (uint32_t)__builtin_amdgcn_readlane((int)mo_mine.y, (int)b));
      const uint32_t prev_l = (uint32_t)__builtin_amdgcn_readlane((int)prev_mine, (int)b);
      uint32_t c0, c1, t0, t1f;
      bool alive0, alive1;
      if (mo_l.x == META_TAIL) {
        decode_docs<USE_DPP>(idx, lead, mo_l, prev_l, lane, c0, c1);
        decode_tfs(idx, lead, mo_l, lane, t0, t1f);
        alive0 = c0 != TQD_TERMINATED;
        alive1 = c1 != TQD_TERMINATED;
        if (prune) {
          const uint32_t tfmin = (uint32_t)__builtin_amdgcn_readlane((int)tfmin_mine, (int)b);
          alive0 = alive0 && t0 >= tfmin;
          alive1 = alive1 && t1f >= tfmin;
        }
      } else {
        const uint32_t doc_bits = mo_l.x & 31u;
        const uint32_t strict = (mo_l.x >> 6) & 1u;
        const uint32_t tf_bits = lead.has_freq ? (mo_l.x >> 8) & 0xFFu : 0u;
        wave_mem_fence();
        stage_payload(L.pay, idx + lead.payload_base + mo_l.y, 16u * (doc_bits + tf_bits), lane);
        wave_mem_fence();
        if (lead.has_freq) {
          unpack2_lds(L.pay + 4u * doc_bits, tf_bits, lane, t0, t1f);
          t0 += strict;  // minus-one encoding is tied to the strict flag
          t1f += strict;
        } else {
          t0 = 1u;
          t1f = 1u;
        }
        alive0 = true;  // full blocks have no padding
        alive1 = true;
        if (prune) {
          // tf-only bound first (block_wand_intersection.rs:112-125 filters on the exact leader
          // score; this is the same test with the norm replaced by its lower bound, folded into
          // an integer compare by the pre-filter).  Most blocks end here without a prefix sum.
          const uint32_t tfmin = (uint32_t)__builtin_amdgcn_readlane((int)tfmin_mine, (int)b);
          alive0 = t0 >= tfmin;
          alive1 = t1f >= tfmin;
          if (!(__ballot(alive0) | __ballot(alive1))) return;
        }
        uint32_t x0, x1;
        unpack2_lds(L.pay, doc_bits, lane, x0, x1);
        finish_docs<USE_DPP>(x0, x1, strict, prev_l, lane, c0, c1);
      }
      const uint64_t m0 = __ballot(alive0), m1 = __ballot(alive1);
      if (!(m0 | m1)) return;
      const uint32_t n0 = (uint32_t)__popcll(m0);
      const uint32_t pos0 = q1n + mbcnt64(m0);
      const uint32_t pos1 = q1n + n0 + mbcnt64(m1);
      wave_mem_fence();
      if (alive0) {
        L.q1_doc[pos0] = c0;
        L.q1_tf[pos0] = t0;
      }
      if (alive1) {
        L.q1_doc[pos1] = c1;
        L.q1_tf[pos1] = t1f;
      }
      wave_mem_fence();
      q1n += n0 + (uint32_t)__popcll(m1);
      while (q1n >= 64u) {
        stageB(64u);
        while (q2n >= 64u) stageC(64u);
      }
    };
    while (todo) {
      const uint32_t b = (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      stageA(b);
    }
  }
  // final flush
  if (q_tile_end > q_tile_start) {
    drain();
    const uint32_t part = sload(&Q->part_start) + (chunk - sload(&Q->chunk_first));
    flush_partial<KPL>(tk, sload(&p.sinks->partials), part, lane);
        if (lane == 0 && n_q) atomicAdd(sload(&p.sinks->query_matches) + sload(sload(&p.sinks->out_index) + q), n_q);
        n_q = 0;
  }
  if (lane == 0 && n_matches) atomicAdd(sload(&p.sinks->match_counter), (unsigned long long)n_matches);
}

// =================================================================== OR kernel
// Union with MaxScore pruning, the window-parallel form of block_wand
// (src/query/boolean_query/block_wand_union.rs:16-265) and BufferedUnionScorer
// (src/query/union/buffered_union.rs:63-158).  Terms are ordered by weight DESCENDING (the weight
// bounds a term's score: tf/(tf+norm) < 1).  One workgroup = one chunk of consecutive 4096-doc
// windows (HORIZON, buffered_union.rs:11-12).  Per window:
//   * the suffix of terms that are dense (bitmap) and whose weights together stay below the
//     threshold is NON-ESSENTIAL: a doc found only in those lists cannot reach the top-k, so
//     their postings are never enumerated (find_pivot_doc's prefix rule, :16-43, with the
//     reference's per-doc pivot replaced by a per-window one);
//   * the other (essential) lists are decoded block by block, scored and summed into f32
//     accumulators in LDS, the 4 waves splitting each list's blocks;
//   * every present doc whose partial score plus the non-essential weights can reach the
//     threshold is compacted into a per-wave queue; batches of 64 probe the non-essential lists'
//     bitmaps in order (one 8-byte load = membership + posting index -> tf), stopping as soon as
//     the remaining weights cannot lift the score over the threshold (:49-80);
//   * survivors are offered to the wave's register top-k and published to the query's threshold
//     slots (same scheme as the AND kernel; 64 slots for k <= 64, 128 for k <= 128).
// Scores are summed in term order in both modes, so pruned and exhaustive runs are bit-identical;
// against the reference the sum order of 3+ terms is not canonical (1e-5 relative).
template <bool PRUNE>
struct OrLds {
  float acc[TQD_OR_WINDOW];
  uint32_t present[TQD_OR_WINDOW / 32];
  float cache[256];
  float suffix[TQD_MAX_TERMS + 1];  // suffix[m] = sum of the weights of terms m..
  uint32_t thr_shared;
  // first block of every list for 64 consecutive windows (+1): planned lane-parallel once per
  // 64 windows, so the per-window loops carry no dependent seek
  uint32_t wj[TQD_MAX_TERMS][65];
  uint32_t cq_doc[PRUNE ? TQD_WAVES_PER_WG : 1][127];  // per-wave candidate queue
  uint32_t cq_s[PRUNE ? TQD_WAVES_PER_WG : 1][127];
};

template <int KPL, bool PRUNE>
__global__ __launch_bounds__(TQD_WAVES_PER_WG * 64) void or_kernel(TqkScanParams p) {
  constexpr bool USE_DPP = true;
  __shared__ OrLds<PRUNE> L;
  const int lane = (int)__lane_id();
  const uint32_t wave = uni(threadIdx.x >> 6);
  const uint32_t tid = threadIdx.x;
  if (blockIdx.x >= p.n_chunks) return;
  const uint32_t chunk = blockIdx.x;
  const uint32_t t_begin = sload(p.chunk_starts + chunk);
  const uint32_t t_end = sload(p.chunk_starts + chunk + 1u);

  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;
  uint32_t q = uni(find_query(p.tile_starts, p.n_queries, t_begin));
  uint32_t q_tile_start = 0, q_tile_end = 0;
  const TqdQuery *Q = nullptr;
  uint32_t nt = 0, dense_mask = 0, n_slot_rows = 0;
  bool prune = false, query_done = false;
  uint32_t *slots = nullptr;
  uint32_t thr = 0, thr_g = 0;
  uint32_t cache_loaded = 0xFFFFFFFFu;
  TopK<KPL> tk;
  uint32_t n_matches = 0, n_q = 0;  // docs scored: whole chunk / current query
  uint32_t cqn = 0;  // this wave's candidate queue fill
  uint32_t plan_begin = 0xFFFFFFFFu, plan_end = 0;  // windows [plan_begin, plan_end) are planned

  auto setup_query = [&]() __attribute__((always_inline)) {
    q_tile_start = sload(p.tile_starts + q);
    q_tile_end = sload(p.tile_starts + q + 1u);
    Q = p.queries + q;
    nt = sload(&Q->n_terms);
    prune = PRUNE && (sload(&Q->flags) & TQD_QF_PRUNE) != 0u;
    const uint32_t thr_index = sload(&Q->thr_index);
    const uint32_t k = sload(&Q->k);
    n_slot_rows = k <= 64u ? 1u : 2u;
    slots = (prune && thr_index != 0xFFFFFFFFu) ? p.thr_slots + (size_t)thr_index * TQD_THR_SLOTS
                                                : nullptr;
    const uint32_t ci = sload(&Q->cache_idx);
    __syncthreads();  // nobody still reads the previous query's cache / suffix sums
    if (ci != cache_loaded) {
      const float *cg = p.caches + (size_t)ci * 256u;
      for (uint32_t i = tid; i < 256u; i += TQD_WAVES_PER_WG * 64) L.cache[i] = cg[i];
      cache_loaded = ci;
    }
    dense_mask = 0;
    float suf = 0.0f;
    if (tid == 0) {
      L.suffix[nt] = 0.0f;
      L.thr_shared = 0u;
    }
    for (uint32_t m = nt; m-- > 0u;) {
      suf += sload(&Q->weight[m]);
      if (tid == 0) L.suffix[m] = suf;
      const TermRef tr = load_term(p.terms, sload(&Q->term[m]));
      if (tr.dense && p.use_dense) dense_mask |= 1u << m;
    }
    __syncthreads();
    thr = 0;
    thr_g = 0;
    query_done = false;
    plan_begin = 0xFFFFFFFFu;
    plan_end = 0;
    tk.reset(k);
  };

  // probe the non-essential lists [E, nt) for a batch of <= 64 candidates (one per lane)
  auto probe_batch = [&](uint32_t n, uint32_t E) __attribute__((always_inline)) {
    const uint32_t base = cqn - n;
    cqn = base;
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0;
    float s = 0.0f, norm = 0.0f;
    if (alive) {
      doc = L.cq_doc[PRUNE ? wave : 0][base + lane];
      s = __uint_as_float(L.cq_s[PRUNE ? wave : 0][base + lane]);
      norm = L.cache[fieldnorm_id(seg, doc)];
    }
    for (uint32_t m = E; m < nt; ++m) {
      if (alive) alive = sortable((s + L.suffix[m]) * 1.000001f) >= thr;
      if (!__ballot(alive)) break;
      const TermRef tr = load_term(p.terms, sload(&Q->term[m]));
      const float w = sload(&Q->weight[m]);
      if (alive) {
        const uint2 wd = tr.dense[doc >> 5];
        const uint32_t bit = doc & 31u;
        if ((wd.x >> bit) & 1u) {
          const uint32_t pi = wd.y + (uint32_t)__popc(wd.x & ((1u << bit) - 1u));
          const uint4 r = tr.rec[pi >> 7];
          s = s + bm25(w, norm, block_tf_at(idx, tr, make_uint2(r.y, r.z), pi & 127u));
        }
      }
    }
    if (alive) alive = doc_is_alive(seg, doc);
    const uint64_t hit = __ballot(alive);
    if (hit) {
      n_matches += (uint32_t)__popcll(hit);
      n_q += (uint32_t)__popcll(hit);
      const uint64_t key = alive ? make_key(s, doc) : 0ull;
      if (slots) {
        const uint32_t sb = (uint32_t)(key >> 32);
        const uint32_t h = (doc * 0x9E3779B1u) >> (n_slot_rows == 1u ? 26 : 25);
        if (alive && sb > thr_g) atomicMax(slots + h, sb);
      }
      tk.offer(alive, key, lane);
      const uint32_t own = (uint32_t)(tk.thr >> 32);
      if (own > thr) thr = own;
    }
  };

  setup_query();
  for (uint32_t t = t_begin; t < t_end; ++t) {
    while (t >= q_tile_end) {
      if (q_tile_end > q_tile_start && q_tile_end > t_begin) {
        const uint32_t part = sload(&Q->part_start) +
                              (chunk - sload(&Q->chunk_first)) * TQD_WAVES_PER_WG + wave;
        flush_partial<KPL>(tk, sload(&p.sinks->partials), part, lane);
        if (lane == 0 && n_q) atomicAdd(sload(&p.sinks->query_matches) + sload(sload(&p.sinks->out_index) + q), n_q);
        n_q = 0;
      }
      ++q;
      setup_query();
    }
    if (query_done) continue;  // the weights of all lists together are below the threshold
    const uint32_t base = (t - q_tile_start) * TQD_OR_WINDOW;
    const uint32_t win_hi = base + (TQD_OR_WINDOW - 1u);
    if (t >= plan_end || t < plan_begin) {  // plan the next 64 windows: one lane per window
      __syncthreads();
      plan_begin = t;
      plan_end = t + 64u;
      for (uint32_t m = wave; m < nt; m += TQD_WAVES_PER_WG) {
        const TermRef tr = load_term(p.terms, sload(&Q->term[m]));
        const uint64_t b0 = (uint64_t)base + (uint64_t)lane * TQD_OR_WINDOW;
        L.wj[m][lane] = b0 < seg.max_doc ? seek_block(tr, (uint32_t)b0) : tr.n_blocks;
        if (lane == 0) {
          const uint64_t b64 = (uint64_t)base + 64ull * TQD_OR_WINDOW;
          L.wj[m][64] = b64 < seg.max_doc ? seek_block(tr, (uint32_t)b64) : tr.n_blocks;
        }
      }
      __syncthreads();
    }
    const uint32_t wl = t - plan_begin;

    // threshold: wave 0 reads the shared slots, everybody takes max(shared, own k-th key)
    if (slots && wave == 0u) {
      uint32_t sv[2] = {0u, 0u};
      sv[0] = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t g;
      if (n_slot_rows == 2u) {
        sv[1] = __hip_atomic_load(slots + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g = kth_largest_multi<2>(sv, tk.k);
      } else {
        g = kth_largest64(sv[0], tk.k);
      }
      if (lane == 0) L.thr_shared = g;
    }
    for (uint32_t i = tid; i < TQD_OR_WINDOW / 4; i += TQD_WAVES_PER_WG * 64)
      reinterpret_cast<float4 *>(L.acc)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (tid < TQD_OR_WINDOW / 32) L.present[tid] = 0u;
    __syncthreads();
    if (slots) {
      thr_g = L.thr_shared;  // (stays from the last refresh in between)
      if (thr_g > thr) thr = thr_g;
    }
    // all 4 waves must agree on E: use the shared threshold only (own thresholds differ)
    uint32_t E = nt;
    if (prune) {
      const uint32_t thr_w = slots ? thr_g : 0u;
      for (uint32_t m = nt; m-- > 0u;) {
        if (((dense_mask >> m) & 1u) && sortable(L.suffix[m] * 1.000001f) < thr_w)
          E = m;
        else
          break;
      }
      if (E == 0u) {  // no doc of this query can reach the top-k any more
        query_done = true;
        __syncthreads();
        continue;
      }
    }

    // ---- essential lists: decode, score, accumulate (term order = score sum order)
    for (uint32_t m = 0; m < E; ++m) {
      const TermRef tr = load_term(p.terms, sload(&Q->term[m]));
      const float w = sload(&Q->weight[m]);
      // blocks [first block reaching into this window, first block reaching into the next one]
      const uint32_t jb = uni(L.wj[m][wl]);
      uint32_t je = uni(L.wj[m][wl + 1u]);
      if (je >= tr.n_blocks) je = tr.n_blocks ? tr.n_blocks - 1u : 0u;
      for (uint32_t j = jb + wave; j <= je && j < tr.n_blocks; j += TQD_WAVES_PER_WG) {
        const Dec d = decode_block<USE_DPP, false>(idx, tr, j, lane);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint32_t doc = e ? d.d1 : d.d0;
          const uint32_t tf = e ? d.t1 : d.t0;
          if (doc >= base && doc <= win_hi) {
            const uint32_t o = doc - base;
            const float s = bm25(w, L.cache[fieldnorm_id(seg, doc)], tf);
            L.acc[o] = L.acc[o] + s;  // one posting per (term, doc): no intra-phase conflict
            atomicOr(&L.present[o >> 5], 1u << (o & 31u));
          }
        }
      }
      __syncthreads();
    }
    // ---- harvest
    const float rest = L.suffix[E];
    for (uint32_t i = tid; i < TQD_OR_WINDOW; i += TQD_WAVES_PER_WG * 64) {
      bool has = (L.present[i >> 5] >> (i & 31u)) & 1u;
      if (has) has = doc_is_alive(seg, base + i);
      const float s = L.acc[i];
      if (E == nt) {  // every list was enumerated: final score
        const uint64_t key = has ? make_key(s, base + i) : 0ull;
        const uint64_t hit = __ballot(has);
        if (hit) {
          n_matches += (uint32_t)__popcll(hit);
          n_q += (uint32_t)__popcll(hit);
          if (slots) {
            const uint32_t sb = (uint32_t)(key >> 32);
            const uint32_t h = ((base + i) * 0x9E3779B1u) >> (n_slot_rows == 1u ? 26 : 25);
            if (has && sb > thr_g) atomicMax(slots + h, sb);
          }
          tk.offer(has, key, lane);
          if (prune) {
            const uint32_t own = (uint32_t)(tk.thr >> 32);
            if (own > thr) thr = own;
          }
        }
      } else if (PRUNE) {
        if (has) has = sortable((s + rest) * 1.000001f) >= thr;
        const uint64_t m = __ballot(has);
        if (m) {
          const uint32_t pos = cqn + mbcnt64(m);
          wave_mem_fence();
          if (has) {
            L.cq_doc[PRUNE ? wave : 0][pos] = base + i;
            L.cq_s[PRUNE ? wave : 0][pos] = __float_as_uint(s);
          }
          wave_mem_fence();
          cqn += (uint32_t)__popcll(m);
          if (cqn >= 64u) probe_batch(64u, E);
        }
      }
    }
    if (PRUNE && cqn) probe_batch(cqn, E);
    __syncthreads();
  }
  if (q_tile_end > q_tile_start) {
    const uint32_t part = sload(&Q->part_start) +
                          (chunk - sload(&Q->chunk_first)) * TQD_WAVES_PER_WG + wave;
    flush_partial<KPL>(tk, sload(&p.sinks->partials), part, lane);
        if (lane == 0 && n_q) atomicAdd(sload(&p.sinks->query_matches) + sload(sload(&p.sinks->out_index) + q), n_q);
        n_q = 0;
  }
  if (lane == 0 && n_matches) atomicAdd(sload(&p.sinks->match_counter), (unsigned long long)n_matches);
}

// =================================================================== union kernel (candidate-driven)
// The same union, driven by candidates instead of windows — the form that suits the sparse
// (high-weight) lists MaxScore keeps essential.  Terms by weight descending.  A tile is a run of
// blocks of ONE list i ("the leader of the tile"); every doc of the union is scored exactly once,
// by the tile of the FIRST list that holds it:
//   * tiles of a list that is non-essential by now (the weights of lists i.. together are below
//     the threshold) are skipped whole; so are leader blocks whose block-max plus the other
//     lists' weights cannot reach it (block_wand_union.rs:16-43,49-80);
//   * stage A decodes a leader block (as in the AND kernel, tf_min integer pre-filter included);
//   * stage B, 64 candidates per step: probe lists 0..i-1 — found there means the doc belongs to
//     that list's tile and the candidate is dropped — then lists i+1.. in order, adding their
//     BM25 terms, with the bound "score so far + remaining weights" checked before every probe.
//     Dense lists are probed through their bitmap, the others by seek_block + find_in_blocks.
// The sum runs over the lists holding the doc in ascending list index, whichever mode and
// threshold history: results are bit-identical across modes and runs.  One wavefront per chunk.
struct UnionLds {  // per wavefront
  uint32_t pay[516];
  uint32_t q1_doc[191], q1_tf[191];
  float cache[256];
  float suffix[TQD_MAX_TERMS + 1];
};

template <int KPL, bool PRUNE, bool BOOL>
__device__ __forceinline__ void union_body(const TqkScanParams &p) {
  constexpr bool USE_DPP = true;
  __shared__ UnionLds L;
  const int lane = (int)__lane_id();
  if (blockIdx.x >= p.n_chunks) return;
  const uint32_t chunk = sload(p.chunk_perm + blockIdx.x);
  const uint32_t t_begin = sload(p.chunk_starts + chunk);
  const uint32_t t_end = sload(p.chunk_starts + chunk + 1u);
  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;

  uint32_t q = uni(find_query(p.tile_starts, p.n_queries, t_begin));
  uint32_t q_tile_start = 0, q_tile_end = 0;
  const TqdQuery *Q = nullptr;
  uint32_t nt = 0, tile_blocks = TQD_AND_TILE, n_slot_rows = 1;
  uint32_t roles = 0, clause_end = 0, n_lead = 0, n_opt_lead = 0, min_should = 0;
  bool prune = false;
  uint32_t *slots = nullptr;
  uint32_t thr = 0, thr_g = 0;
  uint32_t cache_loaded = 0xFFFFFFFFu;
  float min_norm = 0.0f;
  TopK<KPL> tk;
  uint32_t n_matches = 0, n_q = 0;  // docs scored: whole chunk / current query
  uint32_t q1n = 0;
  // leader of the current tile
  uint32_t li = 0, li_end = 0;
  bool dead = false;
  TermRef lead{};
  float w_lead = 0.0f;

  auto setup_query = [&]() __attribute__((always_inline)) {
    q_tile_start = sload(p.tile_starts + q);
    q_tile_end = sload(p.tile_starts + q + 1u);
    Q = p.queries + q;
    nt = sload(&Q->n_terms);
    tile_blocks = sload(&Q->tile_blocks);
    prune = PRUNE && (sload(&Q->flags) & TQD_QF_PRUNE) != 0u;
    const uint32_t thr_index = sload(&Q->thr_index);
    const uint32_t k = sload(&Q->k);
    // 64 slots hold the top 16 well; beyond that the k-th largest slot is loose (the top k docs
    // collide): 256 slots (128: still loose at k = 100; 512: no better)
    n_slot_rows = k <= 16u ? 1u : 4u;
    slots = (prune && thr_index != 0xFFFFFFFFu) ? p.thr_slots + (size_t)thr_index * TQD_THR_SLOTS
                                                : nullptr;
    const uint32_t ci = sload(&Q->cache_idx);
    wave_mem_fence();
    if (ci != cache_loaded) {
      const float *cg = p.caches + (size_t)ci * 256u;
      for (int i = lane; i < 256; i += WAVE) L.cache[i] = cg[i];
      cache_loaded = ci;
    }
    if (BOOL) {
      roles = sload(&Q->roles);
      clause_end = sload(&Q->clause_end);
      n_lead = sload(&Q->n_lead);
      n_opt_lead = sload(&Q->n_opt_lead);
      min_should = sload(&Q->min_should);
    } else {  // pure union: one leading clause of Should terms
      n_lead = nt;
    }
    // suffix[m]: what the lists m.. can add at most (MustNot lists carry weight 0)
    float suf = 0.0f;
    if (lane == 0) L.suffix[nt] = 0.0f;
    for (uint32_t m = nt; m-- > 0u;) {
      suf += sload(&Q->weight[m]);
      if (lane == 0) L.suffix[m] = suf;
    }
    wave_mem_fence();
    min_norm = sload(p.caches + (size_t)ci * 256u +
                     (seg.fieldnorm ? seg.min_fieldnorm_id : seg.const_fieldnorm_id));
    thr = 0;
    thr_g = 0;
    li = 0xFFFFFFFFu;
    li_end = 0;
    dead = false;
    tk.reset(k);
  };

  // ---- stage B: the other lists of <= 64 candidates of leader li
  auto stageB = [&](uint32_t n) __attribute__((always_inline)) {
    const uint32_t base = q1n - n;
    q1n = base;
    if (p.debug & 64u) n_matches += n;  // COUNTERS
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, tf = 0;
    float norm = 0.0f, s = 0.0f;
    if (alive) {
      doc = L.q1_doc[base + lane];
      tf = L.q1_tf[base + lane];
      norm = L.cache[fieldnorm_id(seg, doc)];
      s = bm25(w_lead, norm, tf);
      if (prune) alive = sortable((s + L.suffix[li + 1u]) * 1.000001f) >= thr;
    }
    // The leader set is a union of n_lead lists: a doc is scored by the tile of the first list of
    // the set that holds it.  Pure unions: all the Should terms.  With Must clauses: the cheapest
    // Must clause, preceded (n_opt_lead) by the optional Should lists in weight order — MaxScore
    // for RequiredOptionalScorer: a Should list drives the docs it holds (which must also be in the
    // Must clause), the Must clause drives the rest with a bound that no longer carries the
    // Should weights, and its tiles die once the threshold passes what the Must part alone can
    // score.  Then come the other Must clauses (each a union of terms; cheapest first, summed as
    // Intersection::score does: left + right + sum(others), intersection.rs:325-329), the MustNot
    // terms (Exclude, exclude.rs) and the remaining optional Should terms
    // (RequiredOptionalScorer::score = req + opt, reqopt_scorer.rs:85-98).
    float opt = 0.0f, oth = 0.0f, csum = 0.0f;
    bool cfound = false;
    bool lcfound = !BOOL || li >= n_opt_lead;  // the lead Must clause holds the doc
    if (BOOL && li < n_opt_lead) {             // an optional list leads: its score is optional
      opt = s;
      s = 0.0f;
    }
    uint32_t clause = 1u;
    uint32_t n_should = ((roles >> (2u * li)) & 3u) == TQD_ROLE_SHOULD ? 1u : 0u;
    // lists after the leader first (they add to the score and tighten the bound), the lists
    // before it last: those only decide whether another tile owns the doc, and most candidates
    // are gone by then without the (sparse, expensive) probes into the high-weight lists
    for (uint32_t mm = 1; mm < nt; ++mm) {
      uint32_t m = li + mm;
      if (m >= nt) m -= nt;
      const uint32_t role = BOOL ? (roles >> (2u * m)) & 3u : TQD_ROLE_SHOULD;
      const float w = sload(&Q->weight[m]);
      // what the lists m.. can still add (lists below li add nothing: found there = dropped)
      if (prune && m > li && alive)
        alive = sortable((((s + oth) + (csum + opt)) + L.suffix[m]) * 1.000001f) >= thr;
      if (!__ballot(alive)) break;
      TermRef tr = load_term(p.terms, sload(&Q->term[m]));
      if (!p.use_dense) tr.dense = nullptr;
      bool found = false;
      uint32_t jb = 0, at = NOT_FOUND;
      if (tr.dense) {
        if (alive) {
          const uint2 wd = tr.dense[doc >> 5];
          const uint32_t bit = doc & 31u;
          found = (wd.x >> bit) & 1u;
          const uint32_t pi = wd.y + (uint32_t)__popc(wd.x & ((1u << bit) - 1u));
          jb = pi >> 7;
          at = pi & 127u;
        }
      } else {
        bool cand = alive;
        if (cand) {
          jb = seek_block(tr, doc);
          cand = jb < tr.n_blocks;
        }
        uint32_t unused;
        at = lookup_in_blocks<false>(idx, tr, jb, doc, cand, L.pay, lane, &unused);
        found = cand && at != NOT_FOUND;
      }
      float sc = 0.0f;
      if (found && alive && role != TQD_ROLE_MUST_NOT && !(m < li && (!BOOL || m < n_lead))) {
        const uint4 r = tr.rec[jb];
        sc = bm25(w, norm, block_tf_at(idx, tr, make_uint2(r.y, r.z), at));
      }
      if (role == TQD_ROLE_MUST_NOT) {
        if (found) alive = false;
      } else if (!BOOL || m < n_lead) {
        if (found) {
          if (m < li) {
            alive = false;  // this doc is scored by list m's tile
          } else if (BOOL && m < n_opt_lead) {
            opt = opt + sc;
            ++n_should;
          } else {
            s = s + sc;
            lcfound = true;
            if (role == TQD_ROLE_SHOULD) ++n_should;
          }
        }
        if (BOOL && m + 1u == n_lead && !lcfound) alive = false;  // not in the lead Must clause
      } else if (role == TQD_ROLE_MUST) {
        cfound = cfound || found;
        if (found) csum = csum + sc;
        if ((clause_end >> m) & 1u) {
          if (!cfound) alive = false;
          if (clause == 1u)
            s = s + csum;
          else
            oth = oth + csum;
          ++clause;
          csum = 0.0f;
          cfound = false;
        }
      } else if (found) {
        opt = opt + sc;
        ++n_should;
      }
    }
    if (BOOL) {
      s = (s + oth) + opt;
      if (n_should < min_should) alive = false;
    }
    if (alive) alive = doc_is_alive(seg, doc);
    const uint64_t hit = __ballot(alive);
    if (hit) {
      if (!(p.debug & 224u)) n_matches += (uint32_t)__popcll(hit);  // COUNTERS
      n_q += (uint32_t)__popcll(hit);
      const uint64_t key = alive ? make_key(s, doc) : 0ull;
      if (slots) {
        const uint32_t sb = (uint32_t)(key >> 32);
        const uint32_t h = (doc * 0x9E3779B1u) >> (n_slot_rows == 1u ? 26 : 24);
        if (alive && sb > thr_g) atomicMax(slots + h, sb);
      }
      tk.offer(alive, key, lane);
      if (prune) {
        const uint32_t own = (uint32_t)(tk.thr >> 32);
        if (own > thr) thr = own;
      }
    }
  };

  setup_query();
  for (uint32_t t = t_begin; t < t_end; ++t) {
    while (t >= q_tile_end) {
      if (q_tile_end > q_tile_start && q_tile_end > t_begin) {
        while (q1n) stageB(q1n < 64u ? q1n : 64u);
        const uint32_t part = sload(&Q->part_start) + (chunk - sload(&Q->chunk_first));
        flush_partial<KPL>(tk, sload(&p.sinks->partials), part, lane);
        if (lane == 0 && n_q) atomicAdd(sload(&p.sinks->query_matches) + sload(sload(&p.sinks->out_index) + q), n_q);
        n_q = 0;
      }
      ++q;
      setup_query();
    }
    // ---- which list leads this tile (tiles of a chunk come in order: advance, never search)
    if (dead) {  // lists li.. are non-essential for good (the threshold only rises): nothing
      t = (q_tile_end < t_end ? q_tile_end : t_end) - 1u;  // left for this query in this chunk
      continue;
    }
    const uint32_t tl = t - q_tile_start;
    bool new_leader = li == 0xFFFFFFFFu;
    uint32_t nli = new_leader ? 0u : li;
    if (new_leader) li_end = sload(&Q->lead_tile_start[1]);
    while (tl >= li_end && nli + 1u < nt) {
      ++nli;
      li_end = sload(&Q->lead_tile_start[nli + 1u]);
      new_leader = true;
    }
    if (new_leader) {
      while (q1n) stageB(q1n < 64u ? q1n : 64u);  // the queue belongs to the previous leader
      li = nli;
      lead = load_term(p.terms, sload(&Q->term[li]));
      w_lead = sload(&Q->weight[li]);
    }
    // threshold: on a new leader and every 8th tile
    if (slots && (new_leader || (tl & 7u) == 0u)) {
      uint32_t sv[4] = {0u, 0u, 0u, 0u};
      sv[0] = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (n_slot_rows == 4u) {
#pragma unroll
        for (int r = 1; r < 4; ++r)
          sv[r] = __hip_atomic_load(slots + 64 * r + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        thr_g = kth_largest_multi<4>(sv, tk.k);
      } else {
        thr_g = kth_largest64(sv[0], tk.k);
      }
      if (thr_g > thr) thr = thr_g;
    }
    // non-essential by now: every doc first seen in list li scores at most the weights of li..
    if (prune && sortable(L.suffix[li] * 1.000001f) < thr) {
      while (q1n) stageB(q1n < 64u ? q1n : 64u);
      dead = true;
      continue;
    }

    // ---- pre-filter: lane <-> leader block
    const uint32_t i_base = (tl - sload(&Q->lead_tile_start[li])) * tile_blocks;
    const uint32_t i_mine = i_base + (uint32_t)lane;
    bool surv = (uint32_t)lane < tile_blocks && i_mine < lead.n_blocks;
    uint4 rec_mine = make_uint4(0u, 0u, 0u, 0u);
    uint32_t prev_mine = 0, tfmin_mine = 1u;
    {
      if (surv) rec_mine = lead.rec[i_mine];
      prev_mine = __shfl_up(rec_mine.x, 1, WAVE);
      if (lane == 0) prev_mine = block_prev_last(lead, i_base);
      if (prune) {
        // what the lists after li can add inside this leader block's doc range (the lists before
        // li hold none of this tile's docs): block-max over the <= 4 blocks the range spans,
        // else the weight; a single-term Must clause that ends before the range drops the block
        const uint32_t first = i_mine ? prev_mine + 1u : 0u;
        const uint32_t last = rec_mine.x;
        float ub = 0.0f, rest_mine = 0.0f;
        if (surv) {
          ub = block_max_score(rec_mine.y, w_lead, L.cache, lead.has_freq);
          surv = sortable((ub + L.suffix[li + 1u] * 1.000001f) * 1.000001f) >= thr;
        }
        // (pure unions: their dense lists span far more than 4 blocks, the seeks do not pay)
        const bool block_bounds = BOOL && n_lead < nt;
        if (!block_bounds) rest_mine = L.suffix[li + 1u];
        if (block_bounds && __ballot(surv)) {
          for (uint32_t m = li + 1u; m < nt; ++m) {
            const uint32_t role = (roles >> (2u * m)) & 3u;
            if (role == TQD_ROLE_MUST_NOT) continue;
            const bool single = role == TQD_ROLE_MUST && m >= n_lead && ((clause_end >> m) & 1u) &&
                                (m == n_lead || ((clause_end >> (m - 1u)) & 1u));
            const TermRef tr = load_term(p.terms, sload(&Q->term[m]));
            const float w = sload(&Q->weight[m]);
            if (surv) {
              const uint32_t j0 = seek_block(tr, first);
              if (j0 >= tr.n_blocks) {
                if (single) surv = false;
              } else {
                float bound = 0.0f;
                bool closed = false;
                for (uint32_t k = 0; k < 4u && !closed; ++k) {
                  const uint32_t j = j0 + k;
                  const uint4 r = tr.rec[j];
                  const float b2 = block_max_score(r.y, w, L.cache, tr.has_freq);
                  bound = b2 > bound ? b2 : bound;
                  closed = r.x >= last || j + 1u >= tr.n_blocks;
                }
                if (!closed) bound = w;
                rest_mine = rest_mine + bound;
              }
            }
          }
        }
        rest_mine *= 1.000001f;
        if (surv) surv = sortable((ub + rest_mine) * 1.000001f) >= thr;
        if (surv) {
          auto pass = [&](uint32_t tfv) __attribute__((always_inline)) {
            return sortable((bm25(w_lead, min_norm, tfv) + rest_mine) * 1.000001f) >= thr;
          };
          if (!pass(0xFFFFFFFFu)) {
            surv = false;
          } else {
            uint32_t tfm = 1u;
            while (tfm < 64u && !pass(tfm)) ++tfm;  // tfs are small; beyond 64 keep everything
            tfmin_mine = tfm < 64u ? tfm : 1u;
          }
        }
      }
    }
    uint64_t todo = __ballot(surv);
    if (p.debug & 32u) n_matches += (uint32_t)__popcll(todo);  // COUNTERS
    if (p.debug & 128u) n_matches += 1u;  // COUNTERS
    while (todo) {
      const uint32_t b = (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      const uint2 mo_l = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)rec_mine.y, (int)b),
                                    (uint32_t)__builtin_amdgcn_readlane((int)rec_mine.z, (int)b));
      const uint32_t prev_l = (uint32_t)__builtin_amdgcn_readlane((int)prev_mine, (int)b);
      uint32_t c0, c1, t0, t1f;
      decode_tfs(idx, lead, mo_l, lane, t0, t1f);  // tail padding reads as tf 0
      bool alive0 = true, alive1 = true;
      if (prune) {
        const uint32_t tfmin = (uint32_t)__builtin_amdgcn_readlane((int)tfmin_mine, (int)b);
        alive0 = t0 >= tfmin;
        alive1 = t1f >= tfmin;
        if (!(__ballot(alive0) | __ballot(alive1))) continue;
      }
      decode_docs<USE_DPP>(idx, lead, mo_l, prev_l, lane, c0, c1);
      alive0 = alive0 && c0 != TQD_TERMINATED;
      alive1 = alive1 && c1 != TQD_TERMINATED;
      const uint64_t m0 = __ballot(alive0), m1 = __ballot(alive1);
      if (!(m0 | m1)) continue;
      const uint32_t n0 = (uint32_t)__popcll(m0);
      const uint32_t pos0 = q1n + mbcnt64(m0);
      const uint32_t pos1 = q1n + n0 + mbcnt64(m1);
      wave_mem_fence();
      if (alive0) {
        L.q1_doc[pos0] = c0;
        L.q1_tf[pos0] = t0;
      }
      if (alive1) {
        L.q1_doc[pos1] = c1;
        L.q1_tf[pos1] = t1f;
      }
      wave_mem_fence();
      q1n += n0 + (uint32_t)__popcll(m1);
      while (q1n >= 64u) stageB(64u);
    }
  }
  if (q_tile_end > q_tile_start) {
    while (q1n) stageB(q1n < 64u ? q1n : 64u);
    const uint32_t part = sload(&Q->part_start) + (chunk - sload(&Q->chunk_first));
    flush_partial<KPL>(tk, sload(&p.sinks->partials), part, lane);
        if (lane == 0 && n_q) atomicAdd(sload(&p.sinks->query_matches) + sload(sload(&p.sinks->out_index) + q), n_q);
        n_q = 0;
  }
  if (lane == 0 && n_matches) atomicAdd(sload(&p.sinks->match_counter), (unsigned long long)n_matches);
}

// k <= 128 (KPL <= 2) instantiations are compiled for 6 waves/SIMD (<= 84 registers) instead of 4
template <int KPL, bool PRUNE, bool BOOL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(96))) void union_kernel(TqkScanParams p) {
  union_body<KPL, PRUNE, BOOL>(p);
}
template <int KPL, bool PRUNE, bool BOOL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(96), amdgpu_waves_per_eu(6, 8))) void
union_kernel_small(TqkScanParams p) {
  union_body<KPL, PRUNE, BOOL>(p);
}

// positions: raw deltas of a whole term (PositionReader::read over everything).  pos_blk[pb] =
// absolute byte offset of position block pb | bit width << 56 (positions/reader.rs:84-101).
__device__ __forceinline__ uint32_t position_delta(const uint8_t *pos, const TqdTerm *t,
                                                   uint64_t i) {
  const uint64_t pb = i >> 7;
  if (pb < t->n_pos_blocks) {
    const uint64_t e = t->pos_blk[pb];
    const uint32_t b = (uint32_t)(e >> 56);
    if (b == 0u) return 0u;
    const uint8_t *p = pos + (e & 0x00FFFFFFFFFFFFFFull);
    const uint32_t v = (uint32_t)(i & 127u);
    const uint32_t bitpos = (v >> 2) * b;
    const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
    const uint8_t *q = p + 16u * w + 4u * (v & 3u);
    const uint32_t lo = ld_u1(q);
    uint32_t hi = 0;
    if (sh + b > 32u) hi = ld_u1(q + 16);  // the value straddles two words of its stream
    const uint32_t mask = (b >= 32u) ? 0xFFFFFFFFu : ((1u << b) - 1u);
    return __funnelshift_r(lo, hi, sh) & mask;
  }
  return t->pos_tail[i - ((uint64_t)t->n_pos_blocks << 7)];
}

// =================================================================== phrase kernel
// Exact phrase (slop 0), PhraseScorer (src/query/phrase_query/phrase_scorer.rs:82-136,347-507).
// Built on the AND kernel's stages: terms by doc freq ascending, leader-block tiles, one
// wavefront per chunk, candidates flowing through LDS queues.  Besides doc and tf every candidate
// carries the index of its first position in the leader's position stream (block's first position
// from the block record + exclusive prefix sum of the block's tfs: segment_postings.rs:232-254).
//   A  decode a leader block: docs, tfs and the tf prefix sum;
//   B  locate the candidate in list 1 (bitmap, or O(1) seek_block);
//   C  for every other list: membership (bitmap / find_in_blocks), then tf and position index
//      through lookup_in_blocks<TFS> (the block's tf stream prefix-summed, 4 blocks per step);
//      finally one lane per candidate runs the n-way merge over adjusted positions
//      (position + max_offset - term_offset, :372-385; count = intersection_count, :437-461),
//      fetching single bitpacked deltas by index, and scores bm25(sum-idf weight, norm, count).
// The reference runs phrases through the default for_each_pruning_scorer (a threshold filter on
// finished scores), so there is nothing to prune before the positions are read.
#define TQD_PH_MAX_TERMS 8
template <int NT_MAX>
struct PhraseLds {  // per wavefront
  uint32_t pay[516];  // lookup_in_blocks' staging area
  uint32_t q1_doc[191], q1_tf[191], q1_pi[191];
  uint32_t q2_doc[127], q2_tf[127], q2_pi[127], q2_loc[127];
  uint32_t ph_pi[NT_MAX - 1][64], ph_tf[NT_MAX - 1][64];  // lists 1.. (the leader's stay in q2)
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

template <int KPL, int NT_MAX>
__global__ __launch_bounds__(64) void phrase_kernel(TqkScanParams p) {
  constexpr bool USE_DPP = true;
  __shared__ PhraseLds<NT_MAX> L;
  const int lane = (int)__lane_id();
  if (blockIdx.x >= p.n_chunks) return;
  const uint32_t chunk = sload(p.chunk_perm + blockIdx.x);
  const uint32_t t_begin = sload(p.chunk_starts + chunk);
  const uint32_t t_end = sload(p.chunk_starts + chunk + 1u);
  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;
  const uint8_t *pos = seg.pos;

  uint32_t q = uni(find_query(p.tile_starts, p.n_queries, t_begin));
  uint32_t q_tile_start = 0, q_tile_end = 0;
  const TqdQuery *Q = nullptr;
  uint32_t nt = 0, tile_blocks = TQD_AND_TILE;
  TermRef lead{}, t1{};
  float weight = 0.0f;
  const float *cache_g = nullptr;  // tf cache of the current query (global memory)
  TopK<KPL> tk;
  uint32_t n_matches = 0, n_q = 0;  // docs scored: whole chunk / current query
  uint32_t q1n = 0, q2n = 0;

  auto setup_query = [&]() __attribute__((always_inline)) {
    q_tile_start = sload(p.tile_starts + q);
    q_tile_end = sload(p.tile_starts + q + 1u);
    Q = p.queries + q;
    nt = sload(&Q->n_terms);
    tile_blocks = sload(&Q->tile_blocks);
    lead = load_term(p.terms, sload(&Q->term[0]));
    t1 = load_term(p.terms, sload(&Q->term[1]));
    if (!p.use_dense) t1.dense = nullptr;
    weight = sload(&Q->weight[0]);
    // the tf cache stays in global memory: only phrase matches are scored (7.9 KB of LDS per
    // wavefront instead of 9.5 KB: 5 waves/SIMD)
    cache_g = p.caches + (size_t)sload(&Q->cache_idx) * 256u;
    tk.reset(sload(&Q->k));
  };

  // ---- stage C: the other lists' postings of the candidate, then the positions
  auto stageC = [&](uint32_t n) __attribute__((always_inline)) {
    const uint32_t base = q2n - n;
    q2n = base;
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, loc = 0, lead_tf = 0, lead_pi = 0;
    if (alive) {
      doc = L.q2_doc[base + lane];
      loc = L.q2_loc[base + lane];
      lead_tf = L.q2_tf[base + lane];
      lead_pi = L.q2_pi[base + lane];
    }
    for (uint32_t m = 1; m < nt; ++m) {
      TermRef tr = m == 1u ? t1 : load_term(p.terms, sload(&Q->term[m]));
      if (!p.use_dense) tr.dense = nullptr;
      uint32_t jb = 0, at = NOT_FOUND;
      if (m == 1u) {
        if (tr.dense) {
          jb = loc >> 7;
          at = loc & 127u;
        } else {
          jb = loc;
        }
      } else if (tr.dense) {
        if (alive) {
          const uint2 wd = tr.dense[doc >> 5];
          const uint32_t bit = doc & 31u;
          alive = (wd.x >> bit) & 1u;
          const uint32_t pi = wd.y + (uint32_t)__popc(wd.x & ((1u << bit) - 1u));
          jb = pi >> 7;
          at = pi & 127u;
        }
      } else if (alive) {
        jb = seek_block(tr, doc);
        alive = jb < tr.n_blocks;
      }
      if (!tr.dense) {
        uint32_t unused;
        at = lookup_in_blocks<false>(idx, tr, jb, doc, alive, L.pay, lane, &unused);
        alive = alive && at != NOT_FOUND;
      }
      if (!__ballot(alive)) return;
      uint32_t excl = 0;
      uint32_t tf = 1;
      if (!(p.debug & 2u)) tf = lookup_in_blocks<true>(idx, tr, jb, at, alive, L.pay, lane, &excl);
      if (alive) {
        L.ph_tf[m - 1u][lane] = tf;
        L.ph_pi[m - 1u][lane] = tr.rec[jb].w + excl;
      }
    }
    // ---- position check, one lane per candidate
    bool has = false;
    uint64_t key = 0;
    if (alive && (p.debug & 3u)) {
      has = true;
      key = make_key(1.0f, doc);
    } else if (alive) {
      PosCursor cur[NT_MAX];
#pragma unroll
      for (int m = 0; m < NT_MAX; ++m) {
        cur[m].valid = false;
        cur[m].idx = cur[m].end = cur[m].cur = 0;
        if ((uint32_t)m < nt) {
          const uint32_t pi = m ? L.ph_pi[m ? m - 1 : 0][lane] : lead_pi;
          cur[m].idx = pi + 1u;
          cur[m].end = pi + (m ? L.ph_tf[m ? m - 1 : 0][lane] : lead_tf);
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
        for (int m = 1; m < NT_MAX; ++m) {
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
          for (int m = 1; m < NT_MAX; ++m)
            if ((uint32_t)m < nt) pos_advance(cur[m], pos, p.terms + Q->term[m]);
        }
        pos_advance(cur[0], pos, p.terms + Q->term[0]);
      }
      if (count > 0 && doc_is_alive(seg, doc)) {
        has = true;
        key = make_key(bm25(weight, cache_g[fieldnorm_id(seg, doc)], count), doc);
      }
    }
    const uint64_t hit = __ballot(has);
    if (hit) {
      n_matches += (uint32_t)__popcll(hit);
      n_q += (uint32_t)__popcll(hit);
      tk.offer(has, key, lane);
    }
  };

  // ---- stage B: locate the candidate in list 1
  auto stageB = [&](uint32_t n) __attribute__((always_inline)) {
    const uint32_t base = q1n - n;
    q1n = base;
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, tf = 0, pi0 = 0, loc = 0;
    if (alive) {
      doc = L.q1_doc[base + lane];
      tf = L.q1_tf[base + lane];
      pi0 = L.q1_pi[base + lane];
    }
    if (t1.dense) {
      if (alive) {
        const uint2 wd = t1.dense[doc >> 5];
        const uint32_t bit = doc & 31u;
        alive = (wd.x >> bit) & 1u;
        loc = wd.y + (uint32_t)__popc(wd.x & ((1u << bit) - 1u));
      }
    } else if (alive) {
      loc = seek_block(t1, doc);
      alive = loc < t1.n_blocks;
    }
    const uint64_t m = __ballot(alive);
    if (m) {
      const uint32_t at = q2n + mbcnt64(m);
      wave_mem_fence();
      if (alive) {
        L.q2_doc[at] = doc;
        L.q2_tf[at] = tf;
        L.q2_pi[at] = pi0;
        L.q2_loc[at] = loc;
      }
      wave_mem_fence();
      q2n += (uint32_t)__popcll(m);
    }
  };

  auto drain = [&]() __attribute__((always_inline)) {
    while (q1n) {
      stageB(q1n < 64u ? q1n : 64u);
      while (q2n >= 64u) stageC(64u);
    }
    while (q2n) stageC(q2n < 64u ? q2n : 64u);
  };

  setup_query();
  for (uint32_t t = t_begin; t < t_end; ++t) {
    while (t >= q_tile_end) {
      if (q_tile_end > q_tile_start && q_tile_end > t_begin) {
        drain();
        const uint32_t part = sload(&Q->part_start) + (chunk - sload(&Q->chunk_first));
        flush_partial<KPL>(tk, sload(&p.sinks->partials), part, lane);
        if (lane == 0 && n_q) atomicAdd(sload(&p.sinks->query_matches) + sload(sload(&p.sinks->out_index) + q), n_q);
        n_q = 0;
      }
      ++q;
      setup_query();
    }
    // ---- pre-filter: lane <-> leader block; drop blocks past the end of another list
    const uint32_t i_base = (t - q_tile_start) * tile_blocks;
    const uint32_t i_mine = i_base + (uint32_t)lane;
    bool surv = (uint32_t)lane < tile_blocks && i_mine < lead.n_blocks;
    uint4 rec_mine = make_uint4(0u, 0u, 0u, 0u);
    uint32_t prev_mine = 0;
    {
      if (surv) rec_mine = lead.rec[i_mine];
      prev_mine = __shfl_up(rec_mine.x, 1, WAVE);
      if (lane == 0) prev_mine = block_prev_last(lead, i_base);
      const uint32_t first = i_mine ? prev_mine + 1u : 0u;
      for (uint32_t m = 1; m < nt; ++m) {
        const TermRef tr = m == 1u ? t1 : load_term(p.terms, sload(&Q->term[m]));
        if (surv) surv = seek_block(tr, first) < tr.n_blocks;
      }
    }
    uint64_t todo = __ballot(surv);
    // ---- stage A per surviving leader block
    while (todo) {
      const uint32_t b = (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      const uint2 mo_l = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)rec_mine.y, (int)b),
                                    (uint32_t)__builtin_amdgcn_readlane((int)rec_mine.z, (int)b));
      const uint32_t bp = (uint32_t)__builtin_amdgcn_readlane((int)rec_mine.w, (int)b);
      const uint32_t prev_l = (uint32_t)__builtin_amdgcn_readlane((int)prev_mine, (int)b);
      uint32_t c0, c1, t0, t1f;
      decode_docs<USE_DPP>(idx, lead, mo_l, prev_l, lane, c0, c1);
      decode_tfs(idx, lead, mo_l, lane, t0, t1f);  // tail padding reads as tf 0
      const uint32_t ssum = t0 + t1f;
      const uint32_t incl = wave_inclusive_scan<USE_DPP>(ssum, lane);
      const uint32_t e0 = bp + (incl - ssum), e1 = e0 + t0;
      const bool alive0 = c0 != TQD_TERMINATED, alive1 = c1 != TQD_TERMINATED;
      const uint64_t m0 = __ballot(alive0), m1 = __ballot(alive1);
      if (!(m0 | m1)) continue;
      const uint32_t n0 = (uint32_t)__popcll(m0);
      const uint32_t pos0 = q1n + mbcnt64(m0);
      const uint32_t pos1 = q1n + n0 + mbcnt64(m1);
      wave_mem_fence();
      if (alive0) {
        L.q1_doc[pos0] = c0;
        L.q1_tf[pos0] = t0;
        L.q1_pi[pos0] = e0;
      }
      if (alive1) {
        L.q1_doc[pos1] = c1;
        L.q1_tf[pos1] = t1f;
        L.q1_pi[pos1] = e1;
      }
      wave_mem_fence();
      q1n += n0 + (uint32_t)__popcll(m1);
      while (q1n >= 64u) {
        stageB(64u);
        while (q2n >= 64u) stageC(64u);
      }
    }
  }
  if (q_tile_end > q_tile_start) {
    drain();
    const uint32_t part = sload(&Q->part_start) + (chunk - sload(&Q->chunk_first));
    flush_partial<KPL>(tk, sload(&p.sinks->partials), part, lane);
        if (lane == 0 && n_q) atomicAdd(sload(&p.sinks->query_matches) + sload(sload(&p.sinks->out_index) + q), n_q);
        n_q = 0;
  }
  if (lane == 0 && n_matches) atomicAdd(sload(&p.sinks->match_counter), (unsigned long long)n_matches);
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
  // a full partial list's k-th key is a lower bound of the final k-th key: the largest of them
  // (one 8-byte load per list, 64 lists at a time) keeps almost every other key out of the
  // serial insertions below
  uint64_t floor_key = 0;
  for (uint32_t pi = (uint32_t)lane; pi < n_parts; pi += WAVE) {
    const uint64_t kth = p.partials[(uint64_t)(part_start + pi) * (uint64_t)(KPL * 64) + (k - 1u)];
    floor_key = kth > floor_key ? kth : floor_key;
  }
  for (int o = 32; o; o >>= 1) {
    const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(floor_key >> 32), o, WAVE) << 32) |
                           (uint32_t)__shfl_xor((int)(uint32_t)floor_key, o, WAVE);
    floor_key = other > floor_key ? other : floor_key;
  }
  // the loads of a group of lists are issued together (one wavefront walks hundreds of lists:
  // one round trip per list was the whole cost of this kernel)
  constexpr uint32_t GROUP = KPL <= 2 ? 8u : (KPL <= 4 ? 4u : 1u);
  for (uint32_t pi0 = 0; pi0 < n_parts; pi0 += GROUP) {
    uint64_t keys[GROUP][KPL];
#pragma unroll
    for (uint32_t g = 0; g < GROUP; ++g) {
      const uint32_t pi = pi0 + g < n_parts ? pi0 + g : n_parts - 1u;  // clamped: unconditional loads
      const uint64_t *src = p.partials + (uint64_t)(part_start + pi) * (uint64_t)(KPL * 64);
#pragma unroll
      for (int r = 0; r < KPL; ++r) keys[g][r] = src[(uint32_t)r * 64u + (uint32_t)lane];
    }
#pragma unroll
    for (uint32_t g = 0; g < GROUP; ++g) {
      if (pi0 + g >= n_parts) break;
#pragma unroll
      for (int r = 0; r < KPL; ++r)
        tk.offer(keys[g][r] != 0ull && keys[g][r] >= floor_key, keys[g][r], lane);
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
static void launch_and_t(const TqkScanParams &p, dim3 grid, dim3 block, hipStream_t st) {
  // instantiations: the pruning code costs registers the exhaustive scan does not need (and the
  // kernel names tell the modes apart in a profile); launches whose non-leader lists all have
  // bitmaps drop the seek / block-search code and its LDS
  if (p.exhaustive) {
    if (p.all_dense)
      and_kernel<KPL, false, true><<<grid, block, 0, st>>>(p);
    else
      and_kernel<KPL, false, false><<<grid, block, 0, st>>>(p);
  } else {
    if (p.all_dense)
      and_kernel<KPL, true, true><<<grid, block, 0, st>>>(p);
    else
      and_kernel<KPL, true, false><<<grid, block, 0, st>>>(p);
  }
}
template <int KPL>
static void launch_or_t(const TqkScanParams &p, bool /*dpp*/, dim3 grid, dim3 block, hipStream_t st) {
  if (p.or_windows) {  // window-parallel form: one workgroup per chunk
    if (p.exhaustive)
      or_kernel<KPL, false><<<grid, block, 0, st>>>(p);
    else
      or_kernel<KPL, true><<<grid, block, 0, st>>>(p);
  } else {  // candidate-driven form: one wavefront per chunk
#define TQ_UNION(PR, BO)                                                  \
  do {                                                                    \
    if (KPL <= 2)                                                         \
      union_kernel_small<KPL, PR, BO><<<grid, dim3(64), 0, st>>>(p);      \
    else                                                                  \
      union_kernel<KPL, PR, BO><<<grid, dim3(64), 0, st>>>(p);            \
  } while (0)
    if (p.boolean) {
      if (p.exhaustive)
        TQ_UNION(false, true);
      else
        TQ_UNION(true, true);
    } else if (p.exhaustive) {
      TQ_UNION(false, false);
    } else {
      TQ_UNION(true, false);
    }
#undef TQ_UNION
  }
}

hipError_t tqk_launch_and(const TqkScanParams &p, int kpl, bool /*use_dpp*/, hipStream_t st) {
  if (p.n_chunks == 0) return hipSuccess;
  const dim3 grid(p.n_chunks);
  const dim3 block(64);
  switch (kpl) {
    case 1: launch_and_t<1>(p, grid, block, st); break;
    case 2: launch_and_t<2>(p, grid, block, st); break;
    case 4: launch_and_t<4>(p, grid, block, st); break;
    default: launch_and_t<16>(p, grid, block, st); break;
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
static void launch_phrase_t(const TqkScanParams &p, bool /*dpp*/, dim3 grid, dim3 block, hipStream_t st) {
  if (p.max_terms <= 4u)  // fewer position cursors and half the per-candidate LDS
    phrase_kernel<KPL, 4><<<grid, block, 0, st>>>(p);
  else
    phrase_kernel<KPL, TQD_PH_MAX_TERMS><<<grid, block, 0, st>>>(p);
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
