// tq_common.hpp — device helpers shared by the gfx950 (CDNA4, wave64) kernels of the tantivy
// query-execution path (tq_and.hip, tq_union.hip, tq_phrase.hip, tq_misc.hip; one translation unit
// per kernel family so that they compile in parallel).
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
#pragma once
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
// (a float passed to uni() would be CONVERTED to an integer: its own overload, and no implicit ones)
__device__ __forceinline__ float uni_f(float x) { return __uint_as_float(uni(__float_as_uint(x))); }
__device__ uint32_t uni(float) = delete;
__device__ uint32_t uni(double) = delete;
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
  uint32_t mat_slot;  // column of the list in TqdSegment::docmat, or >= TQD_MAT_SLOTS
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
  r.has_freq = h.has_freq & 1u;
  r.mat_slot = ((h.has_freq >> 8) & 0xFFu) - 1u;  // 0 => 0xFFFFFFFF
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

// ---- range directory of a sparse list (TermHost::rdir_blob, tq_terms.cpp): the list's postings as one u32 each, in
// posting order — (doc & (2^S - 1)) << 16 | min(tf, 0xFFFF) — and dir[r] = the number of postings with doc < r << S
// for r = 0 .. (max_doc >> S) + 1 (S <= 16, chosen per list so that a range holds one to two postings on average).
// "Is doc d in the list, with which tf" is dir[d >> S], dir[(d >> S) + 1] and the few entries between them; docs that
// are close lie close in both arrays (a bitmap's locality at 6-8 bytes per posting instead of max_doc / 4 per list).
__device__ __forceinline__ uint32_t rdir_entry(uint32_t doc, uint32_t tf, uint32_t S) {
  return ((doc & ((1u << S) - 1u)) << 16) | (tf < 0xFFFFu ? tf : 0xFFFFu);
}
// every lane its own list and doc (on = the lane asks): found -> tf (0xFFFF = that or more: the packed value) and the
// posting's index
__device__ __forceinline__ bool rdir_lookup(const uint32_t *dir, const uint32_t *ent, uint32_t S, uint32_t doc, bool on,
                                            uint32_t &tf, uint32_t &pi) {
  uint32_t lo = 0, hi = 0;
  if (on) {
    const uint32_t r = doc >> S;
    lo = dir[r];
    hi = dir[r + 1u];
  }
  const uint32_t key = doc & ((1u << S) - 1u);
  bool found = false;
  on = on && lo < hi;
  while (__ballot(on)) {
    if (on) {
      const uint32_t e0 = ent[lo], e1 = lo + 1u < hi ? ent[lo + 1u] : 0xFFFFFFFFu;  // (two at a time: entries ascend inside a range)
      const bool m0 = (e0 >> 16) == key, m1 = lo + 1u < hi && (e1 >> 16) == key;
      if (m0 | m1) {
        tf = (m0 ? e0 : e1) & 0xFFFFu;
        pi = lo + (m0 ? 0u : 1u);
        found = true;
        on = false;
      } else {
        lo += 2u;
        on = lo < hi && (e1 >> 16) < key;
      }
    }
  }
  return found;
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

__device__ __forceinline__ uint32_t sortable(float x) {
  uint32_t fb = __float_as_uint(x);
  return fb ^ ((uint32_t)((int32_t)fb >> 31) | 0x80000000u);
}
// upper bound of a term's score inside one block (TermScorer::block_max_score,
// term_scorer.rs:58-75; skip.rs:175-184).  tail / no-freq / unknown block-max => weight itself
// (tf/(tf+norm) < 1).  Requires weight >= 0.
// `slack` (>= 1): the stored (fieldnorm id, tf) pair is the block's best under the SEGMENT's own
// average fieldnorm (serializer.rs:130-135,404-428); under the query's global average another doc
// of the block may score up to (1 + d)^2 higher, d = relative difference of the two averages
// (tf/(tf+norm) moves by at most d when the average does).  The reference accepts that risk
// (term_scorer.rs:58-70); here the host passes slack = (1 + d)^2 and pruning stays exact.
__device__ __forceinline__ float block_max_score(uint32_t meta, float w, const float *cache,
                                                 uint32_t has_freq, float slack) {
  const uint32_t tfc = meta >> 24;
  if (meta == META_TAIL || !has_freq || tfc == 0u) return w;
  const uint32_t tf = tfc == 255u ? 0xFFFFFFFFu : tfc;  // skip.rs:31-43
  return bm25(w, cache[(meta >> 16) & 0xFFu], tf) * slack;
}
// block_max_score for threshold tests (see bm25_bound)
__device__ __forceinline__ float block_max_bound(uint32_t meta, float w, const float *cache,
                                                 uint32_t has_freq, float slack) {
  const uint32_t tfc = meta >> 24;
  if (meta == META_TAIL || !has_freq || tfc == 0u) return w;
  const uint32_t tf = tfc == 255u ? 0xFFFFFFFFu : tfc;
  return bm25_bound(w, cache[(meta >> 16) & 0xFFu], tf) * slack;
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
// The same on the upper 16 bits only (first S registers of v): the result, low bits zero, still
// has at least k values at or above it — a slightly lower, equally valid bound at half the steps.
#ifndef TQ_KTH_LOW_BIT
#define TQ_KTH_LOW_BIT 8
#endif
template <int S>
__device__ __forceinline__ uint32_t kth_largest_hi16(const uint32_t (&v)[4], uint32_t k) {
  uint32_t ans = 0;
  for (int bit = 31; bit >= TQ_KTH_LOW_BIT; --bit) {
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

// ---- positions of one posting, one at a time (the phrase kernels' cursor merge; tq_tree.hip's phrase atoms)
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


// Term freq of posting `at` of a block and the sum of the term freqs of the <= 3 postings before
// it in its group of four (slots 4g..4g+3 are value g of the four bit streams: they share one
// 16-byte row of the tf payload, or two when the value straddles a word).  rec = the block's
// record; per-lane arguments.
__device__ __forceinline__ void group_tfs(const uint8_t *idx, const TermRef &t, const uint4 rec,
                                          uint32_t at, uint32_t &tf, uint32_t &excl) {
  uint32_t v[4];
  if (rec.y == META_TAIL) {
    const uint32_t g0 = at & ~3u;
#pragma unroll
    for (uint32_t l = 0; l < 4u; ++l) v[l] = g0 + l < t.n_tail ? t.tail_tfs[g0 + l] : 0u;
  } else {
    const uint32_t doc_bits = rec.y & 31u;
    const uint32_t strict = (rec.y >> 6) & 1u;
    const uint32_t b = (rec.y >> 8) & 0xFFu;
    const uint32_t bitpos = (at >> 2) * b;
    const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
    const uint8_t *q = idx + t.payload_base + rec.z + 16u * doc_bits + 16u * w;
    U4Unaligned r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
    if (b) {  // (b == 0: every tf of the block is `strict`)
      r0 = *reinterpret_cast<const U4Unaligned *>(q);
      r1 = *reinterpret_cast<const U4Unaligned *>(q + 16);  // may over-read: buffers are padded
    }
    const uint32_t mask = b >= 32u ? 0xFFFFFFFFu : ((1u << b) - 1u);
    v[0] = (__funnelshift_r(r0.x, r1.x, sh) & mask) + strict;
    v[1] = (__funnelshift_r(r0.y, r1.y, sh) & mask) + strict;
    v[2] = (__funnelshift_r(r0.z, r1.z, sh) & mask) + strict;
    v[3] = (__funnelshift_r(r0.w, r1.w, sh) & mask) + strict;
  }
  const uint32_t l0 = at & 3u;
  tf = l0 == 0u ? v[0] : (l0 == 1u ? v[1] : (l0 == 2u ? v[2] : v[3]));
  excl = (l0 > 0u ? v[0] : 0u) + (l0 > 1u ? v[1] : 0u) + (l0 > 2u ? v[2] : 0u);
}

}  // namespace
