// Count collector over bitmaps (src/collector/count_collector.rs:39-80; BufferedUnionScorer::count_including_deleted,
// src/query/union/buffered_union.rs:331-351, counts a union's docs out of its 64-bit bitset words the same way):
// a boolean combination of lists that all have a bitmap is a bitwise expression over their bitmap words —
// AND over the Must clauses of (OR over the clause's lists), AND NOT (OR over the MustNot lists), AND the alive
// bits — and its match count the popcount of the result.  No postings are decoded, no doc is scored.
//
// HBM-bound by construction: 8 bytes (bits + rank directory entry) per list per 32 docs.  A workgroup = one
// (query, tile of 65 536 docs); consecutive workgroups are the batch's queries on the SAME tile, so the tile's
// bitmap words of the lists the batch shares come out of the L2s.
#include "tq_common.hpp"
#include "tq_launch.h"

namespace {

constexpr uint32_t COUNT_THREADS = 256;
constexpr uint32_t COUNT_WORDS_PER_THREAD = 8;
constexpr uint32_t COUNT_TILE_WORDS = COUNT_THREADS * COUNT_WORDS_PER_THREAD;  // 2048 words = 65 536 docs

__global__ __launch_bounds__(COUNT_THREADS) void count_bitmap_kernel(TqkCountParams p) {
  const uint32_t q = blockIdx.x % p.n_queries, tile = blockIdx.x / p.n_queries;
  const TqkCountQuery *Q = p.queries + q;
  const uint32_t nt = sload(&Q->n_terms), kinds = sload(&Q->kinds), clause_end = sload(&Q->clause_end),
                 flags = sload(&Q->flags), narrow = sload(&Q->narrow);
  uint32_t cnt = 0;
#pragma unroll 2
  for (uint32_t i = 0; i < COUNT_WORDS_PER_THREAD; ++i) {
    const uint32_t w = tile * COUNT_TILE_WORDS + i * COUNT_THREADS + threadIdx.x;
    if (w >= p.n_words) break;
    uint32_t must = 0xFFFFFFFFu, nots = 0u, should = 0u, clause = 0u;
    for (uint32_t m = 0; m < nt; ++m) {
      const uint32_t bits = ((narrow >> m) & 1u) ? reinterpret_cast<const uint32_t *>(Q->dense[m])[w] : Q->dense[m][w].x;
      const uint32_t kind = (kinds >> (2u * m)) & 3u;
      if (kind == TQK_COUNT_MUST) {
        clause |= bits;
        if ((clause_end >> m) & 1u) {
          must &= clause;
          clause = 0u;
        }
      } else if (kind == TQK_COUNT_NOT) {
        nots |= bits;
      } else {
        should |= bits;
      }
    }
    uint32_t res = ((flags & TQK_COUNT_HAS_MUST) ? must : should) & ~nots;
    if (flags & TQK_COUNT_NEED_SHOULD) res &= should;
    // AliveBitSet: bit d of byte d >> 3 (alive_bitset.rs:58-61) = bit d & 31 of little-endian word d >> 5
    // (the device copy is padded with zero bytes past its last 64-bit word)
    if (p.alive) res &= reinterpret_cast<const uint32_t *>(p.alive)[w];
    cnt += (uint32_t)__popc(res);
  }
  // wave sum, then one atomic per wave
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if ((threadIdx.x & 63u) == 0u && cnt) atomicAdd(p.out_counts + q, cnt);
}

// A list without a bitmap, for the duration of one Count batch: one wave per 128-doc block, every doc a bit.
__global__ __launch_bounds__(256) void count_scatter_kernel(TqdSegment seg, const TqdTerm *terms, const uint4 *wgs,
                                                            uint32_t *bits, uint32_t words_per_list) {
  const int lane = (int)__lane_id();
  const uint32_t wave = uni(threadIdx.x >> 6);
  const uint4 w = sload(wgs + blockIdx.x);
  const TermRef t = load_term(terms, w.x);
  const uint32_t j = w.y + wave;
  if (j >= t.n_blocks) return;
  const Dec d = decode_block<true, false>(uni_ptr(seg.idx), t, j, lane);
  uint32_t *b = bits + (size_t)w.z * words_per_list;
  if (d.d0 != TQD_TERMINATED) atomicOr(b + (d.d0 >> 5), 1u << (d.d0 & 31u));
  if (d.d1 != TQD_TERMINATED) atomicOr(b + (d.d1 >> 5), 1u << (d.d1 & 31u));
}

}  // namespace

hipError_t tqk_launch_count_scatter(const TqdSegment &seg, const TqdTerm *terms, const uint4 *wgs, uint32_t n_wgs,
                                    uint32_t *bits, uint32_t words_per_list, hipStream_t st) {
  if (!n_wgs) return hipSuccess;
  count_scatter_kernel<<<dim3(n_wgs), dim3(256), 0, st>>>(seg, terms, wgs, bits, words_per_list);
  return hipGetLastError();
}

uint32_t tqk_count_tile_words() { return COUNT_TILE_WORDS; }

hipError_t tqk_launch_count_bitmaps(const TqkCountParams &p, hipStream_t st) {
  const uint32_t tiles = (p.n_words + COUNT_TILE_WORDS - 1) / COUNT_TILE_WORDS;
  if (!tiles || !p.n_queries) return hipSuccess;
  count_bitmap_kernel<<<dim3(tiles * p.n_queries), dim3(COUNT_THREADS), 0, st>>>(p);
  return hipGetLastError();
}
