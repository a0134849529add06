// tq_tree.hip — boolean queries whose clauses are themselves boolean queries of terms: the
// `SpecializedScorer::Other` trees of BooleanWeight::complex_scorer (boolean_weight.rs:236-431) that do not
// flatten — an intersection inside a union (`(+b +c) d`), under MustNot (`+a -(+b +c)`), a nested query with its
// own MustNot / optional terms / minimum_number_should_match, minimum_number_should_match over nested groups
// (disjunction.rs:113-139).  Every clause is a term or a BooleanQuery whose members are terms or CONJUNCTIONS of
// terms ("atoms": `+a +((+b +c) d)` — an atom is present where all its terms are and scores their sum,
// Intersection::score): three levels when the innermost is an intersection of terms.
//
// Every list is reached through its bitmap (its own, or the probe tables of tq_terms.cpp): the doc set of such a
// tree is a bitwise expression over the lists' bitmap words — per nested query AND over its Must terms, AND NOT
// over its MustNot terms, "at least m of its Should terms" from a bit-sliced counter; the same one level up over
// the clauses — so a lane evaluates 32 docs per step from one coalesced 8-byte load per list (what tq_count.hip
// does for Count), and only the docs that MATCH are scored: bitmap word -> rank -> tf byte per term of a matching
// clause, BM25 summed clause by clause (Intersection::score: left + right + others, intersection.rs:325-329;
// RequiredOptionalScorer::score = req + opt, reqopt_scorer.rs:85-98; a union's SumCombiner), the per-wave top-k
// in registers, partial lists reduced by merge_kernel.  Nothing is pruned: the reference runs these trees through
// the generic for_each_pruning_scorer (weight.rs:47-60), which only filters by threshold.
//
// HBM-bound by construction: 8 B per list per 32 docs of the segment, whatever the lists hold.
//
// PHRASES inside boolean queries (`+"a b" +c`, `"a b" c`, `+c -"a b"`, `+a +("b c" d)`: VERDICT r04 item 4): an atom
// can be a PhraseQuery of <= 4 terms (tree_kernel<KPL, true>).  Where all its lists hold the doc, the lane walks the
// positions of the doc in every list with one cursor per term (bitmap rank -> tf bytes + position directory -> first
// position index, then PositionReader deltas one at a time: PhraseScorer::phrase_match / compute_phrase_count,
// phrase_scorer.rs:347-587) and the atom is present where the count is > 0, scoring bm25(sum of idfs, norm, count)
// (phrase_scorer.rs:576-587).  The bitmap expression then only proposes docs: a phrase atom enters it as the AND of
// its lists where that can only ADD docs (positive polarity) and as nothing where it could remove some (under an odd
// number of MustNots), and every proposed doc is decided by the exact per-doc evaluation of the scoring stage.
#include "tq_common.hpp"
#include "tq_launch.h"

namespace {

constexpr uint32_t TREE_TILE_WORDS = TQK_TREE_TILE_WORDS;  // bitmap words per (query, tile) wavefront

// bit-sliced counter of one-bit-per-doc inputs (4 planes), SATURATING at 15: add, and "count >= m" for m <= 15.
// (A query holds up to TQ_MAX_TERMS = 16 Should inputs on one level: without the saturation a doc that held all 16
// wrapped to 0 and fell out of the doc set — the best-scoring doc of the query.)
struct SlicedCount {
  uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
  __device__ __forceinline__ void add(uint32_t x) {
    uint32_t c = p0 & x;
    p0 ^= x;
    x = c;
    c = p1 & x;
    p1 ^= x;
    x = c;
    c = p2 & x;
    p2 ^= x;
    x = c;
    c = p3 & x;  // the carry out of the top plane: the count sticks at 15
    p3 ^= x;
    p0 |= c;
    p1 |= c;
    p2 |= c;
    p3 |= c;
  }
  __device__ __forceinline__ uint32_t at_least(uint32_t m) const {  // m wave-uniform, 0..15
    if (m == 0u) return 0xFFFFFFFFu;
    uint32_t gt = 0u, eq = 0xFFFFFFFFu;
    const uint32_t pl[4] = {p0, p1, p2, p3};
#pragma unroll
    for (int i = 3; i >= 0; --i) {
      if ((m >> i) & 1u) {
        eq &= pl[i];
      } else {
        gt |= eq & pl[i];
        eq &= ~pl[i];
      }
    }
    return gt | eq;
  }
};

template <int KPL, bool PH>
__global__ __launch_bounds__(64) void tree_kernel(TqkTreeParams p) {
  const int lane = (int)__lane_id();
  const uint32_t q = blockIdx.x % p.n_queries, tile = blockIdx.x / p.n_queries;
  const TqdTreeQuery *Q = p.queries + q;
  const uint32_t nt = sload(&Q->n_terms), nc = sload(&Q->n_clauses), k = sload(&Q->k);
  const uint32_t top_need = sload(&Q->top_need), top_has_must = sload(&Q->top_has_must);
  const float *cache = p.caches + (size_t)sload(&Q->cache_idx) * 256u;
  const uint8_t *tbase = p.table_base;
  const TqdSegment seg = p.seg;
  TopK<KPL> tk;
  tk.reset(k);
  uint32_t n_matches = 0;
  if (nt == 0 || nc == 0) {  // (a query the planner found empty: an absent Must term, too few Should clauses, ...)
    flush_partial(tk, sload(&p.sinks->partials), sload(&Q->part_start) + tile, lane);
    return;
  }
  const uint32_t w_end = (tile + 1u) * TREE_TILE_WORDS < p.n_words ? (tile + 1u) * TREE_TILE_WORDS : p.n_words;
  for (uint32_t w0 = tile * TREE_TILE_WORDS; w0 < w_end; w0 += 64u) {
    const uint32_t w = w0 + (uint32_t)lane;
    const bool in = w < w_end;
    // ---- the doc set of 32 docs per lane: clause by clause, then the clauses one level up
    uint32_t top_must = 0xFFFFFFFFu, top_not = 0u;
    SlicedCount top_should;
    for (uint32_t c = 0; c < nc; ++c) {
      const uint32_t t0 = sload(Q->first_term + c), t1 = sload(Q->first_term + c + 1u);
      const uint32_t outer_c = sload(Q->outer + c);
      uint32_t must = 0xFFFFFFFFu, nots = 0u;
      SlicedCount should;
      uint32_t atom = 0xFFFFFFFFu;  // the docs that hold every term of the current atom so far
      uint32_t atom_any = 0u;       // ... or any of them (a union one level down: atom_end bit 2)
      for (uint32_t t = t0; t < t1; ++t) {
        const uint2 *bm = reinterpret_cast<const uint2 *>(tbase + ((uint64_t)sload(Q->dense_off + t) << 3));
        const uint32_t bits = in ? bm[w].x : 0u;
        atom &= bits;
        atom_any |= bits;
        const uint32_t ae = sload(Q->atom_end + t);
        if (!(ae & 1u)) continue;
        if (ae & 4u) atom = atom_any;
        atom_any = 0u;
        const uint32_t inner = sload(Q->inner + t);
        if constexpr (PH) {  // a phrase under an odd number of MustNots must not remove docs it only MAY hold
          if ((ae & 2u) && ((outer_c == TQD_ROLE_MUST_NOT) != (inner == TQD_ROLE_MUST_NOT))) atom = 0u;
        }
        if (inner == TQD_ROLE_MUST)
          must &= atom;
        else if (inner == TQD_ROLE_MUST_NOT)
          nots |= atom;
        else
          should.add(atom);
        atom = 0xFFFFFFFFu;
      }
      const uint32_t cm = must & ~nots & should.at_least(sload(Q->inner_need + c));
      const uint32_t outer = outer_c;
      if (outer == TQD_ROLE_MUST)
        top_must &= cm;
      else if (outer == TQD_ROLE_MUST_NOT)
        top_not |= cm;
      else
        top_should.add(cm);
    }
    uint32_t match = (top_has_must ? top_must : 0xFFFFFFFFu) & ~top_not & top_should.at_least(top_need);
    if (!in) match = 0u;
    if (seg.alive) match &= in ? reinterpret_cast<const uint32_t *>(seg.alive)[w] : 0u;  // AliveBitSet (alive_bitset.rs:58-61)
    if constexpr (!PH) n_matches += (uint32_t)__popc(match);
    // ---- score the matching docs: every lane takes the lowest doc of its word until none has one left
    while (__ballot(match != 0u)) {
      const bool has = match != 0u;
      const uint32_t bit = has ? (uint32_t)__builtin_ctz(match) : 0u;
      match &= match - 1u;
      const uint32_t doc = (w << 5) | bit;
      const float norm = cache[has ? fieldnorm_id(seg, doc) : 0u];
      float musts_first = 0.0f, musts_second = 0.0f, musts_others = 0.0f, opt = 0.0f;
      uint32_t n_must_clauses = 0;
      bool all_must = true, any_not = false;  // (PH: the doc's own verdict)
      uint32_t n_should_clauses = 0;
      for (uint32_t c = 0; c < nc; ++c) {
        const uint32_t t0 = sload(Q->first_term + c), t1 = sload(Q->first_term + c + 1u);
        const uint32_t outer = sload(Q->outer + c);
        if (!PH && outer == TQD_ROLE_MUST_NOT) continue;  // (the doc set already excludes them)
        bool must_ok = true, not_ok = true;
        uint32_t ns = 0;
        float csum = 0.0f;
        bool atom_ok = true;    // the doc holds every term of the current atom so far
        bool atom_some = false; // ... or any of them (a union one level down)
        float atom_sum = 0.0f;  // ... and what they score together (Intersection::score / SumCombiner)
        uint32_t atom_t0 = t0;
        for (uint32_t t = t0; t < t1; ++t) {
          const uint32_t inner = sload(Q->inner + t);
          const uint32_t ae = sload(Q->atom_end + t);
          const uint2 *bm = reinterpret_cast<const uint2 *>(tbase + ((uint64_t)sload(Q->dense_off + t) << 3));
          uint2 wd = make_uint2(0u, 0u);
          if (has) wd = bm[w];
          const bool present = has && ((wd.x >> bit) & 1u);
          atom_ok = atom_ok && present;
          atom_some = atom_some || present;
          if (present && inner != TQD_ROLE_MUST_NOT && !(PH && (ae & 2u))) {
            const uint32_t pi = wd.y + (uint32_t)__popc(wd.x & ((1u << bit) - 1u));
            uint32_t tf = (tbase + ((uint64_t)sload(Q->tf8_off + t) << 3))[pi];
            if (tf == 255u) {  // saturated byte: block record -> packed tf (tq_common.hpp)
              const TqdTermHead *h = p.terms + sload(Q->handle + t);
              TermRef tr{};
              tr.rec = h->rec;
              tr.tail_tfs = h->tail_tfs;
              tr.payload_base = h->payload_base;
              tr.has_freq = h->has_freq & 1u;
              const uint4 r = tr.rec[pi >> 7];
              tf = block_tf_at(seg.idx, tr, make_uint2(r.y, r.z), pi & 127u);
            }
            atom_sum = atom_sum + bm25(__uint_as_float(sload(Q->weight_bits + t)), norm, tf);
          }
          if (!(ae & 1u)) continue;
          if (ae & 4u) atom_ok = atom_some;  // (a union: the present terms' scores are already in atom_sum)
          if constexpr (PH) {
            if (ae & 2u) {  // a PhraseQuery: count the positions where its terms line up (lanes that hold them all)
              uint32_t cnt = 0;
              if (atom_ok) {
                PosCursor cur[TQK_TREE_PHRASE_TERMS];
                const uint32_t n_ph = t + 1u - atom_t0;
#pragma unroll
                for (uint32_t m = 0; m < TQK_TREE_PHRASE_TERMS; ++m) {
                  cur[m].valid = false;
                  cur[m].idx = cur[m].end = cur[m].cur = 0;
                  if (m < n_ph) {
                    const uint32_t tt = atom_t0 + m;
                    const uint2 wm = reinterpret_cast<const uint2 *>(tbase + ((uint64_t)sload(Q->dense_off + tt) << 3))[w];
                    const uint32_t pi = wm.y + (uint32_t)__popc(wm.x & ((1u << bit) - 1u));
                    // the four tf bytes of the posting's group of four + the group's directory entry
                    const uint32_t tw = *reinterpret_cast<const uint32_t *>(tbase + ((uint64_t)sload(Q->tf8_off + tt) << 3) + (pi & ~3u));
                    const uint32_t dv = reinterpret_cast<const uint32_t *>(tbase + ((uint64_t)sload(Q->dir_off + tt) << 3))[pi >> 2];
                    const uint32_t l0 = pi & 3u;
                    const uint32_t b0 = tw & 0xFFu, b1 = (tw >> 8) & 0xFFu, b2 = (tw >> 16) & 0xFFu, b3 = tw >> 24;
                    uint32_t tf = l0 == 0u ? b0 : (l0 == 1u ? b1 : (l0 == 2u ? b2 : b3));
                    uint32_t ex = (l0 > 0u ? b0 : 0u) + (l0 > 1u ? b1 : 0u) + (l0 > 2u ? b2 : 0u);
                    if (tf == 255u || (l0 > 0u && b0 == 255u) || (l0 > 1u && b1 == 255u) || (l0 > 2u && b2 == 255u)) {
                      const TqdTermHead *h = p.terms + sload(Q->handle + tt);  // a saturated byte: the packed values
                      TermRef tr{};
                      tr.rec = h->rec;
                      tr.tail_tfs = h->tail_tfs;
                      tr.payload_base = h->payload_base;
                      tr.has_freq = h->has_freq & 1u;
                      tr.n_tail = h->n_tail;
                      group_tfs(seg.idx, tr, tr.rec[pi >> 7], pi & 127u, tf, ex);
                    }
                    const TqdTerm *term = p.terms + sload(Q->handle + tt);
                    const uint32_t fp = dv + ex;  // index of the doc's first position in the term's stream
                    cur[m].idx = fp + 1u;
                    cur[m].end = fp + tf;
                    cur[m].valid = tf >= 1u;
                    if (cur[m].valid) cur[m].cur = sload(Q->phrase_off + tt) + position_delta(seg.pos, term, fp);
                  }
                }
                bool done = false;
                while (cur[0].valid && !done) {
                  const uint32_t av = cur[0].cur;
                  bool okv = true;
#pragma unroll
                  for (uint32_t m = 1; m < TQK_TREE_PHRASE_TERMS; ++m) {
                    if (m < n_ph && !done) {
                      const TqdTerm *term = p.terms + sload(Q->handle + atom_t0 + m);
                      while (cur[m].valid && cur[m].cur < av) pos_advance(cur[m], seg.pos, term);
                      if (!cur[m].valid)
                        done = true;
                      else if (cur[m].cur != av)
                        okv = false;
                    }
                  }
                  if (done) break;
                  if (okv) {
                    ++cnt;
#pragma unroll
                    for (uint32_t m = 1; m < TQK_TREE_PHRASE_TERMS; ++m)
                      if (m < n_ph) pos_advance(cur[m], seg.pos, p.terms + sload(Q->handle + atom_t0 + m));
                  }
                  pos_advance(cur[0], seg.pos, p.terms + sload(Q->handle + atom_t0));
                }
              }
              atom_ok = cnt > 0u;
              atom_sum = atom_ok ? bm25(__uint_as_float(sload(Q->weight_bits + atom_t0)), norm, cnt) : 0.0f;
            }
          }
          if (inner == TQD_ROLE_MUST_NOT) {
            not_ok = not_ok && !atom_ok;
          } else {
            if (inner == TQD_ROLE_MUST) must_ok = must_ok && atom_ok;
            if (atom_ok) {
              if (inner == TQD_ROLE_SHOULD) ++ns;
              csum = csum + atom_sum;
            }
          }
          atom_ok = true;
          atom_some = false;
          atom_sum = 0.0f;
          atom_t0 = t + 1u;
        }
        const bool cmatch = has && must_ok && not_ok && ns >= sload(Q->inner_need + c);
        if (outer == TQD_ROLE_MUST) {  // Intersection::score: left + right + sum(others), clauses cheapest first
          if (n_must_clauses == 0u)
            musts_first = csum;
          else if (n_must_clauses == 1u)
            musts_second = csum;
          else
            musts_others = musts_others + csum;
          ++n_must_clauses;
          all_must = all_must && cmatch;
        } else if (outer == TQD_ROLE_MUST_NOT) {
          any_not = any_not || cmatch;
        } else if (cmatch) {
          opt = opt + csum;  // a Should clause that matches adds its score (SumCombiner / RequiredOptionalScorer)
          ++n_should_clauses;
        }
      }
      float s = musts_first;
      if (n_must_clauses >= 2u) s = s + musts_second;
      if (n_must_clauses >= 3u) s = s + musts_others;
      s = n_must_clauses ? s + opt : opt;
      if constexpr (PH) {  // the bitmap expression only proposed the doc
        const bool doc_ok = has && all_must && !any_not && n_should_clauses >= top_need;
        n_matches += doc_ok ? 1u : 0u;
        tk.offer(doc_ok, make_key(s, doc), lane);
      } else {
        tk.offer(has, make_key(s, doc), lane);
      }
    }
  }
  flush_partial(tk, sload(&p.sinks->partials), sload(&Q->part_start) + tile, lane);
  for (int off = 32; off > 0; off >>= 1) n_matches += __shfl_down(n_matches, off, 64);
  if (lane == 0 && n_matches) {
    atomicAdd(sload(&p.sinks->query_matches) + sload(&p.sinks->out_index)[q], n_matches);
    atomicAdd(sload(&p.sinks->match_counter), (unsigned long long)n_matches);
  }
}

}  // namespace

uint32_t tqk_tree_tiles(uint32_t n_words) { return (n_words + TREE_TILE_WORDS - 1u) / TREE_TILE_WORDS; }

template <bool PH>
static void launch_tree_t(const TqkTreeParams &p, int kpl, dim3 grid, dim3 block, hipStream_t st) {
  switch (kpl) {
    case 1: tree_kernel<1, PH><<<grid, block, 0, st>>>(p); break;
    case 2: tree_kernel<2, PH><<<grid, block, 0, st>>>(p); break;
    case 4: tree_kernel<4, PH><<<grid, block, 0, st>>>(p); break;
    default: tree_kernel<16, PH><<<grid, block, 0, st>>>(p); break;
  }
}
hipError_t tqk_launch_tree(const TqkTreeParams &p, int kpl, hipStream_t st) {
  const uint32_t tiles = tqk_tree_tiles(p.n_words);
  if (!tiles || !p.n_queries) return hipSuccess;
  const dim3 grid(tiles * p.n_queries), block(64);
  if (p.any_phrase)
    launch_tree_t<true>(p, kpl, grid, block, st);
  else
    launch_tree_t<false>(p, kpl, grid, block, st);
  return hipGetLastError();
}
