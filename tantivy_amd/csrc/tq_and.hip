// tq_and.hip — the AND kernel (block_wand_intersection restated for wavefronts).
// Shared device helpers: tq_common.hpp.
#include "tq_common.hpp"

namespace {

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
template <int KPL, bool PRUNE, bool DENSE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(80))) void and_kernel(TqkScanParams p) {
  constexpr bool USE_DPP = true;
  __shared__ AndLdsT<DENSE> L;  // one wavefront per workgroup: finished chunks free their slot at once
  const int lane = (int)__lane_id();
  if (blockIdx.x >= p.n_chunks) return;
  const uint4 crec = sload(p.chunk_recs + blockIdx.x);
  const uint32_t chunk = crec.w, t_begin = crec.x, t_end = crec.y;

  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;

  // ---- per-query state (wave-uniform)
  uint32_t q = crec.z;
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
  // PROFILING (TQ_DEBUG bits 16..19 = phase): wave cycles of one phase, summed into the match
  // counter.  1 setup + flush, 2 threshold + pre-filter, 3 stage A, 4 stage B, 5 stage C
  const uint32_t tphase = (p.debug >> 16) & 15u;
  uint64_t tacc = 0, tlast = tphase ? __builtin_readcyclecounter() : 0ull;
  auto tick = [&](uint32_t done) __attribute__((always_inline)) {
    if (tphase) {
      const uint64_t now = __builtin_readcyclecounter();
      if (done == tphase) tacc += now - tlast;
      tlast = now;
    }
  };

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
    tick(4u);
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
        float ub = bm25_bound(w_lead, norm, tf) + block_max_bound(mo.x, w1, L.cache, t1.has_freq, p.bound_slack);
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
        const float ub = (s + block_max_score(mo.x, w, L.cache, tr.has_freq, p.bound_slack) + rest) * 1.000001f;
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
      if (!(p.debug & (224u | 0xF00000u))) n_matches += (uint32_t)__popcll(hit);  // COUNTERS (bits 20..23: the shared launch's byte counters, nothing here)
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
    tick(5u);
  };

  // ---- stage B: locate in list 1.  Dense list: the bitmap answers membership, which is the
  // strongest filter there is, so nothing else is looked at first.  Other lists: pruned mode
  // scores the leader exactly (one fieldnorm gather) before paying for the seek.
  auto stageB = [&](uint32_t n) __attribute__((always_inline)) {
    tick(3u);
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
    tick(4u);
  };

  auto drain = [&]() __attribute__((always_inline)) {
    while (q1n) {
      stageB(q1n < 64u ? q1n : 64u);
      while (q2n >= 64u) stageC(64u);
    }
    while (q2n) stageC(q2n < 64u ? q2n : 64u);
  };

  setup_query();
  tick(1u);
  for (uint32_t t = t_begin; t < t_end; ++t) {
    tick(3u);
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
      tick(1u);
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
        ub = block_max_score(mo_mine.x, w_lead, L.cache, lead.has_freq, p.bound_slack);
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
              const float b2 = block_max_score(r.y, w, L.cache, tr.has_freq, p.bound_slack);
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
    tick(2u);

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
  tick(1u);
  if (tphase) n_matches = (uint32_t)(tacc >> 4);
  if (lane == 0 && n_matches) atomicAdd(sload(&p.sinks->match_counter), (unsigned long long)n_matches);
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
