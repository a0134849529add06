// tq_phrase.hip — the exact-phrase kernel.
// Shared device helpers: tq_common.hpp.
#include "tq_common.hpp"

namespace {

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
  const uint4 crec = sload(p.chunk_recs + blockIdx.x);
  const uint32_t chunk = crec.w, t_begin = crec.x, t_end = crec.y;
  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;
  const uint8_t *pos = seg.pos;

  uint32_t q = crec.z;
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

}  // namespace

// =================================================================== launch wrappers
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
