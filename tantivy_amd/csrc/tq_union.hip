// tq_union.hip — unions and boolean queries: or_kernel (4096-doc windows) and union_kernel (candidate-driven).
// Shared device helpers: tq_common.hpp.
// (Since round 3 the pure unions of a batch — pruned, k <= 128, <= 8 terms, segment with a doc
// matrix — run term by term in tq_ushare.hip; this file serves the boolean queries (union_kernel<..,
// BOOL = true>, with the doc-matrix pre-stage b0_test in front of the exact walk), the unions the
// shared launch does not take, and the exhaustive mode (or_kernel).)
#include "tq_common.hpp"

// build switches of the union kernel (tools/build_variant.py makes A/B libraries from them)
#ifndef TQ_U_TIMERS
#define TQ_U_TIMERS 0  // region timers cost 2 %: tools/probe_phases.py builds a variant with them
#endif
#ifndef TQ_U_WAVES
#define TQ_U_WAVES 5  // occupancy target of the k <= 128 instantiations: 96 VGPRs, (almost) no scratch.
                      // At 6 (80 VGPRs) a third of the vector-memory instructions were spills: +10..13 % time;
                      // LDS (6.9 KB per wavefront) caps the CU at 22 wavefronts anyway
#endif
#ifndef TQ_U_SWEEP_RATIO
#define TQ_U_SWEEP_RATIO 32u
#endif

namespace {

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
  const uint4 crec = sload(p.chunk_recs + blockIdx.x);
  const uint32_t chunk = crec.w, t_begin = crec.x, t_end = crec.y;

  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;
  uint32_t q = crec.z;
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
template <bool BOOL>
struct UnionLds {  // per wavefront
  uint32_t pay[516];
  uint32_t q1_doc[191], q1_tf[191];
  float cache[256];
  float suffix[TQD_MAX_TERMS + 1];
  // pure unions: survivors of the membership stage (doc, tf, membership bits | fieldnorm id << 16,
  // what the lists after the leader can still add) and per-query tables read by broadcast
  // (boolean queries: survivors of the doc-matrix pre-stage, doc and tf only)
  uint32_t q2_doc[BOOL ? 191 : 127], q2_tf[BOOL ? 191 : 127], q2_mx[BOOL ? 1 : 127];
  float q2_rest[BOOL ? 1 : 127];
  // boolean queries: doc-matrix bits of every Must clause after the leader set whose terms all
  // have a column (a doc with none of them cannot match), and of the MustNot terms with a column
  uint64_t cmask[BOOL ? TQD_MAX_TERMS : 1], csig[BOOL ? TQD_MAX_TERMS : 1];  // (columns / signature bits)
  uint64_t bmask[BOOL ? 3 : 1];  // [0] MustNot columns, [1] / [2] columns / signature bits of the lead Must
                                 // clause (optional leaders; 0 = no test)
  float wgt[TQD_MAX_TERMS];
  const uint2 *dptr[TQD_MAX_TERMS];
  uint32_t mshift[TQD_MAX_TERMS];  // bit of the list's column in a doc-matrix word
};

template <int KPL, bool PRUNE, bool BOOL>
__device__ __forceinline__ void union_body(const TqkScanParams &p) {
  constexpr bool USE_DPP = true;
  __shared__ UnionLds<BOOL> L;
  const int lane = (int)__lane_id();
  if (blockIdx.x >= p.n_chunks) return;
  const uint4 crec = sload(p.chunk_recs + blockIdx.x);
  const uint32_t chunk = crec.w, t_begin = crec.x, t_end = crec.y;
  uint32_t q = crec.z;
  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;
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
  uint32_t q1n = 0, q2n = 0;
  uint32_t dense_mask = 0, sparse_mask = 0;  // pure unions: lists with / without a bitmap
  uint32_t mat_mask = 0;                     // ... with a column in the doc matrix
  uint32_t n_cmask = 0;                      // boolean queries: testable Must clauses after the leader set
  uint32_t sig_mask = 0;                     // ... lists with a signature bit instead of a column
  bool lead_test = false;                    // ... the lead Must clause is testable (optional leaders)
  const bool use_sig = BOOL && !(p.debug & 65536u);
  uint32_t slots_sum = 0;  // checksum of the threshold slots at the last radix select
  float slack_abs = 0.0f;
  // leader of the current tile
  uint32_t li = 0, li_end = 0;
  bool dead = false;
  TermRef lead{};
  float w_lead = 0.0f;
  // bitmap sweep of a dense leader's tiles (pure unions, pruned): see the main loop
  const uint2 *sw_lead = nullptr;  // the leader's bitmap + rank directory, or null = no sweep
  uint32_t sw_after = 0, sw_before = 0, sw_n = 0;  // dense lists after / before the leader
  float sw_ssum = 0.0f;          // lane j < 2^sw_n: weights of the subset j of the lists after
  float sw_sparse_after = 0.0f;  // weights of the lists after the leader without a bitmap
  bool q1_is_pi = false;         // queue 1 carries posting indices instead of term freqs
  // PROFILING (TQ_DEBUG bits 16..19 = region): wave cycles spent inside ONE region per run (the
  // counter is only read at the edges of that region), summed into the match counter.
  // 1 whole chunk, 2 query setup, 3 flush, 4 tile bookkeeping + threshold, 5 pre-filter,
  // 6 bitmap sweep (stages B / C inside included), 7 decode path (B / C included), 8 stage B,
  // 9 stage C
  const uint32_t tphase = TQ_U_TIMERS ? (p.debug >> 16) & 15u : 0u;
  uint64_t tacc = 0, tlast = 0;
  auto tb = [&](uint32_t ph) __attribute__((always_inline)) {
    if (TQ_U_TIMERS && tphase == ph) tlast = __builtin_readcyclecounter();
  };
  auto te = [&](uint32_t ph) __attribute__((always_inline)) {
    if (TQ_U_TIMERS && tphase == ph) tacc += __builtin_readcyclecounter() - tlast;
  };
  tb(1u);

  auto setup_query = [&]() __attribute__((always_inline)) {
    tb(2u);
    Q = p.queries + q;
    q_tile_start = sload(p.tile_starts + q);
    q_tile_end = sload(p.tile_starts + q + 1u);
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
    // (the weights come in with ONE vector load, lane m <-> list m; the sums run over registers
    // in the order nt-1 .. m, the order every bound was derived with)
    float suf = 0.0f;
    {
      const float wv = (uint32_t)lane < nt ? Q->weight[lane] : 0.0f;
      float mine = 0.0f;
      for (uint32_t m = nt; m-- > 0u;) {
        suf += __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(wv), (int)m));
        if ((uint32_t)lane == m) mine = suf;
      }
      if ((uint32_t)lane <= nt) L.suffix[lane] = mine;  // suffix[nt] = 0
    }
    if constexpr (!BOOL) {  // per-list tables of the membership stage, one list per lane
      const uint2 *dp = nullptr;
      uint32_t slot = 0xFFFFFFFFu;
      if ((uint32_t)lane < nt) {
        const TqdTerm *tt = p.terms + Q->term[lane];
        dp = p.use_dense ? tt->dense : nullptr;
        if (dp && seg.docmat) slot = ((tt->has_freq >> 8) & 0xFFu) - 1u;
        L.dptr[lane] = dp;
        L.wgt[lane] = Q->weight[lane];
        L.mshift[lane] = 8u + slot;
      }
      dense_mask = (uint32_t)__ballot(dp != nullptr);
      mat_mask = (uint32_t)__ballot(slot < TQD_MAT_SLOTS);
      sparse_mask = ((1u << nt) - 1u) & ~dense_mask;
      slack_abs = suf * 4.0e-6f;  // bounds summed by add-then-subtract: absolute slack
    } else {
      // boolean queries: the doc-matrix pre-stage (stageB0) needs every list's column and weight,
      // the column masks of the all-column Must clauses and of the MustNot terms
      // a list's membership bit: its doc-matrix column (exact), else its signature bit (a clear
      // bit proves absence, a set one means "maybe"), else none: mshift < 64 / 64..127 / >= 128
      uint32_t slot = 0xFFFFFFFFu, sbit = 0xFFFFFFFFu;
      if ((uint32_t)lane < nt) {
        const TqdTerm *tt = p.terms + Q->term[lane];
        if (p.use_dense && tt->dense && seg.docmat) slot = ((tt->has_freq >> 8) & 0xFFu) - 1u;
        if (slot >= TQD_MAT_SLOTS && use_sig && seg.docmat) sbit = ((tt->has_freq >> 16) & 0xFFu) - 1u;
        L.wgt[lane] = Q->weight[lane];
        L.mshift[lane] = slot < TQD_MAT_SLOTS ? 8u + slot : (sbit < TQD_SIG_BITS ? TQD_SIG_SHIFT + sbit : 128u);
      }
      mat_mask = (uint32_t)__ballot(slot < TQD_MAT_SLOTS);
      sig_mask = (uint32_t)__ballot(sbit < TQD_SIG_BITS);
      slack_abs = suf * 4.0e-6f;
      uint64_t notm = 0, leadm = 0, leads = 0, cm = 0, cs = 0;
      bool call = true;  // every term of the current clause has a column or a signature bit
      n_cmask = 0;
      bool lead_all = true;
      for (uint32_t m = 0; m < nt; ++m) {
        const uint32_t role = (roles >> (2u * m)) & 3u;
        const bool has = (mat_mask >> m) & 1u, hsig = (sig_mask >> m) & 1u;
        const uint64_t bit = has ? 1ull << (8u + (uint32_t)__builtin_amdgcn_readlane((int)slot, (int)m)) : 0ull;
        const uint64_t sgb = hsig ? 1ull << (TQD_SIG_SHIFT + (uint32_t)__builtin_amdgcn_readlane((int)sbit, (int)m)) : 0ull;
        if (role == TQD_ROLE_MUST_NOT) {
          notm |= bit;  // (only an exact bit may exclude)
        } else if (m < n_lead) {
          if (m >= n_opt_lead) {  // the lead Must clause (when optional lists lead in front of it)
            leadm |= bit;
            leads |= sgb;
            lead_all = lead_all && (has || hsig);
          }
        } else if (role == TQD_ROLE_MUST) {
          cm |= bit;
          cs |= sgb;
          call = call && (has || hsig);
          if ((clause_end >> m) & 1u) {
            if (call && n_cmask < TQD_MAX_TERMS) {
              if (lane == 0) {
                L.cmask[n_cmask] = cm;
                L.csig[n_cmask] = cs;
              }
              ++n_cmask;
            }
            cm = 0;
            cs = 0;
            call = true;
          }
        }
      }
      if (lane == 0) {
        L.bmask[0] = notm;
        L.bmask[1] = (n_opt_lead && lead_all) ? leadm : 0ull;
        L.bmask[2] = (n_opt_lead && lead_all) ? leads : 0ull;
      }
      lead_test = n_opt_lead && lead_all;
    }
    wave_mem_fence();
    min_norm = sload(p.caches + (size_t)ci * 256u +
                     (seg.fieldnorm ? seg.min_fieldnorm_id : seg.const_fieldnorm_id));
    thr = 0;
    thr_g = 0;
    slots_sum = 0;
    li = 0xFFFFFFFFu;
    li_end = 0;
    dead = false;
    tk.reset(k);
    te(2u);
  };

  // ---- stage B: the other lists of <= 64 candidates of leader li
  auto stageB = [&](uint32_t n, bool from_q2) __attribute__((always_inline)) {
    const uint32_t base = (from_q2 ? q2n : q1n) - n;
    if (from_q2)
      q2n = base;
    else
      q1n = base;
    if (p.debug & (from_q2 ? 128u : 64u)) n_matches += n;  // COUNTERS
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, tf = 0;
    float norm = 0.0f, s = 0.0f;
    if (alive) {
      doc = from_q2 ? L.q2_doc[base + lane] : L.q1_doc[base + lane];
      tf = from_q2 ? L.q2_tf[base + lane] : L.q1_tf[base + lane];
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
      // what the lists m.. can still add.  Lists below li add nothing (found there = dropped:
      // that list's tile scores the doc), so once the walk wraps around to them the score is
      // final unless the doc is dropped: below the threshold it is dead either way, before
      // the (sparse, expensive) ownership probes
      if (prune && alive) {
        const float rest = m > li ? L.suffix[m] : 0.0f;
        if (m > li || !BOOL)
          alive = sortable((((s + oth) + (csum + opt)) + rest) * 1.000001f) >= thr;
      }
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
      if (!(p.debug & 480u)) n_matches += (uint32_t)__popcll(hit);  // COUNTERS
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

  // ================================================================ pure unions: two stages
  // C (64 survivors): the exact score.  Dense lists after the leader that hold the doc: rank
  // from the bitmap word -> block record -> tf; lists without a bitmap: seek + block search;
  // then the ownership probes into the sparse lists BEFORE the leader (found = dropped).  The
  // sum runs leader first, then ascending list index: the same bits as every other mode.
  auto stageC = [&](uint32_t n) __attribute__((always_inline)) {
    tb(9u);
    const uint32_t base = q2n - n;
    q2n = base;
    if (p.debug & 128u) n_matches += n;  // COUNTERS
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, tf = 0, mx = 0;
    float rest = 0.0f;
    if (alive) {
      doc = L.q2_doc[base + lane];
      tf = L.q2_tf[base + lane];
      mx = L.q2_mx[base + lane];
      rest = L.q2_rest[base + lane];
    }
    const float norm = L.cache[mx >> 16];
    float s = bm25(w_lead, norm, tf);
    for (uint32_t mm = 1; mm < nt; ++mm) {
      uint32_t m = li + mm;
      if (m >= nt) m -= nt;
      const bool after = m > li;
      const bool is_dense = (dense_mask >> m) & 1u;
      if (!after && is_dense) continue;  // ownership by a dense list was settled in stage B
      const bool mine = (mx >> m) & 1u;
      if (is_dense) {
        if (__ballot(alive && mine)) {
          const TermRef tr = load_term(p.terms, sload(&Q->term[m]));
          const float w = L.wgt[m];
          if (alive && mine) {
            const uint2 wd = tr.dense[doc >> 5];
            const uint32_t pi = wd.y + (uint32_t)__popc(wd.x & ((1u << (doc & 31u)) - 1u));
            const uint4 r = tr.rec[pi >> 7];
            s = s + bm25(w, norm, block_tf_at(idx, tr, make_uint2(r.y, r.z), pi & 127u));
            rest -= w;
          }
        }
        continue;
      }
      // no bitmap: the exact (expensive) probe, only for candidates that can still make it
      if (prune && alive) {
        const float r0 = after ? (rest > 0.0f ? rest : 0.0f) : 0.0f;
        alive = sortable((s + r0) * 1.000002f + slack_abs) >= thr;
      }
      if (!__ballot(alive)) break;
      const TermRef tr = load_term(p.terms, sload(&Q->term[m]));
      bool cand = alive;
      uint32_t jb = 0;
      if (cand) {
        jb = seek_block(tr, doc);
        cand = jb < tr.n_blocks;
      }
      uint32_t unused;
      const uint32_t at = lookup_in_blocks<false>(idx, tr, jb, doc, cand, L.pay, lane, &unused);
      const bool found = cand && at != NOT_FOUND;
      if (after) {
        const float w = L.wgt[m];
        if (found) {
          const uint4 r = tr.rec[jb];
          s = s + bm25(w, norm, block_tf_at(idx, tr, make_uint2(r.y, r.z), at));
        }
        rest -= w;
      } else if (found) {
        alive = false;  // list m's tile scores this doc
      }
    }
    // the score is final: below the threshold it cannot enter the top-k (equal scores stay: ties
    // resolve by doc id in the collector)
    if (prune && alive) alive = sortable(s) >= thr;
    if (alive) alive = doc_is_alive(seg, doc);
    const uint64_t hit = __ballot(alive);
    if (hit) {
      if (!(p.debug & 480u)) n_matches += (uint32_t)__popcll(hit);  // COUNTERS
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
    te(9u);
  };
  // B (64 candidates of leader li): membership only.  One fieldnorm byte and one bitmap word per
  // dense list, all independent gathers; a doc held by a list before the leader belongs to that
  // list's tile; the others keep "leader score + the weights of the later lists that hold the
  // doc (bitmap) or may hold it (no bitmap)" as their bound (block_wand_union.rs:16-80 with the
  // pivot test made per doc) and most of them end here, before any tf is fetched.
  auto stageB_pure = [&](uint32_t n) __attribute__((always_inline)) {
    tb(8u);
    const uint32_t base = q1n - n;
    q1n = base;
    if (p.debug & 64u) n_matches += n;  // COUNTERS
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, tf = 0;
    if (alive) {
      doc = L.q1_doc[base + lane];
      tf = L.q1_tf[base + lane];
    }
    if (q1_is_pi) {  // swept tiles: the queue carries the posting index (bitmap rank), not the tf
      if (alive) {
        const uint4 r = lead.rec[tf >> 7];
        tf = block_tf_at(idx, lead, make_uint2(r.y, r.z), tf & 127u);
      }
    }
    uint32_t nid = 0, mask = 0;
    uint32_t dm = dense_mask & ~(1u << li);
    if (seg.docmat) {  // one gather: the fieldnorm id and the membership in every matrix list
      const uint64_t mw = alive ? seg.docmat[doc] : 0ull;
      nid = (uint32_t)mw & 0xFFu;
      uint32_t mm = mat_mask & ~(1u << li);
      dm &= ~mat_mask;
      while (mm) {
        const uint32_t m = (uint32_t)__builtin_ctz(mm);
        mm &= mm - 1u;
        mask |= ((uint32_t)(mw >> L.mshift[m]) & 1u) << m;
      }
    } else if (alive) {
      nid = fieldnorm_id(seg, doc);
    }
    const uint32_t word = doc >> 5, bit = doc & 31u;
    while (dm) {  // four lists per round: the gathers of a round are in flight together
      uint32_t ms[4] = {0u, 0u, 0u, 0u}, bits[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (dm) {
          ms[u] = (uint32_t)__builtin_ctz(dm);
          dm &= dm - 1u;
          const uint2 *dp = L.dptr[ms[u]];
          if (alive) bits[u] = dp[word].x;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) mask |= ((bits[u] >> bit) & 1u) << ms[u];
    }
    if (mask & ((1u << li) - 1u)) alive = false;  // an earlier list's tile scores this doc
    float rest = 0.0f;
    {
      const uint32_t may = mask | sparse_mask;
      for (uint32_t m = nt; m-- > li + 1u;) rest += ((may >> m) & 1u) ? L.wgt[m] : 0.0f;
    }
    if (prune && alive) {
      const float s = bm25(w_lead, L.cache[nid], tf);
      if (p.debug & 256u)  // COUNTERS: the bound without membership knowledge
        n_matches += (uint32_t)__popcll(__ballot(sortable((s + L.suffix[li + 1u]) * 1.000002f + slack_abs) >= thr));
      alive = sortable((s + rest) * 1.000002f + slack_abs) >= thr;
    }
    if (p.debug & 2048u) alive = false;  // ABLATION: no stage C
    const uint64_t mk = __ballot(alive);
    if (mk) {
      const uint32_t pos = q2n + mbcnt64(mk);
      wave_mem_fence();
      if (alive) {
        L.q2_doc[pos] = doc;
        L.q2_tf[pos] = tf;
        L.q2_mx[pos] = mask | (nid << 16);
        L.q2_rest[pos] = rest;
      }
      wave_mem_fence();
      q2n += (uint32_t)__popcll(mk);
    }
    te(8u);
  };
  // Boolean queries, stage B0 (64 candidates of leader li): ONE doc-matrix gather — fieldnorm id
  // and the membership in every list with a column — settles most candidates on registers, before
  // the leader is scored and before any list is probed: a doc held by an earlier list of the leader
  // set belongs to that list's tile; a MustNot term holds it; a Must clause whose terms all have
  // columns holds it in none of them (intersection.rs:120-179 / exclude.rs with the seeks replaced
  // by bit tests); or "leader score + weights of the other lists that hold or may hold it" cannot
  // reach the threshold.  Survivors (11..20 % on the bench shapes) take the exact walk of stage B.
  auto b0_test = [&](uint64_t mw, uint32_t tf, bool alive) __attribute__((always_inline)) -> bool {
    // ownership: lists of the leader set before li
    uint64_t own = 0;
    for (uint32_t mm = mat_mask & ((1u << (li < n_lead ? li : n_lead)) - 1u); mm; mm &= mm - 1u)
      own |= 1ull << L.mshift[__builtin_ctz(mm)];
    if (mw & (own | L.bmask[0])) alive = false;
    if (li < n_opt_lead && lead_test && !(mw & (L.bmask[1] | L.bmask[2]))) alive = false;
    for (uint32_t c = 0; c < n_cmask; ++c)
      if (!(mw & (L.cmask[c] | L.csig[c]))) alive = false;
    if (prune && alive) {
      float rest = 0.0f;
      for (uint32_t m = 0; m < nt; ++m) {
        if (m == li) continue;
        const float w = L.wgt[m];  // (0 for MustNot terms)
        const uint32_t sh = L.mshift[m];
        rest += sh < 64u ? (((mw >> sh) & 1ull) ? w : 0.0f) : w;
      }
      const float sl = bm25_bound(w_lead, L.cache[(uint32_t)mw & 0xFFu], tf);
      alive = sortable((sl + rest) * 1.000004f + slack_abs) >= thr;
    }
    return alive;
  };
  auto stageB0 = [&](uint32_t n) __attribute__((always_inline)) {
    const uint32_t base = q1n - n;
    q1n = base;
    if (p.debug & 64u) n_matches += n;  // COUNTERS
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, tf = 0;
    if (alive) {
      doc = L.q1_doc[base + lane];
      tf = L.q1_tf[base + lane];
    }
    const uint64_t mw = alive ? seg.docmat[doc] : 0ull;  // (column bits, signature bits, fieldnorm id)
    alive = b0_test(mw, tf, alive);
    const uint64_t mk = __ballot(alive);
    if (mk) {
      const uint32_t pos = q2n + mbcnt64(mk);
      wave_mem_fence();
      if (alive) {
        L.q2_doc[pos] = doc;
        L.q2_tf[pos] = tf;
      }
      wave_mem_fence();
      q2n += (uint32_t)__popcll(mk);
    }
  };
  const bool use_b0 = BOOL && PRUNE && seg.docmat != nullptr && !(p.debug & 32768u);
  auto step64 = [&]() __attribute__((always_inline)) {  // q1 holds >= 64 candidates
    if constexpr (BOOL) {
      if (use_b0 && prune) {
        stageB0(64u);
        while (q2n >= 64u) stageB(64u, true);
      } else {
        stageB(64u, false);
      }
    } else {
      stageB_pure(64u);
      while (q2n >= 64u) stageC(64u);
    }
  };
  auto drain = [&]() __attribute__((always_inline)) {
    if constexpr (BOOL) {
      if (use_b0 && prune) {
        while (q1n) {
          stageB0(q1n < 64u ? q1n : 64u);
          while (q2n >= 64u) stageB(64u, true);
        }
        while (q2n) stageB(q2n < 64u ? q2n : 64u, true);
      } else {
        while (q1n) stageB(q1n < 64u ? q1n : 64u, false);
      }
    } else {
      while (q1n) {
        stageB_pure(q1n < 64u ? q1n : 64u);
        while (q2n >= 64u) stageC(64u);
      }
      while (q2n) stageC(q2n < 64u ? q2n : 64u);
    }
  };

  setup_query();
  for (uint32_t t = t_begin; t < t_end; ++t) {
    while (t >= q_tile_end) {
      if (q_tile_end > q_tile_start && q_tile_end > t_begin) {
        drain();
        tb(3u);
        const uint32_t part = sload(&Q->part_start) + (chunk - sload(&Q->chunk_first));
        flush_partial<KPL>(tk, sload(&p.sinks->partials), part, lane);
        if (lane == 0 && n_q) atomicAdd(sload(&p.sinks->query_matches) + sload(sload(&p.sinks->out_index) + q), n_q);
        n_q = 0;
        te(3u);
      }
      ++q;
      setup_query();
    }
    // ---- which list leads this tile (tiles of a chunk come in order: advance, never search)
    tb(4u);
    if (dead) {  // lists li.. are non-essential for good (the threshold only rises): nothing
      t = (q_tile_end < t_end ? q_tile_end : t_end) - 1u;  // left for this query in this chunk
      te(4u);
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
      te(4u);
      drain();  // the queue belongs to the previous leader
      tb(4u);
      li = nli;
      lead = load_term(p.terms, sload(&Q->term[li]));
      w_lead = sload(&Q->weight[li]);
      if constexpr (!BOOL && PRUNE) {
        q1_is_pi = false;
        sw_after = dense_mask & ~((2u << li) - 1u);
        sw_before = dense_mask & ((1u << li) - 1u);
        sw_n = (uint32_t)__popc(sw_after);
        sw_lead = nullptr;
        // only where a 32-doc word holds several postings of the leader: below that the decode
        // path spends fewer instructions per candidate than the sweep spends per word
        const uint32_t sw_ratio = ((p.debug >> 24) & 63u) ? ((p.debug >> 24) & 63u) : TQ_U_SWEEP_RATIO;
        if (((dense_mask >> li) & 1u) && sw_n >= 1u && sw_n <= 5u && !(p.debug & 8192u) &&
            (uint64_t)lead.n_blocks * 128ull * sw_ratio >= seg.max_doc) {
          sw_lead = uni_ptr(L.dptr[li]);
          sw_ssum = 0.0f;
          uint32_t am = sw_after, b = 0;
          while (am) {
            const uint32_t m = (uint32_t)__builtin_ctz(am);
            am &= am - 1u;
            if (((uint32_t)lane >> b) & 1u) sw_ssum += L.wgt[m];
            ++b;
          }
          sw_sparse_after = 0.0f;
          uint32_t sm = sparse_mask & ~((2u << li) - 1u);
          while (sm) {
            const uint32_t m = (uint32_t)__builtin_ctz(sm);
            sm &= sm - 1u;
            sw_sparse_after += L.wgt[m];
          }
        }
      }
    }
    // threshold: on a new leader and every 8th tile.  The radix select over the slots (the bulk of
    // this kernel's scalar work when it ran at every refresh) only runs when the slots changed
    // since the wave last looked (checksum), and on the upper 16 bits only: any v with
    // |{slots >= v}| >= k is a valid bound, the low bits of the k-th largest cost 0.8 % of it.
    if (slots && (new_leader || (tl & 7u) == 0u)) {
      uint32_t sv[4] = {0u, 0u, 0u, 0u};
      sv[0] = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (n_slot_rows == 4u) {
#pragma unroll
        for (int r = 1; r < 4; ++r)
          sv[r] = __hip_atomic_load(slots + 64 * r + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      uint32_t sum = (sv[0] + sv[1]) + (sv[2] + sv[3]);
      sum += dpp_get<0x111, 0xF>(sum);
      sum += dpp_get<0x112, 0xF>(sum);
      sum += dpp_get<0x114, 0xF>(sum);
      sum += dpp_get<0x118, 0xF>(sum);
      sum += dpp_get<0x142, 0xA>(sum);
      sum += dpp_get<0x143, 0xC>(sum);
      sum = (uint32_t)__builtin_amdgcn_readlane((int)sum, 63);
      if (sum != slots_sum) {
        slots_sum = sum;
        const uint32_t g = n_slot_rows == 4u ? kth_largest_hi16<4>(sv, tk.k) : kth_largest_hi16<1>(sv, tk.k);
        if (g > thr_g) thr_g = g;
      }
      if (thr_g > thr) thr = thr_g;
    }
    // non-essential by now: every doc first seen in list li scores at most the weights of li..
    te(4u);
    if (prune && sortable(L.suffix[li] * 1.000001f) < thr) {
      drain();
      dead = true;
      continue;
    }

    if (p.debug & 4096u) continue;  // ABLATION: tile bookkeeping only
    tb(5u);
    // ---- pre-filter: lane <-> leader block
    const uint32_t i_base = (tl - sload(&Q->lead_tile_start[li])) * tile_blocks;
    const uint32_t i_mine = i_base + (uint32_t)lane;
    bool surv = (uint32_t)lane < tile_blocks && i_mine < lead.n_blocks;
    uint4 rec_mine = make_uint4(0u, 0u, 0u, 0u);
    uint32_t prev_mine = 0, tfmin_mine = 1u;
    float ub_mine = 0.0f;  // block-max score of the lane's leader block (valid lanes)
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
          ub = block_max_score(rec_mine.y, w_lead, L.cache, lead.has_freq, p.bound_slack);
          ub_mine = ub;
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
                  const float b2 = block_max_score(r.y, w, L.cache, tr.has_freq, p.bound_slack);
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
    te(5u);
    tb(6u);
    // ---- bitmap sweep (pure unions, pruned, the leader has a bitmap and 1..5 of the lists after
    // it do): the membership test of stage B moved in front of the decode and made word-parallel.
    // A doc of this tile can only reach the threshold if the lists after the leader that hold it
    // carry enough weight: "block-max of the tile + the weights of a subset S of those lists >=
    // threshold" is a monotone function of S, evaluated once per tile for all <= 32 subsets (one
    // per lane); its minimal passing subsets S_1.. turn the per-doc test into
    //   survivors = leader bits & ~(bits of the dense lists before) & OR_i AND_{m in S_i} bits_m
    // over coalesced 32-doc bitmap words — no leader block is decoded and no candidate gathers
    // anything until it has passed.  Survivors enter queue 1 with their posting index (rank
    // directory), stage B fetches the tf.  block_wand_union.rs:16-80 with the pivot test made
    // per 32 docs.  When the leader alone can reach the threshold the tile takes the decode path.
    if constexpr (!BOOL && PRUNE) {
      bool swept = false;
      if (sw_lead && todo && thr != 0u) {
        const uint32_t lo_l = (uint32_t)__builtin_ctzll(todo), hi_l = 63u - (uint32_t)__builtin_clzll(todo);
        uint32_t ubv = ((uint32_t)lane >= lo_l && (uint32_t)lane <= hi_l) ? __float_as_uint(ub_mine) : 0u;
        {  // wave max (scores are >= 0: their bits order like integers)
          uint32_t o;
          o = dpp_get<0x111, 0xF>(ubv); ubv = o > ubv ? o : ubv;
          o = dpp_get<0x112, 0xF>(ubv); ubv = o > ubv ? o : ubv;
          o = dpp_get<0x114, 0xF>(ubv); ubv = o > ubv ? o : ubv;
          o = dpp_get<0x118, 0xF>(ubv); ubv = o > ubv ? o : ubv;
          o = dpp_get<0x142, 0xA>(ubv); ubv = o > ubv ? o : ubv;
          o = dpp_get<0x143, 0xC>(ubv); ubv = o > ubv ? o : ubv;
          ubv = (uint32_t)__builtin_amdgcn_readlane((int)ubv, 63);
        }
        const float base_ub = __uint_as_float(ubv) + sw_sparse_after;
        const bool pj = (uint32_t)lane < (1u << sw_n) &&
                        sortable((base_ub + sw_ssum) * 1.000002f + slack_abs) >= thr;
        const uint32_t tt = (uint32_t)__ballot(pj);
        if (!(tt & 1u)) {  // the leader alone cannot make it: sweep
          swept = true;
          bool minimal = pj;
          for (uint32_t b = 0; b < sw_n; ++b)
            if ((((uint32_t)lane >> b) & 1u) && ((tt >> ((uint32_t)lane & ~(1u << b))) & 1u)) minimal = false;
          uint32_t mins = (uint32_t)__ballot(minimal);
          if (mins) {
            if (!q1_is_pi) {
              drain();
              q1_is_pi = true;
            }
            const uint32_t prev_lo = (uint32_t)__builtin_amdgcn_readlane((int)prev_mine, (int)lo_l);
            const uint32_t first = (i_base + lo_l) ? prev_lo + 1u : 0u;
            const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)rec_mine.x, (int)hi_l);
            const uint32_t w_lo = first >> 5, w_hi = last >> 5;
            uint32_t used = 0;  // lists after the leader that some minimal subset needs
            for (uint32_t mm = mins; mm; mm &= mm - 1u) used |= (uint32_t)__builtin_ctz(mm);
            // pointers of the lists after the leader, by subset bit
            const uint2 *ap[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
            {
              uint32_t am = sw_after;
#pragma unroll
              for (int b = 0; b < 5; ++b) {
                if (am) {
                  ap[b] = uni_ptr(L.dptr[__builtin_ctz(am)]);
                  am &= am - 1u;
                }
              }
            }
            for (uint32_t wb = w_lo; wb <= w_hi; wb += 64u) {
              const uint32_t w = wb + (uint32_t)lane;
              const bool on = w <= w_hi;
              uint2 wl = make_uint2(0u, 0u);
              if (on) wl = sw_lead[w];
              uint32_t before = 0;
              for (uint32_t bm = sw_before; bm; bm &= bm - 1u) {
                const uint2 *dp = uni_ptr(L.dptr[__builtin_ctz(bm)]);
                if (on) before |= dp[w].x;
              }
              uint32_t W[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
              for (int b = 0; b < 5; ++b)
                if (((used >> b) & 1u) && on) W[b] = ap[b][w].x;
              uint32_t cand = wl.x & ~before;
              if (w == w_lo) cand &= 0xFFFFFFFFu << (first & 31u);
              if (w == w_hi) cand &= 0xFFFFFFFFu >> (31u - (last & 31u));
              uint32_t pass = 0;
              for (uint32_t mm = mins; mm; mm &= mm - 1u) {
                const uint32_t j = (uint32_t)__builtin_ctz(mm);
                uint32_t acc = 0xFFFFFFFFu;
#pragma unroll
                for (int b = 0; b < 5; ++b)
                  if ((j >> b) & 1u) acc &= W[b];
                pass |= acc;
              }
              uint32_t sv = cand & pass;
              while (__ballot(sv != 0u)) {
                const bool has = sv != 0u;
                const uint32_t bit = has ? (uint32_t)__builtin_ctz(sv) : 0u;
                sv &= sv - 1u;
                const uint64_t mk = __ballot(has);
                const uint32_t pos = q1n + mbcnt64(mk);
                wave_mem_fence();
                if (has) {
                  L.q1_doc[pos] = (w << 5) + bit;
                  L.q1_tf[pos] = wl.y + (uint32_t)__popc(wl.x & ((1u << bit) - 1u));
                }
                wave_mem_fence();
                q1n += (uint32_t)__popcll(mk);
                while (q1n >= 64u) step64();
              }
            }
          }
        }
      }
      te(6u);
      if (swept) continue;
      if (q1_is_pi) {  // back on the decode path (never happens while the threshold only rises
        drain();       // for one leader, but a fresh chunk starts below its predecessor's)
        q1_is_pi = false;
      }
    }
    tb(7u);
    while (todo) {
      const uint32_t b = (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      const uint2 mo_l = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)rec_mine.y, (int)b),
                                    (uint32_t)__builtin_amdgcn_readlane((int)rec_mine.z, (int)b));
      const uint32_t prev_l = (uint32_t)__builtin_amdgcn_readlane((int)prev_mine, (int)b);
      uint32_t c0, c1, t0, t1f;
      bool alive0 = true, alive1 = true;
      // the doc and tf streams are requested together (the prefix sum still waits for the tf test)
      const bool packed = mo_l.x != META_TAIL;
      uint32_t x0 = 0, x1 = 0;
      if (packed) unpack2(idx + lead.payload_base + mo_l.y, mo_l.x & 31u, lane, x0, x1);
      decode_tfs(idx, lead, mo_l, lane, t0, t1f);  // tail padding reads as tf 0
      if (prune) {
        const uint32_t tfmin = (uint32_t)__builtin_amdgcn_readlane((int)tfmin_mine, (int)b);
        alive0 = t0 >= tfmin;
        alive1 = t1f >= tfmin;
        if (!(__ballot(alive0) | __ballot(alive1))) continue;
      }
      if (packed)
        finish_docs<USE_DPP>(x0, x1, (mo_l.x >> 6) & 1u, prev_l, lane, c0, c1);
      else
        decode_docs<USE_DPP>(idx, lead, mo_l, prev_l, lane, c0, c1);
      alive0 = alive0 && c0 != TQD_TERMINATED;
      alive1 = alive1 && c1 != TQD_TERMINATED;
      const uint64_t m0 = __ballot(alive0), m1 = __ballot(alive1);
      if (!(m0 | m1)) continue;
      if (p.debug & 1024u) continue;  // ABLATION: no stage B / C
      if constexpr (BOOL) {
        if (use_b0 && prune) {
          // the doc-matrix pre-stage on the block's 128 docs at once: both gathers in flight with
          // all lanes live, no queue in between (one round trip per block, not one per 64 candidates)
          if (p.debug & 64u) n_matches += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);  // COUNTERS
          const uint64_t w0 = alive0 ? seg.docmat[c0] : 0ull;
          const uint64_t w1 = alive1 ? seg.docmat[c1] : 0ull;
          alive0 = b0_test(w0, t0, alive0);
          alive1 = b0_test(w1, t1f, alive1);
          const uint64_t k0 = __ballot(alive0), k1 = __ballot(alive1);
          if (!(k0 | k1)) continue;
          const uint32_t kn0 = (uint32_t)__popcll(k0);
          const uint32_t p0 = q2n + mbcnt64(k0);
          const uint32_t p1 = q2n + kn0 + mbcnt64(k1);
          wave_mem_fence();
          if (alive0) {
            L.q2_doc[p0] = c0;
            L.q2_tf[p0] = t0;
          }
          if (alive1) {
            L.q2_doc[p1] = c1;
            L.q2_tf[p1] = t1f;
          }
          wave_mem_fence();
          q2n += kn0 + (uint32_t)__popcll(k1);
          while (q2n >= 64u) stageB(64u, true);
          continue;
        }
      }
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
      while (q1n >= 64u) step64();
    }
    te(7u);
  }
  if (q_tile_end > q_tile_start) {
    drain();
    tb(3u);
    const uint32_t part = sload(&Q->part_start) + (chunk - sload(&Q->chunk_first));
    flush_partial<KPL>(tk, sload(&p.sinks->partials), part, lane);
        if (lane == 0 && n_q) atomicAdd(sload(&p.sinks->query_matches) + sload(sload(&p.sinks->out_index) + q), n_q);
        n_q = 0;
    te(3u);
  }
  te(1u);
  if (tphase) n_matches = (uint32_t)(tacc >> 4);
  if (lane == 0 && n_matches) atomicAdd(sload(&p.sinks->match_counter), (unsigned long long)n_matches);
}

// k <= 128 (KPL <= 2) instantiations are compiled for 6 waves/SIMD (<= 84 registers) instead of 4
template <int KPL, bool PRUNE, bool BOOL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(96))) void union_kernel(TqkScanParams p) {
  union_body<KPL, PRUNE, BOOL>(p);
}
template <int KPL, bool PRUNE, bool BOOL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(96), amdgpu_waves_per_eu(TQ_U_WAVES, 8))) void
union_kernel_small(TqkScanParams p) {
  union_body<KPL, PRUNE, BOOL>(p);
}

}  // namespace

// =================================================================== launch wrappers
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
    // Pure unions run the two-stage form (membership, then exact scores) at every k: with the
    // bitmap sweep in front of it, it beats the single-stage walk of the boolean instantiation also
    // for k <= 16 (4.03 vs 4.55 ms per 1000 5-term queries at k = 10; TQ_DEBUG bit 14 = walk).
    if (p.boolean || (p.small_k && !p.exhaustive && (p.debug & 16384u))) {
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
