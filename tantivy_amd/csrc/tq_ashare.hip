// tq_ashare.hip — the AND queries of a batch, driven LEADER BY LEADER instead of query by query.
// Shared device helpers: tq_common.hpp.
//
// block_wand_intersection (src/query/boolean_query/block_wand_intersection.rs:19-179) as and_kernel
// (tq_and.hip) runs it per query: lists by doc freq ascending, the rarest one (the leader) is walked
// block by block, a block whose block-max bound cannot reach the threshold is skipped (:81-85), the
// docs of the others are tested for membership in the other lists, scored leader first, then in
// ascending doc freq (:144-165), and offered to the collector.
//
// What changes is the loop order.  In a batch many queries lead with the same list (the 10 000
// 2-term queries of the headline batch have 255 distinct leaders: every leader block was decoded
// ~80 times per batch).  Here a TASK is a run of blocks of ONE leader list for a GROUP of up to 32
// LEADS (queries that lead with it):
//   pre-filter, lane <-> block: the block's record, its block-max tf/(tf+norm), and for every live
//      lead "block-max score + weights of the other lists >= threshold" -> a lead mask per block;
//   A  per surviving block: ONE decode (unpack + prefix sum), ONE 8-byte doc-matrix gather per doc
//      (fieldnorm id, membership in the segment's column lists, the signature bits of the others),
//      tf/(tf+norm) of the leader for both docs of a lane;
//   F  per (block, lead in the block's mask): the lead's constants are LDS reads at a uniform
//      address; "every other list of the query holds the doc" is two ORs, an AND and a compare per
//      doc against the lead's column / signature mask (kept across consecutive leads with the
//      same mask: the leads of a group are sorted by it), "leader score + weights of the others
//      >= threshold" one float compare; survivors -> LDS queue, tagged with the lead slot;
//   C  64 survivors, every lane with its own query: list 1 through its bitmap word (exact
//      membership for signature-"maybe" lists, the posting index) -> the tf byte at that index,
//      further lists of a 3+ term query the same way, the exact BM25 sum in the reference's order,
//      then the collector.
// Twins: a batch repeats queries (the headline batch holds 3 877 distinct ones among its 10 000), and
// the planner puts identical queries (same lists, weights, k) next to each other in a group.  A head
// and its twins form a FAMILY that is evaluated once: blocks are tested, docs scored and collected for
// the head; at the end of the task the head's list (<= k entries) is appended to the result list of
// EVERY member, and the threshold to every member's word.  Identical queries also share one row of
// threshold slots across groups (a slot is hash(doc): the same doc lands in the same slot whichever
// group scored it).  Every query keeps its own result list and its own merge.
// Scores are and_kernel's bits (same operations in the same order); thresholds only ever hold scores
// of k distinct real matches, candidates equal to the threshold are kept, so the top-k equals the
// exhaustive run's, ties by doc id as TopNHeap resolves them (sort_by_score.rs:86-161).
//
// Collector and thresholds: tq_ushare.hip's — per-wave staging lists in global memory cut back by a
// radix select, appended to the queries' result lists at the end of a task, merge_lists_kernel;
// hashed atomic-max slots + thr_val[query].
#include "tq_common.hpp"

#ifndef TQ_BS_SGPR
#define TQ_BS_SGPR 96
#endif
#ifndef TQ_BS_WAVES
#define TQ_BS_WAVES 5  // (boolean leads: 96 registers; 6 waves of 80 spilled 60 of them and ran 18 % slower)
#endif
#ifndef TQ_AS_WAVES
// (intersections: 8 waves of 64 registers until round 6 — by then the scoring stage had grown (class matrix, range
// directories) and spilled 150 bytes per lane; 5 waves of 96 registers: and2 0.93 -> 0.83 ms, and2_distinct 1.43 ->
// 1.25, 4 096 terms 1.18 -> 0.90; fewer waves in flight also see higher thresholds (a fifth fewer docs scored).
// 7 / 6 / 4 waves: 0.89 / 0.88 / 0.92 ms)
#define TQ_AS_WAVES 5
#endif
#ifndef TQ_AS_TIMERS
#define TQ_AS_TIMERS 0  // region timers (tools/probe_ashare_regions.sh builds a variant with them)
#endif
#ifndef TQ_AS_PREFETCH
#define TQ_AS_PREFETCH 0  // 1: the next wanted block's payload is fetched into LDS (global_load_lds) under the current block's work — measured: no gain (the chain is the doc-matrix gather), 1 KB of LDS
#endif

namespace {

constexpr uint32_t AS_GROUP = TQD_AS_GROUP;
// TQ_DEBUG / option "debug" bits that turn the launch's match counter into a work counter (one per run): 32
// (block, family) pairs, 64 stage-C candidates, 256 blocks decoded, 512 stage-C steps, 4096 / 8192 / 16384
// compactions / flush selects / flushes; bytes the lanes consume (bench.py's useful_bytes): 0x100000 payload +
// record bytes of the decoded blocks, 0x200000 fieldnorm bytes, 0x400000 doc-matrix words gathered, 0x800000
// range-maxima bytes.  None of them changes a result.
constexpr uint32_t AS_COUNTER_BITS = 0x7FE0u | 0xF00000u;

template <bool BOOLQ>
struct AShareLds {  // per wavefront: 4868 bytes (32 wavefronts per CU fit the 160 KB); boolean leads: + 1 KB
  float cache[256];                            // Bm25Weight.cache of the task's queries
  uint32_t q_doc[127], q_tf[127], q_tag[127];  // survivors: doc, leader tf, lead slot | fieldnorm id << 8
  TqdALeadLds lead[AS_GROUP];                  // the leads of the task (what the scoring stage needs)
  uint32_t lthr[AS_GROUP];                     // the lead's threshold (sortable score bits); only ever rises
  uint32_t lk[AS_GROUP];                       // k of its query (bits 0..7) | its row of threshold slots << 8
  uint32_t cnt[AS_GROUP];                      // bits 0..15: entries in the slot's staging list; 16..31: docs scored
  uint32_t flen[AS_GROUP];                     // per family head: leads in the family (itself + its twins)
  float bw[BOOLQ ? AS_GROUP * TQD_AS_MAX_TERMS : 1];  // boolean leads: what every list of the query can add to the lead's docs
#if TQ_AS_PREFETCH
  uint32_t pay[260];                           // the NEXT wanted block's bitpacked payload (<= 1008 B), landed by LDS-DMA
#endif
};

// k-th largest of the n (<= 64 R) keys held R per lane (0 = empty); n >= k
template <int R>
__device__ __forceinline__ uint64_t as_kth_largest_key(const uint64_t (&v)[R], uint32_t k) {
  uint64_t ans = 0;
  for (int bit = 63; bit >= 0; --bit) {
    const uint64_t trial = ans | (1ull << bit);
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) c += (uint32_t)__popcll(__ballot(v[r] >= trial));
    if (c >= k) ans = trial;
  }
  return ans;
}

// tf of posting `pi` of list `handle` when its tf byte is saturated (>= 255): block record -> packed tf
__device__ __forceinline__ uint32_t as_exact_tf(const uint8_t *idx, const TqdTerm *terms, uint32_t handle,
                                                uint32_t pi) {
  const TqdTermHead *h = terms + handle;
  TermRef tr{};
  tr.rec = h->rec;
  tr.tail_tfs = h->tail_tfs;
  tr.payload_base = h->payload_base;
  tr.has_freq = h->has_freq & 1u;
  const uint4 r = tr.rec[pi >> 7];
  return block_tf_at(idx, tr, make_uint2(r.y, r.z), pi & 127u);
}

// BOOLQ: the leads are (boolean query, leading list) pairs (TQ_MODE_BOOL: Must / Should / MustNot clauses of
// terms and unions of terms, BooleanWeight::complex_scorer, boolean_weight.rs:236-431) — the doc set and the
// score walk of union_kernel<.., BOOL = true> (tq_union.hip), with every list but the leader reached through
// its bitmap.
// RD (boolean leads): some list of the launch is probed through its range directory (TqkAShareParams::rdir_lists; the
// instantiation without that code scores the bench's boolean batch 4 % faster)
template <int KPL, bool BOOLQ, bool RD = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(TQ_BS_SGPR), amdgpu_waves_per_eu(BOOLQ ? TQ_BS_WAVES : TQ_AS_WAVES, 8))) void
ashare_kernel(TqkAShareParams p) {
  constexpr bool USE_DPP = true;
  constexpr int R = KPL + 1;                    // staging registers per lane
  constexpr uint32_t CAPL = (uint32_t)R * 64u;  // staging entries per lead slot
  __shared__ AShareLds<BOOLQ> L;
  const int lane = (int)__lane_id();
  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;
  uint64_t *const my_stage = p.stage + (size_t)blockIdx.x * (size_t)(AS_GROUP * CAPL);
  const uint8_t *const tbase = p.table_base;
  const uint32_t bmode = BOOLQ ? 0u : p.bound_mode;  // TQ_AS_BOUND bits (tq_search.cpp)
  uint32_t cache_loaded = 0xFFFFFFFFu;
  uint32_t qn = 0;        // survivor queue fill
  uint32_t n_scored = 0;  // docs scored by this wave (all tasks)
  uint32_t n_leads = 0;
  // PROFILING (-DTQ_AS_TIMERS=1, TQ_DEBUG bits 16..19 = region): wave cycles spent inside ONE region per
  // run, summed into the match counter (>> 6).  1 everything, 2 task fetch + setup, 3 pre-filter, 4 stage A
  // (payload + unpack + prefix sum), 5 stage A: doc-matrix gather + tf/(tf+norm), 6 stage F, 7 stage C,
  // 8 flush, 9 threshold refresh
  const uint32_t tphase = TQ_AS_TIMERS ? (p.debug >> 16) & 15u : 0u;
  uint64_t tacc = 0, tlast = 0;
  auto tb = [&](uint32_t ph) __attribute__((always_inline)) {
    if (TQ_AS_TIMERS && tphase == ph) tlast = __builtin_readcyclecounter();
  };
  auto te = [&](uint32_t ph) __attribute__((always_inline)) {
    if (TQ_AS_TIMERS && tphase == ph) tacc += __builtin_readcyclecounter() - tlast;
  };
  tb(1u);

  // a staging list is cut back to its k best; returns the k-th key (the list held n > k entries)
  auto compact_slot = [&](uint32_t g, uint32_t n, uint32_t k) __attribute__((always_inline)) -> uint64_t {
    uint64_t *sl = my_stage + (size_t)g * CAPL;
    uint64_t v[R];
    wave_mem_fence();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t i = (uint32_t)r * 64u + (uint32_t)lane;
      v[r] = i < n ? sl[i] : 0ull;
    }
    const uint64_t kth = as_kth_largest_key<R>(v, k);
    uint32_t base = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool keep = v[r] >= kth && v[r] != 0ull;
      const uint64_t m = __ballot(keep);
      if (keep) sl[base + mbcnt64(m)] = v[r];
      base += (uint32_t)__popcll(m);
    }
    wave_mem_fence();
    return kth;
  };

  // ---- stage C: 64 survivors, every lane with its own query
  auto stageC = [&](uint32_t n) __attribute__((always_inline)) {
    te(6u);
    tb(7u);
    const uint32_t base = qn - n;
    qn = base;
    if (p.debug & 64u) n_scored += n;  // COUNTERS
    if (p.debug & 512u) ++n_scored;
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, tf = 0, tag = 0;
    if (alive) {
      doc = L.q_doc[base + lane];
      tf = L.q_tf[base + lane];
      tag = L.q_tag[base + lane];
    }
    const uint32_t g = tag & 31u;  // the family's head
    const float norm = L.cache[(tag >> 8) & 0xFFu];
    const TqdALeadLds ld = L.lead[g];
    // (the family's threshold, k and threshold row are read from LDS where they are first needed: fewer registers live
    // across the probes of the other lists)
    const uint32_t q = ld.query;
    const uint32_t nt = ld.info & 31u;
    float s = bm25(ld.w, norm, tf);
    if constexpr (BOOLQ) {
      // The score walk of union_kernel<.., BOOL = true> (tq_union.hip, "leader set" comment): the lists after
      // the leader in the query's order, wrapping around to the ones before it; a doc found in a lead-set
      // list before the leader belongs to that list's lead; Must clauses (unions of terms) are summed as
      // Intersection::score does (left + right + sum(others), intersection.rs:325-329), MustNot lists
      // exclude (exclude.rs), optional Should lists add (RequiredOptionalScorer::score = req + opt,
      // reqopt_scorer.rs:85-98).  Same operations in the same order: the same bits.
      const TqdQuery *Q = p.queries + q;
      const uint32_t li = (ld.info >> 16) & 15u;
      const uint32_t pb = tag >> 16;  // bit m clear: list m does not hold the doc (its doc-matrix column says so)
      uint32_t roles = 0, clause_end = 0, n_lead = 0, n_opt_lead = 0, min_should = 0;
      if (alive) {
        roles = Q->roles;
        clause_end = Q->clause_end;
        n_lead = Q->n_lead;
        n_opt_lead = Q->n_opt_lead;
        min_should = Q->min_should;
      }
      float opt = 0.0f, oth = 0.0f, csum = 0.0f;
      bool cfound = false;
      bool lcfound = li >= n_opt_lead;  // the lead Must clause holds the doc
      if (li < n_opt_lead) {            // an optional list leads: its score is optional
        opt = s;
        s = 0.0f;
      }
      uint32_t clause = 1u;
      uint32_t n_should = ((roles >> (2u * li)) & 3u) == TQD_ROLE_SHOULD ? 1u : 0u;
      for (uint32_t mm = 1; mm < TQD_AS_MAX_TERMS; ++mm) {
        const bool on = alive && mm < nt;
        if (!__ballot(on)) break;
        if (on) {
          uint32_t m = li + mm;
          if (m >= nt) m -= nt;
          const uint32_t role = (roles >> (2u * m)) & 3u;
          bool found = false;
          float sc = 0.0f;
          if ((pb >> m) & 1u) {  // bitmap word (exact membership, the posting index) -> tf byte
            const uint2 bl = p.qlists[(size_t)q * TQD_AS_MAX_TERMS + m];
            if (RD && (bl.x & 31u)) {  // (a list probed through its range directory: directory | shift, entries — rdir_lookup)
              uint32_t tfm = 0, pm = 0;
              found = rdir_lookup(reinterpret_cast<const uint32_t *>(tbase + ((uint64_t)(bl.x & ~31u) << 3)),
                                  reinterpret_cast<const uint32_t *>(tbase + ((uint64_t)bl.y << 3)), bl.x & 31u, doc, true, tfm, pm);
              if (found && role != TQD_ROLE_MUST_NOT && !(m < li && m < n_lead)) {
                if (tfm == 0xFFFFu) tfm = as_exact_tf(idx, p.terms, Q->term[m], pm);
                sc = bm25(L.bw[g * TQD_AS_MAX_TERMS + m], norm, tfm);
              }
            } else {
              const uint2 wd = reinterpret_cast<const uint2 *>(tbase + ((uint64_t)bl.x << 3))[doc >> 5];
              const uint32_t bit = doc & 31u;
              found = (wd.x >> bit) & 1u;
              if (found && role != TQD_ROLE_MUST_NOT && !(m < li && m < n_lead)) {
                const uint32_t pm = wd.y + (uint32_t)__popc(wd.x & ((1u << bit) - 1u));
                uint32_t tfm = (tbase + ((uint64_t)bl.y << 3))[pm];
                if (tfm == 255u) tfm = as_exact_tf(idx, p.terms, Q->term[m], pm);
                sc = bm25(L.bw[g * TQD_AS_MAX_TERMS + m], norm, tfm);
              }
            }
          }
          if (role == TQD_ROLE_MUST_NOT) {
            if (found) alive = false;
          } else if (m < n_lead) {
            if (found) {
              if (m < li) {
                alive = false;  // this doc is scored by list m's lead
              } else if (m < n_opt_lead) {
                opt = opt + sc;
                ++n_should;
              } else {
                s = s + sc;
                lcfound = true;
                if (role == TQD_ROLE_SHOULD) ++n_should;
              }
            }
            if (m + 1u == n_lead && !lcfound) alive = false;  // not in the lead Must clause
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
      }
      s = (s + oth) + opt;
      if (n_should < min_should) alive = false;
    } else {
    // leader first, then ascending doc freq (block_wand_intersection.rs:144-165)
    // list 1: bitmap word (exact membership, the posting index) -> tf byte
    // (round 6: a list with tf classes in the segment's class matrix — ONE 8-byte gather says "absent", "tf 1", "tf 2"
    // or "three or more: read the tf byte"; 91 % of the bench's postings need nothing else)
    // (round 6: a sparse list 1 with a range directory — rdir_lookup, tq_common.hpp: two directory slots, then the
    // range's few entries — says "absent" or the tf and the posting's index)
    uint32_t pi = 0, tf1 = 0;
    bool chain = alive;
    uint32_t esc = 255u;  // the tf value that means "that or more: read the packed value"
    if (__ballot(alive && (ld.info & TQD_AL_RDIR))) {
      const bool ron = alive && (ld.info & TQD_AL_RDIR);
      const bool found = rdir_lookup(reinterpret_cast<const uint32_t *>(tbase + ((uint64_t)ld.dense_off << 3)),
                                     reinterpret_cast<const uint32_t *>(tbase + ((uint64_t)ld.tf8_off << 3)), (ld.info >> 16) & 31u,
                                     doc, ron, tf1, pi);
      if (ron) {
        alive = found;
        chain = false;
        esc = 0xFFFFu;
      }
    }
    {
      const uint32_t cslot1 = (ld.info >> 10) & 0x3Fu;
      if (__ballot(alive && cslot1)) {
        uint64_t cw = 0;
        if (alive && cslot1) cw = seg.doccls[doc];
        const uint32_t c = (uint32_t)(cw >> (2u * (cslot1 - 1u))) & 3u;
        if (alive && cslot1) {
          if (c == 0u) alive = false;  // (exact: the class matrix holds every posting of the list)
          tf1 = c;
          chain = c == 3u;
        }
      }
    }
    if (__ballot(chain)) {
      uint2 wd = make_uint2(0u, 0u);
      if (chain) wd = reinterpret_cast<const uint2 *>(tbase + ((uint64_t)ld.dense_off << 3))[doc >> 5];
      const uint32_t bit = doc & 31u;
      if (chain) alive = alive && ((wd.x >> bit) & 1u);
      if (chain) pi = wd.y + (uint32_t)__popc(wd.x & ((1u << bit) - 1u));
      if (chain && alive) tf1 = (tbase + ((uint64_t)ld.tf8_off << 3))[pi];
    }
    const bool more = alive && nt > 2u;
    const uint64_t more_m = __ballot(more);
    float w1 = ld.rest;
    if (more_m) {
      if (more) w1 = p.queries[q].weight[1];
    }
    if (__ballot(alive && tf1 == esc)) {  // tf >= 255 (65535): block record -> packed tf
      if (alive && tf1 == esc) tf1 = as_exact_tf(idx, p.terms, p.queries[q].term[1], pi);
    }
    if (alive) s = s + bm25(w1, norm, tf1);
    if (more_m) {  // lists 2.. of a 3+ term query: term table -> bitmap word -> tf byte, one list at a time
      float rest = ld.rest - w1;
      for (uint32_t m = 2; m < TQD_AS_MAX_TERMS; ++m) {
        bool on = alive && m < nt;
        if (on) {  // what the lists m.. can still add
          const float r0 = rest > 0.0f ? rest : 0.0f;
          if (!(sortable((s + r0) * 1.000002f + ld.rest * 4.0e-6f) >= L.lthr[g])) {
            alive = false;
            on = false;
          }
        }
        if (!__ballot(on)) break;  // (lanes with more lists than m are among the lanes with more than m - 1)
        if (on) {
          const uint32_t h = p.queries[q].term[m];
          const float wm = p.queries[q].weight[m];
          const TqdTerm *T = p.terms + h;
          const uint2 *dn = T->dense;
          const uint8_t *t8 = T->tf8;
          const uint2 wd = dn[doc >> 5];
          const uint32_t bit = doc & 31u;
          if ((wd.x >> bit) & 1u) {
            const uint32_t pm = wd.y + (uint32_t)__popc(wd.x & ((1u << bit) - 1u));
            uint32_t tfm = t8[pm];
            if (tfm == 255u) tfm = as_exact_tf(idx, p.terms, h, pm);
            s = s + bm25(wm, norm, tfm);
            rest -= wm;
          } else {
            alive = false;
          }
        }
      }
    }
    }
    // the score is final: below the threshold it cannot enter the top-k (equal scores stay: ties
    // resolve by doc id in the collector)
    if (alive) alive = sortable(s) >= L.lthr[g];
    if (alive) alive = doc_is_alive(seg, doc);
    const uint64_t hit = __ballot(alive);
    if (!hit) {
      te(7u);
      tb(6u);
      return;
    }
    if (!(p.debug & AS_COUNTER_BITS)) n_scored += (uint32_t)__popcll(hit);  // COUNTERS (TQ_DEBUG): 32 (block, family) pairs,
    const uint64_t key = alive ? make_key(s, doc) : 0ull;           // 64 stage-C candidates, 256 blocks decoded
    const uint32_t sb = (uint32_t)(key >> 32);
    if (alive) {
      // the query's hashed score slots: fire and forget (the k-th largest slot is selected once per
      // (task, family) at the end of the task, not once per change)
      const uint32_t kr = L.lk[g];  // k | threshold row << 8
      const uint32_t hsh = (doc * 0x9E3779B1u) >> ((kr & 0xFFu) <= 16u ? 26 : 24);
      (void)atomicMax(p.thr_slots + (size_t)(kr >> 8) * TQD_THR_SLOTS + hsh, sb);
      const uint32_t pos = atomicAdd(&L.cnt[g], 0x10001u) & 0xFFFFu;  // (a list never overflows: see the cut below)
      my_stage[(size_t)g * CAPL + pos] = key;
    }
    // staging lists that could overflow with the next batch are cut back to their k best now
    wave_mem_fence();
    const uint32_t cn = (uint32_t)lane < AS_GROUP ? L.cnt[lane] & 0xFFFFu : 0u;
    uint64_t full = __ballot(cn > CAPL - 64u);
    while (full) {
      const uint32_t gs = (uint32_t)__builtin_ctzll(full);
      full &= full - 1ull;
      const uint32_t ns = (uint32_t)__builtin_amdgcn_readlane((int)cn, (int)gs);
      const uint32_t ks = uni(L.lk[gs]) & 0xFFu;
      if (p.debug & 4096u) ++n_scored;  // COUNTERS
      const uint64_t kth = compact_slot(gs, ns, ks);
      const uint32_t t = (uint32_t)(kth >> 32);
      if ((uint32_t)lane == gs) {
        L.cnt[gs] = (L.cnt[gs] & 0xFFFF0000u) | ks;
        if (t > L.lthr[gs]) L.lthr[gs] = t;
        atomicMax(p.thr_val + L.lead[gs].query, t);  // k distinct docs of this query score >= t
      }
    }
    wave_mem_fence();
    te(7u);
    tb(6u);
  };

  // Task queues (experiment, TQ_AS_QUEUES; default ONE queue): the launch's tasks are in doc-slice order;
  // queue x = the x-th of n_queues equal shares of them = one contiguous part of the doc-id space, a
  // workgroup's home queue = the XCD it runs on, a workgroup whose queue is empty moves on to the next.
  // Meant to keep an XCD's L2 on one part of the index; measured 11-19 % slower than one queue.
  const uint32_t n_launch = p.n_tasks - p.task_begin;
  const uint32_t nq = p.n_queues ? p.n_queues : 1u;
  // (XCC_ID: hardware register 20, bits 0..3 — the XCD this wavefront runs on)
  const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
  const uint32_t home = xcc % nq;
  uint32_t q_tried = 0;
  for (;;) {
    tb(2u);
    uint32_t task = 0xFFFFFFFFu;
    while (q_tried < nq) {
      uint32_t qx = home + q_tried;
      if (qx >= nq) qx -= nq;
      const uint32_t q0 = (uint32_t)((uint64_t)n_launch * qx / nq), q1 = (uint32_t)((uint64_t)n_launch * (qx + 1u) / nq);
      uint32_t t = 0;
      if (lane == 0) t = atomicAdd(p.task_counter + qx, 1u);
      t = uni(t) + q0;
      if (t < q1) {
        task = t + p.task_begin;
        break;
      }
      ++q_tried;
    }
    if (task == 0xFFFFFFFFu) break;
    const uint4 trec = sload(p.tasks + task);
    const uint32_t j0 = trec.y, nb_task = trec.z & 0xFFFFu, ci = trec.z >> 24, lead0 = trec.w;
    n_leads = (trec.z >> 16) & 0xFFu;
    const TermRef lead = load_term(p.terms, trec.x);
    if (ci != cache_loaded) {
      const float *cg = p.caches + (size_t)ci * 256u;
      wave_mem_fence();
      for (int i = lane; i < 256; i += WAVE) L.cache[i] = cg[i];
      wave_mem_fence();
      cache_loaded = ci;
    }
    // ---- the group's leads, one per lane
    wave_mem_fence();
    // lane g also keeps lead g's constants in registers: the loops over the leads of a block fetch
    // them with v_readlane (no LDS round trip per (block, lead) pair); stage C, where every lane
    // works for another lead, gathers them from the LDS copy
    float my_w = 0.0f, my_rest = 0.0f;
    uint32_t my_mlo = 0, my_mhi = 0;
    uint32_t my_xlo = 0, my_xhi = 0, my_a1lo = 0, my_a1hi = 0, my_a2lo = 0, my_a2hi = 0;  // (boolean leads)
    uint32_t my_blo = 0, my_bhi = 0;  // (boolean leads) byte m: doc-matrix bit of the query's list m, 0 = none
    bool twin = false;
    const bool is_lead = (uint32_t)lane < n_leads;
    if (is_lead) {
      const TqdALead mine = p.leads[lead0 + lane];
      L.lead[lane] = TqdALeadLds{mine.query, mine.info, mine.w, mine.rest, mine.dense_off, mine.tf8_off};
      my_w = mine.w;
      my_rest = mine.rest;
      my_mlo = mine.mask_lo;
      my_mhi = mine.mask_hi;
      if constexpr (BOOLQ) {
        my_xlo = mine.excl_lo;
        my_xhi = mine.excl_hi;
        my_a1lo = mine.any1_lo;
        my_a1hi = mine.any1_hi;
        my_a2lo = mine.any2_lo;
        my_a2hi = mine.any2_hi;
        my_blo = mine.dense_off;
        my_bhi = mine.tf8_off;
        // what the query's other lists can add to a doc of this lead: the lists after the leader (a doc held
        // by a list of the lead set before it belongs to that list's lead; MustNot lists weigh 0)
        const TqdQuery *Q = p.queries + mine.query;
        const uint32_t li = (mine.info >> 16) & 15u, nt = mine.info & 31u;
#pragma unroll
        for (uint32_t m = 0; m < TQD_AS_MAX_TERMS; ++m) L.bw[(uint32_t)lane * TQD_AS_MAX_TERMS + m] = (m > li && m < nt) ? Q->weight[m] : 0.0f;
      }
      if constexpr (!BOOLQ) {
        // 2-term intersections: list 1's range maxima (TqdALead comment); `rest` = list 1's weight — in the
        // lead's LDS record — becomes what the list can add anywhere: its largest range maximum
        my_xlo = (bmode && (mine.info & 31u) == 2u) ? mine.excl_lo : 0u;
        // (list 1 probed through its range directory: the directory — which leader blocks the list has postings in —
        // and its shift; the list's largest tf/(tf + norm) bounds it as the largest range maximum does)
        const bool rdir = bmode && (mine.info & TQD_AL_RDIR) != 0u;
        my_xhi = rdir ? mine.dense_off | ((mine.info >> 16) & 31u) : 0u;  // (the directory is 256-byte aligned)
        if (my_xlo || rdir) {
          const float r1 = mine.rest * p.bound_slack * (1.00002f / 255.0f) * (float)(mine.any1_hi & 0xFFu);
          my_rest = r1 < mine.rest ? r1 : mine.rest;
        }
      }
      twin = lane != 0 && (mine.info & 0x200u) != 0u;
      L.lk[lane] = mine.k | (mine.thr_row << 8);
      L.lthr[lane] = __hip_atomic_load(p.thr_val + mine.query, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((uint32_t)lane < AS_GROUP) L.cnt[lane] = 0u;
    // families: a head and the twins behind it (the same query: same lists, weights, mask)
    const bool is_head = is_lead && !twin;
    const uint32_t heads = (uint32_t)__ballot(is_head);
    if (is_head) {  // leads in the family
      const uint32_t above = (uint32_t)lane >= 31u ? 0u : heads & ~((2u << lane) - 1u);
      L.flen[lane] = (above ? (uint32_t)__builtin_ctz(above) : n_leads) - (uint32_t)lane;
    }
    wave_mem_fence();
    // a lead whose best possible score is below its threshold is done with the whole list
    auto lead_alive = [&]() __attribute__((always_inline)) {
      return (uint32_t)lane < n_leads && sortable((my_w + my_rest) * 1.000001f) >= L.lthr[lane];
    };
    auto refresh_thr = [&]() __attribute__((always_inline)) {  // one word per lead
      if ((uint32_t)lane < n_leads) {
        const uint32_t t = __hip_atomic_load(p.thr_val + L.lead[lane].query, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t > L.lthr[lane]) L.lthr[lane] = t;
      }
      wave_mem_fence();
    };
    // live families (heads), their thresholds and the bound derived from them: "leader score + weights
    // of the other lists >= threshold" is one float compare per doc (scores are >= 0: the float order is
    // the order of the sortable bits): tfn >= (thr - rest) / w, every factor widened by 1e-5 for the
    // reciprocal-based tf/(tf+norm) and the summation order
    uint32_t live = 0, live_heads = 0, fam_thr = 0;
    float fam_need = -1.0f, fam_thr_f = -1.0f;
    auto update_families = [&]() __attribute__((always_inline)) {
      live = (uint32_t)__ballot(is_head && lead_alive());
      live_heads = live;
      fam_thr = is_head ? L.lthr[lane] : 0xFFFFFFFFu;
      fam_need = -1.0f;
      fam_thr_f = -1.0f;
      if (is_head && fam_thr != 0u) {
        const float thr_f = __uint_as_float(fam_thr ^ ((fam_thr >> 31) ? 0x80000000u : 0xFFFFFFFFu));
        fam_thr_f = thr_f * 0.99999f;
        const float num = thr_f * 0.99999f - my_rest * 1.00001f;
        if (num > 0.0f) fam_need = num * __builtin_amdgcn_rcpf(my_w) * 0.99999f;
      }
    };
    update_families();
    te(2u);

    for (uint32_t jt = 0; jt < nb_task && live; jt += TQD_AS_TILE) {
      if (jt) {  // thresholds may have risen since the last step
        tb(9u);
        refresh_thr();
        update_families();
        te(9u);
        if (!live) break;
      }
      tb(3u);
      // ---- pre-filter: lane <-> block
      const uint32_t nb = nb_task - jt < TQD_AS_TILE ? nb_task - jt : TQD_AS_TILE;
      const uint32_t i_base = j0 + jt;
      const uint32_t i_mine = i_base + (uint32_t)lane;
      const bool in_tile = (uint32_t)lane < nb && i_mine < lead.n_blocks;
      uint4 rec_mine = make_uint4(0u, 0u, 0u, 0u);
      if (in_tile) rec_mine = lead.rec[i_mine];
      // block-max of tf/(tf+norm), the weight-free part of block_max_score (term_scorer.rs:58-75)
      float tfn_max = 1.0f;
      {
        const uint32_t tfc = rec_mine.y >> 24;
        if (!(rec_mine.y == META_TAIL || !lead.has_freq || tfc == 0u)) {
          const float f = (float)(tfc == 255u ? 0xFFFFFFFFu : tfc);
          tfn_max = f * __builtin_amdgcn_rcpf(f + L.cache[(rec_mine.y >> 16) & 0xFFu]);
        }
      }
      // the block's record stays with its lane: {meta, payload offset, last doc of the block before}
      uint32_t prev_mine = __shfl_up(rec_mine.x, 1, WAVE);
      if (lane == 0) prev_mine = block_prev_last(lead, i_base);
      // (intersections) the level of list 1's range maxima at which this block's doc span [first, last] touches at
      // most two entries, and the two entries' indices there (no such level: the span is wider than the coarsest
      // level's entries, and the list's largest entry — already in the lead's `rest` — stands in)
      // (blk_first / blk_last = the block's doc span, rm_pack = the level's byte offset | its shift << 24, shift 0 = none)
      uint32_t rm_pack = 0, blk_first = 1u, blk_last = 0u;
      if constexpr (!BOOLQ) {
        if ((bmode & 1u) && in_tile) {
          blk_first = i_mine ? prev_mine + 1u : 0u;
          blk_last = rec_mine.x;
          const uint32_t wide = blk_last >= blk_first ? (blk_last - blk_first) >> TQD_RM_SHIFT : 0xFFFFFFFFu;  // span in level-0 entries
          const uint32_t lvl = wide ? (33u - (uint32_t)__builtin_clz(wide)) >> 1 : 0u;                      // 0 | 1..3 | 4..15 | ... -> 0 | 1 | 2 | ...
          if (lvl < TQD_RM_LEVELS) rm_pack = tqd_rm_level_off(seg.max_doc, lvl) | ((TQD_RM_SHIFT + 2u * lvl) << 24);
        }
      }
      uint32_t pass_mask = 0;  // families that still want this block (block_wand_intersection.rs:81-85)
      float need_blk = 3.0e38f;  // the loosest leader tf/(tf+norm) any of them still accepts from this block
      for (uint32_t lm = live_heads; lm; lm &= lm - 1u) {
        const uint32_t g = (uint32_t)__builtin_ctz(lm);
        const float w = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(my_w), (int)g));
        float rest = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(my_rest), (int)g));
        const uint32_t thr = (uint32_t)__builtin_amdgcn_readlane((int)fam_thr, (int)g);
        if constexpr (!BOOLQ) {
          // list 1 over this block's doc span: the larger of the (at most) two range maxima the span touches
          // (block_wand_intersection.rs:59-85: leader block-max + the secondaries' block-max)
          const uint32_t rmo = (uint32_t)__builtin_amdgcn_readlane((int)my_xlo, (int)g);
          if ((bmode & 1u) && rmo) {
            const uint8_t *rt = tbase + ((uint64_t)rmo << 3) + (rm_pack & 0xFFFFFFu);
            const uint32_t sh = rm_pack >> 24;
            const uint32_t qa = sh ? (uint32_t)rt[blk_first >> sh] : 255u, qb = sh ? (uint32_t)rt[blk_last >> sh] : 255u;
            if (p.debug & 0x800000u) n_scored += 2u * (uint32_t)__popcll(__ballot(sh != 0u));  // COUNTERS
            const float w1 = uni_f(L.lead[g].rest);  // (a 2-term query: the weight of list 1)
            const float r1 = w1 * p.bound_slack * (1.00002f / 255.0f) * (float)(qa > qb ? qa : qb);
            rest = r1 < rest ? r1 : rest;
          }
        }
        bool empty = false;  // list 1 has no posting in this block's doc span (its range directory says so)
        if constexpr (!BOOLQ) {
          const uint32_t rdo = (uint32_t)__builtin_amdgcn_readlane((int)my_xhi, (int)g);  // (directory | shift: 256-byte aligned)
          if ((bmode & 1u) && rdo) {
            const uint32_t S = rdo & 31u;
            const uint32_t *dr = reinterpret_cast<const uint32_t *>(tbase + ((uint64_t)(rdo & ~31u) << 3));
            if (blk_last >= blk_first) empty = dr[blk_first >> S] == dr[(blk_last >> S) + 1u];
          }
        }
        const float ub = w * tfn_max * p.bound_slack;
        if (in_tile && !empty && sortable((ub + rest) * 1.000004f + (w + rest) * 4.0e-6f) >= thr) {
          pass_mask |= 1u << g;
          if constexpr (!BOOLQ) {
            const float thr_f = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(fam_thr_f), (int)g));
            const float num = thr_f - rest * 1.00001f;
            const float nd = (thr_f > 0.0f && num > 0.0f) ? num * __builtin_amdgcn_rcpf(w) * 0.99999f : -1.0f;
            need_blk = nd < need_blk ? nd : need_blk;
          }
        }
      }
      uint64_t todo = __ballot(pass_mask != 0u);
      uint32_t since_refresh = 0;
      te(3u);
#if TQ_AS_PREFETCH
      // The payload of the NEXT wanted block travels to LDS (global_load_lds: one 16-byte row per lane,
      // no registers) while the current block's doc-matrix gathers, tests and scoring run: a block's
      // chain of dependent round trips loses its first link.
      auto prefetch = [&](uint64_t rest) __attribute__((always_inline)) {
        if (!rest) return;
        const uint32_t nb2 = (uint32_t)__builtin_ctzll(rest);
        const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)rec_mine.y, (int)nb2);
        const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)rec_mine.z, (int)nb2);
        if (meta == META_TAIL) return;
        const uint32_t nbytes = 16u * ((meta & 31u) + (lead.has_freq ? (meta >> 8) & 0xFFu : 0u));
        const uint32_t o = 16u * (uint32_t)lane;
        if (o < nbytes)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(idx + lead.payload_base + off + o),
                                           (__attribute__((address_space(3))) void *)L.pay, 16, 0, 0);
      };
      wave_mem_fence();
      prefetch(todo);
#endif
      while (todo) {
        const uint32_t b = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        if (++since_refresh == 8u) {  // thresholds rise while the tile is walked
          since_refresh = 0;
          tb(9u);
          refresh_thr();
          update_families();
          te(9u);
          if (!live) break;
        }
        tb(4u);
        uint32_t lm = (uint32_t)__builtin_amdgcn_readlane((int)pass_mask, (int)b) & live_heads;
        const uint32_t prev_l = (uint32_t)__builtin_amdgcn_readlane((int)prev_mine, (int)b);
        const uint2 mo_l = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)rec_mine.y, (int)b),
                                      (uint32_t)__builtin_amdgcn_readlane((int)rec_mine.z, (int)b));
#if TQ_AS_PREFETCH
        if (!lm) {  // (its families died since the pre-filter: the landed payload is dropped)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          wave_mem_fence();
          prefetch(todo);
          te(4u);
          continue;
        }
#else
        if (!lm) {
          te(4u);
          continue;
        }
#endif
        if (p.debug & 256u) ++n_scored;  // COUNTERS
        if (p.debug & 0x100000u)
          n_scored += 16u + (mo_l.x == META_TAIL ? 8u * lead.n_tail : 16u * ((mo_l.x & 31u) + (lead.has_freq ? (mo_l.x >> 8) & 0xFFu : 0u)));
        // ---- stage A: decode the block once
        uint32_t c0, c1, t0, t1;
#if TQ_AS_PREFETCH
        if (mo_l.x == META_TAIL) {
          decode_docs<USE_DPP>(idx, lead, mo_l, prev_l, lane, c0, c1);
          decode_tfs(idx, lead, mo_l, lane, t0, t1);
        } else {
          const uint32_t doc_bits = mo_l.x & 31u;
          const uint32_t strict = (mo_l.x >> 6) & 1u;
          const uint32_t tf_bits = lead.has_freq ? (mo_l.x >> 8) & 0xFFu : 0u;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prefetched payload has landed
          wave_mem_fence();
          if (lead.has_freq) {
            unpack2_lds(L.pay + 4u * doc_bits, tf_bits, lane, t0, t1);
            t0 += strict;  // minus-one encoding is tied to the strict flag
            t1 += strict;
          } else {
            t0 = 1u;
            t1 = 1u;
          }
          uint32_t x0, x1;
          unpack2_lds(L.pay, doc_bits, lane, x0, x1);
          finish_docs<USE_DPP>(x0, x1, strict, prev_l, lane, c0, c1);
        }
        wave_mem_fence();  // (every lane has read its payload words: the next block's may land)
        prefetch(todo);
#else
        decode_docs<USE_DPP>(idx, lead, mo_l, prev_l, lane, c0, c1);
        decode_tfs(idx, lead, mo_l, lane, t0, t1);
#endif
        if (TQ_AS_TIMERS && tphase == 4u && c0 == 0xFFFFFFFEu) ++n_scored;  // (the decode has to land inside the region)
        te(4u);
        tb(5u);
        const bool v0 = c0 != TQD_TERMINATED, v1 = c1 != TQD_TERMINATED;
        // The leader's tf/(tf+norm) from the fieldnorm BYTES first (1 B per doc: the part of the file the
        // chip is working on stays in the L2s), and the loosest bound any family of this block still
        // accepts: only docs that pass it pay the doc-matrix gather (8 B per doc out of a table that does
        // not fit the L2s: a 128-byte fabric request each — with every doc gathered the launch moved
        // 5.5 TB/s of them and waited for that).
        float block_need = 3.0e38f;
        for (uint32_t x = lm; x; x &= x - 1u) {
          const float nd = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(fam_need), (int)__builtin_ctz(x)));
          block_need = nd < block_need ? nd : block_need;
        }
        if constexpr (!BOOLQ) {
          // (the pre-filter's cut knows list 1's range maxima over this block, the loop above the thresholds as of
          // the last refresh: both are lower bounds of what any family of the block accepts)
          if (bmode & 2u) {
            const float nb = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(need_blk), (int)b));
            block_need = nb > block_need ? nb : block_need;
          }
        }
        const uint32_t nid0 = v0 ? fieldnorm_id(seg, c0) : 0u, nid1 = v1 ? fieldnorm_id(seg, c1) : 0u;
        const float f0 = (float)t0, f1 = (float)t1;
        const float tfn0 = f0 * __builtin_amdgcn_rcpf(f0 + L.cache[nid0]);
        const float tfn1 = f1 * __builtin_amdgcn_rcpf(f1 + L.cache[nid1]);
        const bool g0 = v0 && tfn0 >= block_need, g1 = v1 && tfn1 >= block_need;
        // ONE gather per doc that may matter: membership in every column list + signature bits
        const uint64_t mw0 = g0 ? seg.docmat[c0] : 0ull;
        const uint64_t mw1 = g1 ? seg.docmat[c1] : 0ull;
        const uint64_t valid0 = __ballot(g0), valid1 = __ballot(g1);
        if (p.debug & 0x200000u) n_scored += (uint32_t)(__popcll(__ballot(v0)) + __popcll(__ballot(v1)));  // COUNTERS
        if (p.debug & 0x400000u) n_scored += (uint32_t)(__popcll(valid0) + __popcll(valid1));
        if (TQ_AS_TIMERS && tphase == 5u && (uint32_t)mw0 == 0xFFFFFFFEu) ++n_scored;  // (the gathers have to land inside the region)
        te(5u);
        tb(6u);
        // ---- stage F: every family that wants the block
        uint32_t pm_lo = 0, pm_hi = 0;
        uint64_t mem0 = valid0, mem1 = valid1;  // (mask 0: every doc)
        uint32_t pb0 = 0xFFu, pb1 = 0xFFu;       // (boolean leads) lists the doc's doc-matrix word does not rule out
        if (p.debug & 2048u) lm = 0;  // ABLATION: decode only
        for (; lm; lm &= lm - 1u) {
          const uint32_t g = (uint32_t)__builtin_ctz(lm);
          if (p.debug & 32u) ++n_scored;  // COUNTERS
          const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)my_mlo, (int)g);
          const uint32_t mhi = (uint32_t)__builtin_amdgcn_readlane((int)my_mhi, (int)g);
          const float need = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(fam_need), (int)g));
          if constexpr (BOOLQ) {  // none of the excluded lists, one list of every Must clause (columns: exact;
                                  // signature bits: maybe), per family
            const uint32_t xlo = (uint32_t)__builtin_amdgcn_readlane((int)my_xlo, (int)g);
            const uint32_t xhi = (uint32_t)__builtin_amdgcn_readlane((int)my_xhi, (int)g);
            const uint32_t a1lo = (uint32_t)__builtin_amdgcn_readlane((int)my_a1lo, (int)g);
            const uint32_t a1hi = (uint32_t)__builtin_amdgcn_readlane((int)my_a1hi, (int)g);
            const uint32_t a2lo = (uint32_t)__builtin_amdgcn_readlane((int)my_a2lo, (int)g);
            const uint32_t a2hi = (uint32_t)__builtin_amdgcn_readlane((int)my_a2hi, (int)g);
            const uint32_t l0 = (uint32_t)mw0, h0 = (uint32_t)(mw0 >> 32), l1 = (uint32_t)mw1, h1 = (uint32_t)(mw1 >> 32);
            bool b0 = !((l0 & xlo) | (h0 & xhi));
            bool b1 = !((l1 & xlo) | (h1 & xhi));
            if (a1lo | a1hi) {
              b0 = b0 && ((l0 & a1lo) | (h0 & a1hi));
              b1 = b1 && ((l1 & a1lo) | (h1 & a1hi));
            }
            if (a2lo | a2hi) {
              b0 = b0 && ((l0 & a2lo) | (h0 & a2hi));
              b1 = b1 && ((l1 & a2lo) | (h1 & a2hi));
            }
            // (the family's bound with every other list in it first: most (block, family) pairs end here)
            b0 = b0 && tfn0 >= need;
            b1 = b1 && tfn1 >= need;
            if (!((__ballot(b0) & valid0) | (__ballot(b1) & valid1))) continue;
            // the doc's own bound: the leader's score + the weights of the lists its doc-matrix word does not
            // rule out (mask_lo: the sum over the lists without a bit) — `+a b -c` sends a doc of a that is
            // not in b on with a's score alone, not with a's + the weight of b.  Bytes: the lists after the
            // leader that can add to the score and have a bit; info bits 20-27: clear = a list the masks above
            // have already ruled out (the scoring stage does not probe it)
            const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)my_blo, (int)g);
            const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)my_bhi, (int)g);
            pb0 = pb1 = (uni(L.lead[g].info) >> 20) & 0xFFu;
            if (blo | bhi) {
              const float wl = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(my_w), (int)g));
              const float thr_f = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(fam_thr_f), (int)g));
              const float4 wa = *reinterpret_cast<const float4 *>(&L.bw[g * TQD_AS_MAX_TERMS]);
              const float4 wb = *reinterpret_cast<const float4 *>(&L.bw[g * TQD_AS_MAX_TERMS + 4u]);
              const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
              float ps0 = __uint_as_float(mlo), ps1 = ps0;
#pragma unroll
              for (uint32_t m = 0; m < TQD_AS_MAX_TERMS; ++m) {
                const uint32_t bp = ((m < 4u ? blo : bhi) >> (8u * (m & 3u))) & 0xFFu;
                if (!bp) continue;  // (uniform)
                const uint32_t sh = bp & 31u;
                const uint32_t p0 = ((bp & 32u ? h0 : l0) >> sh) & 1u, p1 = ((bp & 32u ? h1 : l1) >> sh) & 1u;
                ps0 += p0 ? wv[m] : 0.0f;
                ps1 += p1 ? wv[m] : 0.0f;
                pb0 &= ~((p0 ^ 1u) << m);
                pb1 &= ~((p1 ^ 1u) << m);
              }
              b0 = b0 && (wl * tfn0 + ps0) * 1.00001f >= thr_f;
              b1 = b1 && (wl * tfn1 + ps1) * 1.00001f >= thr_f;
            }
            mem0 = __ballot(b0) & valid0;
            mem1 = __ballot(b1) & valid1;
          } else if (mlo != pm_lo || mhi != pm_hi) {  // "every other list holds (or may hold) the doc"
            pm_lo = mlo;
            pm_hi = mhi;
            const uint32_t x0 = ((uint32_t)mw0 | ~mlo) & ((uint32_t)(mw0 >> 32) | ~mhi);
            const uint32_t x1 = ((uint32_t)mw1 | ~mlo) & ((uint32_t)(mw1 >> 32) | ~mhi);
            mem0 = __ballot(x0 == 0xFFFFFFFFu) & valid0;
            mem1 = __ballot(x1 == 0xFFFFFFFFu) & valid1;
          }
          if (!(mem0 | mem1)) continue;
          uint64_t a0m = 0, a1m = 0;
          bool ranged = false;
          if constexpr (!BOOLQ) {
            const uint32_t rmo = (uint32_t)__builtin_amdgcn_readlane((int)my_xlo, (int)g);
            if ((bmode & 4u) && rmo) {
              // list 1 can add at most its range maximum at the doc's own range (one byte per 1024 docs out of a
              // 10 KB table per list: L2-resident), not its weight: "leader + list 1's share >= threshold - others"
              ranged = true;
              const uint8_t *rt = tbase + ((uint64_t)rmo << 3);
              const bool m0 = (mem0 >> lane) & 1ull, m1 = (mem1 >> lane) & 1ull;
              const uint32_t q0 = m0 ? (uint32_t)rt[c0 >> TQD_RM_SHIFT] : 0u, q1 = m1 ? (uint32_t)rt[c1 >> TQD_RM_SHIFT] : 0u;
              if (p.debug & 0x800000u) n_scored += (uint32_t)(__popcll(mem0) + __popcll(mem1));  // COUNTERS
              // tfn0 + min(cq * q, cw) >= threshold / w, every factor widened like `need` above
              const float w1 = uni_f(L.lead[g].rest);  // (a 2-term query: the weight of list 1)
              const float rw = __builtin_amdgcn_rcpf(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(my_w), (int)g)));
              const float thr_f = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(fam_thr_f), (int)g));
              const float base = thr_f > 0.0f ? thr_f * rw * 0.99999f : -1.0f;
              const float cw = w1 * rw * 1.00002f, cq = cw * p.bound_slack * (1.00001f / 255.0f);
              const float s0 = cq * (float)q0, s1 = cq * (float)q1;
              a0m = __ballot(m0 && tfn0 + (s0 < cw ? s0 : cw) >= base);
              a1m = __ballot(m1 && tfn1 + (s1 < cw ? s1 : cw) >= base);
            }
          }
          if (!ranged) {
            a0m = __ballot(tfn0 >= need) & mem0;
            a1m = __ballot(tfn1 >= need) & mem1;
          }
          if (!(a0m | a1m)) continue;
          if (p.debug & 1024u) continue;  // ABLATION: no queue, no stage C
          // the two docs of a lane are queued one after the other: the queue holds < 64 leftovers
          // plus <= 64 new entries and is drained below 64 before the next push
#pragma unroll 1
          for (uint32_t e = 0; e < 2u; ++e) {
            const uint64_t m = e ? a1m : a0m;
            if (!m) continue;
            const bool a = (m >> lane) & 1ull;
            const uint32_t pos = qn + mbcnt64(m);
            wave_mem_fence();
            if (a) {
              L.q_doc[pos] = e ? c1 : c0;
              L.q_tf[pos] = e ? t1 : t0;
              L.q_tag[pos] = g | ((e ? nid1 : nid0) << 8) | (BOOLQ ? (e ? pb1 : pb0) << 16 : 0u);
            }
            wave_mem_fence();
            qn += (uint32_t)__popcll(m);
            while (qn >= 64u) stageC(64u);
          }
        }
        te(6u);
      }
    }
    tb(6u);
    while (qn) stageC(qn < 64u ? qn : 64u);
    te(6u);
    tb(8u);

    // ---- flush: the heads' staging lists go to the result lists of every member of their families
    wave_mem_fence();
    const uint32_t cw = (uint32_t)lane < AS_GROUP ? L.cnt[lane] : 0u;
    const uint32_t cn = cw & 0xFFFFu, sc = cw >> 16;
    uint64_t upd = __ballot(sc != 0u);  // families that got new scores
    while (upd) {
      const uint32_t gs = (uint32_t)__builtin_ctzll(upd);
      upd &= upd - 1ull;
      if (p.debug & 16384u) ++n_scored;  // COUNTERS
      const uint32_t lkv = uni(L.lk[gs]);
      const uint32_t ks = lkv & 0xFFu, rs = lkv >> 8;
      const uint32_t fl = uni(L.flen[gs]);
      const uint32_t scs = (uint32_t)__builtin_amdgcn_readlane((int)sc, (int)gs);
      uint32_t ns = (uint32_t)__builtin_amdgcn_readlane((int)cn, (int)gs);
      // the k-th largest of the query's slots is the new shared threshold of the family's members
      const uint32_t *slots = p.thr_slots + (size_t)rs * TQD_THR_SLOTS;
      uint32_t sv[4] = {0u, 0u, 0u, 0u};
      sv[0] = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t gth;
      if (ks > 16u) {
#pragma unroll
        for (int r = 1; r < 4; ++r)
          sv[r] = __hip_atomic_load(slots + 64 * r + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gth = kth_largest_hi16<4>(sv, ks);
      } else {
        gth = kth_largest_hi16<1>(sv, ks);
      }
      // lane i < fl <-> member i of the family
      uint32_t mq = 0, thr_now = 0;
      if ((uint32_t)lane < fl) {
        mq = L.lead[gs + (uint32_t)lane].query;
        if (gth) atomicMax(p.thr_val + mq, gth);
        atomicAdd(sload(&p.sinks->query_matches) + sload(&p.sinks->out_index)[mq], scs);
      }
      // (what this wave knows: the head's threshold as of its last refresh and the select just done;
      // another round trip for the word's current value bought nothing)
      thr_now = uni(L.lthr[gs]);
      if (gth > thr_now) thr_now = gth;
      if (!ns) continue;
      const uint64_t *sl = my_stage + (size_t)gs * CAPL;
      uint64_t v[R];
      uint32_t keep_n = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t i = (uint32_t)r * 64u + (uint32_t)lane;
        v[r] = i < ns ? sl[i] : 0ull;
        if ((uint32_t)(v[r] >> 32) < thr_now) v[r] = 0ull;  // k docs of the query score higher by now
        keep_n += (uint32_t)__popcll(__ballot(v[r] != 0ull));
      }
      // (the threshold just selected from the slots drops most entries: the exact select over the
      // list — 64 dependent steps — is only paid by the lists that still hold more than k)
      if (keep_n > ks) {
        if (p.debug & 8192u) ++n_scored;  // COUNTERS
        const uint64_t kth = as_kth_largest_key<R>(v, ks);
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (v[r] < kth) v[r] = 0ull;
        keep_n = ks;
      }
      if (!keep_n) continue;
      // one append per member: its place in its query's result list
      uint32_t at_mine = 0;
      if ((uint32_t)lane < fl) at_mine = p.queries[mq].part_start + atomicAdd(p.list_count + mq, keep_n);
      for (uint32_t i = 0; i < fl; ++i) {
        const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)at_mine, (int)i);
        uint64_t *dst = p.lists + (size_t)at;
        uint32_t base = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint64_t m = __ballot(v[r] != 0ull);
          if (v[r] != 0ull) dst[base + mbcnt64(m)] = v[r];
          base += (uint32_t)__popcll(m);
        }
      }
    }
    te(8u);
  }
  te(1u);
  if (tphase) n_scored = (uint32_t)(tacc >> 6);
  if (lane == 0 && n_scored) atomicAdd(sload(&p.sinks->match_counter), (unsigned long long)n_scored);
}

}  // namespace

// =================================================================== launch wrappers
uint32_t tqk_ashare_waves_per_cu() { return 4u * TQ_AS_WAVES; }
uint32_t tqk_bshare_waves_per_cu() { return 4u * TQ_BS_WAVES; }

hipError_t tqk_launch_ashare(const TqkAShareParams &p, int kpl, hipStream_t st) {
  if (p.n_tasks <= p.task_begin || p.grid == 0) return hipSuccess;
  const dim3 grid(p.grid), block(64);
  if (p.boolean) {
    switch (kpl) {
      case 1:
        if (p.rdir_lists)
          ashare_kernel<1, true, true><<<grid, block, 0, st>>>(p);
        else
          ashare_kernel<1, true><<<grid, block, 0, st>>>(p);
        break;
      default:
        if (p.rdir_lists)
          ashare_kernel<2, true, true><<<grid, block, 0, st>>>(p);
        else
          ashare_kernel<2, true><<<grid, block, 0, st>>>(p);
        break;
    }
  } else {
    switch (kpl) {
      case 1: ashare_kernel<1, false><<<grid, block, 0, st>>>(p); break;
      default: ashare_kernel<2, false><<<grid, block, 0, st>>>(p); break;
    }
  }
  return hipGetLastError();
}
