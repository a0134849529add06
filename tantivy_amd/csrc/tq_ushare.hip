// tq_ushare.hip — the pure unions of a batch, driven TERM BY TERM instead of query by query.
// Shared device helpers: tq_common.hpp.
//
// block_wand (src/query/boolean_query/block_wand_union.rs:16-80,158-214) in its MaxScore set form,
// as union_kernel (tq_union.hip) runs it per query: terms by weight descending, every list leads
// the docs it is the FIRST list to hold, a doc first seen in list i can score at most the weights
// of lists i.. ("suffix"), lists whose suffix is below the threshold are never enumerated.
//
// What changes is the loop order.  In a batch many queries lead with the same term (nine per term
// in the 1000-query or5 batch, 45 in the mixed one): each of them used to decode the same blocks
// and gather the same doc-matrix rows.  Here a TASK is a run of blocks of ONE term for a GROUP of
// up to 32 LEADS — (query, list) pairs whose list is that term:
//   pre-filter, lane <-> block: the block's record, its block-max tf/(tf+norm), and for every live
//      lead "block-max score + weights of the later lists >= threshold" -> a lead mask per block;
//   A  per surviving block: ONE staged load + unpack + prefix sum, ONE 8-byte doc-matrix gather
//      per doc (fieldnorm id + membership in every dense list of the segment);
//   F  per (block, lead in the block's mask): the lead's constants come from lane g's registers
//      (readlane -> scalar operands); two docs per lane test "another list before i holds the doc"
//      (owned elsewhere) and "leader score + weights of the later lists that HOLD it (column bits)
//      or MAY hold it (no column) >= threshold" on registers; survivors -> LDS queue, tagged g;
//   C  64 survivors, each lane with its own query: exact BM25 of the leader and of the later
//      lists (bitmap rank -> block record -> tf bits; lists without a bitmap: seek + block search,
//      up to four distinct (list, block) pairs per step), ownership probes into the earlier lists
//      without a column, then the collector step.
// Scores are summed leader first, then ascending list index: the bits of every other union
// kernel and mode.
//
// Collector: a wave works for up to 32 queries at once, so the top-k cannot live in registers.
// Every lead slot has a staging list in global memory private to the wave (k + 64.. entries);
// a list that fills up is cut back to its k best (64-bit radix select) — which also yields a
// threshold: k real docs score at least that.  At the end of a task the lists are appended to
// their queries' result lists (one atomic per list), which merge_lists_kernel reduces.
// Thresholds: the hashed atomic-max slots of the other pruned kernels, plus thr_val[query], the
// k-th largest slot as last computed by a wave that changed a slot — tasks read ONE word per
// lead instead of selecting over 256 slots.  Both only ever rise and every value is the score of
// k distinct real matches, so pruning stays exact: the top-k equals the exhaustive run's bits.
#include "tq_common.hpp"

#ifndef TQ_US_REFRESH
#define TQ_US_REFRESH 8u  // blocks between two looks at the leads' thresholds (other waves raise them)
#endif
#ifndef TQ_US_TIMERS
#define TQ_US_TIMERS 0  // region timers (tools/probe_ushare.sh builds a variant with them)
#endif
#ifndef TQ_US_CHUNK
#define TQ_US_CHUNK 2  // lists after the leader resolved together in the scoring stage (4 spills more than it hides)
#endif
#ifndef TQ_US_WAVES
#define TQ_US_WAVES 4
#endif
#ifndef TQ_US_FWIDE
#define TQ_US_FWIDE 1  // stage F reads a lead's constants as four 16-byte LDS reads (0: field by field, round 3)
#endif

namespace {

constexpr uint32_t US_GROUP = TQD_US_GROUP;
constexpr uint32_t CH = TQ_US_CHUNK;

struct alignas(16) ShareLds {  // per wavefront
  // The leads of the task, one per slot: kept in LDS, not in lane g's registers — stage C needs
  // its registers for four lists' worth of loads in flight, and 20 live lead registers across it
  // ended in scratch (3 spill reloads per (block, lead) pair: 7000 cycles each, measured).
  // First member: 16-byte aligned, so that stage F reads a record's first 64 bytes as four ds_read_b128.
  TqdLead lead[US_GROUP];
  uint32_t pay[516];  // staged payload of the leader block / four 512-byte regions of the block search
  float cache[256];   // Bm25Weight.cache of the task's queries
  uint32_t q_doc[127], q_tf[127], q_tag[127];  // survivors: doc, leader tf, g | fieldnorm id << 8 | column bits << 16
  uint32_t lthr[US_GROUP];     // the lead's threshold (sortable score bits); only ever rises
  uint32_t lk[US_GROUP];       // k of its query (bits 0..7: k <= 128) | its row of threshold slots << 8
  // per block of the current tile: the leads that still want it; per doc of the current block:
  // tf/(tf+norm).  (In LDS because the scoring stage needs the registers: what the compiler spilled
  // instead cost a scratch round trip per (block, lead) pair.)
  uint32_t pmask[64];
  float tfn[2][64];
  uint32_t cnt[US_GROUP];      // bits 0..15: entries in the lead slot's staging list; bits 16..31: docs
                               // scored for the slot by this task (Count / statistics)
};

// per-lane view of a posting list (fields of TqdTermHead fetched with vector loads)
__device__ __forceinline__ TermRef load_term_lane(const TqdTerm *terms, uint32_t handle) {
  const TqdTermHead *h = terms + handle;
  TermRef r;
  r.rec = h->rec;
  r.coarse = h->coarse;
  r.dense = h->dense;
  r.tail_docs = h->tail_docs;
  r.tail_tfs = h->tail_tfs;
  r.payload_base = h->payload_base;
  r.n_blocks = h->n_blocks;
  r.n_tail = h->n_tail;
  const uint32_t hf = h->has_freq;
  r.has_freq = hf & 1u;
  r.mat_slot = ((hf >> 8) & 0xFFu) - 1u;
  r.shift = h->coarse_shift;
  return r;
}

// lookup_in_blocks<false> (tq_common.hpp) for candidates whose LISTS differ per lane: where is
// `doc` inside block jb of list t?  Up to four distinct (list, block) pairs are decoded per step,
// one per 16-lane row; then every candidate binary-searches its block (block_search.rs:38-76).
__device__ __forceinline__ uint32_t lookup_docs_multi(const uint8_t *idx, const TermRef &t,
                                                      uint32_t jb, uint32_t doc, bool alive,
                                                      uint32_t *P, int lane) {
  uint32_t result = NOT_FOUND;
  uint64_t pend = __ballot(alive);
  const uint32_t row = (uint32_t)lane >> 4, l16 = (uint32_t)lane & 15u;
  while (pend) {
    uint32_t gid = 4u, n_groups = 0;
    // the row's block: record pointer, block index, payload base, vint tail of its list
    uint64_t my_rec = 0, my_pb = 0, my_td = 0;
    uint32_t my_j = 0, my_nt = 0;
#pragma unroll
    for (uint32_t g = 0; g < 4u; ++g) {
      if (pend) {
        const uint32_t l = (uint32_t)__builtin_ctzll(pend);
        const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)jb, (int)l);
        const uint64_t rp = readlane64((uint64_t)t.rec, l);
        const uint64_t pb = readlane64(t.payload_base, l);
        const uint64_t td = readlane64((uint64_t)t.tail_docs, l);
        const uint32_t nt = (uint32_t)__builtin_amdgcn_readlane((int)t.n_tail, (int)l);
        const bool in = alive && gid == 4u && jb == j && (uint64_t)t.rec == rp;
        if (in) gid = g;
        pend &= ~__ballot(in);
        if (row == g) {
          my_rec = rp;
          my_pb = pb;
          my_td = td;
          my_j = j;
          my_nt = nt;
        }
        n_groups = g + 1u;
      }
    }
    const bool row_on = row < n_groups;
    uint4 rec = make_uint4(0u, META_TAIL, 0u, 0u);
    uint32_t prev = 0;
    if (row_on) {
      const uint4 *rp = (const uint4 *)my_rec;
      rec = rp[my_j];
      if (my_j) prev = rp[my_j - 1u].x;
    }
    const bool is_tail = rec.y == META_TAIL;
    const uint32_t b = is_tail ? 0u : (rec.y & 31u);
    const uint32_t strict = is_tail ? 0u : (rec.y >> 6) & 1u;
    wave_mem_fence();
    if (row_on && !is_tail) {
      const uint8_t *src = idx + my_pb + rec.z + 16u * l16;
      if (l16 < b) {
        const U4Unaligned v = *reinterpret_cast<const U4Unaligned *>(src);
        *reinterpret_cast<uint4 *>(P + row * 128u + 4u * l16) = make_uint4(v.x, v.y, v.z, v.w);
      }
      if (16u + l16 < b) {
        const U4Unaligned v = *reinterpret_cast<const U4Unaligned *>(src + 256);
        *reinterpret_cast<uint4 *>(P + row * 128u + 64u + 4u * l16) = make_uint4(v.x, v.y, v.z, v.w);
      }
    }
    wave_mem_fence();
    uint32_t d[8];
    {
      const uint32_t mask = b >= 32u ? 0xFFFFFFFFu : (1u << b) - 1u;
#pragma unroll
      for (uint32_t kk = 0; kk < 2u; ++kk) {
        const uint32_t bitpos = (2u * l16 + kk) * b;
        const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
        const uint4 lo = *reinterpret_cast<const uint4 *>(P + row * 128u + 4u * w);
        const uint4 hi = *reinterpret_cast<const uint4 *>(P + row * 128u + 4u * w + 4u);
        d[4 * kk + 0] = (__funnelshift_r(lo.x, hi.x, sh) & mask) + strict;
        d[4 * kk + 1] = (__funnelshift_r(lo.y, hi.y, sh) & mask) + strict;
        d[4 * kk + 2] = (__funnelshift_r(lo.z, hi.z, sh) & mask) + strict;
        d[4 * kk + 3] = (__funnelshift_r(lo.w, hi.w, sh) & mask) + strict;
      }
    }
    if (__ballot(row_on && is_tail)) {  // the pre-decoded vint tail of the list
      if (row_on && is_tail) {
        const uint32_t *td = (const uint32_t *)my_td;
#pragma unroll
        for (uint32_t e = 0; e < 8u; ++e) {
          const uint32_t i = 8u * l16 + e;
          d[e] = i < my_nt ? td[i] : TQD_TERMINATED;
        }
      }
    }
    if (!is_tail) {  // (uniform per 16-lane row)
      uint32_t loc[8];
      loc[0] = d[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) loc[e] = loc[e - 1] + d[e];
      uint32_t incl = loc[7];
      incl += dpp_get<0x111, 0xF>(incl);
      incl += dpp_get<0x112, 0xF>(incl);
      incl += dpp_get<0x114, 0xF>(incl);
      incl += dpp_get<0x118, 0xF>(incl);
      // compression/mod.rs:36-39,112-121: offset 0 <=> None <=> seed u32::MAX (wrapping)
      const uint32_t base = ((strict && prev == 0u) ? 0xFFFFFFFFu : prev) + (incl - loc[7]);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = loc[e] + base;
    }
    wave_mem_fence();
    if (row_on) {
      *reinterpret_cast<uint4 *>(P + row * 128u + 8u * l16) = make_uint4(d[0], d[1], d[2], d[3]);
      *reinterpret_cast<uint4 *>(P + row * 128u + 8u * l16 + 4u) = make_uint4(d[4], d[5], d[6], d[7]);
    }
    wave_mem_fence();
    if (gid < 4u) {
      const uint32_t *blk = P + gid * 128u;
      uint32_t pos = 0;
#pragma unroll
      for (uint32_t step = 64u; step > 0u; step >>= 1)
        if (blk[pos + step - 1u] < doc) pos += step;
      result = blk[pos] == doc ? pos : NOT_FOUND;
    }
  }
  return result;
}

// k-th largest of the n (<= 64 R) keys held R per lane (0 = empty); n >= k
template <int R>
__device__ __forceinline__ uint64_t kth_largest_key(const uint64_t (&v)[R], uint32_t k) {
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

template <int KPL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(96), amdgpu_waves_per_eu(TQ_US_WAVES, 8))) void
ushare_kernel(TqkShareParams p) {
  constexpr bool USE_DPP = true;
  constexpr int R = KPL + 1;                    // staging registers per lane
  constexpr uint32_t CAPL = (uint32_t)R * 64u;  // staging entries per lead slot
  __shared__ ShareLds L;
  const int lane = (int)__lane_id();
  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;
  uint64_t *const my_stage = p.stage + (size_t)blockIdx.x * (size_t)(US_GROUP * CAPL);
  uint32_t cache_loaded = 0xFFFFFFFFu;
  uint32_t qn = 0;        // survivor queue fill
  uint32_t n_scored = 0;  // docs scored by this wave (all tasks)

  // ---- state of the current task (lane g < n_leads <-> lead g)
  uint32_t n_leads = 0;
  // PROFILING (TQ_DEBUG bits 16..19 = region): wave cycles spent inside ONE region per run, summed
  // into the match counter.  1 everything, 2 task fetch + setup, 3 pre-filter, 4 stage A (decode +
  // doc-matrix gather), 5 stage F, 6 stage C, 7 stage C: lists after the leader, 8 stage C: ownership
  // probes, 9 stage C: collector (slots, select, staging), 10 staging cuts, 11 flush
  const uint32_t tphase = TQ_US_TIMERS ? (p.debug >> 16) & 15u : 0u;
  uint64_t tacc = 0, tlast = 0;
  auto tb = [&](uint32_t ph) __attribute__((always_inline)) {
    if (TQ_US_TIMERS && tphase == ph) tlast = __builtin_readcyclecounter();
  };
  auto te = [&](uint32_t ph) __attribute__((always_inline)) {
    if (TQ_US_TIMERS && tphase == ph) tacc += __builtin_readcyclecounter() - tlast;
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
    const uint64_t kth = kth_largest_key<R>(v, k);
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
    te(5u);
    tb(6u);
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
    const uint32_t g = tag & 31u;
    // fieldnorm id + membership in the dense lists: the doc-matrix word again (a gather for the
    // survivors only; stage F does not carry it through the queue)
    const uint64_t mw = alive ? seg.docmat[doc] : 0ull;
    const float norm = L.cache[(uint32_t)mw & 0xFFu];
    uint32_t bits = 0;
    {
      const uint32_t cl = L.lead[g].cols_lo, ch = L.lead[g].cols_hi;
#pragma unroll
      for (uint32_t c = 0; c < 7u; ++c) {
        const uint32_t col = ((c < 4u ? cl >> (8u * c) : ch >> (8u * (c - 4u)))) & 0xFFu;
        bits |= (col ? (uint32_t)(mw >> col) & 1u : 0u) << c;
      }
    }
    // the lead's constants from lane g's registers
    const uint32_t q = L.lead[g].query;
    const uint32_t info = L.lead[g].info;
    const float w_lead = L.lead[g].w;
    const float suffix = L.lead[g].suffix;
    const uint32_t thr = L.lthr[g];
    const uint32_t k = L.lk[g] & 0xFFu;
    const uint32_t thr_row = L.lk[g] >> 8;
    const uint32_t li = info & 15u, nt = (info >> 8) & 15u, nocol = (info >> 16) & 0xFFu, nopc = info >> 24;
    const float slack_abs = suffix * 4.0e-6f;
    float s = bm25(w_lead, norm, tf);
    float rest = suffix - w_lead;  // what the lists after the current one can still add
    uint32_t c = 0;                // column counter of the lists after the leader
    uint32_t max_after = 0, max_before = 0;
    {
      const uint32_t na = alive ? nt - 1u - li : 0u, nbf = alive ? li : 0u;
#pragma unroll
      for (uint32_t x = 1; x < TQD_US_MAX_TERMS; ++x) {  // (independent ballots, no shuffle chain)
        if (__ballot(na >= x)) max_after = x;
        if (__ballot(nbf >= x)) max_before = x;
      }
    }
    // Lists after the leader, ascending: they add to the score.  CH lists at a time, and every
    // level of the chain is issued for all of them before the next one is awaited: bitmap word
    // (membership of the lists without a column + the posting index) -> the tf byte at that index.
    // The tables' addresses come from the lead record in LDS: two dependent round trips per group
    // of lists (it was four per list: term table -> bitmap word -> block record -> packed tf).
    // The adds then run in list order.
    const uint8_t *const tbase = p.table_base;
    for (uint32_t a0 = 1; a0 <= max_after; a0 += CH) {
      bool on[CH], probe_sparse[CH], found[CH], has_w[CH];
      float w[CH];
      uint32_t tfv[CH], pi[CH], doff[CH], toff[CH];
#pragma unroll
      for (uint32_t u = 0; u < CH; ++u) {
        const uint32_t m = li + a0 + u;
        on[u] = alive && m < nt;
        // a list with a membership bit: exact (doc-matrix column) or "maybe" (signature bit)
        const bool has_bit = on[u] && !((nopc >> m) & 1u);
        const bool exact = on[u] && !((nocol >> m) & 1u);
        bool set = false;
        w[u] = 0.0f;
        has_w[u] = has_bit;
        if (has_bit) {
          set = (bits >> c) & 1u;
          w[u] = L.lead[g].aw[c < 7u ? c : 6u];
          ++c;
        }
        doff[u] = 0;
        toff[u] = 0;
        if (on[u] && (a0 + u) <= 7u) {
          doff[u] = L.lead[g].dense_off[a0 + u - 1u];
          toff[u] = L.lead[g].tf8_off[a0 + u - 1u];
        }
        found[u] = exact && set;  // (column lists: membership is known already; the rank is not)
        // no column: bitmap or seek — unless the signature says "not in it"
        probe_sparse[u] = on[u] && !exact && (set || !has_bit);
        tfv[u] = 0;
        pi[u] = 0;
      }
      tb(12u);
      // level 1: bitmap word + rank — or, a list with a range directory (doff's low bits = its shift): the two
      // directory slots around the doc's range (round 6: such a list used to be seeked and block-searched)
      uint2 wd[CH];
#pragma unroll
      for (uint32_t u = 0; u < CH; ++u) {
        wd[u] = make_uint2(0u, 0u);
        if ((found[u] || probe_sparse[u]) && doff[u]) {
          const uint32_t S = doff[u] & 31u;
          if (!S) {
            wd[u] = reinterpret_cast<const uint2 *>(tbase + ((uint64_t)doff[u] << 3))[doc >> 5];
          } else {
            const uint32_t *dir = reinterpret_cast<const uint32_t *>(tbase + ((uint64_t)(doff[u] & ~31u) << 3)) + (doc >> S);
            wd[u] = make_uint2(dir[0], dir[1]);
          }
        }
      }
      bool rd[CH];  // the list is probed through its range directory: entries wd.x .. wd.y - 1 are its range's
#pragma unroll
      for (uint32_t u = 0; u < CH; ++u) {
        const uint32_t bit = doc & 31u;
        rd[u] = probe_sparse[u] && (doff[u] & 31u);
        if (probe_sparse[u] && doff[u] && !rd[u]) {
          found[u] = (wd[u].x >> bit) & 1u;
          probe_sparse[u] = false;
        }
        if (rd[u]) {
          probe_sparse[u] = false;
          rd[u] = wd[u].x < wd[u].y;
        } else {
          pi[u] = wd[u].y + (uint32_t)__popc(wd[u].x & ((1u << bit) - 1u));
        }
      }
      // level 2: the tf byte / the range's first two entries
      uint32_t e1[CH];
#pragma unroll
      for (uint32_t u = 0; u < CH; ++u) {
        e1[u] = 0;
        if (found[u]) tfv[u] = (tbase + ((uint64_t)toff[u] << 3))[pi[u]];
        if (rd[u]) {
          const uint32_t *ent = reinterpret_cast<const uint32_t *>(tbase + ((uint64_t)toff[u] << 3)) + wd[u].x;
          tfv[u] = ent[0];
          if (wd[u].x + 1u < wd[u].y) e1[u] = ent[1];
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < CH; ++u) {
        if (__ballot(rd[u])) {
          const uint32_t key = doc & ((1u << (doff[u] & 31u)) - 1u);
          uint32_t lo = wd[u].x, e = tfv[u];
          bool on2 = rd[u];
          if (on2) {
            tfv[u] = 0;
            if ((e >> 16) != key && lo + 1u < wd[u].y) {  // (the second entry, already here)
              e = e1[u];
              ++lo;
            }
          }
          while (__ballot(on2)) {  // (entries ascend inside a range: one to two on average)
            if (on2) {
              if ((e >> 16) == key) {
                found[u] = true;
                tfv[u] = e & 0xFFFFu;
                pi[u] = lo;
                on2 = false;
              } else if ((e >> 16) > key || lo + 1u >= wd[u].y) {
                on2 = false;
              } else {
                ++lo;
                e = (reinterpret_cast<const uint32_t *>(tbase + ((uint64_t)toff[u] << 3)))[lo];
              }
            }
          }
          if (__ballot(rd[u] && found[u] && tfv[u] == 0xFFFFu)) {  // tf >= 65535: block record -> packed tf
            if (rd[u] && found[u] && tfv[u] == 0xFFFFu) {
              const TermRef tr = load_term_lane(p.terms, p.queries[q].term[li + a0 + u]);
              const uint4 r = tr.rec[pi[u] >> 7];
              tfv[u] = block_tf_at(idx, tr, make_uint2(r.y, r.z), pi[u] & 127u);
            }
          }
        }
      }
      if (TQ_US_TIMERS && tphase == 12u) {  // (the loads have to land inside the timed region)
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t u = 0; u < CH; ++u) acc += tfv[u];
        if (acc == 0xFFFFFFFFu) ++n_scored;
      }
      te(12u);
      tb(13u);
      // in list order: saturated tf bytes are read exactly, lists without a bitmap are searched
      // (only while the doc can still make it), scores are added
#pragma unroll
      for (uint32_t u = 0; u < CH; ++u) {
        const uint32_t m = li + a0 + u;
        if (__ballot(found[u] && tfv[u] == 255u)) {  // tf >= 255: block record -> packed tf
          if (found[u] && tfv[u] == 255u) {
            const TermRef tr = load_term_lane(p.terms, p.queries[q].term[m]);
            const uint4 r = tr.rec[pi[u] >> 7];
            tfv[u] = block_tf_at(idx, tr, make_uint2(r.y, r.z), pi[u] & 127u);
          }
        }
        if (__ballot(probe_sparse[u])) {
          bool cand = probe_sparse[u] && alive;
          if (cand) {
            const float r0 = rest > 0.0f ? rest : 0.0f;
            if (!(sortable((s + r0) * 1.000002f + slack_abs) >= thr)) {
              alive = false;
              cand = false;
            }
          }
          if (__ballot(cand)) {
            TermRef tr{};
            uint32_t jb = 0;
            if (cand) {
              tr = load_term_lane(p.terms, p.queries[q].term[m]);
              jb = seek_block(tr, doc);
              cand = jb < tr.n_blocks;
            }
            if (p.debug & 128u) n_scored += (uint32_t)__popcll(__ballot(cand));  // COUNTERS
            const uint32_t at = lookup_docs_multi(idx, tr, jb, doc, cand, L.pay, lane);
            if (cand && at != NOT_FOUND) {
              const uint4 r = tr.rec[jb];
              tfv[u] = block_tf_at(idx, tr, make_uint2(r.y, r.z), at);
              found[u] = true;
            }
          }
        }
        if (on[u] && !has_w[u]) w[u] = p.queries[q].weight[m];  // (a list with neither column nor signature bit)
        if (found[u] && alive) s = s + bm25(w[u], norm, tfv[u]);
        if (on[u]) rest -= w[u];
      }
      te(13u);
    }
    te(7u);
    tb(8u);
    // lists before the leader without a column: found there = that list's tasks score the doc
    for (uint32_t m = 0; m < max_before; ++m) {
      bool probe = alive && m < li && ((nocol >> m) & 1u);
      if (probe) {  // the signature word may already say "not in that list"
        const uint32_t sb1 = L.lead[g].sig[m & 7u];
        if (sb1 && !((mw >> (TQD_SIG_SHIFT + sb1 - 1u)) & 1u)) probe = false;
      }
      if (probe && !(sortable(s * 1.000002f + slack_abs) >= thr)) {  // the score is final: dead either way
        alive = false;
        probe = false;
      }
      if (!__ballot(probe)) continue;
      TermRef tr{};
      if (probe) tr = load_term_lane(p.terms, p.queries[q].term[m]);
      const bool bitmap = probe && tr.dense != nullptr;
      bool found = false;
      if (bitmap) found = (tr.dense[doc >> 5].x >> (doc & 31u)) & 1u;
      bool cand = probe && !bitmap;
      if (__ballot(cand)) {
        uint32_t jb = 0;
        if (cand) {
          jb = seek_block(tr, doc);
          cand = jb < tr.n_blocks;
        }
        if (p.debug & 128u) n_scored += (uint32_t)__popcll(__ballot(cand));  // COUNTERS
        const uint32_t a2 = lookup_docs_multi(idx, tr, jb, doc, cand, L.pay, lane);
        if (cand && a2 != NOT_FOUND) found = true;
      }
      if (found) alive = false;
    }
    // the score is final: below the threshold it cannot enter the top-k (equal scores stay: ties
    // resolve by doc id in the collector)
    te(8u);
    if (alive) alive = sortable(s) >= thr;
    if (alive) alive = doc_is_alive(seg, doc);
    const uint64_t hit = __ballot(alive);
    if (!hit) {
      te(6u);
      tb(5u);
      return;
    }
    tb(9u);
    if (!(p.debug & 992u)) n_scored += (uint32_t)__popcll(hit);  // COUNTERS (TQ_DEBUG): 32 (block, lead) pairs,
    const uint64_t key = alive ? make_key(s, doc) : 0ull;        // 64 stage-C candidates, 128 block searches, 256 blocks decoded
    const uint32_t sb = (uint32_t)(key >> 32);
    bool changed = false;
    if (alive) {

      const uint32_t hsh = (doc * 0x9E3779B1u) >> (k <= 16u ? 26 : 24);
      const uint32_t old = atomicMax(p.thr_slots + (size_t)thr_row * TQD_THR_SLOTS + hsh, sb);
      changed = old < sb;
      const uint32_t pos = atomicAdd(&L.cnt[g], 0x10001u) & 0xFFFFu;  // (a list never overflows: see the cut below)
      my_stage[(size_t)g * CAPL + pos] = key;
    }
    // queries whose slots changed: their k-th largest slot is the new shared threshold
    uint64_t chg = __ballot(changed);
    while (chg) {
      const uint32_t l = (uint32_t)__builtin_ctzll(chg);
      const uint32_t qs = (uint32_t)__builtin_amdgcn_readlane((int)q, (int)l);
      const uint32_t ks = (uint32_t)__builtin_amdgcn_readlane((int)k, (int)l);
      const uint32_t rs = (uint32_t)__builtin_amdgcn_readlane((int)thr_row, (int)l);
      chg &= ~__ballot(changed && q == qs);
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
      if (gth) {
        if (lane == 0) atomicMax(p.thr_val + qs, gth);
        if ((uint32_t)lane < n_leads && L.lead[lane].query == qs && gth > L.lthr[lane]) L.lthr[lane] = gth;
      }
    }
    te(9u);
    tb(10u);
    // staging lists that could overflow with the next batch are cut back to their k best now
    wave_mem_fence();
    const uint32_t cn = (uint32_t)lane < US_GROUP ? L.cnt[lane] & 0xFFFFu : 0u;
    uint64_t full = __ballot(cn > CAPL - 64u);
    while (full) {
      const uint32_t gs = (uint32_t)__builtin_ctzll(full);
      full &= full - 1ull;
      const uint32_t ns = (uint32_t)__builtin_amdgcn_readlane((int)cn, (int)gs);
      const uint32_t ks = uni(L.lk[gs]) & 0xFFu;
      const uint64_t kth = compact_slot(gs, ns, ks);
      const uint32_t t = (uint32_t)(kth >> 32);
      if ((uint32_t)lane == gs) {
        L.cnt[gs] = (L.cnt[gs] & 0xFFFF0000u) | ks;
        if (t > L.lthr[gs]) L.lthr[gs] = t;
        atomicMax(p.thr_val + L.lead[gs].query, t);  // k distinct docs of this query score >= t
      }
    }
    wave_mem_fence();
    te(10u);
    te(6u);
    tb(5u);
  };

  for (;;) {
    tb(2u);
    uint32_t task = 0;
    if (lane == 0) task = atomicAdd(p.task_counter, 1u);
    task = uni(task) + p.task_begin;
    if (task >= p.n_tasks) break;
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
    if ((uint32_t)lane < n_leads) {
      const TqdLead mine = p.leads[lead0 + lane];
      L.lead[lane] = mine;
      const TqdQuery *Q = p.queries + mine.query;
      L.lk[lane] = Q->k | (Q->thr_index << 8);
      L.lthr[lane] = __hip_atomic_load(p.thr_val + mine.query, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wave_mem_fence();
    if ((uint32_t)lane < US_GROUP) L.cnt[lane] = 0u;
    wave_mem_fence();
    auto lead_alive = [&]() __attribute__((always_inline)) {
      return (uint32_t)lane < n_leads && sortable(L.lead[lane].suffix * 1.000001f) >= L.lthr[lane];
    };
    uint32_t live = (uint32_t)__ballot(lead_alive());
    te(2u);

    for (uint32_t jt = 0; jt < nb_task && live; jt += TQD_US_TILE) {
      // thresholds may have risen since the last step (one word per lead)
      if (jt) {
        if ((uint32_t)lane < n_leads) {
          const uint32_t t = __hip_atomic_load(p.thr_val + L.lead[lane].query, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (t > L.lthr[lane]) L.lthr[lane] = t;
        }
        wave_mem_fence();
        live = (uint32_t)__ballot(lead_alive());
        if (!live) break;
      }
      tb(3u);
      // ---- pre-filter: lane <-> block
      const uint32_t nb = nb_task - jt < TQD_US_TILE ? nb_task - jt : TQD_US_TILE;
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
      uint32_t pass_mask = 0;  // leads that still want this block
      for (uint32_t lm = live; lm; lm &= lm - 1u) {
        const uint32_t g = (uint32_t)__builtin_ctz(lm);
        const float w = L.lead[g].w, suf = L.lead[g].suffix;  // (uniform address: LDS broadcast)
        const uint32_t thr = L.lthr[g];
        const float ub = w * tfn_max * p.bound_slack;
        if (in_tile && sortable((ub + (suf - w)) * 1.000004f + suf * 4.0e-6f) >= thr) pass_mask |= 1u << g;
      }
      uint64_t todo = __ballot(pass_mask != 0u);
      wave_mem_fence();
      L.pmask[lane] = pass_mask;
      wave_mem_fence();
      te(3u);
      uint32_t since_refresh = 0;
      // (Tried: software-pipelining the walk — the next wanted block's record and payload requested
      // under the current block's doc-matrix gathers.  The four registers of the payload in flight
      // cost more in spills than the two hidden round trips won: 2.79..2.98 against 2.81 ms.)
      while (todo) {
        const uint32_t b = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        if (++since_refresh == TQ_US_REFRESH) {  // thresholds rise while the tile is walked: one word per lead
          since_refresh = 0;
          if ((uint32_t)lane < n_leads) {
            const uint32_t t = __hip_atomic_load(p.thr_val + L.lead[lane].query, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t > L.lthr[lane]) L.lthr[lane] = t;
          }
          wave_mem_fence();
          live = (uint32_t)__ballot(lead_alive());
          if (!live) break;
        }
        uint32_t lm = uni(L.pmask[b]) & live;
        if (!lm) continue;
        // the block's record again, by scalar loads (the 64 records of the tile are not kept in registers)
        const uint4 rec_b = sload(lead.rec + (i_base + b));
        const uint32_t prev_l = block_prev_last(lead, i_base + b);
        const uint2 mo_l = make_uint2(rec_b.y, rec_b.z);
        tb(4u);
        if (p.debug & 256u) ++n_scored;  // COUNTERS
        // ---- stage A: decode the block once
        uint32_t c0, c1, t0, t1;
        if (mo_l.x == META_TAIL) {
          decode_docs<USE_DPP>(idx, lead, mo_l, prev_l, lane, c0, c1);
          decode_tfs(idx, lead, mo_l, lane, t0, t1);
        } else {
          const uint32_t doc_bits = mo_l.x & 31u;
          const uint32_t strict = (mo_l.x >> 6) & 1u;
          const uint32_t tf_bits = lead.has_freq ? (mo_l.x >> 8) & 0xFFu : 0u;
          wave_mem_fence();
          stage_payload(L.pay, idx + lead.payload_base + mo_l.y, 16u * (doc_bits + tf_bits), lane);
          wave_mem_fence();
          if (lead.has_freq) {
            unpack2_lds(L.pay + 4u * doc_bits, tf_bits, lane, t0, t1);
            t0 += strict;
            t1 += strict;
          } else {
            t0 = 1u;
            t1 = 1u;
          }
          uint32_t x0, x1;
          unpack2_lds(L.pay, doc_bits, lane, x0, x1);
          finish_docs<USE_DPP>(x0, x1, strict, prev_l, lane, c0, c1);
        }
        const bool v0 = c0 != TQD_TERMINATED, v1 = c1 != TQD_TERMINATED;
        // ONE gather per doc: fieldnorm id + membership in every dense list of the segment
        const uint64_t mw0 = v0 ? seg.docmat[c0] : 0ull;
        const uint64_t mw1 = v1 ? seg.docmat[c1] : 0ull;
        const uint32_t nid0 = (uint32_t)mw0 & 0xFFu, nid1 = (uint32_t)mw1 & 0xFFu;
        const float f0 = (float)t0, f1 = (float)t1;
        wave_mem_fence();
        L.tfn[0][lane] = f0 * __builtin_amdgcn_rcpf(f0 + L.cache[nid0]);
        L.tfn[1][lane] = f1 * __builtin_amdgcn_rcpf(f1 + L.cache[nid1]);
        wave_mem_fence();
        te(4u);
        tb(5u);
        // ---- stage F: every lead that wants the block
        for (; lm; lm &= lm - 1u) {
          const uint32_t g = (uint32_t)__builtin_ctz(lm);
          if (p.debug & 32u) ++n_scored;  // COUNTERS
          // the lead's constants: the first 64 bytes of its record as FOUR 16-byte LDS reads at a uniform address
          // (broadcast) issued together with the threshold word and the two tf/(tf+norm) of the lane — one wait for
          // the lot (field by field the loop waited for LDS five times per (block, lead) pair)
#if TQ_US_FWIDE
          const uint4 *lp = reinterpret_cast<const uint4 *>(&L.lead[g]);
          const uint4 r0 = lp[0], r1 = lp[1], r2 = lp[2], r3 = lp[3];
          const uint32_t thr = uni(L.lthr[g]);
          const float tfn0 = L.tfn[0][lane], tfn1 = L.tfn[1][lane];
          const float w = __uint_as_float(r0.z), suf = __uint_as_float(r0.w), sp = __uint_as_float(r1.x);
          const uint32_t ncols = uni((r0.y >> 4) & 15u);
          const uint32_t b_lo = r3.z, b_hi = r3.w;
          const uint32_t cols_lo = r1.y, cols_hi = r1.z;
          const float awv[7] = {__uint_as_float(r1.w), __uint_as_float(r2.x), __uint_as_float(r2.y), __uint_as_float(r2.z),
                                __uint_as_float(r2.w), __uint_as_float(r3.x), __uint_as_float(r3.y)};
#else  // (A/B: round 3's field-by-field reads)
          const TqdLead &ld = L.lead[g];
          const uint32_t thr = uni(L.lthr[g]);
          const float w = ld.w, suf = ld.suffix, sp = ld.sparse_after;
          const uint32_t ncols = uni((ld.info >> 4) & 15u);
          const uint32_t b_lo = (uint32_t)ld.before_mask, b_hi = (uint32_t)(ld.before_mask >> 32);
          const uint32_t cols_lo = uni(ld.cols_lo), cols_hi = uni(ld.cols_hi);
          const float *awv = ld.aw;
#define tfn0 L.tfn[0][lane]
#define tfn1 L.tfn[1][lane]
#endif
          // "leader score + weights of the later lists that hold / may hold the doc >= threshold" as
          // one fused multiply-add and one float compare per doc: scores are >= 0, so the float
          // order is the order of the sortable bits; the slack that the reciprocal-based tfn and the
          // different summation order need is folded into the threshold once per (block, lead)
          const float thr_f = thr ? __uint_as_float(thr ^ ((thr >> 31) ? 0x80000000u : 0xFFFFFFFFu)) : -1.0f;
          const float need = (thr_f - suf * 4.0e-6f) * 0.999995f;
          float rest0 = sp, rest1 = sp;
#pragma unroll
          for (uint32_t c = 0; c < 7u; ++c) {
            if (c < ncols) {  // (wave-uniform)
              const uint32_t col = ((c < 4u ? cols_lo >> (8u * c) : cols_hi >> (8u * (c - 4u)))) & 0xFFu;
              const float aw = awv[c];
              rest0 = fmaf((float)((uint32_t)(mw0 >> col) & 1u), aw, rest0);
              rest1 = fmaf((float)((uint32_t)(mw1 >> col) & 1u), aw, rest1);
            }
          }
          const bool a0 = v0 && !(((uint32_t)mw0 & b_lo) | ((uint32_t)(mw0 >> 32) & b_hi)) && fmaf(w, tfn0, rest0) >= need;
          const bool a1 = v1 && !(((uint32_t)mw1 & b_lo) | ((uint32_t)(mw1 >> 32) & b_hi)) && fmaf(w, tfn1, rest1) >= need;
          if (!(__ballot(a0) | __ballot(a1))) continue;
          // the two docs of a lane are queued one after the other: the queue holds < 64 leftovers
          // plus <= 64 new entries and is drained below 64 before the next push
#pragma unroll 1
          for (uint32_t e = 0; e < 2u; ++e) {
            const bool a = e ? a1 : a0;
            const uint64_t m = __ballot(a);
            if (!m) continue;
            const uint32_t pos = qn + mbcnt64(m);
            wave_mem_fence();
            if (a) {
              L.q_doc[pos] = e ? c1 : c0;
              L.q_tf[pos] = e ? t1 : t0;
              L.q_tag[pos] = g;
            }
            wave_mem_fence();
            qn += (uint32_t)__popcll(m);
            while (qn >= 64u) stageC(64u);
          }
        }
        te(5u);
      }
    }
    tb(5u);
    while (qn) stageC(qn < 64u ? qn : 64u);
    te(5u);
    tb(11u);

    // ---- flush: the staging lists go to their queries' result lists
    wave_mem_fence();
    const uint32_t cw = (uint32_t)lane < US_GROUP ? L.cnt[lane] : 0u;
    const uint32_t cn = cw & 0xFFFFu, sc = cw >> 16;
    if (sc) atomicAdd(sload(&p.sinks->query_matches) + sload(&p.sinks->out_index)[L.lead[lane].query], sc);
    uint64_t have = __ballot(cn != 0u);
    while (have) {
      const uint32_t gs = (uint32_t)__builtin_ctzll(have);
      have &= have - 1ull;
      uint32_t ns = (uint32_t)__builtin_amdgcn_readlane((int)cn, (int)gs);
      const uint32_t ks = uni(L.lk[gs]) & 0xFFu;
      const uint32_t qs = uni(L.lead[gs].query);
      if (ns > ks) {
        (void)compact_slot(gs, ns, ks);
        ns = ks;
      }
      const uint32_t thr_now = __hip_atomic_load(p.thr_val + qs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
      if (!keep_n) continue;
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(p.list_count + qs, keep_n);
      at = uni(at);
      uint64_t *dst = p.lists + (size_t)sload(&p.queries[qs].part_start) + at;
      uint32_t base = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint64_t m = __ballot(v[r] != 0ull);
        if (v[r] != 0ull) dst[base + mbcnt64(m)] = v[r];
        base += (uint32_t)__popcll(m);
      }
    }
    te(11u);
  }
  te(1u);
  if (tphase) n_scored = (uint32_t)(tacc >> 6);
  if (lane == 0 && n_scored) atomicAdd(sload(&p.sinks->match_counter), (unsigned long long)n_scored);
}

// One wavefront per query: the query's result list (list_count entries) -> sorted top-k.
template <int KPL>
__global__ __launch_bounds__(64) void merge_lists_kernel(TqkMergeParams p, const uint32_t *list_count) {
  const int lane = (int)__lane_id();
  const uint32_t q = blockIdx.x;
  if (q >= p.n_queries) return;
  const TqdQuery *Q = uni_ptr(p.queries + q);
  const uint32_t k = uni(Q->k);
  // (chunk_first: the query that owns the list — q itself, or the identical query of the batch that was
  // evaluated in its place, build_ashare_plan)
  const uint32_t n = uni(list_count[uni(Q->chunk_first)]);
  const uint64_t *src = p.partials + (size_t)uni(Q->part_start);
  TopK<KPL> tk;
  tk.reset(k);
  for (uint32_t i0 = 0; i0 < n; i0 += 256u) {  // four loads in flight per step
    uint64_t keys[4];
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      const uint32_t i = i0 + 64u * u + (uint32_t)lane;
      keys[u] = i < n ? src[i] : 0ull;
    }
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) tk.offer(keys[u] != 0ull, keys[u], lane);
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

}  // namespace

// =================================================================== launch wrappers
uint32_t tqk_share_capl(int kpl) { return (uint32_t)(kpl + 1) * 64u; }

hipError_t tqk_launch_share(const TqkShareParams &p, int kpl, hipStream_t st) {
  if (p.n_tasks <= p.task_begin || p.grid == 0) return hipSuccess;
  const dim3 grid(p.grid), block(64);
  switch (kpl) {
    case 1: ushare_kernel<1><<<grid, block, 0, st>>>(p); break;
    default: ushare_kernel<2><<<grid, block, 0, st>>>(p); break;
  }
  return hipGetLastError();
}

hipError_t tqk_launch_merge_lists(const TqkMergeParams &m, const uint32_t *list_count, int kpl,
                                  hipStream_t st) {
  if (m.n_queries == 0) return hipSuccess;
  const dim3 grid(m.n_queries), block(64);
  switch (kpl) {
    case 1: merge_lists_kernel<1><<<grid, block, 0, st>>>(m, list_count); break;
    default: merge_lists_kernel<2><<<grid, block, 0, st>>>(m, list_count); break;
  }
  return hipGetLastError();
}
