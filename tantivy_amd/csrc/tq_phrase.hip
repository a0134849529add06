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
#ifndef TQ_PH_HOIST
#define TQ_PH_HOIST 0  // 1: the block records of every term's position run are loaded together (one round
                       // trip instead of one per term; 16 more live registers: spills at 96 VGPRs)
#endif
#ifndef TQ_PH_WAVES_DENSE
#define TQ_PH_WAVES_DENSE 4  // occupancy target of the all-dense instantiation (waves per SIMD): the level-parallel
                             // list stage needs ~100 VGPRs (5 waves: 56..84 B of scratch, 2.2..2.5 ms against 1.85)
                             // without scratch beat 80 with 12 B of it by 2 %
#endif
// DENSE: every non-leader list of every query of the launch has a bitmap, a doc-matrix column and
// a position directory (the planner checks): no seek / block-search code, half the staging area
template <int NT_MAX, bool DENSE>
struct PhraseLds {  // per wavefront
  uint32_t pay[DENSE ? 260 : 516];  // stage A's payload / lookup_in_blocks' staging area
  uint32_t q1_doc[DENSE ? 1 : 191], q1_tf[DENSE ? 1 : 191], q1_pi[DENSE ? 1 : 191];
  // (DENSE: stage A tests its 128 docs against the doc matrix itself and feeds queue 2 directly)
  uint32_t q2_doc[DENSE ? 191 : 127], q2_tf[DENSE ? 191 : 127], q2_pi[DENSE ? 191 : 127], q2_loc[DENSE ? 1 : 127];
  uint32_t ph_pi[NT_MAX - 1][64], ph_tf[NT_MAX - 1][64];  // lists 1.. (the leader's stay in q2)
  // per-term position stream tables of the current query, read by broadcast
  const uint64_t *pt_blk[NT_MAX];
  const uint32_t *pt_tail[NT_MAX];
  uint32_t pt_nblk[NT_MAX], pt_off[NT_MAX];
  // DENSE: bitmap + rank directory, byte-wide tfs and position directory of every list
  const uint2 *pt_dense[DENSE ? NT_MAX : 1];
  const uint8_t *pt_tf8[DENSE ? NT_MAX : 1];
  const uint32_t *pt_dir[DENSE ? NT_MAX : 1];
};

// The position deltas [i, i + n), n <= 8, of a term, fetched with independent loads: they lie in
// position block i >> 7 and possibly the next one, each either a bitpacked block (pos_blk record:
// byte offset | width << 56, positions/reader.rs:84-101) or the pre-decoded vint tail.
struct PosRun {
  const uint8_t *base[2];  // block payload, or the tail array
  uint32_t b[2];           // bit width; 0xFFFFFFFF = tail (plain u32 values)
  uint32_t v0;             // index of delta 0 inside block 0 (tail: inside the tail array)
};
// the two block records of a run (loaded unconditionally, clamped: independent loads)
__device__ __forceinline__ void pos_run_records(const uint64_t *pos_blk, uint32_t n_pb, uint32_t i,
                                                uint64_t &e0, uint64_t &e1) {
  const uint32_t pb = i >> 7;
  e0 = pos_blk[pb < n_pb ? pb : 0u];
  e1 = pos_blk[pb + 1u < n_pb ? pb + 1u : 0u];
}
__device__ __forceinline__ PosRun pos_run_of(const uint8_t *pos, uint64_t e0, uint64_t e1,
                                             const uint32_t *pos_tail, uint32_t n_pb, uint32_t i) {
  PosRun r;
  const uint32_t pb = i >> 7;
  const bool t0 = pb >= n_pb, t1 = pb + 1u >= n_pb;
  r.base[0] = t0 ? reinterpret_cast<const uint8_t *>(pos_tail) : pos + (e0 & 0x00FFFFFFFFFFFFFFull);
  r.b[0] = t0 ? 0xFFFFFFFFu : (uint32_t)(e0 >> 56);
  r.v0 = t0 ? i - (n_pb << 7) : (i & 127u);
  // block 1 is indexed from v0 as well: index v of block 0 is index v - 128 of block 1; as a tail
  // it starts at element 0 of the tail array
  r.base[1] = t1 ? reinterpret_cast<const uint8_t *>(pos_tail) : pos + (e1 & 0x00FFFFFFFFFFFFFFull);
  r.b[1] = t1 ? 0xFFFFFFFFu : (uint32_t)(e1 >> 56);
  return r;
}
__device__ __forceinline__ PosRun pos_run(const uint8_t *pos, const uint64_t *pos_blk,
                                          const uint32_t *pos_tail, uint32_t n_pb, uint32_t i) {
  uint64_t e0, e1;
  pos_run_records(pos_blk, n_pb, i, e0, e1);
  return pos_run_of(pos, e0, e1, pos_tail, n_pb, i);
}
// delta k of the run (k clamped by the caller): two independent 4-byte loads
__device__ __forceinline__ uint32_t pos_run_delta(const PosRun &r, uint32_t k) {
  uint32_t v = r.v0 + k;
  const bool second = r.b[0] != 0xFFFFFFFFu && v >= 128u;  // (a tail never overflows)
  if (second) v -= 128u;
  const uint8_t *base = second ? r.base[1] : r.base[0];
  const uint32_t b = second ? r.b[1] : r.b[0];
  const bool tail = b == 0xFFFFFFFFu;
  const uint32_t bb = tail ? 32u : b;
  const uint32_t bitpos = tail ? 0u : (v >> 2) * bb;
  const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
  const uint8_t *q = tail ? base + 4u * v : base + 16u * w + 4u * (v & 3u);
  const uint32_t lo = ld_u1(q), hi = ld_u1(q + 16);  // hi may over-read: buffers are padded
  const uint32_t mask = bb >= 32u ? 0xFFFFFFFFu : ((1u << bb) - 1u);
  return bb ? (__funnelshift_r(lo, tail ? 0u : hi, sh) & mask) : 0u;
}

// the largest x (1..8) of the lanes that are `on`, 0 if none: wave-uniform, so that the position loops below run to
// the longest run of the 64 candidates instead of to 8 (term freqs are mostly 1 or 2: the unrolled 8 x 8 compare and
// the 8 clamped delta fetches per term were two thirds of the sweep kernel's vector instructions)
#ifndef TQ_PH_KMAX
#define TQ_PH_KMAX 1  // 0: the loops run to 8 whatever the candidates hold (rounds 2-5)
#endif
__device__ __forceinline__ uint32_t wave_max_le8(uint32_t x, bool on) {
  if (!TQ_PH_KMAX) return 8u;
  uint32_t m = 0;
#pragma unroll
  for (uint32_t v = 1; v <= 8u; ++v)
    if (__ballot(on && x >= v)) m = v;
  return m;
}

// (5 waves per SIMD: what the 8 KB of LDS per wavefront admit)
template <int KPL, int NT_MAX, bool DENSE>
__global__ __launch_bounds__(64, DENSE ? TQ_PH_WAVES_DENSE : 5) void phrase_kernel(TqkScanParams p) {
  constexpr bool USE_DPP = true;
  __shared__ PhraseLds<NT_MAX, DENSE> L;
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
  uint64_t mat_need = 0;  // != 0: stage B tests every non-leader list with one doc-matrix word
  // PROFILING (TQ_DEBUG bits 16..19 = phase): wave cycles of one phase, summed into the match
  // counter.  1 setup + flush, 2 pre-filter, 3 stage A, 4 stage B, 5 stage C lists, 6 positions,
  // 7 collector
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
    weight = sload(&Q->weight[0]);
    // the tf cache stays in global memory: only phrase matches are scored (7.9 KB of LDS per
    // wavefront instead of 9.5 KB: 5 waves/SIMD)
    cache_g = p.caches + (size_t)sload(&Q->cache_idx) * 256u;
    tk.reset(sload(&Q->k));
    // per-term tables (one term per lane) and the doc-matrix columns of the non-leader lists
    wave_mem_fence();
    uint32_t slot = 0xFFFFFFFFu;
    if ((uint32_t)lane < nt) {
      const TqdTerm *tt = p.terms + Q->term[lane];
      L.pt_blk[lane] = tt->pos_blk;
      L.pt_tail[lane] = tt->pos_tail;
      L.pt_nblk[lane] = tt->n_pos_blocks;
      L.pt_off[lane] = Q->phrase_off[lane];
      if constexpr (DENSE) {
        L.pt_dense[lane] = tt->dense;
        L.pt_tf8[lane] = tt->tf8;
        L.pt_dir[lane] = tt->pos_dir;
      }
      if (lane && seg.docmat && tt->dense) slot = ((tt->has_freq >> 8) & 0xFFu) - 1u;
    }
    wave_mem_fence();
    const uint64_t have = __ballot(slot < TQD_MAT_SLOTS);
    mat_need = 0;
    if (seg.docmat && have == (((1ull << nt) - 1ull) & ~1ull)) {  // every non-leader list has a column
      uint64_t need = slot < TQD_MAT_SLOTS ? 1ull << (8u + slot) : 0ull;
      for (int o = 32; o; o >>= 1)
        need |= ((uint64_t)(uint32_t)__shfl_xor((int)(need >> 32), o, WAVE) << 32) |
                (uint32_t)__shfl_xor((int)(uint32_t)need, o, WAVE);
      mat_need = uni64(need);
    }
  };

  // ---- stage C: the other lists' postings of the candidate, then the positions
  auto stageC = [&](uint32_t n) __attribute__((always_inline)) {
    tick(4u);
    const uint32_t base = q2n - n;
    q2n = base;
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, loc = 0, lead_tf = 0, lead_pi = 0;
    if (alive) {
      doc = L.q2_doc[base + lane];
      if constexpr (!DENSE) loc = L.q2_loc[base + lane];
      lead_tf = L.q2_tf[base + lane];
      lead_pi = L.q2_pi[base + lane];
    }
    if constexpr (DENSE) {
      // Every other list has a bitmap, byte-wide tfs and a position directory: the chain per list
      // is bitmap word (posting index) -> {the four tf bytes of its group, the group's directory
      // entry}, and the levels of ALL lists are in flight together: two dependent round trips for
      // the whole candidate (it was bitmap word -> block record -> packed tf row, list after list).
      uint2 wd[NT_MAX - 1];
#pragma unroll
      for (int m = 1; m < NT_MAX; ++m) {
        wd[m - 1] = make_uint2(0u, 0u);
        if ((uint32_t)m < nt && alive) wd[m - 1] = L.pt_dense[m][doc >> 5];
      }
      uint32_t piv[NT_MAX - 1], tw[NT_MAX - 1], dv[NT_MAX - 1];
#pragma unroll
      for (int m = 1; m < NT_MAX; ++m) {
        const uint32_t bit = doc & 31u;
        if ((uint32_t)m < nt) alive = alive && ((wd[m - 1].x >> bit) & 1u);
        piv[m - 1] = wd[m - 1].y + (uint32_t)__popc(wd[m - 1].x & ((1u << bit) - 1u));
      }
#pragma unroll
      for (int m = 1; m < NT_MAX; ++m) {
        tw[m - 1] = 0;
        dv[m - 1] = 0;
        if ((uint32_t)m < nt && alive) {
          tw[m - 1] = *reinterpret_cast<const uint32_t *>(L.pt_tf8[m] + (piv[m - 1] & ~3u));
          dv[m - 1] = L.pt_dir[m][piv[m - 1] >> 2];
        }
      }
      if (!__ballot(alive)) {
        tick(5u);
        return;
      }
#pragma unroll
      for (int m = 1; m < NT_MAX; ++m) {
        if ((uint32_t)m < nt) {
          const uint32_t l0 = piv[m - 1] & 3u;
          const uint32_t b0 = tw[m - 1] & 0xFFu, b1 = (tw[m - 1] >> 8) & 0xFFu, b2 = (tw[m - 1] >> 16) & 0xFFu,
                         b3 = tw[m - 1] >> 24;
          uint32_t tf = l0 == 0u ? b0 : (l0 == 1u ? b1 : (l0 == 2u ? b2 : b3));
          uint32_t ex = (l0 > 0u ? b0 : 0u) + (l0 > 1u ? b1 : 0u) + (l0 > 2u ? b2 : 0u);
          // a saturated byte (tf >= 255) among the ones used: the packed values are read instead
          const bool sat = alive && (tf == 255u || (l0 > 0u && b0 == 255u) || (l0 > 1u && b1 == 255u) ||
                                     (l0 > 2u && b2 == 255u));
          if (__ballot(sat)) {
            const TermRef tr = load_term(p.terms, sload(&Q->term[m]));
            if (sat) group_tfs(idx, tr, tr.rec[piv[m - 1] >> 7], piv[m - 1] & 127u, tf, ex);
          }
          if (alive) {
            L.ph_tf[m - 1][lane] = tf;
            L.ph_pi[m - 1][lane] = dv[m - 1] + ex;
          }
        }
      }
    }
    for (uint32_t m = 1; !DENSE && m < nt; ++m) {
      TermRef tr = m == 1u ? t1 : load_term(p.terms, sload(&Q->term[m]));
      if (!p.use_dense) tr.dense = nullptr;
      uint32_t jb = 0, at = NOT_FOUND;
      if (!DENSE && m == 1u && !mat_need) {
        if (tr.dense) {
          jb = loc >> 7;
          at = loc & 127u;
        } else {
          jb = loc;
        }
      } else if (DENSE || tr.dense) {
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
      if constexpr (!DENSE) {
        if (!tr.dense) {
          uint32_t unused;
          at = lookup_in_blocks<false>(idx, tr, jb, doc, alive, L.pay, lane, &unused);
          alive = alive && at != NOT_FOUND;
        }
      }
      if (!__ballot(alive)) {
        tick(5u);
        return;
      }
      uint32_t excl = 0;
      uint32_t tf = 1;
      const uint32_t *pdir =
          (DENSE || tr.dense) ? sload(&(p.terms + sload(&Q->term[m]))->pos_dir) : nullptr;
      if (DENSE || pdir) {
        // dense list: posting index = 128 * jb + at (from the bitmap's rank); its first position
        // index = directory entry of its group of four + the term freqs before it in the group
        if (alive) {
          const uint4 r = tr.rec[jb];
          const uint32_t dirv = pdir[(jb << 5) + (at >> 2)];
          uint32_t ex;
          group_tfs(idx, tr, r, at, tf, ex);
          L.ph_tf[m - 1u][lane] = tf;
          L.ph_pi[m - 1u][lane] = dirv + ex;
        }
      } else if constexpr (!DENSE) {
        if (!(p.debug & 2u)) tf = lookup_in_blocks<true>(idx, tr, jb, at, alive, L.pay, lane, &excl);
        if (alive) {
          L.ph_tf[m - 1u][lane] = tf;
          L.ph_pi[m - 1u][lane] = tr.rec[jb].w + excl;
        }
      }
    }
    // the n-way cursor merge over adjusted positions (phrase_scorer.rs:372-385,437-461), one
    // position fetched at a time: candidates with a long position list or a run that crosses a
    // position block
    auto phrase_count_serial = [&]() __attribute__((always_inline)) -> uint32_t {
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
          cur[m].valid = cur[m].end >= cur[m].idx;  // tf >= 1 (0 only in a corrupt index: no position)
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
      return count;
    };
    tick(5u);
    // ---- position check, one lane per candidate
    bool has = false;
    uint64_t key = 0;
    if (alive && (p.debug & 3u)) {
      has = true;
      key = make_key(1.0f, doc);
    } else if (NT_MAX <= 4 && alive) {
      // Fast path (every term's tf <= 8 — all but 2e-4 of the candidates): the deltas of a term
      // are fetched with independent loads (one round trip per term instead of one per position)
      // and the adjusted positions intersected in registers.
      // Everything else takes the cursor merge below.
      constexpr uint32_t TM = 8;
      // (tf - 1 < TM: a term freq of 0 — only a corrupt index has one — must not reach the clamped
      // `tf - 1` delta index below; it takes the cursor merge, which finds no position)
      bool fast = lead_tf - 1u < TM;
      for (uint32_t m = 1; m < nt; ++m) fast = fast && L.ph_tf[m - 1u][lane] - 1u < TM;
      uint32_t count = 0xFFFFFFFFu;  // = resolved by the cursor merge
      const uint32_t km0 = wave_max_le8(lead_tf, fast);
      if (fast) {
        // adjusted positions of the leader term, then one term at a time: bit i of `ok` stays set
        // while every term seen so far has a position equal to the leader's i-th
        uint32_t a[TM], d[TM];
#pragma unroll
        for (uint32_t k = 0; k < TM; ++k) a[k] = d[k] = 0;
        // the block records of every term's run first (one round trip for all of them)
        uint64_t e0[NT_MAX], e1[NT_MAX];
        uint32_t pis[NT_MAX];
#pragma unroll
        for (int m = 0; m < NT_MAX; ++m) {
          pis[m] = 0;
          e0[m] = 0;
          e1[m] = 0;
          if ((uint32_t)m < nt) {
            pis[m] = m ? L.ph_pi[m ? m - 1 : 0][lane] : lead_pi;
            if (TQ_PH_HOIST) pos_run_records(L.pt_blk[m], L.pt_nblk[m], pis[m], e0[m], e1[m]);
          }
        }
        {
          if (!TQ_PH_HOIST) pos_run_records(L.pt_blk[0], L.pt_nblk[0], lead_pi, e0[0], e1[0]);
          const PosRun run = pos_run_of(pos, e0[0], e1[0], L.pt_tail[0], L.pt_nblk[0], lead_pi);
#pragma unroll
          for (uint32_t k = 0; k < TM; ++k) {  // clamped: unconditional, independent loads
            if (k >= km0) break;               // (wave-uniform)
            d[k] = pos_run_delta(run, k < lead_tf ? k : lead_tf - 1u);
          }
          uint32_t c = L.pt_off[0];
#pragma unroll
          for (uint32_t k = 0; k < TM; ++k) {
            if (k >= km0) break;
            c += d[k];
            a[k] = c;
          }
        }
        uint32_t ok = (1u << lead_tf) - 1u;
#pragma unroll
        for (int m = 1; m < NT_MAX; ++m) {
          if ((uint32_t)m >= nt) break;
          const uint32_t tfm = L.ph_tf[m - 1][lane];
          if (!TQ_PH_HOIST) pos_run_records(L.pt_blk[m], L.pt_nblk[m], pis[m], e0[m], e1[m]);
          const PosRun run = pos_run_of(pos, e0[m], e1[m], L.pt_tail[m], L.pt_nblk[m], pis[m]);
          const uint32_t kmm = wave_max_le8(tfm, true);  // (inside `if (fast)`: the lanes here are the fast ones)
#pragma unroll
          for (uint32_t k = 0; k < TM; ++k) {
            if (k >= kmm) break;
            d[k] = pos_run_delta(run, k < tfm ? k : tfm - 1u);
          }
          uint32_t c = L.pt_off[m], hit = 0;
#pragma unroll
          for (uint32_t k = 0; k < TM; ++k) {
            if (k >= kmm) break;
            c += d[k];  // (k >= tfm repeats the last delta; those sums are masked out below)
            uint32_t eq = 0;
#pragma unroll
            for (uint32_t i = 0; i < TM; ++i) {
              if (i >= km0) break;
              eq |= (a[i] == c ? 1u : 0u) << i;
            }
            hit |= k < tfm ? eq : 0u;
          }
          ok &= hit;
        }
        count = (uint32_t)__popc(ok);
      }
      if (__ballot(count == 0xFFFFFFFFu)) {
        if (count == 0xFFFFFFFFu) count = phrase_count_serial();
      }
      if (count > 0 && doc_is_alive(seg, doc)) {
        has = true;
        key = make_key(bm25(weight, cache_g[fieldnorm_id(seg, doc)], count), doc);
      }
    } else if (alive) {
      const uint32_t count = phrase_count_serial();
      if (count > 0 && doc_is_alive(seg, doc)) {
        has = true;
        key = make_key(bm25(weight, cache_g[fieldnorm_id(seg, doc)], count), doc);
      }
    }
    tick(6u);
    const uint64_t hit = __ballot(has);
    if (hit) {
      n_matches += (uint32_t)__popcll(hit);
      n_q += (uint32_t)__popcll(hit);
      tk.offer(has, key, lane);
    }
    tick(7u);
  };

  // ---- stage B: locate the candidate in list 1
  auto stageB = [&](uint32_t n) __attribute__((always_inline)) {
    tick(3u);
    const uint32_t base = q1n - n;
    q1n = base;
    bool alive = (uint32_t)lane < n;
    uint32_t doc = 0, tf = 0, pi0 = 0, loc = 0;
    if (alive) {
      doc = L.q1_doc[base + lane];
      tf = L.q1_tf[base + lane];
      pi0 = L.q1_pi[base + lane];
    }
    if (DENSE || mat_need) {  // one gather answers every other list (all have bitmaps + columns)
      if (alive) alive = (seg.docmat[doc] & mat_need) == mat_need;
    } else if (t1.dense) {
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
        if constexpr (!DENSE) L.q2_loc[at] = loc;
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
    tick(2u);
    // ---- stage A per surviving leader block
    while (todo) {
      const uint32_t b = (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      const uint2 mo_l = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)rec_mine.y, (int)b),
                                    (uint32_t)__builtin_amdgcn_readlane((int)rec_mine.z, (int)b));
      const uint32_t bp = (uint32_t)__builtin_amdgcn_readlane((int)rec_mine.w, (int)b);
      const uint32_t prev_l = (uint32_t)__builtin_amdgcn_readlane((int)prev_mine, (int)b);
      uint32_t c0, c1, t0, t1f;
      if (mo_l.x == META_TAIL) {
        decode_docs<USE_DPP>(idx, lead, mo_l, prev_l, lane, c0, c1);
        decode_tfs(idx, lead, mo_l, lane, t0, t1f);  // tail padding reads as tf 0
      } else {
        // one 16-byte load per lane brings the whole doc + tf payload (<= 1008 B) into LDS
        const uint32_t doc_bits = mo_l.x & 31u;
        const uint32_t strict = (mo_l.x >> 6) & 1u;
        const uint32_t tf_bits = lead.has_freq ? (mo_l.x >> 8) & 0xFFu : 0u;
        wave_mem_fence();
        stage_payload(L.pay, idx + lead.payload_base + mo_l.y, 16u * (doc_bits + tf_bits), lane);
        wave_mem_fence();
        uint32_t x0, x1;
        unpack2_lds(L.pay, doc_bits, lane, x0, x1);
        finish_docs<USE_DPP>(x0, x1, strict, prev_l, lane, c0, c1);
        if (lead.has_freq) {
          unpack2_lds(L.pay + 4u * doc_bits, tf_bits, lane, t0, t1f);
          t0 += strict;  // minus-one encoding is tied to the strict flag
          t1f += strict;
        } else {
          t0 = 1u;
          t1f = 1u;
        }
      }
      const uint32_t ssum = t0 + t1f;
      const uint32_t incl = wave_inclusive_scan<USE_DPP>(ssum, lane);
      const uint32_t e0 = bp + (incl - ssum), e1 = e0 + t0;
      bool alive0 = c0 != TQD_TERMINATED, alive1 = c1 != TQD_TERMINATED;
      if constexpr (DENSE) {
        // every doc of the block is a candidate and one doc-matrix word answers all the other
        // lists: the two gathers run with all lanes live, no queue in between
        const uint64_t w0 = alive0 ? seg.docmat[c0] : 0ull;
        const uint64_t w1 = alive1 ? seg.docmat[c1] : 0ull;
        alive0 = alive0 && (w0 & mat_need) == mat_need;
        alive1 = alive1 && (w1 & mat_need) == mat_need;
        const uint64_t k0 = __ballot(alive0), k1 = __ballot(alive1);
        if (!(k0 | k1)) continue;
        const uint32_t n0 = (uint32_t)__popcll(k0);
        const uint32_t pos0 = q2n + mbcnt64(k0);
        const uint32_t pos1 = q2n + n0 + mbcnt64(k1);
        wave_mem_fence();
        if (alive0) {
          L.q2_doc[pos0] = c0;
          L.q2_tf[pos0] = t0;
          L.q2_pi[pos0] = e0;
        }
        if (alive1) {
          L.q2_doc[pos1] = c1;
          L.q2_tf[pos1] = t1f;
          L.q2_pi[pos1] = e1;
        }
        wave_mem_fence();
        q2n += n0 + (uint32_t)__popcll(k1);
        while (q2n >= 64u) stageC(64u);
        continue;
      }
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
  tick(1u);
  if (tphase) n_matches = (uint32_t)(tacc >> 4);
  if (lane == 0 && n_matches) atomicAdd(sload(&p.sinks->match_counter), (unsigned long long)n_matches);
}


// =================================================================== phrase sweep (all lists dense)
// Exact phrases whose lists ALL have a bitmap + rank directory, byte-wide term freqs and a position
// directory, and whose rarest list still holds about a posting per bitmap word: the doc set of
// PhraseScorer's intersection (phrase_scorer.rs:82-136) is the AND of the lists' bitmaps, read as
// coalesced 8-byte words — 32 docs per load and list — instead of one doc-matrix gather per leader
// posting (the block-driven kernel above spends 61 % of its time there: 427 M gathers per
// 1000-query batch).  No posting block is decoded at all: a surviving doc's posting index in
// every list falls out of the rank directory, its tf and the index of its first position out of
// the tf bytes of its group of four and the position directory.  Tile = SWEEP_WORDS bitmap words.
// Stage C (64 docs, one per lane): {four tf bytes, directory entry} of every list in flight
// together, then the position runs (block records of all terms together, deltas term by term)
// and the n-way intersection of adjusted positions over registers (phrase_scorer.rs:437-507);
// long runs take the cursor merge.
constexpr uint32_t SWEEP_NT = 4;       // terms per phrase in this kernel
constexpr uint32_t SWEEP_WORDS = 2048; // bitmap words per tile (65536 docs)
#ifndef TQ_PH_SWEEP_UNROLL
#define TQ_PH_SWEEP_UNROLL 2  // (4 and 8: the word arrays end in scratch, 3.9 ms; 1: 1.67 ms; 2: 1.53 ms)
#endif
#ifndef TQ_PH_SWEEP_WAVES
#define TQ_PH_SWEEP_WAVES 5
#endif
constexpr uint32_t SWEEP_UNROLL = TQ_PH_SWEEP_UNROLL;  // 64-word steps whose loads are in flight together
struct SweepLds {  // per wavefront
  uint32_t q_doc[191];  // (< 64 leftovers + up to 64 new docs per extraction step)
  uint32_t q0_doc[127]; // pruned mode with tf classes: the extraction's docs before the class test (stage P)
  uint32_t cls_sh[SWEEP_NT];  // ... and the shifts of the phrase's lists in a doc's class word
  const uint32_t *bits[SWEEP_NT];
  const uint2 *dense[SWEEP_NT];
  const uint8_t *tf8[SWEEP_NT];
  const uint32_t *dir[SWEEP_NT];
  const uint64_t *blk[SWEEP_NT];
  const uint32_t *tail[SWEEP_NT];
  uint32_t nblk[SWEEP_NT], off[SWEEP_NT];
};

template <int KPL>
__global__ __launch_bounds__(64, TQ_PH_SWEEP_WAVES) void phrase_sweep_kernel(TqkScanParams p) {
  __shared__ SweepLds L;
  const int lane = (int)__lane_id();
  if (blockIdx.x >= p.n_chunks) return;
  const uint4 crec = sload(p.chunk_recs + blockIdx.x);
  const uint32_t chunk = crec.w, t_begin = crec.x, t_end = crec.y;
  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;
  const uint8_t *pos = seg.pos;
  const uint32_t n_words = (seg.max_doc + 31u) >> 5;
  uint32_t q = crec.z;
  uint32_t q_tile_start = 0, q_tile_end = 0;
  const TqdQuery *Q = nullptr;
  uint32_t nt = 0;
  float weight = 0.0f;
  const float *cache_g = nullptr;
  TopK<KPL> tk;
  uint32_t n_matches = 0, n_q = 0, qn = 0;
  // pruned mode: the query's shared threshold slots (the other pruned kernels' protocol: every match that beats the
  // bound does a fire-and-forget atomicMax into slot[hash(doc)]; the k-th largest slot is a lower bound of the final
  // k-th best score) and the bound itself, sortable score bits
  uint32_t *slots = nullptr;
  uint32_t thr_g = 0;
  // ... and, when every list of the phrase has its tf classes in the segment's class matrix (TqdSegment::doccls), the
  // shifts of its lists in a doc's class word: stage P bounds a candidate with ONE 8-byte gather + its fieldnorm byte
  bool cls_ok = false;
  uint32_t q0n = 0;

  auto setup_query = [&]() __attribute__((always_inline)) {
    q_tile_start = sload(p.tile_starts + q);
    q_tile_end = sload(p.tile_starts + q + 1u);
    Q = p.queries + q;
    nt = sload(&Q->n_terms);
    weight = sload(&Q->weight[0]);
    cache_g = p.caches + (size_t)sload(&Q->cache_idx) * 256u;
    {
      const uint32_t thr_index = sload(&Q->thr_index);
      const bool prune = !p.exhaustive && (sload(&Q->flags) & TQD_QF_PRUNE) != 0u && thr_index != 0xFFFFFFFFu;
      slots = prune ? p.thr_slots + (size_t)thr_index * TQD_THR_SLOTS : nullptr;
      thr_g = 0;
      cls_ok = slots != nullptr && seg.doccls != nullptr;
      for (uint32_t m = 0; m < nt; ++m) {
        const uint32_t hf = sload(&(p.terms + sload(&Q->term[m]))->has_freq);
        cls_ok = cls_ok && ((hf >> 24) & 1u);
        if (lane == 0) L.cls_sh[m] = 2u * (((hf >> 8) & 0xFFu) - 1u);
      }
    }
    tk.reset(sload(&Q->k));
    wave_mem_fence();
    if ((uint32_t)lane < nt) {
      const TqdTerm *tt = p.terms + Q->term[lane];
      L.bits[lane] = tt->bits;
      L.dense[lane] = tt->dense;
      L.tf8[lane] = tt->tf8;
      L.dir[lane] = tt->pos_dir;
      L.blk[lane] = tt->pos_blk;
      L.tail[lane] = tt->pos_tail;
      L.nblk[lane] = tt->n_pos_blocks;
      L.off[lane] = Q->phrase_off[lane];
    }
    wave_mem_fence();
  };

  auto stageC = [&](uint32_t n) __attribute__((always_inline)) {
    const uint32_t base = qn - n;
    qn = base;
    const bool alive = (uint32_t)lane < n;
    uint32_t doc = 0;
    uint32_t pi[SWEEP_NT] = {0u, 0u, 0u, 0u};
    if (alive) doc = L.q_doc[base + lane];
    const uint32_t nid = alive ? fieldnorm_id(seg, doc) : 0u;  // (requested with the level-0 words)
    {
      // level 0: the doc's {bits, rank} word of every list (the sweep itself streams the bits alone) -> posting index
      uint2 wd[SWEEP_NT];
#pragma unroll
      for (uint32_t m = 0; m < SWEEP_NT; ++m) {
        wd[m] = make_uint2(0u, 0u);
        if (m < nt && alive) wd[m] = L.dense[m][doc >> 5];
      }
      const uint32_t below = (1u << (doc & 31u)) - 1u;
#pragma unroll
      for (uint32_t m = 0; m < SWEEP_NT; ++m) pi[m] = wd[m].y + (uint32_t)__popc(wd[m].x & below);
    }
    // level 1: the four tf bytes of the posting's group + the group's directory entry, every list
    uint32_t tw[SWEEP_NT], dv[SWEEP_NT];
#pragma unroll
    for (uint32_t m = 0; m < SWEEP_NT; ++m) {
      tw[m] = 0;
      dv[m] = 0;
      if (m < nt && alive) {
        tw[m] = *reinterpret_cast<const uint32_t *>(L.tf8[m] + (pi[m] & ~3u));
        dv[m] = L.dir[m][pi[m] >> 2];
      }
    }
    uint32_t tf[SWEEP_NT], fp[SWEEP_NT];  // term freq, index of the first position
    constexpr uint32_t TM = 8;
    bool fast = alive;
#pragma unroll
    for (uint32_t m = 0; m < SWEEP_NT; ++m) {
      tf[m] = 1;
      fp[m] = 0;
      if (m < nt) {
        const uint32_t l0 = pi[m] & 3u;
        const uint32_t b0 = tw[m] & 0xFFu, b1 = (tw[m] >> 8) & 0xFFu, b2 = (tw[m] >> 16) & 0xFFu, b3 = tw[m] >> 24;
        uint32_t t = l0 == 0u ? b0 : (l0 == 1u ? b1 : (l0 == 2u ? b2 : b3));
        uint32_t ex = (l0 > 0u ? b0 : 0u) + (l0 > 1u ? b1 : 0u) + (l0 > 2u ? b2 : 0u);
        // a saturated byte (tf >= 255) among the ones used: the packed values are read instead
        const bool sat = alive && (t == 255u || (l0 > 0u && b0 == 255u) || (l0 > 1u && b1 == 255u) ||
                                   (l0 > 2u && b2 == 255u));
        if (__ballot(sat)) {
          const TermRef tr = load_term(p.terms, sload(&Q->term[m]));
          if (sat) group_tfs(idx, tr, tr.rec[pi[m] >> 7], pi[m] & 127u, t, ex);
        }
        tf[m] = t;
        fp[m] = dv[m] + ex;
        fast = fast && t - 1u < TM;  // (tf 0 — a corrupt index — takes the cursor merge: no position)
      }
    }
    // pruned mode: count <= min tf, so bm25(weight, norm, min tf) bounds the doc — below the query's threshold its
    // positions are never read (1.000002: the bound is compared across a different rounding path than the score)
    bool walk = alive;
    if (slots) {
      uint32_t mintf = tf[0];
#pragma unroll
      for (uint32_t m = 1; m < SWEEP_NT; ++m)
        if (m < nt) mintf = tf[m] < mintf ? tf[m] : mintf;
      uint32_t thr_here = thr_g;
      if (tk.thr) {
        const uint32_t own = (uint32_t)(tk.thr >> 32);
        thr_here = own > thr_here ? own : thr_here;
      }
      if (walk && mintf && sortable(bm25(weight, cache_g[nid], mintf) * 1.000002f) < thr_here) walk = false;
      if (!__ballot(walk)) return;
    }
    fast = fast && walk;
    uint32_t count = 0xFFFFFFFFu;  // = resolved by the cursor merge
    const uint32_t km0 = wave_max_le8(tf[0], fast);
    if (fast) {
      // level 2: the block records of every term's run; then the deltas, term by term
      uint64_t e0[SWEEP_NT], e1[SWEEP_NT];
#pragma unroll
      for (uint32_t m = 0; m < SWEEP_NT; ++m) {
        e0[m] = 0;
        e1[m] = 0;
        if (m < nt) pos_run_records(L.blk[m], L.nblk[m], fp[m], e0[m], e1[m]);
      }
      uint32_t a[TM], d[TM];
#pragma unroll
      for (uint32_t k = 0; k < TM; ++k) a[k] = d[k] = 0;
      {
        const PosRun run = pos_run_of(pos, e0[0], e1[0], L.tail[0], L.nblk[0], fp[0]);
#pragma unroll
        for (uint32_t k = 0; k < TM; ++k) {
          if (k >= km0) break;  // (wave-uniform: the longest run among the 64 candidates)
          d[k] = pos_run_delta(run, k < tf[0] ? k : tf[0] - 1u);
        }
        uint32_t c = L.off[0];
#pragma unroll
        for (uint32_t k = 0; k < TM; ++k) {
          if (k >= km0) break;
          c += d[k];
          a[k] = c;
        }
      }
      uint32_t ok = (1u << tf[0]) - 1u;
#pragma unroll
      for (uint32_t m = 1; m < SWEEP_NT; ++m) {
        if (m >= nt) break;
        const PosRun run = pos_run_of(pos, e0[m], e1[m], L.tail[m], L.nblk[m], fp[m]);
        const uint32_t kmm = wave_max_le8(tf[m], true);  // (inside `if (fast)`)
#pragma unroll
        for (uint32_t k = 0; k < TM; ++k) {
          if (k >= kmm) break;
          d[k] = pos_run_delta(run, k < tf[m] ? k : tf[m] - 1u);
        }
        uint32_t c = L.off[m], hit = 0;
#pragma unroll
        for (uint32_t k = 0; k < TM; ++k) {
          if (k >= kmm) break;
          c += d[k];  // (k >= tf repeats the last delta; those sums are masked out below)
          uint32_t eq = 0;
#pragma unroll
          for (uint32_t i = 0; i < TM; ++i) {
            if (i >= km0) break;
            eq |= (a[i] == c ? 1u : 0u) << i;
          }
          hit |= k < tf[m] ? eq : 0u;
        }
        ok &= hit;
      }
      count = (uint32_t)__popc(ok);
    }
    if (__ballot(walk && count == 0xFFFFFFFFu)) {
      if (walk && count == 0xFFFFFFFFu) {  // the n-way cursor merge, one position at a time
        PosCursor cur[SWEEP_NT];
#pragma unroll
        for (uint32_t m = 0; m < SWEEP_NT; ++m) {
          cur[m].valid = false;
          cur[m].idx = cur[m].end = cur[m].cur = 0;
          if (m < nt) {
            cur[m].idx = fp[m] + 1u;
            cur[m].end = fp[m] + tf[m];
            cur[m].cur = L.off[m] + position_delta(pos, p.terms + Q->term[m], fp[m]);
            cur[m].valid = cur[m].end >= cur[m].idx;
          }
        }
        uint32_t cnt = 0;
        bool done = false;
        while (cur[0].valid && !done) {
          const uint32_t av = cur[0].cur;
          bool okv = true;
#pragma unroll
          for (uint32_t m = 1; m < SWEEP_NT; ++m) {
            if (m < nt && !done) {
              while (cur[m].valid && cur[m].cur < av) pos_advance(cur[m], pos, p.terms + Q->term[m]);
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
            for (uint32_t m = 1; m < SWEEP_NT; ++m)
              if (m < nt) pos_advance(cur[m], pos, p.terms + Q->term[m]);
          }
          pos_advance(cur[0], pos, p.terms + Q->term[0]);
        }
        count = cnt;
      }
    }
    bool has = false;
    uint64_t key = 0;
    if (walk && count > 0u && doc_is_alive(seg, doc)) {
      has = true;
      key = make_key(bm25(weight, cache_g[nid], count), doc);
    }
    const uint64_t hit = __ballot(has);
    if (hit) {
      n_matches += (uint32_t)__popcll(hit);
      n_q += (uint32_t)__popcll(hit);
      if (slots) {
        const uint32_t sb = (uint32_t)(key >> 32);
        if (has && sb > thr_g) atomicMax(slots + ((doc * 0x9E3779B1u) >> 26), sb);
      }
      tk.offer(has, key, lane);
    }
  };

  // ---- stage P (pruned mode, every list has tf classes): 64 docs of the AND, one per lane — ONE class word + the
  // fieldnorm byte per doc: count <= min tf, so bm25(weight, norm, min tf) bounds the doc; below the query's threshold it
  // never reaches the scoring stage (ten gathers: rank words, tf bytes, directory entries).  Class 3 = "three or more":
  // no bound.  Survivors -> q_doc.
  auto stageP = [&](uint32_t n) __attribute__((always_inline)) {
    const uint32_t base = q0n - n;
    q0n = base;
    const bool alive = (uint32_t)lane < n;
    uint32_t doc = 0;
    if (alive) doc = L.q0_doc[base + lane];
    const uint64_t cw = alive ? seg.doccls[doc] : 0ull;
    const uint32_t nid = alive ? fieldnorm_id(seg, doc) : 0u;
    uint32_t mincls = 3u;
#pragma unroll
    for (uint32_t m = 0; m < SWEEP_NT; ++m)
      if (m < nt) {
        const uint32_t c = (uint32_t)(cw >> L.cls_sh[m]) & 3u;
        mincls = c < mincls ? c : mincls;
      }
    uint32_t thr_here = thr_g;
    if (tk.thr) {
      const uint32_t own = (uint32_t)(tk.thr >> 32);
      thr_here = own > thr_here ? own : thr_here;
    }
    // (mincls 0 cannot happen for a doc of the AND; it is kept, like class 3)
    const bool keep = alive && !(mincls - 1u < 2u && sortable(bm25(weight, cache_g[nid], mincls) * 1.000002f) < thr_here);
    const uint64_t mk = __ballot(keep);
    if (!mk) return;
    const uint32_t at = qn + mbcnt64(mk);
    wave_mem_fence();
    if (keep) L.q_doc[at] = doc;
    wave_mem_fence();
    qn += (uint32_t)__popcll(mk);
  };
  // the two stages, each as long as it has a full batch (final: whatever is left) — ONE place, so that the scoring
  // stage is instantiated twice (here for the extraction loop, here for the flush), not once per caller of a caller
  auto pump = [&](bool final) __attribute__((always_inline)) {
    for (;;) {
      if (q0n >= 64u || (final && q0n)) {
        stageP(q0n < 64u ? q0n : 64u);
      } else if (qn >= 64u || (final && qn)) {
        stageC(qn < 64u ? qn : 64u);
      } else {
        break;
      }
    }
  };

  auto flush_query = [&]() __attribute__((always_inline)) {
    pump(true);
    const uint32_t part = sload(&Q->part_start) + (chunk - sload(&Q->chunk_first));
    flush_partial<KPL>(tk, sload(&p.sinks->partials), part, lane);
    if (lane == 0 && n_q) atomicAdd(sload(&p.sinks->query_matches) + sload(sload(&p.sinks->out_index) + q), n_q);
    n_q = 0;
  };

  setup_query();
  for (uint32_t t = t_begin; t < t_end; ++t) {
    while (t >= q_tile_end) {
      if (q_tile_end > q_tile_start && q_tile_end > t_begin) flush_query();
      ++q;
      setup_query();
    }
    const uint32_t w_begin = (t - q_tile_start) * SWEEP_WORDS;
    const uint32_t w_end = w_begin + SWEEP_WORDS < n_words ? w_begin + SWEEP_WORDS : n_words;
    if (slots) {  // the other wavefronts of the query may have raised its threshold
      const uint32_t sv = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t g = kth_largest64(sv, tk.k);
      if (g > thr_g) thr_g = g;
    }
    // Every lane takes FOUR consecutive words (128 docs) of every list's doc bits per step — one 16-byte load per list,
    // SWEEP_UNROLL steps requested together — ANDs them and hands the surviving docs to the queue, lowest first.
    // (Rounds 3-5 streamed the {bits, rank} words, one per lane and list: twice the bytes and four times the
    // steps per tile; the loop around them was two thirds of the kernel's vector instructions.)
    for (uint32_t wb0 = w_begin; wb0 < w_end; wb0 += 256u * SWEEP_UNROLL) {
      uint4 bu[SWEEP_UNROLL][SWEEP_NT];
#pragma unroll
      for (uint32_t u = 0; u < SWEEP_UNROLL; ++u) {
        const uint32_t w = wb0 + 256u * u + 4u * (uint32_t)lane;
#pragma unroll
        for (uint32_t m = 0; m < SWEEP_NT; ++m) {
          const uint32_t fill = m < nt ? 0u : 0xFFFFFFFFu;
          bu[u][m] = make_uint4(fill, fill, fill, fill);
          if (m < nt && w < w_end) bu[u][m] = *reinterpret_cast<const uint4 *>(L.bits[m] + w);  // (zero-padded past n_words)
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < SWEEP_UNROLL; ++u) {
        const uint32_t w = wb0 + 256u * u + 4u * (uint32_t)lane;
        uint32_t c0 = bu[u][0].x & bu[u][1].x & bu[u][2].x & bu[u][3].x;
        uint32_t c1 = bu[u][0].y & bu[u][1].y & bu[u][2].y & bu[u][3].y;
        uint32_t c2 = bu[u][0].z & bu[u][1].z & bu[u][2].z & bu[u][3].z;
        uint32_t c3 = bu[u][0].w & bu[u][1].w & bu[u][2].w & bu[u][3].w;
        while (__ballot((c0 | c1 | c2 | c3) != 0u)) {
          const bool has = (c0 | c1 | c2 | c3) != 0u;
          // the lane's lowest doc: first non-empty word, its lowest bit
          const uint32_t j = c0 ? 0u : (c1 ? 1u : (c2 ? 2u : 3u));
          const uint32_t cw = c0 ? c0 : (c1 ? c1 : (c2 ? c2 : c3));
          const uint32_t bit = has ? (uint32_t)__builtin_ctz(cw) : 0u;
          const uint32_t rest = cw & (cw - 1u);
          if (c0) c0 = rest; else if (c1) c1 = rest; else if (c2) c2 = rest; else c3 = rest;
          const uint64_t mk = __ballot(has);
          // (cls_ok, wave-uniform: through the class test first)
          uint32_t *const qd = cls_ok ? L.q0_doc : L.q_doc;
          const uint32_t at = (cls_ok ? q0n : qn) + mbcnt64(mk);
          wave_mem_fence();
          if (has) qd[at] = ((w + j) << 5) + bit;
          wave_mem_fence();
          if (cls_ok)
            q0n += (uint32_t)__popcll(mk);
          else
            qn += (uint32_t)__popcll(mk);
          pump(false);
        }
      }
    }
  }
  if (q_tile_end > q_tile_start) flush_query();
  if (lane == 0 && n_matches) atomicAdd(sload(&p.sinks->match_counter), (unsigned long long)n_matches);
}

}  // namespace

// =================================================================== launch wrappers
template <int KPL>
static void launch_phrase_t(const TqkScanParams &p, bool /*dpp*/, dim3 grid, dim3 block, hipStream_t st) {
  if (p.max_terms <= 4u) {  // fewer position cursors and half the per-candidate LDS
    if (p.all_dense)
      phrase_kernel<KPL, 4, true><<<grid, block, 0, st>>>(p);
    else
      phrase_kernel<KPL, 4, false><<<grid, block, 0, st>>>(p);
  } else {
    phrase_kernel<KPL, TQD_PH_MAX_TERMS, false><<<grid, block, 0, st>>>(p);
  }
}
hipError_t tqk_launch_phrase(const TqkScanParams &p, int kpl, bool use_dpp, hipStream_t st) {
  if (p.n_chunks == 0) return hipSuccess;
  const dim3 grid(p.n_chunks), block(64);
  if (p.or_windows == 2u) {  // (the planner's "phrase sweep" launch group: tiles are bitmap-word ranges)
    switch (kpl) {
      case 1: phrase_sweep_kernel<1><<<grid, block, 0, st>>>(p); break;
      case 2: phrase_sweep_kernel<2><<<grid, block, 0, st>>>(p); break;
      case 4: phrase_sweep_kernel<4><<<grid, block, 0, st>>>(p); break;
      default: phrase_sweep_kernel<16><<<grid, block, 0, st>>>(p); break;
    }
    return hipGetLastError();
  }
  switch (kpl) {
    case 1: launch_phrase_t<1>(p, use_dpp, grid, block, st); break;
    case 2: launch_phrase_t<2>(p, use_dpp, grid, block, st); break;
    case 4: launch_phrase_t<4>(p, use_dpp, grid, block, st); break;
    default: launch_phrase_t<16>(p, use_dpp, grid, block, st); break;
  }
  return hipGetLastError();
}
