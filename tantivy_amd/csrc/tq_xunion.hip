// tq_xunion.hip — pure unions WITHOUT pruning (every match is scored: what tantivy does under a
// (TopDocs, Count) collector pair or a tweaked score), evaluated doc-major for a whole batch.
// Shared device helpers: tq_common.hpp.
//
// The reference walks each query's lists through BufferedUnionScorer (src/query/union/
// buffered_union.rs:63-158): every posting of every list of every query is decoded, scored
// (bm25.rs:179-193) and added into a 4096-doc window — 6.7 G postings for the 1000-query or5
// batch, 90 vector instructions per 64 of them in or_kernel<.., false> (tq_union.hip).  But a
// posting's tf/(tf+norm) does not depend on the query, only its weight does, and a batch touches
// few distinct lists (256 here) — so here the loop is turned inside out:
//   * a workgroup owns a TILE of 128 consecutive docs and builds, for EVERY list of the batch,
//     its BM25 score w * tf/(tf+norm) for the tile's docs in LDS (0 where the doc is not in the
//     list): one row of 128 floats per list — a term's weight is the same in every query of a
//     batch (idf * (1 + k1) * boost); were it not, the list takes one row per weight.  Lists with a bitmap: bitmap word + rank -> byte-wide tf -> one IEEE
//     division per doc; lists without one are kept as plain doc/tf arrays (built on first use)
//     and scattered into their row by one lane per list;
//   * then each of its 16 waves takes its share of the queries: lane <-> two docs, per list of the
//     query one 8-byte LDS read and an add — summed in the query's list order (weight
//     descending), i.e. the order of every other union kernel; adding the 0.0 of an absent list
//     is exact, so the bits are those of the per-query kernels;
//   * score > 0 <=> the doc is in the union (weights > 0, tf >= 1; a posting with tf == 0 puts
//     -0.0 in its row — x + -0.0 is x — and its tile tests the rows' bits instead).  Matches are counted; those at or above the
//     query's threshold go to the collector.
// Cost per (query, 128 docs): ~46 vector instructions whatever the lists hold — against 2.3 per
// posting before; a query pays off here when its lists together hold more than a sixth of the
// segment (the planner's test), and the rows are built once for all queries of the batch.
//
// Collector: as in the shared-union launch (tq_ushare.hip).  A query belongs to ONE wave of each
// workgroup, which keeps a staging list for it in global memory (k + 64.. entries, cut back to
// the k best by a radix select when it fills up); hashed atomic-max slots give every workgroup the
// same lower bound of the k-th best score (thr_val[query]); at the end the staging lists are
// appended to the query's result list, which merge_lists_kernel reduces.  Nothing is pruned: the
// threshold only decides which scored docs are worth queueing, like TopNComputer's own test
// (src/collector/top_score_collector.rs).
#include "tq_common.hpp"

namespace {

constexpr uint32_t XT = TQK_XU_TILE;
constexpr uint32_t XW = TQK_XU_WAVES;
#ifndef TQ_XU_BATCH
#define TQ_XU_BATCH 4
#endif
#ifndef TQ_XU_TIMERS
#define TQ_XU_TIMERS 0  // region timers (experiments): TQ_DEBUG bits 16..19 = 1 build, 2 evaluation, 3 barriers, 4 all
#endif
#ifndef TQ_XU_UNROLL
#define TQ_XU_UNROLL 4
#endif
constexpr uint32_t XU_DEFAULT = TQ_XU_UNROLL;  // queries scored together (two when they have up to 8 lists: registers)
constexpr uint32_t XB = TQ_XU_BATCH;  // bitmap rows whose tf bytes are requested together (8: scratch)

struct XuLds {
  float T[TQK_XU_MAX_ROWS][XT];        // w * tf/(tf+norm) of doc d0+i in row r's list, 0 = not in the list
  float cache[256];                    // Bm25Weight.cache
  uint16_t cnt[TQK_XU_MAX_QUERIES];    // entries in this workgroup's staging list of the query
  uint32_t zflag[2];                   // the tile (by parity) met a posting with tf == 0
  uint32_t task;
  uint2 wsc[XW][64];                   // per wave: the tile's bitmap words of its 16 rows ([4 * i + word])
};
static_assert(sizeof(XuLds) <= 160 * 1024, "one workgroup per CU: all of its LDS");

// per-lane view of a posting list (fields of TqdTermHead fetched with vector loads)
__device__ __forceinline__ TermRef term_of_lane(const TqdTerm *terms, uint32_t handle) {
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
// the exact tf of posting i of a list (a saturated byte: 255 or more)
__device__ __forceinline__ uint32_t exact_tf(const uint8_t *idx, const TqdTerm *terms, uint32_t handle,
                                             uint32_t i) {
  const TermRef tr = term_of_lane(terms, handle);
  const uint4 r = tr.rec[i >> 7];
  return block_tf_at(idx, tr, make_uint2(r.y, r.z), i & 127u);
}

template <int R>
__device__ __forceinline__ uint64_t kth_key(const uint64_t (&v)[R], uint32_t k) {
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

__device__ __forceinline__ float thr_as_float(uint32_t thr) {  // sortable bits -> the score (scores are >= 0)
  return thr ? __uint_as_float(thr ^ ((thr >> 31) ? 0x80000000u : 0xFFFFFFFFu)) : -1.0f;
}

// MT: the most lists a query of the launch has.  Queries with fewer are padded with the all-zero
// row (x + 0 is x), so that the scoring loop has no branch and all of a query's
// LDS reads are in flight together.
// DEL: the segment has deletes (the alive bits are applied to the matches; without them a doc beyond
// max_doc has all-zero rows and needs no test).
template <int KPL, int MT, bool DEL>
__global__ __launch_bounds__(XW * 64) void xunion_kernel(TqkDenseParams p) {
  constexpr int R = KPL + 1;
  constexpr uint32_t CAPL = (uint32_t)R * 64u;
  constexpr uint32_t XU = MT > 5 ? (XU_DEFAULT < 2u ? XU_DEFAULT : 2u) : XU_DEFAULT;
  __shared__ XuLds L;
  const int lane = (int)__lane_id();
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = uni(tid >> 6);
  const TqdSegment seg = p.seg;
  const uint8_t *idx = seg.idx;
  const uint32_t nq = p.n_queries;
  uint64_t *const wg_stage = p.stage + (size_t)blockIdx.x * (size_t)nq * CAPL;

  for (uint32_t i = tid; i < 256u; i += XW * 64u) L.cache[i] = p.cache[i];
  for (uint32_t i = tid; i < nq; i += XW * 64u) L.cnt[i] = 0;
  if (tid < 2u) L.zflag[tid] = 0u;
  if (tid < XT) L.T[p.n_rows][tid] = 0.0f;  // the padding row (n_rows < TQK_XU_MAX_ROWS)

  // ---- this wave's queries: q = wave + XW * j, j = 64 * chunk + lane
  const uint32_t n_mine = nq > wave ? (nq - wave + XW - 1u) / XW : 0u;
  const uint32_t n_chunks = (n_mine + 63u) >> 6;
  uint32_t d_off[MT], d_ntk = 0, d_trow = 0, d_thr = 0;  // d_off[t]: LDS byte offset of the row of list t
#pragma unroll
  for (int t = 0; t < MT; ++t) d_off[t] = 0;
  float d_thr_f = -1.0f;
  uint32_t mc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // docs matched, per chunk and lane (<-> query)
  auto load_chunk = [&](uint32_t c) __attribute__((always_inline)) {
    const uint32_t j = 64u * c + (uint32_t)lane;
    d_ntk = 0;
    d_thr = 0;
    if (j < n_mine) {
      const uint32_t q = wave + XW * j;
      const uint4 a = *reinterpret_cast<const uint4 *>(p.queries + q);
#pragma unroll
      for (uint32_t t = 0; t < (uint32_t)MT; ++t)
        d_off[t] = (((t < 4u ? a.x >> (8u * t) : a.y >> (8u * (t - 4u)))) & 0xFFu) * (XT * 4u);
      d_ntk = a.z;
      d_trow = a.w;
      d_thr = __hip_atomic_load(p.thr_val + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    d_thr_f = thr_as_float(d_thr);
  };
  if (n_chunks == 1u) load_chunk(0u);

  // a staging list is cut back to its k best; returns the k-th key (the list held n > k entries)
  auto compact = [&](uint64_t *sl, uint32_t n, uint32_t k) __attribute__((always_inline)) -> uint64_t {
    uint64_t v[R];
    wave_mem_fence();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t i = (uint32_t)r * 64u + (uint32_t)lane;
      v[r] = i < n ? sl[i] : 0ull;
    }
    const uint64_t kth = kth_key<R>(v, k);
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

  // lists without a bitmap: lane <-> row n_bitmap_rows + wave + XW * lane; its cursor into flat_docs
  const uint32_t n_rows = p.n_rows, n_a = p.n_bitmap_rows;
  const uint32_t b_row = n_a + wave + XW * (uint32_t)lane;
  const bool b_on = b_row < n_rows;
  const uint32_t *b_docs = nullptr;
  const uint8_t *b_tf8 = nullptr;
  uint32_t b_df = 0, b_handle = 0, b_cur = 0;
  float b_w = 0.0f;
  if (b_on) {
    const TqkDenseRow *rr = p.rows + b_row;
    b_docs = rr->flat_docs;
    b_tf8 = rr->tf8;
    b_df = rr->doc_freq;
    b_handle = rr->handle;
    b_w = rr->w;
  }
  const uint32_t n_b_mine = n_rows > n_a + wave ? (n_rows - n_a - wave + XW - 1u) / XW : 0u;  // (<= 16)
  // lists with a bitmap: this wave builds rows wave + XW * i, i < n_a_mine (<= 16).  Lane l fetches
  // word (l & 3) of row slot (l >> 2) — ONE load per tile for all of the wave's rows; lane i < 16
  // keeps row slot i's byte-wide tfs and term handle
  const uint32_t n_a_mine = n_a > wave ? (n_a - wave + XW - 1u) / XW : 0u;
  const uint2 *a_dense_w = nullptr;
  const uint8_t *a_tf8 = nullptr;
  uint32_t a_handle = 0;
  float a_w = 0.0f;
  {
    const uint32_t rw = wave + XW * ((uint32_t)lane >> 2);
    if (rw < n_a) a_dense_w = p.rows[rw].dense;
    const uint32_t ri = wave + XW * (uint32_t)lane;
    if ((uint32_t)lane < 16u && ri < n_a) {
      a_tf8 = p.rows[ri].tf8;
      a_handle = p.rows[ri].handle;
      a_w = p.rows[ri].w;
    }
  }
  uint32_t b_next = 0xFFFFFFFFu;  // the doc at b_cur (requested ahead of the tile that needs it)
  uint2 pre_w = make_uint2(0u, 0u);  // the bitmap words of tile pre_d0, requested during the previous tile
  uint32_t pre_d0 = 0xFFFFFFFFu;
  const uint32_t n_words = (seg.max_doc + 31u) >> 5;
  uint32_t n_scored = 0;
  const uint32_t tphase = TQ_XU_TIMERS ? (p.debug >> 16) & 15u : 0u;
  uint64_t tacc = 0, tlast = 0;
  auto tb = [&](uint32_t ph) __attribute__((always_inline)) {
    if (TQ_XU_TIMERS && (tphase == ph || tphase == 4u)) tlast = __builtin_readcyclecounter();
  };
  auto te = [&](uint32_t ph) __attribute__((always_inline)) {
    if (TQ_XU_TIMERS && (tphase == ph || tphase == 4u)) tacc += __builtin_readcyclecounter() - tlast;
  };
  uint32_t seq = 0;  // tiles this workgroup has processed (parity: which zflag)
  __syncthreads();

  for (;;) {
    if (tid == 0) L.task = atomicAdd(p.task_counter, 1u);
    __syncthreads();
    const uint32_t task = uni(L.task);
    if (task >= p.n_tasks) break;
    const uint32_t tile0 = task * p.tiles_per_task;
    const uint32_t d_first = tile0 * XT;
    if (b_on) {  // lower bound of the task's first doc
      uint32_t lo = 0, hi = b_df;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (b_docs[mid] < d_first)
          lo = mid + 1u;
        else
          hi = mid;
      }
      b_cur = lo;
      b_next = lo < b_df ? b_docs[lo] : 0xFFFFFFFFu;
    }
    for (uint32_t ti = 0; ti < p.tiles_per_task; ++ti) {
      const uint32_t d0 = d_first + ti * XT;
      if (d0 >= seg.max_doc) break;  // (uniform)
      const uint32_t par = seq++ & 1u;
      // the thresholds of the wave's queries as other workgroups left them: requested now, used
      // after the build
      uint32_t thr_new = 0;
      if (n_chunks == 1u && (uint32_t)lane < n_mine)
        thr_new = __hip_atomic_load(p.thr_val + wave + XW * (uint32_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tb(3u);
      __syncthreads();  // (A) every wave is done with the previous tile's rows
      te(3u);
      tb(1u);
      if (tid == 0) L.zflag[par ^ 1u] = 0u;
      // ---- build: the lane's two docs
      const uint32_t doc0 = d0 + 2u * (uint32_t)lane;
      const bool in0 = doc0 < seg.max_doc, in1 = doc0 + 1u < seg.max_doc;
      float nrm0, nrm1;
      {
        const uint32_t f0 = in0 ? fieldnorm_id(seg, doc0) : 0u, f1 = in1 ? fieldnorm_id(seg, doc0 + 1u) : 0u;
        nrm0 = L.cache[f0];
        nrm1 = L.cache[f1];
      }
      bool al0 = in0, al1 = in1;
      if (DEL && seg.alive) {
        const uint32_t ab = in0 ? (uint32_t)seg.alive[doc0 >> 3] : 0u;
        al0 = in0 && ((ab >> (doc0 & 7u)) & 1u);
        al1 = in1 && ((ab >> ((doc0 & 7u) + 1u)) & 1u);
      }
      const uint64_t valid0 = __ballot(al0), valid1 = __ballot(al1);
      bool zero_tf = false;
      // rows with a bitmap.  Three round trips per tile: the words of all rows (one load), then the
      // tf bytes of XB rows at a time
      {
        uint2 wdw = pre_w;
        if (pre_d0 != d0) {  // (the first tile of a task)
          const uint32_t wi = (d0 >> 5) + ((uint32_t)lane & 3u);
          wdw = make_uint2(0u, 0u);
          if (a_dense_w && wi < n_words) wdw = a_dense_w[wi];
        }
        L.wsc[wave][lane] = wdw;
      }
      wave_mem_fence();
      const uint32_t b0 = 2u * ((uint32_t)lane & 15u);
#pragma unroll 1
      for (uint32_t i0 = 0; i0 < n_a_mine; i0 += XB) {
        uint32_t wx[XB], tf0[XB], tf1[XB], pi0[XB];
#pragma unroll
        for (uint32_t u = 0; u < XB; ++u) {
          const uint32_t i = i0 + u;
          wx[u] = 0u;
          tf0[u] = 1u;
          tf1[u] = 1u;
          pi0[u] = 0u;
          if (i < n_a_mine) {  // (uniform)
            const uint2 wd = L.wsc[wave][4u * i + ((uint32_t)lane >> 4)];
            wx[u] = wd.x;
            pi0[u] = wd.y + (uint32_t)__popc(wd.x & ((1u << b0) - 1u));
            const uint8_t *tp = (const uint8_t *)readlane64((uint64_t)a_tf8, i);
            if ((wd.x >> b0) & 1u) tf0[u] = tp[pi0[u]];
            if ((wd.x >> (b0 + 1u)) & 1u) tf1[u] = tp[pi0[u] + ((wd.x >> b0) & 1u)];
          }
        }
#pragma unroll
        for (uint32_t u = 0; u < XB; ++u) {
          const uint32_t i = i0 + u;
          if (i < n_a_mine) {
            const uint32_t r = wave + XW * i;
            const bool p0 = (wx[u] >> b0) & 1u, p1 = (wx[u] >> (b0 + 1u)) & 1u;
            uint32_t t0 = tf0[u], t1 = tf1[u];
            const uint32_t odd = (p0 && (t0 == 255u || t0 == 0u)) || (p1 && (t1 == 255u || t1 == 0u));
            if (__ballot(odd)) {  // saturated bytes: the packed value
              const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)a_handle, (int)i);
              if (p0 && t0 == 255u) t0 = exact_tf(idx, p.terms, h, pi0[u]);
              if (p1 && t1 == 255u) t1 = exact_tf(idx, p.terms, h, pi0[u] + (p0 ? 1u : 0u));
              zero_tf = zero_tf || (p0 && t0 == 0u) || (p1 && t1 == 0u);
            }
            // bm25.rs:179-193, the bits of bm25() (tq_common.hpp).  (Tried: tf/(tf+norm) from a
            // 256 x 256 table built per batch — one gather where the IEEE division is ten
            // instructions; the second dependent load per row cost more than it saved: 10.2 vs 9.9 ms.)
            const float w = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(a_w), (int)i));
            const float f0 = (float)t0, f1 = (float)t1;
            float v0 = p0 ? w * (f0 / (f0 + nrm0)) : 0.0f;
            float v1 = p1 ? w * (f1 / (f1 + nrm1)) : 0.0f;
            if (p0 && t0 == 0u) v0 = -0.0f;  // tf == 0: present with score 0 (x + -0.0 is x)
            if (p1 && t1 == 0u) v1 = -0.0f;
            *reinterpret_cast<float2 *>(&L.T[r][2 * lane]) = make_float2(v0, v1);
          }
        }
      }
      // rows without one: cleared by the wave, then filled by the row's lane
      for (uint32_t i = 0; i < n_b_mine; ++i)
        *reinterpret_cast<float2 *>(&L.T[n_a + wave + XW * i][2 * lane]) = make_float2(0.0f, 0.0f);
      wave_mem_fence();
      if (b_on) {
        const uint32_t d_end = d0 + XT;
        while (b_next < d_end) {  // (0xFFFFFFFF past the list's end)
          const uint32_t d = b_next;
          const uint32_t tf = b_tf8 ? (uint32_t)b_tf8[b_cur] : 1u;
          float v;
          if (tf == 0u) {
            v = -0.0f;
            zero_tf = true;
          } else {
            const float f = (float)(tf == 255u ? exact_tf(idx, p.terms, b_handle, b_cur) : tf);
            v = b_w * (f / (f + L.cache[fieldnorm_id(seg, d)]));
          }
          L.T[b_row][d - d0] = v;
          ++b_cur;
          b_next = b_cur < b_df ? b_docs[b_cur] : 0xFFFFFFFFu;
        }
      }
      if (zero_tf) L.zflag[par] = 1u;
      // the next tile's bitmap words travel while this one is evaluated
      pre_d0 = 0xFFFFFFFFu;
      if (ti + 1u < p.tiles_per_task && d0 + XT < seg.max_doc) {
        pre_d0 = d0 + XT;
        const uint32_t wi = (pre_d0 >> 5) + ((uint32_t)lane & 3u);
        pre_w = make_uint2(0u, 0u);
        if (a_dense_w && wi < n_words) pre_w = a_dense_w[wi];
      }
      te(1u);
      tb(3u);
      __syncthreads();  // (B) the tile's rows are complete
      te(3u);
      tb(2u);
      const uint32_t generic = uni(L.zflag[par]);

      // ---- evaluate: this wave's queries against the tile
      for (uint32_t c = 0; c < n_chunks; ++c) {
        if (n_chunks > 1u) {
          load_chunk(c);
        } else if (thr_new > d_thr) {
          d_thr = thr_new;
          d_thr_f = thr_as_float(thr_new);
        }
        const uint32_t n_in = n_mine - 64u * c < 64u ? n_mine - 64u * c : 64u;
        uint32_t tilecnt = 0;  // lane jj: docs of the tile matched by query jj
        // the scores of query jj for the lane's two docs
        auto score = [&](uint32_t jj, float &s0, float &s1) __attribute__((always_inline)) {
          float2 v[MT];
#pragma unroll
          for (uint32_t t = 0; t < (uint32_t)MT; ++t) {
            const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)d_off[t], (int)jj);
            v[t] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(&L.T[0][2 * lane]) + off);
          }
          s0 = v[0].x;  // (0 + x is x)
          s1 = v[0].y;
#pragma unroll
          for (uint32_t t = 1; t < (uint32_t)MT; ++t) {
            s0 = s0 + v[t].x;
            s1 = s1 + v[t].y;
          }
        };
        // count the matches, queue those at or above the query's threshold
        auto finish = [&](uint32_t jj, float s0, float s1) __attribute__((always_inline)) {
          const float thr_f = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d_thr_f), (int)jj));
          uint64_t pres0 = __ballot(s0 > 0.0f), pres1 = __ballot(s1 > 0.0f);
          if (generic != 0u) {  // a tf of 0 scores 0 (the row holds -0.0): any non-zero bits = present
            uint32_t o0 = 0, o1 = 0;
#pragma unroll
            for (uint32_t t = 0; t < (uint32_t)MT; ++t) {
              const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)d_off[t], (int)jj);
              const float2 v = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(&L.T[0][2 * lane]) + off);
              o0 |= __float_as_uint(v.x);
              o1 |= __float_as_uint(v.y);
            }
            pres0 = __ballot(o0 != 0u);
            pres1 = __ballot(o1 != 0u);
          }
          if (DEL) {
            pres0 &= valid0;
            pres1 &= valid1;
          }
          const uint32_t nm = (uint32_t)__popcll(pres0) + (uint32_t)__popcll(pres1);
          tilecnt = (uint32_t)lane == jj ? nm : tilecnt;
          // (one compare and branch for both docs; scores are >= 0: their order is that of their bits)
          const uint32_t b0s = __float_as_uint(s0), b1s = __float_as_uint(s1);
          if (!__ballot(__uint_as_float(b0s > b1s ? b0s : b1s) >= thr_f)) return;
          const uint64_t pass0 = __ballot(s0 >= thr_f) & pres0;
          const uint64_t pass1 = __ballot(s1 >= thr_f) & pres1;
          if (!(pass0 | pass1)) return;

          // ---- collector (rare): threshold slots, the staging list, its cut
          const uint32_t q = wave + XW * (64u * c + jj);
          const uint32_t k = ((uint32_t)__builtin_amdgcn_readlane((int)d_ntk, (int)jj) >> 8) & 0xFFu;
          const uint32_t thr_row = (uint32_t)__builtin_amdgcn_readlane((int)d_trow, (int)jj);
          uint64_t *sl = wg_stage + (size_t)q * CAPL;
          uint32_t thr_up = 0;  // a higher lower bound of the query's k-th best score, if one was found
#pragma unroll 1
          for (uint32_t e = 0; e < 2u; ++e) {
            const uint64_t m = e ? pass1 : pass0;
            if (!m) continue;
            const bool a = (m >> lane) & 1ull;
            const uint32_t doc = doc0 + e;
            const uint64_t key = make_key(e ? s1 : s0, doc);
            const uint32_t sb = (uint32_t)(key >> 32);
            bool changed = false;
            if (a) {
              const uint32_t hsh = (doc * 0x9E3779B1u) >> (k <= 16u ? 26 : 24);
              const uint32_t old = atomicMax(p.thr_slots + (size_t)thr_row * TQD_THR_SLOTS + hsh, sb);
              changed = old < sb;
            }
            wave_mem_fence();
            const uint32_t n0 = L.cnt[q];
            if (a) sl[n0 + mbcnt64(m)] = key;
            const uint32_t n1 = n0 + (uint32_t)__popcll(m);
            wave_mem_fence();
            if (__ballot(changed)) {  // the k-th largest slot is a bound every workgroup can use
              const uint32_t *slots = p.thr_slots + (size_t)thr_row * TQD_THR_SLOTS;
              uint32_t sv[4] = {0u, 0u, 0u, 0u};
              sv[0] = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              uint32_t gth;
              if (k > 16u) {
#pragma unroll
                for (int r = 1; r < 4; ++r)
                  sv[r] = __hip_atomic_load(slots + 64 * r + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gth = kth_largest_hi16<4>(sv, k);
              } else {
                gth = kth_largest_hi16<1>(sv, k);
              }
              if (gth > thr_up) thr_up = gth;
            }
            uint32_t n2 = n1;
            if (n1 > CAPL - 64u) {  // the next push may bring 64 more
              const uint64_t kth = compact(sl, n1, k);
              n2 = k;
              const uint32_t t = (uint32_t)(kth >> 32);
              if (t > thr_up) thr_up = t;
            }
            if (lane == 0) L.cnt[q] = (uint16_t)n2;
            wave_mem_fence();
          }
          if (thr_up) {
            if (lane == 0) atomicMax(p.thr_val + q, thr_up);
            if ((uint32_t)lane == jj && thr_up > d_thr) {
              d_thr = thr_up;
              d_thr_f = thr_as_float(thr_up);
            }
          }
        };
        // XU queries per step: their LDS reads travel together (the kernel waits for LDS and barriers
        // two thirds of the time at 16 waves per CU: rocprofv3 SQ_WAIT_ANY / SQ_WAVE_CYCLES)
        uint32_t jj = 0;
        for (; jj + XU <= n_in; jj += XU) {
          float a0[XU], a1[XU];
          {  // (all lane reads first: a row offset read from a lane cannot feed the very next instruction)
            uint32_t off[XU][MT];
#pragma unroll
            for (uint32_t u = 0; u < XU; ++u)
#pragma unroll
              for (uint32_t t = 0; t < (uint32_t)MT; ++t)
                off[u][t] = (uint32_t)__builtin_amdgcn_readlane((int)d_off[t], (int)(jj + u));
            float2 v[XU][MT];
#pragma unroll
            for (uint32_t u = 0; u < XU; ++u)
#pragma unroll
              for (uint32_t t = 0; t < (uint32_t)MT; ++t)
                v[u][t] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(&L.T[0][2 * lane]) + off[u][t]);
#pragma unroll
            for (uint32_t u = 0; u < XU; ++u) {
              a0[u] = v[u][0].x;  // (0 + x is x)
              a1[u] = v[u][0].y;
#pragma unroll
              for (uint32_t t = 1; t < (uint32_t)MT; ++t) {
                a0[u] = a0[u] + v[u][t].x;
                a1[u] = a1[u] + v[u][t].y;
              }
            }
          }
#pragma unroll
          for (uint32_t u = 0; u < XU; ++u) finish(jj + u, a0[u], a1[u]);
        }
        for (; jj < n_in; ++jj) {
          float a0, a1;
          score(jj, a0, a1);
          finish(jj, a0, a1);
        }
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i)
          if (i == c) mc[i] += tilecnt;
      }
      te(2u);
    }
  }

  // ---- flush: match counts, then the staging lists go to their queries' result lists
  const TqkSinks sk = sload(p.sinks);
  for (uint32_t c = 0; c < n_chunks; ++c) {
    uint32_t v = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i)
      if (i == c) v = mc[i];
    const uint32_t j = 64u * c + (uint32_t)lane;
    if (j < n_mine && v) {
      atomicAdd(sk.query_matches + sk.out_index[wave + XW * j], v);
      n_scored += v;
    }
  }
  for (uint32_t j = 0; j < n_mine; ++j) {
    const uint32_t q = wave + XW * j;
    uint32_t ns = L.cnt[q];
    if (!ns) continue;
    const uint32_t k = (sload(&p.queries[q].nt_k) >> 8) & 0xFFu;
    uint64_t *sl = wg_stage + (size_t)q * CAPL;
    if (ns > k) {
      (void)compact(sl, ns, k);
      ns = k;
    }
    const uint32_t thr_now = __hip_atomic_load(p.thr_val + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t v[R];
    uint32_t keep_n = 0;
    wave_mem_fence();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t i = (uint32_t)r * 64u + (uint32_t)lane;
      v[r] = i < ns ? sl[i] : 0ull;
      if ((uint32_t)(v[r] >> 32) < thr_now) v[r] = 0ull;  // k docs of the query score higher
      keep_n += (uint32_t)__popcll(__ballot(v[r] != 0ull));
    }
    if (!keep_n) continue;
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(p.list_count + q, keep_n);
    at = uni(at);
    uint64_t *dst = p.lists + (size_t)q * p.list_stride + at;
    uint32_t base = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t m = __ballot(v[r] != 0ull);
      if (v[r] != 0ull) dst[base + mbcnt64(m)] = v[r];
      base += (uint32_t)__popcll(m);
    }
  }
  // docs scored by the launch
  for (int o = 32; o > 0; o >>= 1) n_scored += __shfl_down(n_scored, o, WAVE);
  if (tphase) n_scored = (uint32_t)(tacc >> 6);
  if (lane == 0 && n_scored) atomicAdd(sk.match_counter, (unsigned long long)n_scored);
}

// A list without a bitmap as plain arrays (one wave per 128-doc block): doc ids and min(tf, 255)
__global__ __launch_bounds__(256) void flat_list_kernel(TqdSegment seg, const TqdTerm *terms, uint32_t handle,
                                                        uint32_t *docs, uint8_t *tf8) {
  const int lane = (int)__lane_id();
  const uint32_t wave = uni(threadIdx.x >> 6);
  const TermRef t = load_term(terms, handle);
  const uint32_t j = blockIdx.x * 4u + wave;
  if (j >= t.n_blocks) return;
  const Dec d = decode_block<true, false>(uni_ptr(seg.idx), t, j, lane);
  const uint32_t i0 = j * 128u + 2u * (uint32_t)lane;
  const uint32_t df = uni(terms[handle].doc_freq);
  if (i0 < df) {
    docs[i0] = d.d0;
    tf8[i0] = (uint8_t)(d.t0 < 255u ? d.t0 : 255u);
  }
  if (i0 + 1u < df) {
    docs[i0 + 1u] = d.d1;
    tf8[i0 + 1u] = (uint8_t)(d.t1 < 255u ? d.t1 : 255u);
  }
}

}  // namespace

hipError_t tqk_launch_xunion(const TqkDenseParams &p, int kpl, hipStream_t st) {
  if (p.n_tasks == 0 || p.grid == 0 || p.n_queries == 0) return hipSuccess;
  const dim3 grid(p.grid), block(XW * 64);
#define TQ_XU(K, D)                                                           \
  do {                                                                        \
    if (p.max_terms <= 2)                                                     \
      xunion_kernel<K, 2, D><<<grid, block, 0, st>>>(p);                      \
    else if (p.max_terms <= 3)                                                \
      xunion_kernel<K, 3, D><<<grid, block, 0, st>>>(p);                      \
    else if (p.max_terms <= 5)                                                \
      xunion_kernel<K, 5, D><<<grid, block, 0, st>>>(p);                      \
    else                                                                      \
      xunion_kernel<K, 8, D><<<grid, block, 0, st>>>(p);                      \
  } while (0)
  if (p.seg.alive) {
    if (kpl == 1)
      TQ_XU(1, true);
    else
      TQ_XU(2, true);
  } else {
    if (kpl == 1)
      TQ_XU(1, false);
    else
      TQ_XU(2, false);
  }
#undef TQ_XU
  return hipGetLastError();
}

hipError_t tqk_launch_flat_list(const TqdSegment &seg, const TqdTerm *terms, uint32_t handle,
                                uint32_t n_blocks, uint32_t *docs, uint8_t *tf8, hipStream_t st) {
  if (n_blocks == 0) return hipSuccess;
  flat_list_kernel<<<dim3((n_blocks + 3) / 4), dim3(256), 0, st>>>(seg, terms, handle, docs, tf8);
  return hipGetLastError();
}
