/* to_query.c — CPU ORACLE (test infrastructure): TermScorer, Intersection, union,
 * block-WAND executors, PhraseScorer, TopNHeap, merge_top_k.  See tantivy_oracle.h. */
#include "tantivy_oracle.h"

#include <assert.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ TopNHeap
 * src/collector/sort_key/sort_by_score.rs:86-161.
 * Min-heap under ScoreHeapEntry::cmp: score asc, then doc DESC (higher doc = "smaller"). */
static int entry_less(const to_hit *a, const to_hit *b) {
  /* a < b under ScoreHeapEntry::cmp */
  if (a->score < b->score) return 1;
  if (a->score > b->score) return 0;
  return a->doc > b->doc; /* other.doc.cmp(&self.doc) */
}
void to_topn_init(to_topn_heap *h, size_t top_n) {
  h->heap = (to_hit *)malloc(sizeof(to_hit) * (top_n ? top_n : 1));
  h->len = 0;
  h->top_n = top_n;
  h->has_threshold = 0;
  h->threshold = 0.0f;
}
void to_topn_free(to_topn_heap *h) {
  free(h->heap);
  h->heap = NULL;
}
static void sift_up(to_hit *a, size_t i) {
  while (i > 0) {
    size_t p = (i - 1) / 2;
    if (entry_less(&a[i], &a[p])) {
      to_hit t = a[i];
      a[i] = a[p];
      a[p] = t;
      i = p;
    } else
      break;
  }
}
static void sift_down(to_hit *a, size_t n, size_t i) {
  for (;;) {
    size_t l = 2 * i + 1, r = l + 1, m = i;
    if (l < n && entry_less(&a[l], &a[m])) m = l;
    if (r < n && entry_less(&a[r], &a[m])) m = r;
    if (m == i) break;
    to_hit t = a[i];
    a[i] = a[m];
    a[m] = t;
    i = m;
  }
}
void to_topn_push(to_topn_heap *h, float score, uint32_t doc) {
  /* :137-152 */
  if (h->len < h->top_n) {
    h->heap[h->len].score = score;
    h->heap[h->len].doc = doc;
    sift_up(h->heap, h->len);
    h->len++;
    if (h->len == h->top_n) {
      h->has_threshold = 1;
      h->threshold = h->heap[0].score;
    }
  } else if (h->has_threshold) {
    if (score > h->threshold) {
      h->heap[0].score = score;
      h->heap[0].doc = doc;
      sift_down(h->heap, h->len, 0);
      h->threshold = h->heap[0].score;
    }
  }
}
size_t to_topn_into_vec(const to_topn_heap *h, to_hit *out) {
  memcpy(out, h->heap, h->len * sizeof(to_hit));
  return h->len;
}
static int hit_cmp(const void *pa, const void *pb) {
  const to_hit *a = (const to_hit *)pa, *b = (const to_hit *)pb;
  if (a->score > b->score) return -1;
  if (a->score < b->score) return 1;
  return (a->doc > b->doc) - (a->doc < b->doc);
}
void to_sort_hits(to_hit *hits, size_t n) { qsort(hits, n, sizeof(to_hit), hit_cmp); }

static int ghit_cmp(const void *pa, const void *pb) {
  /* top_score_collector.rs:590-600: sort_key desc, then DocAddress (segment_ord, doc) asc */
  const to_global_hit *a = (const to_global_hit *)pa, *b = (const to_global_hit *)pb;
  if (a->score > b->score) return -1;
  if (a->score < b->score) return 1;
  if (a->segment_ord != b->segment_ord) return a->segment_ord < b->segment_ord ? -1 : 1;
  return (a->doc > b->doc) - (a->doc < b->doc);
}
size_t to_merge_top_k(const to_global_hit *hits, size_t n, size_t offset, size_t limit,
                      to_global_hit *out) {
  /* sort_key_top_collector.rs:76-95.  TopNComputer::into_sorted_vec of the top (offset+limit)
   * == the first (offset+limit) of the full sort under compare_for_top_k. */
  if (limit == 0) return 0;
  to_global_hit *tmp = (to_global_hit *)malloc(sizeof(to_global_hit) * (n ? n : 1));
  memcpy(tmp, hits, n * sizeof(to_global_hit));
  qsort(tmp, n, sizeof(to_global_hit), ghit_cmp);
  size_t end = offset + limit < n ? offset + limit : n;
  size_t w = 0;
  for (size_t i = offset; i < end; i++) out[w++] = tmp[i];
  free(tmp);
  return w;
}

/* ------------------------------------------------------------------ TermScorer
 * src/query/term_query/term_scorer.rs:9-150 */
typedef struct {
  to_segment_postings sp;
  const uint8_t *fieldnorm; /* NULL => constant id */
  uint8_t const_fieldnorm_id;
  to_bm25 w;
  float max_score; /* TermScorerWithMaxScore cache (block_wand_union.rs:267-277) */
} term_scorer;

static uint8_t ts_fieldnorm_id(const term_scorer *t, uint32_t doc) {
  return t->fieldnorm ? t->fieldnorm[doc] : t->const_fieldnorm_id;
}
static uint32_t ts_doc(const term_scorer *t) { return to_sp_doc(&t->sp); }
static float ts_score(const term_scorer *t) {
  return to_bm25_score(&t->w, ts_fieldnorm_id(t, ts_doc(t)), to_sp_term_freq(&t->sp));
}
static float ts_block_max_score(term_scorer *t) {
  return to_block_postings_block_max_score(&t->sp.bp, t->fieldnorm, t->const_fieldnorm_id, &t->w);
}
static uint32_t ts_last_doc_in_block(const term_scorer *t) {
  return t->sp.bp.skip.last_doc_in_block;
}
static int ts_open(term_scorer *t, const to_segment_view *seg, const to_term_info *ti,
                   const to_bm25 *w, int requested) {
  const uint8_t *body = seg->idx + 8; /* inverted_index_reader.rs:72-73 */
  const uint8_t *pos = seg->pos ? seg->pos + ti->positions_start : NULL;
  size_t pos_len = seg->pos ? (size_t)(ti->positions_end - ti->positions_start) : 0;
  if (to_segment_postings_open(&t->sp, ti->doc_freq, body + ti->postings_start,
                               (size_t)(ti->postings_end - ti->postings_start), pos, pos_len,
                               seg->record_option, requested))
    return -1;
  t->fieldnorm = seg->fieldnorm;
  t->const_fieldnorm_id = 1; /* FieldNormReader::constant(max_doc, 1): term_weight.rs:209-219 */
  if (w) t->w = *w;
  t->max_score = w ? to_bm25_max_score(w) : 0.0f;
  return 0;
}

/* ------------------------------------------------------------------ callback plumbing */
typedef struct {
  to_topn_heap *heap;
} pruning_cb;
static float cb_push(pruning_cb *cb, uint32_t doc, float score) {
  /* sort_by_score.rs:55-58 */
  to_topn_push(cb->heap, score, doc);
  return cb->heap->has_threshold ? cb->heap->threshold : -3.40282347e+38f;
}

/* ------------------------------------------------------------------ block_wand_intersection
 * src/query/boolean_query/block_wand_intersection.rs:19-179 */
static void sort_by_size_hint(term_scorer **s, size_t n) {
  /* stable insertion sort on doc_freq (Vec::sort_by_key is stable) */
  for (size_t i = 1; i < n; i++) {
    term_scorer *x = s[i];
    size_t j = i;
    while (j > 0 && s[j - 1]->sp.bp.doc_freq > x->sp.bp.doc_freq) {
      s[j] = s[j - 1];
      j--;
    }
    s[j] = x;
  }
}
static void block_wand_intersection(term_scorer **scorers, size_t n, float threshold,
                                    pruning_cb *cb) {
  assert(n >= 2);
  sort_by_size_hint(scorers, n);
  term_scorer *leader = scorers[0];
  term_scorer **sec = scorers + 1;
  size_t ns = n - 1;
  float leader_max = leader->max_score;
  float sec_global_max_sum = 0.0f;
  for (size_t i = 0; i < ns; i++) sec_global_max_sum += sec[i]->max_score;
  if (leader_max + sec_global_max_sum <= threshold) return;

  uint32_t doc = ts_doc(leader);
  float sec_bms[64], sec_suffix[64];
  assert(ns <= 64);
  while (doc < TO_TERMINATED) {
    to_block_postings_seek_block(&leader->sp.bp, doc);
    float leader_block_max = ts_block_max_score(leader);
    uint32_t window_end = ts_last_doc_in_block(leader);
    float sec_bm_sum = 0.0f;
    for (size_t i = 0; i < ns; i++) {
      to_block_postings_seek_block(&sec[i]->sp.bp, doc);
      if (sec[i]->sp.bp.skip.remaining_docs == 0) return;
      uint32_t ld = ts_last_doc_in_block(sec[i]);
      if (ld < window_end) window_end = ld;
      float bms = ts_block_max_score(sec[i]);
      sec_bms[i] = bms;
      sec_bm_sum += bms;
    }
    if (leader_block_max + sec_bm_sum <= threshold) {
      doc = window_end + 1;
      continue;
    }
    to_block_postings *bc = &leader->sp.bp;
    size_t start_idx = to_block_postings_seek(bc, doc);
    size_t end_idx = to_search_block(bc->docs, window_end + 1);
    if (end_idx > bc->block_len) end_idx = bc->block_len;

    float score_threshold = threshold - sec_bm_sum;
    uint32_t cand_docs[TO_BLOCK_LEN];
    float cand_scores[TO_BLOCK_LEN];
    size_t ncand = 0;
    for (size_t i = start_idx; i < end_idx; i++) {
      uint32_t d = bc->docs[i];
      float ls = to_bm25_score(&leader->w, ts_fieldnorm_id(leader, d), bc->freqs[i]);
      cand_docs[ncand] = d;
      cand_scores[ncand] = ls;
      ncand += (ls > score_threshold) ? 1u : 0u;
    }
    if (ncand == 0) {
      doc = window_end + 1;
      continue;
    }
    float running = 0.0f;
    for (size_t i = ns; i-- > 0;) {
      sec_suffix[i] = running;
      running += sec_bms[i];
    }
    for (size_t c = 0; c < ncand; c++) {
      uint32_t cd = cand_docs[c];
      float total = cand_scores[c];
      int ok = 1;
      for (size_t i = 0; i < ns; i++) {
        if (ts_doc(sec[i]) > cd) {
          ok = 0;
          break;
        }
        uint32_t r = to_sp_seek(&sec[i]->sp, cd);
        if (r != cd) {
          ok = 0;
          break;
        }
        total += ts_score(sec[i]);
        if (total + sec_suffix[i] <= threshold) {
          ok = 0;
          break;
        }
      }
      if (!ok) continue;
      if (total > threshold) {
        threshold = cb_push(cb, cd, total);
        if (leader_max + sec_global_max_sum <= threshold) return;
      }
    }
    doc = window_end + 1;
  }
}

/* ------------------------------------------------------------------ block_wand (union)
 * src/query/boolean_query/block_wand_union.rs:16-265 */
static void bw_sort_by_doc(term_scorer **s, size_t n) {
  for (size_t i = 1; i < n; i++) { /* stable */
    term_scorer *x = s[i];
    size_t j = i;
    while (j > 0 && ts_doc(s[j - 1]) > ts_doc(x)) {
      s[j] = s[j - 1];
      j--;
    }
    s[j] = x;
  }
}
static void bw_restore_ordering(term_scorer **s, size_t n, size_t ord) {
  uint32_t doc = ts_doc(s[ord]);
  for (size_t i = ord + 1; i < n; i++) {
    if (ts_doc(s[i]) >= doc) break;
    term_scorer *t = s[i];
    s[i] = s[i - 1];
    s[i - 1] = t;
  }
}
static void bw_swap_remove(term_scorer **s, size_t *n, size_t i) {
  s[i] = s[*n - 1];
  (*n)--;
}
static void block_wand_single_scorer(term_scorer *sc, float threshold, pruning_cb *cb) {
  /* :226-265 */
  uint32_t doc = ts_doc(sc);
  for (;;) {
    while (ts_block_max_score(sc) <= threshold) {
      uint32_t last = ts_last_doc_in_block(sc);
      if (last == TO_TERMINATED) return;
      doc = last + 1;
      to_block_postings_seek_block(&sc->sp.bp, doc);
    }
    doc = to_sp_seek(&sc->sp, doc);
    if (doc == TO_TERMINATED) break;
    for (;;) {
      float s = ts_score(sc);
      if (s > threshold) threshold = cb_push(cb, doc, s);
      if (doc == ts_last_doc_in_block(sc)) break;
      doc = to_sp_advance(&sc->sp);
      if (doc == TO_TERMINATED) return;
    }
    doc += 1;
    to_block_postings_seek_block(&sc->sp.bp, doc);
  }
}
static void block_wand(term_scorer **s, size_t n, float threshold, pruning_cb *cb) {
  /* retain non-terminated (:152) */
  size_t w = 0;
  for (size_t i = 0; i < n; i++)
    if (ts_doc(s[i]) < TO_TERMINATED) s[w++] = s[i];
  n = w;
  if (n == 0) return;
  if (n == 1) {
    block_wand_single_scorer(s[0], threshold, cb);
    return;
  }
  bw_sort_by_doc(s, n);
  for (;;) {
    /* find_pivot_doc :16-43 */
    float max_score = 0.0f;
    size_t before = 0;
    uint32_t pivot_doc = TO_TERMINATED;
    while (before < n) {
      max_score += s[before]->max_score;
      if (max_score > threshold) {
        pivot_doc = ts_doc(s[before]);
        break;
      }
      before++;
    }
    if (pivot_doc == TO_TERMINATED) return;
    size_t pivot_len = before + 1;
    while (pivot_len < n && ts_doc(s[pivot_len]) == pivot_doc) pivot_len++;

    float ub = 0.0f;
    for (size_t i = 0; i < pivot_len; i++) {
      to_block_postings_seek_block(&s[i]->sp.bp, pivot_doc);
      ub += ts_block_max_score(s[i]);
    }
    if (ub <= threshold) {
      /* block_max_was_too_low_advance_one_scorer :49-80 */
      size_t to_seek = pivot_len - 1;
      float gmax = s[to_seek]->max_score;
      uint32_t after = ts_last_doc_in_block(s[to_seek]);
      for (size_t o = pivot_len - 1; o-- > 0;) {
        if (ts_last_doc_in_block(s[o]) <= after) after = ts_last_doc_in_block(s[o]);
        if (s[o]->max_score > gmax) {
          gmax = s[o]->max_score;
          to_seek = o;
        }
      }
      if (after != TO_TERMINATED) after += 1;
      for (size_t i = pivot_len; i < n; i++)
        if (ts_doc(s[i]) <= after) after = ts_doc(s[i]);
      to_sp_seek(&s[to_seek]->sp, after);
      bw_restore_ordering(s, n, to_seek);
      continue;
    }
    /* align_scorers :101-124 */
    int aligned = 1;
    for (size_t i = before; i-- > 0;) {
      uint32_t nd = to_sp_seek(&s[i]->sp, pivot_doc);
      if (nd != pivot_doc) {
        if (nd == TO_TERMINATED) bw_swap_remove(s, &n, i);
        if (i < n) bw_restore_ordering(s, n, i);
        aligned = 0;
        break;
      }
    }
    if (!aligned) continue;
    float score = 0.0f;
    for (size_t i = 0; i < pivot_len; i++) score += ts_score(s[i]);
    if (score > threshold) threshold = cb_push(cb, pivot_doc, score);
    /* advance_all_scorers_on_pivot :129-143 */
    for (size_t i = 0; i < pivot_len; i++) to_sp_advance(&s[i]->sp);
    size_t i = 0;
    while (i != n) {
      if (ts_doc(s[i]) == TO_TERMINATED)
        bw_swap_remove(s, &n, i);
      else
        i++;
    }
    /* sort_by_key is a stable sort */
    bw_sort_by_doc(s, n);
  }
}

/* ------------------------------------------------------------------ exhaustive AND
 * Doc set: Intersection (src/query/intersection.rs:120-179).  Score: summed leader-first then
 * secondaries in ascending doc_freq, the order block_wand_intersection uses (:144-165), which
 * is what TopDocs actually observes for an all-term AND. */
typedef void (*match_fn)(void *ctx, uint32_t doc, float score);
static void exhaustive_and(term_scorer **s, size_t n, match_fn fn, void *ctx) {
  sort_by_size_hint(s, n);
  /* go_to_first_doc :66-79 */
  uint32_t cand = 0;
  for (size_t i = 0; i < n; i++)
    if (ts_doc(s[i]) > cand) cand = ts_doc(s[i]);
  for (;;) {
    int again = 0;
    for (size_t i = 0; i < n; i++) {
      uint32_t d = to_sp_seek(&s[i]->sp, cand);
      if (d > cand) {
        cand = d;
        again = 1;
        break;
      }
    }
    if (!again) break;
  }
  while (cand < TO_TERMINATED) {
    float total = ts_score(s[0]);
    for (size_t i = 1; i < n; i++) total += ts_score(s[i]);
    fn(ctx, cand, total);
    /* advance :121-179 */
    uint32_t c = ts_doc(s[0]) + 1;
    for (;;) {
      if (c >= TO_TERMINATED) {
        cand = TO_TERMINATED;
        break;
      }
      c = to_sp_seek(&s[0]->sp, c);
      if (c == TO_TERMINATED) {
        cand = TO_TERMINATED;
        break;
      }
      int all = 1;
      for (size_t i = 1; i < n; i++) {
        uint32_t d = to_sp_seek(&s[i]->sp, c);
        if (d != c) {
          c = d;
          all = 0;
          break;
        }
      }
      if (all) {
        cand = c;
        break;
      }
    }
  }
}

/* ------------------------------------------------------------------ exhaustive OR
 * BufferedUnionScorer (src/query/union/buffered_union.rs:11-12,63-158) with SumCombiner
 * (score_combiner.rs:39-56): 4096-doc horizon, per-doc accumulators updated scorer by scorer
 * in the given term order. */
#define TO_HORIZON 4096u
static void exhaustive_or(term_scorer **s, size_t n, match_fn fn, void *ctx) {
  static __thread float acc[TO_HORIZON];
  static __thread uint8_t present[TO_HORIZON];
  for (;;) {
    uint32_t min_doc = TO_TERMINATED;
    for (size_t i = 0; i < n; i++)
      if (ts_doc(s[i]) < min_doc) min_doc = ts_doc(s[i]);
    if (min_doc == TO_TERMINATED) return;
    memset(present, 0, sizeof present);
    for (uint32_t i = 0; i < TO_HORIZON; i++) acc[i] = 0.0f;
    for (size_t i = 0; i < n; i++) {
      for (;;) {
        uint32_t d = ts_doc(s[i]);
        if (d == TO_TERMINATED || d - min_doc >= TO_HORIZON) break;
        acc[d - min_doc] += ts_score(s[i]);
        present[d - min_doc] = 1;
        to_sp_advance(&s[i]->sp);
      }
    }
    for (uint32_t i = 0; i < TO_HORIZON; i++)
      if (present[i]) fn(ctx, min_doc + i, acc[i]);
  }
}

/* ------------------------------------------------------------------ exact phrase
 * src/query/phrase_query/phrase_scorer.rs:82-136,347-587 (slop = 0) */
typedef struct {
  term_scorer *ts;
  uint32_t offset; /* max_offset - term_offset (:372-385) */
} phrase_term;

static size_t pos_intersection(uint32_t *left, size_t ln, const uint32_t *right, size_t rn) {
  size_t li = 0, ri = 0, c = 0;
  while (li < ln && ri < rn) {
    if (left[li] < right[ri])
      li++;
    else if (left[li] == right[ri]) {
      left[c++] = left[li];
      li++;
      ri++;
    } else
      ri++;
  }
  return c;
}
static size_t pos_intersection_count(const uint32_t *left, size_t ln, const uint32_t *right,
                                     size_t rn) {
  size_t li = 0, ri = 0, c = 0;
  while (li < ln && ri < rn) {
    if (left[li] < right[ri])
      li++;
    else if (left[li] == right[ri]) {
      c++;
      li++;
      ri++;
    } else
      ri++;
  }
  return c;
}
typedef struct {
  uint32_t *left, *right;
  size_t cap;
} pos_bufs;
static void pos_bufs_reserve(pos_bufs *b, size_t n) {
  if (n <= b->cap) return;
  size_t cap = b->cap ? b->cap : 256;
  while (cap < n) cap *= 2;
  b->left = (uint32_t *)realloc(b->left, cap * sizeof(uint32_t));
  b->right = (uint32_t *)realloc(b->right, cap * sizeof(uint32_t));
  b->cap = cap;
}
static uint32_t phrase_count(phrase_term *pt, size_t n, pos_bufs *b) {
  /* compute_phrase_match :463-507 + intersection_count :437-461 */
  pos_bufs_reserve(b, to_sp_term_freq(&pt[0].ts->sp));
  size_t ln = to_sp_positions_with_offset(&pt[0].ts->sp, pt[0].offset, b->left);
  for (size_t i = 1; i + 1 < n; i++) {
    pos_bufs_reserve(b, to_sp_term_freq(&pt[i].ts->sp));
    size_t rn = to_sp_positions_with_offset(&pt[i].ts->sp, pt[i].offset, b->right);
    ln = pos_intersection(b->left, ln, b->right, rn);
    if (ln == 0) return 0;
  }
  pos_bufs_reserve(b, to_sp_term_freq(&pt[n - 1].ts->sp));
  size_t rn = to_sp_positions_with_offset(&pt[n - 1].ts->sp, pt[n - 1].offset, b->right);
  return (uint32_t)pos_intersection_count(b->left, ln, b->right, rn);
}
static void exhaustive_phrase(phrase_term *pt, size_t n, const to_bm25 *w, match_fn fn,
                              void *ctx) {
  /* Intersection::new sorts by cost (= doc_freq for SegmentPostings), stable */
  for (size_t i = 1; i < n; i++) {
    phrase_term x = pt[i];
    size_t j = i;
    while (j > 0 && pt[j - 1].ts->sp.bp.doc_freq > x.ts->sp.bp.doc_freq) {
      pt[j] = pt[j - 1];
      j--;
    }
    pt[j] = x;
  }
  pos_bufs b = {0};
  uint32_t cand = 0;
  for (size_t i = 0; i < n; i++)
    if (ts_doc(pt[i].ts) > cand) cand = ts_doc(pt[i].ts);
  for (;;) {
    int again = 0;
    for (size_t i = 0; i < n; i++) {
      uint32_t d = to_sp_seek(&pt[i].ts->sp, cand);
      if (d > cand) {
        cand = d;
        again = 1;
        break;
      }
    }
    if (!again) break;
  }
  while (cand < TO_TERMINATED) {
    uint32_t cnt = phrase_count(pt, n, &b);
    if (cnt > 0) {
      uint8_t fid = ts_fieldnorm_id(pt[0].ts, cand);
      fn(ctx, cand, to_bm25_score(w, fid, cnt)); /* :578-586 */
    }
    uint32_t c = ts_doc(pt[0].ts) + 1;
    for (;;) {
      if (c >= TO_TERMINATED) {
        cand = TO_TERMINATED;
        break;
      }
      c = to_sp_seek(&pt[0].ts->sp, c);
      if (c == TO_TERMINATED) {
        cand = TO_TERMINATED;
        break;
      }
      int all = 1;
      for (size_t i = 1; i < n; i++) {
        uint32_t d = to_sp_seek(&pt[i].ts->sp, c);
        if (d != c) {
          c = d;
          all = 0;
          break;
        }
      }
      if (all) {
        cand = c;
        break;
      }
    }
  }
  free(b.left);
  free(b.right);
}

/* ------------------------------------------------------------------ top-level */
typedef struct {
  to_topn_heap heap;
  float threshold;
} topk_ctx;
static void topk_match(void *ctx, uint32_t doc, float score) {
  /* for_each_pruning_scorer (query/weight.rs:47-60): score > threshold => callback */
  topk_ctx *c = (topk_ctx *)ctx;
  if (score > c->threshold) {
    to_topn_push(&c->heap, score, doc);
    c->threshold = c->heap.has_threshold ? c->heap.threshold : -3.40282347e+38f;
  }
}
typedef struct {
  uint32_t *docs;
  float *scores;
  size_t cap, n;
} all_ctx;
static void all_match(void *ctx, uint32_t doc, float score) {
  all_ctx *c = (all_ctx *)ctx;
  if (c->n < c->cap) {
    c->docs[c->n] = doc;
    c->scores[c->n] = score;
  }
  c->n++;
}

#define TO_MAX_TERMS 64
static int open_scorers(const to_segment_view *seg, const to_query *q, term_scorer *store,
                        term_scorer **ptrs, int requested) {
  if (q->n_terms > TO_MAX_TERMS) return -1;
  for (uint32_t i = 0; i < q->n_terms; i++) {
    const to_bm25 *w = (q->mode == TO_MODE_PHRASE) ? &q->weights[0] : &q->weights[i];
    if (ts_open(&store[i], seg, &q->terms[i], w, requested)) return -1;
    ptrs[i] = &store[i];
  }
  return 0;
}
static void run_bool(const to_segment_view *seg, const to_query *q, match_fn fn, void *ctx);
static void run_exhaustive(const to_segment_view *seg, const to_query *q, match_fn fn,
                           void *ctx) {
  if (q->mode == TO_MODE_BOOL) {
    run_bool(seg, q, fn, ctx);
    return;
  }
  term_scorer *store = (term_scorer *)malloc(sizeof(term_scorer) * (q->n_terms ? q->n_terms : 1));
  term_scorer *ptrs[TO_MAX_TERMS];
  if (q->mode == TO_MODE_AND) {
    /* an absent term => empty intersection (EmptyScorer, term_weight.rs:179-190) */
    for (uint32_t i = 0; i < q->n_terms; i++)
      if (q->terms[i].doc_freq == 0) goto done;
    if (open_scorers(seg, q, store, ptrs, TO_WITH_FREQS)) goto done;
    if (q->n_terms == 1) {
      term_scorer *t = ptrs[0];
      for (uint32_t d = ts_doc(t); d < TO_TERMINATED; d = to_sp_advance(&t->sp))
        fn(ctx, d, ts_score(t));
    } else {
      exhaustive_and(ptrs, q->n_terms, fn, ctx);
    }
  } else if (q->mode == TO_MODE_OR) {
    size_t n = 0;
    for (uint32_t i = 0; i < q->n_terms; i++) {
      if (q->terms[i].doc_freq == 0) continue;
      if (ts_open(&store[n], seg, &q->terms[i], &q->weights[i], TO_WITH_FREQS)) goto done;
      ptrs[n] = &store[n];
      n++;
    }
    if (n) exhaustive_or(ptrs, n, fn, ctx);
  } else {
    /* phrase_weight.rs:53-62: any absent term => no scorer */
    for (uint32_t i = 0; i < q->n_terms; i++)
      if (q->terms[i].doc_freq == 0) goto done;
    if (open_scorers(seg, q, store, ptrs, TO_WITH_FREQS_AND_POSITIONS)) goto done;
    phrase_term pt[TO_MAX_TERMS];
    uint32_t max_off = 0;
    for (uint32_t i = 0; i < q->n_terms; i++)
      if (q->phrase_offsets[i] > max_off) max_off = q->phrase_offsets[i];
    for (uint32_t i = 0; i < q->n_terms; i++) {
      pt[i].ts = ptrs[i];
      pt[i].offset = max_off - q->phrase_offsets[i];
    }
    exhaustive_phrase(pt, q->n_terms, &q->weights[0], fn, ctx);
  }
done:
  free(store);
}

size_t to_search_exhaustive(const to_segment_view *seg, const to_query *q, to_hit *out) {
  topk_ctx c;
  to_topn_init(&c.heap, q->k);
  c.threshold = -3.40282347e+38f;
  run_exhaustive(seg, q, topk_match, &c);
  size_t n = to_topn_into_vec(&c.heap, out);
  to_topn_free(&c.heap);
  return n;
}
size_t to_match_all(const to_segment_view *seg, const to_query *q, uint32_t *docs, float *scores,
                    size_t cap) {
  all_ctx c = {docs, scores, cap, 0};
  run_exhaustive(seg, q, all_match, &c);
  return c.n;
}
size_t to_search_pruned(const to_segment_view *seg, const to_query *q, to_hit *out) {
  /* the generic scorer tree has no block-max executor: for_each_pruning_scorer scores every doc */
  if (q->mode == TO_MODE_BOOL) return to_search_exhaustive(seg, q, out);
  to_topn_heap heap;
  to_topn_init(&heap, q->k);
  pruning_cb cb = {&heap};
  const float MINF = -3.40282347e+38f; /* Score::MIN */
  term_scorer *store = (term_scorer *)malloc(sizeof(term_scorer) * (q->n_terms ? q->n_terms : 1));
  term_scorer *ptrs[TO_MAX_TERMS];
  if (q->mode == TO_MODE_AND) {
    int absent = 0;
    for (uint32_t i = 0; i < q->n_terms; i++)
      if (q->terms[i].doc_freq == 0) absent = 1;
    if (!absent && open_scorers(seg, q, store, ptrs, TO_WITH_FREQS) == 0) {
      if (q->n_terms == 1)
        block_wand_single_scorer(ptrs[0], MINF, &cb); /* term_weight.rs:118-141 */
      else
        block_wand_intersection(ptrs, q->n_terms, MINF, &cb);
    }
  } else if (q->mode == TO_MODE_OR) {
    size_t n = 0;
    int bad = 0;
    for (uint32_t i = 0; i < q->n_terms; i++) {
      if (q->terms[i].doc_freq == 0) continue;
      if (ts_open(&store[n], seg, &q->terms[i], &q->weights[i], TO_WITH_FREQS)) bad = 1;
      ptrs[n] = &store[n];
      n++;
    }
    if (!bad && n) block_wand(ptrs, n, MINF, &cb);
  } else {
    free(store);
    to_topn_free(&heap);
    /* PhraseWeight has no specialised pruning: for_each_pruning_scorer over the scorer */
    return to_search_exhaustive(seg, q, out);
  }
  size_t n = to_topn_into_vec(&heap, out);
  free(store);
  to_topn_free(&heap);
  return n;
}

/* ------------------------------------------------------------------ decode helpers */
size_t to_decode_postings(const to_segment_view *seg, const to_term_info *ti, uint32_t *docs,
                          uint32_t *tfs) {
  term_scorer t;
  if (ti->doc_freq == 0) return 0;
  if (ts_open(&t, seg, ti, NULL, TO_WITH_FREQS)) return 0;
  size_t n = 0;
  for (uint32_t d = ts_doc(&t); d < TO_TERMINATED; d = to_sp_advance(&t.sp)) {
    docs[n] = d;
    if (tfs) tfs[n] = to_sp_term_freq(&t.sp);
    n++;
  }
  return n;
}
size_t to_decode_positions(const to_segment_view *seg, const to_term_info *ti, uint32_t *out,
                           size_t cap) {
  term_scorer t;
  if (ti->doc_freq == 0) return 0;
  if (ts_open(&t, seg, ti, NULL, TO_WITH_FREQS_AND_POSITIONS)) return 0;
  size_t n = 0;
  uint32_t *tmp = NULL;
  size_t tmp_cap = 0;
  for (uint32_t d = ts_doc(&t); d < TO_TERMINATED; d = to_sp_advance(&t.sp)) {
    uint32_t tf = to_sp_term_freq(&t.sp);
    if (tf > tmp_cap) {
      tmp_cap = tf * 2;
      tmp = (uint32_t *)realloc(tmp, tmp_cap * sizeof(uint32_t));
    }
    to_sp_positions_with_offset(&t.sp, 0, tmp);
    for (uint32_t i = 0; i < tf; i++) {
      if (n < cap) out[n] = tmp[i];
      n++;
    }
  }
  free(tmp);
  return n;
}

/* ------------------------------------------------------------------ TermScorer handle API
 * (lets tests replay the reference's TermScorer unit tests step by step) */
void *to_ts_new(const to_segment_view *seg, const to_term_info *ti, const to_bm25 *w) {
  term_scorer *t = (term_scorer *)malloc(sizeof *t);
  if (ts_open(t, seg, ti, w, TO_WITH_FREQS)) {
    free(t);
    return NULL;
  }
  return t;
}
void to_ts_free(void *t) { free(t); }
uint32_t to_ts_doc(void *t) { return ts_doc((term_scorer *)t); }
uint32_t to_ts_advance(void *t) { return to_sp_advance(&((term_scorer *)t)->sp); }
uint32_t to_ts_seek(void *t, uint32_t target) { return to_sp_seek(&((term_scorer *)t)->sp, target); }
void to_ts_seek_block(void *t, uint32_t target) {
  to_block_postings_seek_block(&((term_scorer *)t)->sp.bp, target);
}
uint32_t to_ts_term_freq(void *t) { return to_sp_term_freq(&((term_scorer *)t)->sp); }
float to_ts_score(void *t) { return ts_score((term_scorer *)t); }
float to_ts_block_max_score(void *t) { return ts_block_max_score((term_scorer *)t); }
float to_ts_max_score(void *t) { return ((term_scorer *)t)->max_score; }
uint32_t to_ts_last_doc_in_block(void *t) { return ts_last_doc_in_block((term_scorer *)t); }

/* ================================================================== generic scorer tree
 * What BooleanWeight::complex_scorer builds when the query is not a plain term intersection /
 * union (boolean_weight.rs:236-431): Intersection (intersection.rs:20-56,120-234),
 * BufferedUnionScorer with SumCombiner (union/buffered_union.rs:63-316), Disjunction
 * (disjunction.rs:85-170), RequiredOptionalScorer (reqopt_scorer.rs:36-99), Exclude
 * (exclude.rs:31-115), driven by for_each_pruning_scorer (weight.rs:47-60).  Clauses are terms or
 * unions of terms (the shapes the device path takes, include/tantivy_amd.h TQ_MODE_BOOL). */
typedef struct gscorer gscorer;
enum { GS_TERM, GS_UNION, GS_INTER, GS_REQOPT, GS_EXCLUDE, GS_DISJ, GS_EMPTY, GS_LIST };
#define GS_HORIZON 4096u
struct gscorer {
  int kind;
  term_scorer *term;           /* GS_TERM */
  gscorer **kids;              /* UNION: active docsets; INTER: left,right,others; DISJ: heap */
  size_t n_kids;
  /* union window */
  uint64_t *bits;              /* 64 words */
  float *acc;                  /* 4096 sums (SumCombiner) */
  size_t bucket_idx;
  uint32_t window_start, doc;
  float score;
  /* reqopt / exclude */
  gscorer *req, *opt;
  gscorer **excl;
  size_t n_excl;
  int has_cache;
  float cache;
  /* disjunction */
  size_t min_match;
  uint32_t *heap_doc;          /* ScorerWrapper::current_doc per kid */
  /* GS_LIST: a PhraseScorer as a docset — its docs / scores (exhaustive_phrase, pinned by the reference's phrase
   * KATs) walked by a cursor, its cost = PhraseScorer::cost (phrase_scorer.rs:566-573) */
  uint32_t *l_docs;
  float *l_scores;
  size_t l_n, l_at;
  uint64_t l_cost;
  int invalid; /* GS_INTER: a seek_danger that failed half-way left the members on different docs (see below) */
};
/* Intersection::seek_danger (intersection.rs:193-210) returns at the first member that misses: the members before it
 * stand on `target`, the missing one beyond it, the others where they were — and Intersection::doc() (= left.doc())
 * names a doc >= target that the intersection need not hold: the "danger zone" contract (docset.rs:70-111) lets the
 * CALLER re-seek.  BufferedUnionScorer::seek does so only `if docset.doc() < target` (buffered_union.rs:254-259): when
 * ANOTHER member of the union hit `target`, a half-seeked intersection is refilled as it stands and adds left.score()
 * + right.score() of two DIFFERENT docs to the sum of left's doc — `+a +((+b +c) d)` on a doc that holds a, b, d but
 * not c; or, when left's doc is one no member holds, reports that doc as a match of the union (a doc too many in
 * `+a +(...)` if a holds it).  g_reseek_invalid = 1 (default): such a member is re-seeked,
 * i.e. the scorer tree's intended semantics — what the dense restatement (oracle.py tree_match_all_general) and
 * the device compute; 0: the reference's code path as written (tests/test_tree_oracle_cpu.py shows both). */
static int g_reseek_invalid = 1;
void to_set_union_reseek_invalid(int on) { g_reseek_invalid = on; }
static int gs_is_invalid(const gscorer *s) {
  switch (s->kind) {
    case GS_INTER: return s->invalid;
    case GS_REQOPT: case GS_EXCLUDE: return gs_is_invalid(s->req);
    default: return 0;
  }
}
static uint32_t gs_doc(gscorer *s);
static uint32_t gs_advance(gscorer *s);
static uint32_t gs_seek(gscorer *s, uint32_t target);
static float gs_score(gscorer *s);
/* SeekDangerResult: returns 1 = Found, else 0 with *lower = SeekLowerBound */
static int gs_seek_danger(gscorer *s, uint32_t target, uint32_t *lower);

static uint64_t gs_cost(gscorer *s) {
  switch (s->kind) {
    case GS_TERM: return s->term->sp.bp.doc_freq; /* SegmentPostings::size_hint = doc_freq */
    case GS_UNION: { /* buffered_union.rs:326-328 */
      uint64_t c = 0;
      for (size_t i = 0; i < s->n_kids; i++) c += gs_cost(s->kids[i]);
      return c;
    }
    case GS_INTER: return gs_cost(s->kids[0]); /* intersection.rs:227-232 */
    case GS_REQOPT: return gs_cost(s->req);
    case GS_EXCLUDE: return gs_cost(s->req); /* DocSet::cost default = size_hint of underlying */
    case GS_DISJ: { /* disjunction.rs:149-155 */
      uint64_t c = 0;
      for (size_t i = 0; i < s->n_kids; i++) {
        uint64_t k = gs_cost(s->kids[i]);
        if (k > c) c = k;
      }
      return c;
    }
    case GS_LIST: return s->l_cost;
    default: return 0;
  }
}
static int default_seek_danger(gscorer *s, uint32_t target, uint32_t *lower) {
  /* docset.rs:90-111 */
  if (target >= TO_TERMINATED) {
    *lower = target;
    return 0;
  }
  uint32_t doc = gs_doc(s);
  if (doc < target) doc = gs_seek(s, target);
  if (doc == target) return 1;
  *lower = doc;
  return 0;
}
static uint32_t default_seek(gscorer *s, uint32_t target) {
  /* docset.rs: advance until >= target */
  uint32_t doc = gs_doc(s);
  while (doc < target) doc = gs_advance(s);
  return doc;
}

/* ---- union */
static void union_drain_refill(gscorer *u, uint32_t min_doc) {
  /* refill() helper, buffered_union.rs:63-87, with unordered_drain_filter's swap_remove */
  size_t i = 0;
  while (i < u->n_kids) {
    gscorer *sc = u->kids[i];
    const uint32_t horizon = min_doc + GS_HORIZON;
    int consumed = 0;
    for (;;) {
      uint32_t doc = gs_doc(sc);
      if (doc >= horizon) break;
      uint32_t delta = doc - min_doc;
      u->bits[delta / 64] |= 1ull << (delta % 64);
      u->acc[delta] += gs_score(sc);
      if (gs_advance(sc) == TO_TERMINATED) {
        consumed = 1;
        break;
      }
    }
    if (consumed) {
      u->kids[i] = u->kids[u->n_kids - 1];
      u->n_kids--;
    } else {
      i++;
    }
  }
}
static int union_refill(gscorer *u) { /* :118-137 */
  if (u->n_kids == 0) return 0;
  uint32_t min_doc = TO_TERMINATED;
  for (size_t i = 0; i < u->n_kids; i++) {
    uint32_t d = gs_doc(u->kids[i]);
    if (d < min_doc) min_doc = d;
  }
  u->window_start = min_doc;
  u->bucket_idx = 0;
  u->doc = min_doc;
  union_drain_refill(u, min_doc);
  return 1;
}
static int union_advance_buffered(gscorer *u) { /* :139-155 */
  while (u->bucket_idx < 64) {
    uint64_t w = u->bits[u->bucket_idx];
    if (w) {
      uint32_t val = (uint32_t)__builtin_ctzll(w);
      u->bits[u->bucket_idx] = w & (w - 1);
      uint32_t delta = val + (uint32_t)u->bucket_idx * 64u;
      u->doc = u->window_start + delta;
      u->score = u->acc[delta];
      u->acc[delta] = 0.0f;
      return 1;
    }
    u->bucket_idx++;
  }
  return 0;
}
static uint32_t union_advance(gscorer *u) { /* :170-182 */
  if (union_advance_buffered(u)) return u->doc;
  if (!union_refill(u)) {
    u->doc = TO_TERMINATED;
    return TO_TERMINATED;
  }
  if (!union_advance_buffered(u)) return TO_TERMINATED;
  return u->doc;
}
static uint32_t union_seek(gscorer *u, uint32_t target) { /* :226-275 */
  if (u->doc >= target) return u->doc;
  uint32_t gap = target - u->window_start;
  if (gap < GS_HORIZON) {
    size_t nb = gap / 64;
    for (size_t b = u->bucket_idx; b < nb; b++) u->bits[b] = 0;
    for (size_t k = u->bucket_idx * 64; k < nb * 64; k++) u->acc[k] = 0.0f;
    u->bucket_idx = nb;
    uint32_t doc = u->doc;
    while (doc < target) doc = union_advance(u);
    return doc;
  }
  memset(u->bits, 0, 64 * sizeof(uint64_t));
  memset(u->acc, 0, GS_HORIZON * sizeof(float));
  size_t i = 0;
  while (i < u->n_kids) {
    gscorer *sc = u->kids[i];
    if (gs_doc(sc) < target || (g_reseek_invalid && gs_is_invalid(sc))) gs_seek(sc, target);
    if (gs_doc(sc) == TO_TERMINATED) {
      u->kids[i] = u->kids[u->n_kids - 1];
      u->n_kids--;
    } else {
      i++;
    }
  }
  if (!union_refill(u)) {
    u->doc = TO_TERMINATED;
    return TO_TERMINATED;
  }
  return union_advance(u);
}
static int union_seek_danger(gscorer *u, uint32_t target, uint32_t *lower) { /* :277-321 */
  if (target >= TO_TERMINATED) {
    *lower = TO_TERMINATED;
    return 0;
  }
  /* A second place where the code as written and the scorer tree's semantics part (found by a 500-seed fuzz soak,
   * round 6).  A union that is a MEMBER of another union is asked seek_danger(target) by its parent whatever its own
   * position (:296-306 has no `docset.doc() < target` guard, seek() :258-262 has one).  If it stands PAST the target —
   * the parent's refill drained it to the parent's horizon, its own next window starts at its first doc beyond —
   * then target < window_start, is_in_horizon (wrapping_sub, :157-161) is false, and it answers with the lower bound of
   * its MEMBERS, which its own refill has already drained a whole window ahead: the docs it holds buffered, its
   * current doc first, are skipped.  `+a (b c) (d e)` with minimum_number_should_match = 1 loses up to a 4096-doc
   * window of matches at a window boundary (seed 575: 1432 of 24406 docs); with one term and one union as Should
   * clauses, one doc.  g_reseek_invalid = 1 (default, the intended semantics: what the dense restatement and the
   * device compute): a union standing at or past the target answers with its own position. */
  if (g_reseek_invalid && u->doc >= target) {
    if (u->doc == target) return 1;
    *lower = u->doc;
    return 0;
  }
  if ((uint32_t)(target - u->window_start) < GS_HORIZON) {
    uint32_t d = union_seek(u, target);
    if (d == target) return 1;
    *lower = d;
    return 0;
  }
  int hit = 0;
  uint32_t min_new = TO_TERMINATED;
  for (size_t i = 0; i < u->n_kids; i++) {
    uint32_t lb;
    if (gs_seek_danger(u->kids[i], target, &lb)) {
      hit = 1;
      break;
    }
    if (lb < min_new) min_new = lb;
  }
  if (hit) {
    union_seek(u, target);
    return 1;
  }
  *lower = min_new;
  return 0;
}

/* ---- intersection */
static uint32_t go_to_first_doc(gscorer **ds, size_t n) { /* intersection.rs:66-79 */
  uint32_t candidate = 0;
  for (size_t i = 0; i < n; i++) {
    uint32_t d = gs_doc(ds[i]);
    if (d > candidate) candidate = d;
  }
  for (;;) {
    int again = 0;
    for (size_t i = 0; i < n; i++) {
      uint32_t sd = gs_seek(ds[i], candidate);
      if (sd > candidate) {
        candidate = gs_doc(ds[i]);
        again = 1;
        break;
      }
    }
    if (!again) return candidate;
  }
}
static uint32_t inter_advance(gscorer *s) { /* :120-175 */
  gscorer *left = s->kids[0], *right = s->kids[1];
  uint32_t candidate = gs_doc(left) + 1;
  while (candidate < TO_TERMINATED) {
    candidate = gs_seek(left, candidate);
    uint32_t lb;
    if (!gs_seek_danger(right, candidate, &lb)) {
      candidate = lb;
      continue;
    }
    int restart = 0;
    for (size_t i = 2; i < s->n_kids; i++) {
      if (!gs_seek_danger(s->kids[i], candidate, &lb)) {
        candidate = lb;
        restart = 1;
        break;
      }
    }
    if (restart) continue;
    return candidate;
  }
  gs_seek(left, TO_TERMINATED);
  return TO_TERMINATED;
}

/* ---- disjunction: a binary heap on ScorerWrapper::current_doc (min first) */
static void disj_sift_down(gscorer *s, size_t i) {
  size_t n = s->n_kids;
  for (;;) {
    size_t l = 2 * i + 1, r = l + 1, m = i;
    if (l < n && s->heap_doc[l] < s->heap_doc[m]) m = l;
    if (r < n && s->heap_doc[r] < s->heap_doc[m]) m = r;
    if (m == i) return;
    gscorer *t = s->kids[i];
    s->kids[i] = s->kids[m];
    s->kids[m] = t;
    uint32_t d = s->heap_doc[i];
    s->heap_doc[i] = s->heap_doc[m];
    s->heap_doc[m] = d;
    i = m;
  }
}
static uint32_t disj_advance(gscorer *s) { /* disjunction.rs:113-139 */
  size_t matches = 0;
  float sum = 0.0f;
  /* the reference pops / pushes; here the root is updated in place (same visiting order up to
   * ties, whose order the reference's BinaryHeap leaves unspecified) */
  while (s->n_kids) {
    uint32_t next = s->heap_doc[0];
    if (next == TO_TERMINATED) { /* drop exhausted chains */
      s->kids[0] = s->kids[s->n_kids - 1];
      s->heap_doc[0] = s->heap_doc[s->n_kids - 1];
      s->n_kids--;
      if (s->n_kids) disj_sift_down(s, 0);
      continue;
    }
    if (s->doc != next) {
      if (matches >= s->min_match) {
        s->score = sum;
        return s->doc;
      }
      matches = 0;
      s->doc = next;
      sum = 0.0f;
    }
    matches++;
    sum += gs_score(s->kids[0]);
    s->heap_doc[0] = gs_advance(s->kids[0]);
    disj_sift_down(s, 0);
  }
  if (matches < s->min_match) s->doc = TO_TERMINATED;
  s->score = sum;
  return s->doc;
}

/* ---- exclude */
static int excl_contains(gscorer *s, uint32_t doc) { /* exclude.rs:9-27 */
  for (size_t i = 0; i < s->n_excl; i++) {
    uint32_t lb;
    if (gs_seek_danger(s->excl[i], doc, &lb)) return 1;
  }
  return 0;
}
static uint32_t excl_advance(gscorer *s) { /* :63-73 */
  for (;;) {
    uint32_t c = gs_advance(s->req);
    if (c == TO_TERMINATED) return TO_TERMINATED;
    if (!excl_contains(s, c)) return c;
  }
}

static uint32_t gs_doc(gscorer *s) {
  switch (s->kind) {
    case GS_TERM: return ts_doc(s->term);
    case GS_UNION: case GS_DISJ: return s->doc;
    case GS_INTER: return gs_doc(s->kids[0]);
    case GS_REQOPT: case GS_EXCLUDE: return gs_doc(s->req);
    case GS_LIST: return s->l_at < s->l_n ? s->l_docs[s->l_at] : TO_TERMINATED;
    default: return TO_TERMINATED;
  }
}
static uint32_t gs_advance(gscorer *s) {
  switch (s->kind) {
    case GS_LIST:
      if (s->l_at < s->l_n) s->l_at++;
      return gs_doc(s);
    case GS_TERM: return to_sp_advance(&s->term->sp);
    case GS_UNION: return union_advance(s);
    case GS_INTER: s->invalid = 0; return inter_advance(s);
    case GS_REQOPT: s->has_cache = 0; return gs_advance(s->req);
    case GS_EXCLUDE: return excl_advance(s);
    case GS_DISJ: return disj_advance(s);
    default: return TO_TERMINATED;
  }
}
static uint32_t gs_seek(gscorer *s, uint32_t target) {
  switch (s->kind) {
    case GS_TERM: return to_sp_seek(&s->term->sp, target);
    case GS_UNION: return union_seek(s, target);
    case GS_INTER: { /* :177-187 */
      s->invalid = 0;
      gs_seek(s->kids[0], target);
      return go_to_first_doc(s->kids, s->n_kids);
    }
    case GS_REQOPT: s->has_cache = 0; return gs_seek(s->req, target);
    case GS_EXCLUDE: { /* :75-84 */
      uint32_t c = gs_seek(s->req, target);
      if (c == TO_TERMINATED) return TO_TERMINATED;
      if (!excl_contains(s, c)) return c;
      return excl_advance(s);
    }
    case GS_DISJ: return default_seek(s, target);
    case GS_LIST: return default_seek(s, target);
    default: return TO_TERMINATED;
  }
}
static int gs_seek_danger(gscorer *s, uint32_t target, uint32_t *lower) {
  switch (s->kind) {
    case GS_UNION: return union_seek_danger(s, target, lower);
    case GS_INTER: { /* :193-210 */
      for (size_t i = 0; i < s->n_kids; i++)
        if (!gs_seek_danger(s->kids[i], target, lower)) {
          s->invalid = 1; /* members 0..i moved (member i past the target), the others did not */
          return 0;
        }
      s->invalid = 0;
      return 1;
    }
    case GS_REQOPT: s->has_cache = 0; return gs_seek_danger(s->req, target, lower);
    case GS_EMPTY: *lower = TO_TERMINATED; return 0;
    default: return default_seek_danger(s, target, lower);
  }
}
static float gs_score(gscorer *s) {
  switch (s->kind) {
    case GS_TERM: return ts_score(s->term);
    case GS_UNION: case GS_DISJ: return s->score;
    case GS_INTER: { /* :325-329: left + right + sum(others) */
      float oth = 0.0f;
      for (size_t i = 2; i < s->n_kids; i++) oth += gs_score(s->kids[i]);
      return gs_score(s->kids[0]) + gs_score(s->kids[1]) + oth;
    }
    case GS_REQOPT: { /* reqopt_scorer.rs:85-98 */
      if (s->has_cache) return s->cache;
      uint32_t doc = gs_doc(s->req);
      float sc = 0.0f;
      sc += gs_score(s->req);
      if (gs_doc(s->opt) <= doc && gs_seek(s->opt, doc) == doc) sc += gs_score(s->opt);
      s->cache = sc;
      s->has_cache = 1;
      return sc;
    }
    case GS_EXCLUDE: return gs_score(s->req);
    case GS_LIST: return s->l_at < s->l_n ? s->l_scores[s->l_at] : 0.0f;
    default: return 0.0f;
  }
}

/* arena of nodes for one query */
typedef struct {
  gscorer nodes[8 * TO_MAX_TERMS + 8];
  size_t n_nodes;
  gscorer *ptrs[16 * TO_MAX_TERMS + 16];
  size_t n_ptrs;
  uint64_t *bits;
  float *acc;
  size_t n_windows;
  uint32_t heap_docs[4 * TO_MAX_TERMS]; /* one run per disjunction node */
  size_t n_heap;
} gs_arena;
static gscorer *gs_new(gs_arena *a, int kind) {
  gscorer *s = &a->nodes[a->n_nodes++];
  memset(s, 0, sizeof *s);
  s->kind = kind;
  return s;
}
static gscorer **gs_ptrs(gs_arena *a, size_t n) {
  gscorer **p = &a->ptrs[a->n_ptrs];
  a->n_ptrs += n;
  return p;
}
static void sort_by_cost(gscorer **v, size_t n) { /* stable insertion sort = sort_by_key(cost) */
  for (size_t i = 1; i < n; i++) {
    gscorer *x = v[i];
    uint64_t c = gs_cost(x);
    size_t j = i;
    while (j > 0 && gs_cost(v[j - 1]) > c) {
      v[j] = v[j - 1];
      j--;
    }
    v[j] = x;
  }
}
/* BufferedUnionScorer::build (:92-116); a single scorer stays itself (into_box_scorer :94-101) */
static gscorer *gs_make_union(gs_arena *a, gscorer **kids, size_t n) {
  if (n == 1) return kids[0];
  gscorer *u = gs_new(a, GS_UNION);
  u->kids = gs_ptrs(a, n);
  for (size_t i = 0; i < n; i++)
    if (gs_doc(kids[i]) != TO_TERMINATED) u->kids[u->n_kids++] = kids[i];
  u->bits = a->bits + 64 * a->n_windows;
  u->acc = a->acc + GS_HORIZON * a->n_windows;
  a->n_windows++;
  memset(u->bits, 0, 64 * sizeof(uint64_t));
  memset(u->acc, 0, GS_HORIZON * sizeof(float));
  u->bucket_idx = 64;
  if (union_refill(u))
    union_advance(u);
  else
    u->doc = TO_TERMINATED;
  return u;
}
/* intersect_scorers (intersection.rs:20-56) */
static gscorer *gs_make_inter(gs_arena *a, gscorer **kids, size_t n) {
  if (n == 0) return gs_new(a, GS_EMPTY);
  if (n == 1) return kids[0];
  gscorer *s = gs_new(a, GS_INTER);
  s->kids = gs_ptrs(a, n);
  memcpy(s->kids, kids, n * sizeof(gscorer *));
  s->n_kids = n;
  sort_by_cost(s->kids, n);
  if (go_to_first_doc(s->kids, n) == TO_TERMINATED) return gs_new(a, GS_EMPTY);
  return s;
}

/* BooleanWeight::complex_scorer (boolean_weight.rs:236-431) over sub-scorers that already exist (EmptyScorers removed by
 * the caller: an empty Must member makes the caller return NULL).  Returns NULL for an EmptyScorer.  must / should may
 * be extended in place (minimum == number of Should scorers turns them into Must, :293-298): room for both. */
static gscorer *gs_complex(gs_arena *a, gscorer **must, size_t n_must, gscorer **should, size_t n_should,
                           gscorer **excl, size_t n_excl, size_t msm) {
  if (msm > n_should) return NULL; /* :275-279 */
  gscorer *should_sc = NULL;
  int should_required = 0;
  if (n_should == 0) {
    /* Ignored */
  } else if (msm == 0 || msm == 1) {
    /* scorer_union over the clause scorers: all terms -> one BufferedUnionScorer over the terms;
     * otherwise a union over the clause scorers (:44-86) */
    should_sc = gs_make_union(a, should, n_should);
    should_required = msm == 1;
  } else if (msm == n_should) {
    for (size_t i = 0; i < n_should; i++) must[n_must++] = should[i]; /* :293-298 */
    n_should = 0;
  } else {
    gscorer *d = gs_new(a, GS_DISJ); /* scorer_disjunction :23-41 */
    d->kids = gs_ptrs(a, n_should);
    d->heap_doc = a->heap_docs + a->n_heap;
    a->n_heap += n_should;
    d->n_kids = n_should;
    for (size_t i = 0; i < n_should; i++) d->kids[i] = should[i];
    /* heapify on the scorers' current docs */
    for (size_t i = 0; i < n_should; i++) d->heap_doc[i] = gs_doc(d->kids[i]);
    for (size_t i = n_should; i-- > 0;) disj_sift_down(d, i);
    d->min_match = msm;
    d->doc = TO_TERMINATED;
    disj_advance(d);
    should_sc = d;
    should_required = 1;
  }
  gscorer *include = NULL;
  if (!should_sc) {
    if (n_must == 0) return NULL; /* no include scorer: EmptyScorer */
    include = gs_make_inter(a, must, n_must);
  } else if (!should_required) {
    if (n_must == 0) {
      include = should_sc; /* promoted to required (:361-372) */
    } else {
      gscorer *m = gs_make_inter(a, must, n_must);
      gscorer *r = gs_new(a, GS_REQOPT); /* :373-388 */
      r->req = m;
      r->opt = should_sc;
      include = r;
    }
  } else {
    if (n_must == 0) {
      include = should_sc;
    } else {
      gscorer *pair[2];
      pair[0] = gs_make_inter(a, must, n_must);
      pair[1] = should_sc;
      include = gs_make_inter(a, pair, 2); /* :401-409 */
    }
  }
  if (include->kind == GS_EMPTY) return NULL;
  if (n_excl == 0) return include;
  gscorer *e = gs_new(a, GS_EXCLUDE); /* :416-431; Exclude::new skips excluded heads (:38-52) */
  e->req = include;
  e->excl = gs_ptrs(a, n_excl);
  memcpy(e->excl, excl, n_excl * sizeof(gscorer *));
  e->n_excl = n_excl;
  while (gs_doc(include) != TO_TERMINATED) {
    if (!excl_contains(e, gs_doc(include))) break;
    gs_advance(include);
  }
  return e;
}

/* BooleanWeight::complex_scorer for clauses that are terms or unions of terms.  Returns NULL for
 * an empty result. */
static gscorer *gs_build(gs_arena *a, const to_segment_view *seg, const to_query *q,
                         term_scorer *store) {
  gscorer *clause[TO_MAX_TERMS];
  uint8_t clause_occ[TO_MAX_TERMS];
  uint32_t clause_id[TO_MAX_TERMS];
  gscorer *members[TO_MAX_TERMS][TO_MAX_TERMS > 16 ? 16 : TO_MAX_TERMS];
  size_t n_members[TO_MAX_TERMS];
  size_t n_clauses = 0;
  if (q->n_terms > 16) return NULL;
  for (uint32_t i = 0; i < q->n_terms; i++) {
    uint32_t id = q->clause_of ? q->clause_of[i] : i;
    size_t c = 0;
    while (c < n_clauses && clause_id[c] != id) c++;
    if (c == n_clauses) {
      clause_id[c] = id;
      clause_occ[c] = q->occurs[i];
      n_members[c] = 0;
      n_clauses++;
    }
    if (q->terms[i].doc_freq == 0) continue; /* EmptyScorer: drops out of its union */
    if (ts_open(&store[i], seg, &q->terms[i], &q->weights[i], TO_WITH_FREQS)) return NULL;
    gscorer *t = gs_new(a, GS_TERM);
    t->term = &store[i];
    members[c][n_members[c]++] = t;
  }
  gscorer *must[2 * TO_MAX_TERMS], *should[TO_MAX_TERMS], *excl[TO_MAX_TERMS];
  size_t n_must = 0, n_should = 0, n_excl = 0;
  for (size_t c = 0; c < n_clauses; c++) {
    if (n_members[c] == 0) { /* the clause's scorer is an EmptyScorer */
      if (clause_occ[c] == 1) return NULL; /* boolean_weight.rs:249-251 */
      continue;                            /* removed from should / exclude (:253-262) */
    }
    clause[c] = gs_make_union(a, members[c], n_members[c]);
    if (clause_occ[c] == 1) must[n_must++] = clause[c];
    else if (clause_occ[c] == 0) should[n_should++] = clause[c];
    else excl[n_excl++] = clause[c];
  }
  return gs_complex(a, must, n_must, should, n_should, excl, n_excl, q->min_should_match);
}

static void run_bool(const to_segment_view *seg, const to_query *q, match_fn fn, void *ctx) {
  term_scorer *store = (term_scorer *)malloc(sizeof(term_scorer) * (q->n_terms ? q->n_terms : 1));
  gs_arena *a = (gs_arena *)malloc(sizeof(gs_arena));
  a->n_nodes = a->n_ptrs = a->n_windows = a->n_heap = 0;
  a->bits = (uint64_t *)malloc((size_t)(q->n_terms + 2) * 64 * sizeof(uint64_t));
  a->acc = (float *)malloc((size_t)(q->n_terms + 2) * GS_HORIZON * sizeof(float));
  gscorer *s = gs_build(a, seg, q, store);
  if (s) {
    /* for_each_pruning_scorer (weight.rs:47-60); fn applies the threshold */
    for (uint32_t d = gs_doc(s); d != TO_TERMINATED; d = gs_advance(s)) fn(ctx, d, gs_score(s));
  }
  free(a->bits);
  free(a->acc);
  free(a);
  free(store);
}

/* ================================================================== nested boolean queries (any depth)
 * BooleanWeight::complex_scorer applied recursively (boolean_weight.rs:225-233: a clause's scorer is
 * `weight.scorer(...)` of ITS query — another complex_scorer for a nested BooleanQuery, a PhraseScorer for a
 * PhraseQuery, a TermScorer for a term), the tree given in prefix order: a node, then the subtrees of its children.
 * A PhraseScorer enters as a docset over its docs and scores (exhaustive_phrase above: the restatement the
 * reference's phrase KATs pin) with PhraseScorer::cost (phrase_scorer.rs:566-573: size_hint of the intersection
 * of its lists, size_hint.rs:11-36, * 10 * terms), which orders it among the Must scorers (intersection.rs:31). */
typedef struct {
  uint32_t *docs;
  float *scores;
  size_t n, cap;
} list_ctx;
static void list_match(void *ctx, uint32_t doc, float score) {
  list_ctx *c = (list_ctx *)ctx;
  if (c->n == c->cap) {
    c->cap = c->cap ? 2 * c->cap : 1024;
    c->docs = (uint32_t *)realloc(c->docs, c->cap * sizeof(uint32_t));
    c->scores = (float *)realloc(c->scores, c->cap * sizeof(float));
  }
  c->docs[c->n] = doc;
  c->scores[c->n] = score;
  c->n++;
}
typedef struct {
  gs_arena *a;
  const to_segment_view *seg;
  const to_tree_node *nodes;
  size_t n_nodes;
  const to_term_info *terms;
  const to_bm25 *weights;
  const uint32_t *phrase_offsets;
  term_scorer *store;   /* one per term */
  size_t n_terms;
  void *owned[4 * TO_MAX_TERMS]; /* arrays of the phrase leaves, freed by the caller */
  size_t n_owned;
  int error;
} tree_build;

/* the scorer of the subtree at *pos (advanced past it); NULL = EmptyScorer */
static gscorer *gs_build_node(tree_build *tb, size_t *pos) {
  if (*pos >= tb->n_nodes) {
    tb->error = 1;
    return NULL;
  }
  const to_tree_node *nd = &tb->nodes[(*pos)++];
  if (nd->kind == TO_TREE_TERM) {
    if (nd->first >= tb->n_terms) {
      tb->error = 1;
      return NULL;
    }
    if (tb->terms[nd->first].doc_freq == 0) return NULL; /* EmptyScorer (term_weight.rs:179-190) */
    if (ts_open(&tb->store[nd->first], tb->seg, &tb->terms[nd->first], &tb->weights[nd->first], TO_WITH_FREQS)) {
      tb->error = 1;
      return NULL;
    }
    gscorer *t = gs_new(tb->a, GS_TERM);
    t->term = &tb->store[nd->first];
    return t;
  }
  if (nd->kind == TO_TREE_PHRASE) {
    const uint32_t n = nd->n_kids;
    if (n < 2 || n > TO_MAX_TERMS || nd->first + n > tb->n_terms || tb->n_owned + 2 > 4 * TO_MAX_TERMS) {
      tb->error = 1;
      return NULL;
    }
    for (uint32_t i = 0; i < n; i++)
      if (tb->terms[nd->first + i].doc_freq == 0) return NULL; /* phrase_weight.rs:53-62 */
    phrase_term pt[TO_MAX_TERMS];
    uint32_t max_off = 0;
    double est = 0.0, smallest = 0.0, f = 1.3;
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t ti = nd->first + i;
      if (ts_open(&tb->store[ti], tb->seg, &tb->terms[ti], &tb->weights[nd->first], TO_WITH_FREQS_AND_POSITIONS)) {
        tb->error = 1;
        return NULL;
      }
      if (tb->phrase_offsets[ti] > max_off) max_off = tb->phrase_offsets[ti];
      /* estimate_intersection (size_hint.rs:11-36), lists in the caller's order as the device planner counts them */
      const double df = (double)tb->terms[ti].doc_freq;
      if (i == 0) {
        est = smallest = df;
      } else {
        f = f - 0.1 > 1.0 ? f - 0.1 : 1.0;
        est *= df / (double)(tb->seg->max_doc ? tb->seg->max_doc : 1u) * f;
        if (df < smallest) smallest = df;
      }
    }
    for (uint32_t i = 0; i < n; i++) {
      pt[i].ts = &tb->store[nd->first + i];
      pt[i].offset = max_off - tb->phrase_offsets[nd->first + i];
    }
    list_ctx lc = {NULL, NULL, 0, 0};
    exhaustive_phrase(pt, n, &tb->weights[nd->first], list_match, &lc);
    tb->owned[tb->n_owned++] = lc.docs;
    tb->owned[tb->n_owned++] = lc.scores;
    if (lc.n == 0) return NULL; /* (a PhraseScorer positioned on TERMINATED: removed like an EmptyScorer) */
    gscorer *l = gs_new(tb->a, GS_LIST);
    l->l_docs = lc.docs;
    l->l_scores = lc.scores;
    l->l_n = lc.n;
    l->l_at = 0;
    double r = round(est);
    if (smallest < r) r = smallest;
    l->l_cost = (uint64_t)r * 10u * n;
    return l;
  }
  if (nd->kind != TO_TREE_BOOL || nd->n_kids > TO_MAX_TERMS) {
    tb->error = 1;
    return NULL;
  }
  gscorer *must[2 * TO_MAX_TERMS], *should[TO_MAX_TERMS], *excl[TO_MAX_TERMS];
  size_t n_must = 0, n_should = 0, n_excl = 0;
  int empty_must = 0;
  for (uint32_t c = 0; c < nd->n_kids; c++) {
    if (*pos >= tb->n_nodes) {
      tb->error = 1;
      return NULL;
    }
    const uint8_t occ = tb->nodes[*pos].occur;
    gscorer *k = gs_build_node(tb, pos); /* (every subtree is walked: the positions of the later ones follow) */
    if (tb->error) return NULL;
    if (k && k->kind == GS_EMPTY) k = NULL;
    if (k && gs_doc(k) == TO_TERMINATED) k = NULL; /* :249-262 test `scorer.is::<EmptyScorer>()`; an exhausted scorer behaves alike */
    if (!k) {
      if (occ == 1) empty_must = 1; /* boolean_weight.rs:249-251 */
      continue;
    }
    if (occ == 1) must[n_must++] = k;
    else if (occ == 0) should[n_should++] = k;
    else excl[n_excl++] = k;
  }
  if (empty_must) return NULL;
  return gs_complex(tb->a, must, n_must, should, n_should, excl, n_excl, nd->msm);
}

size_t to_tree_match_all(const to_segment_view *seg, const to_tree_node *nodes, size_t n_nodes,
                         const to_term_info *terms, const to_bm25 *weights, const uint32_t *phrase_offsets,
                         size_t n_terms, uint32_t *docs, float *scores, size_t cap) {
  if (!n_nodes || n_terms > TO_MAX_TERMS) return 0;
  tree_build tb;
  memset(&tb, 0, sizeof tb);
  tb.a = (gs_arena *)malloc(sizeof(gs_arena));
  tb.a->n_nodes = tb.a->n_ptrs = tb.a->n_windows = tb.a->n_heap = 0;
  tb.a->bits = (uint64_t *)malloc((n_nodes + n_terms + 2) * 64 * sizeof(uint64_t));
  tb.a->acc = (float *)malloc((n_nodes + n_terms + 2) * GS_HORIZON * sizeof(float));
  tb.seg = seg;
  tb.nodes = nodes;
  tb.n_nodes = n_nodes;
  tb.terms = terms;
  tb.weights = weights;
  tb.phrase_offsets = phrase_offsets;
  tb.n_terms = n_terms;
  tb.store = (term_scorer *)malloc(sizeof(term_scorer) * (n_terms ? n_terms : 1));
  all_ctx out = {docs, scores, cap, 0};
  size_t pos = 0;
  gscorer *s = gs_build_node(&tb, &pos);
  if (s && !tb.error && s->kind != GS_EMPTY)
    for (uint32_t d = gs_doc(s); d != TO_TERMINATED; d = gs_advance(s)) all_match(&out, d, gs_score(s));
  for (size_t i = 0; i < tb.n_owned; i++) free(tb.owned[i]);
  free(tb.store);
  free(tb.a->bits);
  free(tb.a->acc);
  free(tb.a);
  return tb.error ? (size_t)-1 : out.n;
}
