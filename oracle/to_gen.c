/* to_gen.c — CPU ORACLE (test infrastructure): synthetic Zipf index generator,
 * query-stream sampler and the multi-threaded CPU baseline driver.  See tantivy_oracle.h. */
#include "tantivy_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ rng */
static uint64_t splitmix64(uint64_t *x) {
  uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
void to_rng_seed(to_rng *r, uint64_t seed) {
  uint64_t x = seed;
  for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&x);
}
static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t to_rng_next(to_rng *r) { /* xoshiro256** */
  uint64_t *s = r->s;
  uint64_t result = rotl(s[1] * 5, 7) * 9;
  uint64_t t = s[1] << 17;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl(s[3], 45);
  return result;
}
double to_rng_uniform(to_rng *r) { return (double)(to_rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

uint32_t to_zipf_rank(to_rng *r, uint32_t n) {
  static __thread double *cdf = NULL;
  static __thread uint32_t cdf_n = 0;
  if (cdf_n != n) {
    free(cdf);
    cdf = (double *)malloc(sizeof(double) * n);
    double h = 0.0;
    for (uint32_t i = 0; i < n; i++) {
      h += 1.0 / (double)(i + 1);
      cdf[i] = h;
    }
    for (uint32_t i = 0; i < n; i++) cdf[i] /= h;
    cdf_n = n;
  }
  double u = to_rng_uniform(r);
  uint32_t lo = 0, hi = n - 1;
  while (lo < hi) {
    uint32_t mid = (lo + hi) / 2;
    if (cdf[mid] > u)
      hi = mid;
    else
      lo = mid + 1;
  }
  return lo + 1;
}

/* ------------------------------------------------------------------ generator
 * Mirrors the shape of benches/intersection_bench.rs:20-93 at SURVEY §8d's parameters:
 * df_r = floor(0.5 N / r); per-doc Bernoulli via geometric gaps; tf = 1 + Geometric(cont 0.3)
 * capped at 10; fieldnorm = sum tf + U[5,30) filler; optional positions with planted phrases. */
static uint64_t mix64(uint64_t a, uint64_t b) {
  uint64_t x = a * 0x9E3779B97F4A7C15ull + b;
  return splitmix64(&x);
}
static int u32_cmp(const void *a, const void *b) {
  uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return (x > y) - (x < y);
}

to_synth_segment *to_synth_build(uint32_t max_doc, uint32_t n_terms, uint32_t segment_ord,
                                 int with_positions, uint32_t phrase_terms) {
  to_synth_segment *s = (to_synth_segment *)calloc(1, sizeof *s);
  s->max_doc = max_doc;
  s->n_terms = n_terms;
  s->record_option = with_positions ? TO_WITH_FREQS_AND_POSITIONS : TO_WITH_FREQS;
  to_buf_init(&s->idx);
  to_buf_init(&s->pos);
  to_buf_init(&s->fieldnorm);
  s->terms = (to_term_info *)calloc(n_terms, sizeof(to_term_info));

  uint32_t **docs = (uint32_t **)calloc(n_terms, sizeof(uint32_t *));
  uint8_t **tfs = (uint8_t **)calloc(n_terms, sizeof(uint8_t *));
  uint32_t *dfs = (uint32_t *)calloc(n_terms, sizeof(uint32_t));
  uint32_t *tokens = (uint32_t *)calloc(max_doc ? max_doc : 1, sizeof(uint32_t));

  /* pass 1: postings */
  for (uint32_t t = 0; t < n_terms; t++) {
    uint32_t r = t + 1;
    uint32_t target_df = (uint32_t)(((uint64_t)max_doc / 2u) / r);
    double p = max_doc ? (double)target_df / (double)max_doc : 0.0;
    to_rng rng;
    to_rng_seed(&rng, 0x7A6E7469ull + 1000ull * segment_ord + r);
    size_t cap = (size_t)target_df + (size_t)target_df / 8 + 64;
    docs[t] = (uint32_t *)malloc(cap * sizeof(uint32_t));
    tfs[t] = (uint8_t *)malloc(cap);
    size_t n = 0;
    if (p > 0.0) {
      double log1mp = log(1.0 - p);
      int64_t d = -1;
      for (;;) {
        double u = to_rng_uniform(&rng);
        if (u <= 0.0) u = 1e-300;
        int64_t gap = 1 + (int64_t)floor(log(u) / log1mp);
        d += gap;
        if (d >= (int64_t)max_doc) break;
        if (n == cap) {
          cap = cap * 2;
          docs[t] = (uint32_t *)realloc(docs[t], cap * sizeof(uint32_t));
          tfs[t] = (uint8_t *)realloc(tfs[t], cap);
        }
        uint32_t tf = 1;
        while (tf < 10 && to_rng_uniform(&rng) < 0.3) tf++;
        docs[t][n] = (uint32_t)d;
        tfs[t][n] = (uint8_t)tf;
        tokens[d] += tf;
        n++;
      }
    }
    dfs[t] = (uint32_t)n;
  }
  /* fieldnorms */
  uint8_t *fn_ids = (uint8_t *)malloc(max_doc ? max_doc : 1);
  uint64_t total_tokens = 0;
  {
    to_rng rng;
    to_rng_seed(&rng, 0xF1E1D00Dull + segment_ord);
    for (uint32_t d = 0; d < max_doc; d++) {
      uint32_t filler = 5u + (uint32_t)(to_rng_next(&rng) % 25u);
      tokens[d] += filler;
      total_tokens += tokens[d];
      fn_ids[d] = to_fieldnorm_to_id(tokens[d]);
    }
  }
  to_buf_push(&s->fieldnorm, fn_ids, max_doc);
  s->total_num_tokens = total_tokens;

  /* pass 2: serialize.  FieldSerializer::create (serializer.rs:119-152): u64 LE
   * total_num_tokens header, avg fieldnorm = total_num_tokens / num_docs. */
  {
    uint8_t hdr[8];
    for (int i = 0; i < 8; i++) hdr[i] = (uint8_t)(total_tokens >> (8 * i));
    to_buf_push(&s->idx, hdr, 8);
  }
  float avg = max_doc ? (float)total_tokens / (float)max_doc : 0.0f;
  to_postings_serializer *ps = to_postings_serializer_new(avg, s->record_option, fn_ids, max_doc);
  to_position_serializer *pp = with_positions ? to_position_serializer_new(&s->pos) : NULL;
  uint32_t posbuf[64];
  for (uint32_t t = 0; t < n_terms; t++) {
    uint32_t r = t + 1;
    to_term_info *ti = &s->terms[t];
    ti->doc_freq = dfs[t];
    ti->postings_start = s->idx.len - 8;
    ti->positions_start = s->pos.len;
    to_postings_serializer_new_term(ps, dfs[t], 1);
    for (uint32_t i = 0; i < dfs[t]; i++) {
      uint32_t d = docs[t][i], tf = tfs[t][i];
      if (pp) {
        /* tf distinct sorted positions in [0, tokens[d]); planted phrase: in 1/20 of the docs the
         * first position of rank r <= phrase_terms is base(doc) + (r - 1). */
        uint32_t len = tokens[d];
        to_rng prng;
        to_rng_seed(&prng, mix64(((uint64_t)segment_ord << 40) ^ ((uint64_t)r << 32), d));
        for (uint32_t k = 0; k < tf; k++) posbuf[k] = (uint32_t)(to_rng_next(&prng) % len);
        uint64_t hd = mix64(0xABCDEFull + segment_ord, d);
        if (r <= phrase_terms && (hd % 20u) == 0u) {
          uint32_t span = len > phrase_terms ? len - phrase_terms : 1u;
          posbuf[0] = (uint32_t)((hd >> 8) % span) + (r - 1u);
        }
        qsort(posbuf, tf, sizeof(uint32_t), u32_cmp);
        for (uint32_t k = 1; k < tf; k++)
          if (posbuf[k] <= posbuf[k - 1]) posbuf[k] = posbuf[k - 1] + 1u;
        /* deltas within the doc: postings serializer.rs:204-217 */
        uint32_t prev = 0;
        for (uint32_t k = 0; k < tf; k++) {
          uint32_t pabs = posbuf[k];
          posbuf[k] = pabs - prev;
          prev = pabs;
        }
        to_position_serializer_write_positions_delta(pp, posbuf, tf);
      }
      to_postings_serializer_write_doc(ps, d, tf);
    }
    to_postings_serializer_close_term(ps, dfs[t], &s->idx);
    if (pp) to_position_serializer_close_term(pp);
    ti->postings_end = s->idx.len - 8;
    ti->positions_end = s->pos.len;
    free(docs[t]);
    free(tfs[t]);
  }
  to_postings_serializer_free(ps);
  to_position_serializer_free(pp);
  free(docs);
  free(tfs);
  free(dfs);
  free(tokens);
  free(fn_ids);
  return s;
}
void to_synth_free(to_synth_segment *s) {
  if (!s) return;
  to_buf_free(&s->idx);
  to_buf_free(&s->pos);
  to_buf_free(&s->fieldnorm);
  free(s->terms);
  free(s);
}
void to_synth_view(const to_synth_segment *s, to_segment_view *v) {
  v->max_doc = s->max_doc;
  v->record_option = s->record_option;
  v->idx = s->idx.data;
  v->idx_len = s->idx.len;
  v->pos = s->pos.len ? s->pos.data : NULL;
  v->pos_len = s->pos.len;
  v->fieldnorm = s->fieldnorm.data;
  v->total_num_tokens = s->total_num_tokens;
}

/* ------------------------------------------------------------------ baseline driver */
typedef struct {
  const to_segment_view *seg;
  const to_query *queries;
  size_t n;
  size_t next;
  pthread_mutex_t mu;
  double *lat;
  to_hit *hits;
  uint32_t *counts;
} bl_shared;
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void *bl_worker(void *arg) {
  bl_shared *sh = (bl_shared *)arg;
  to_hit local[1024];
  for (;;) {
    pthread_mutex_lock(&sh->mu);
    size_t i = sh->next++;
    pthread_mutex_unlock(&sh->mu);
    if (i >= sh->n) break;
    const to_query *q = &sh->queries[i];
    double t0 = now_s();
    to_hit *dst = local;
    to_hit *heapbuf = NULL;
    if (q->k > 1024) dst = heapbuf = (to_hit *)malloc(sizeof(to_hit) * q->k);
    size_t n = to_search_pruned(sh->seg, q, dst);
    double t1 = now_s();
    if (sh->lat) sh->lat[i] = t1 - t0;
    if (sh->counts) sh->counts[i] = (uint32_t)n;
    if (sh->hits) {
      to_sort_hits(dst, n);
      memcpy(sh->hits + i * q->k, dst, n * sizeof(to_hit));
    }
    free(heapbuf);
  }
  return NULL;
}
double to_baseline_run(const to_segment_view *seg, const to_query *queries, size_t n_queries,
                       int n_threads, double *latency_s, to_hit *out_hits, uint32_t *out_counts) {
  bl_shared sh;
  sh.seg = seg;
  sh.queries = queries;
  sh.n = n_queries;
  sh.next = 0;
  sh.lat = latency_s;
  sh.hits = out_hits;
  sh.counts = out_counts;
  pthread_mutex_init(&sh.mu, NULL);
  if (n_threads < 1) n_threads = 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
  double t0 = now_s();
  for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, bl_worker, &sh);
  for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
  double t1 = now_s();
  free(th);
  pthread_mutex_destroy(&sh.mu);
  return t1 - t0;
}
