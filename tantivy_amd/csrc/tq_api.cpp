// tq_api.cpp — the C ABI of include/tantivy_amd.h: contexts, segment residency, options, statistics,
// deletes and counting, codec access, the cross-segment merge.  Compiled with hipcc.  The other entry
// points: tq_terms.cpp (tq_term_prepare), tq_search.cpp (tq_search_batch*), tq_submit.cpp (tq_submit /
// tq_wait / tq_search_one), tq_encode.hip, tq_comm.cpp.
#include "tq_internal.hpp"
#include "../host/bm25.hpp"

namespace tqi {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}
}  // namespace tqi

int tq_internal_fail(int code, const char *where, const char *what) {
  return fail(code, "%s: %s", where, what);
}
bool tq_internal_ctx_has_device(const tq_ctx *ctx, int device) {
  return std::find(ctx->devices.begin(), ctx->devices.end(), device) != ctx->devices.end();
}

extern "C" {

const char *tq_last_error(void) { return g_last_error.c_str(); }

int tq_init(const int *device_ids, int n_devices, tq_ctx **out) {
  if (!out) return fail(TQ_ERR_INVALID, "tq_init: out is null");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(TQ_ERR_NO_DEVICE, "tq_init: no HIP device visible (%s)",
                e == hipSuccess ? "count=0" : hipGetErrorString(e));
  tq_ctx *ctx = new tq_ctx();
  if (!device_ids || n_devices <= 0) {
    ctx->devices.push_back(0);
  } else {
    for (int i = 0; i < n_devices; ++i) {
      if (device_ids[i] < 0 || device_ids[i] >= count) {
        delete ctx;
        return fail(TQ_ERR_INVALID, "tq_init: device %d out of range (%d visible)", device_ids[i],
                    count);
      }
      ctx->devices.push_back(device_ids[i]);
    }
  }
  *out = ctx;
  return TQ_OK;
}

void tq_shutdown(tq_ctx *ctx) { delete ctx; }

static int segment_upload_common(tq_ctx *ctx, int device, uint32_t max_doc, const uint8_t *idx,
                                 size_t idx_len, const uint8_t *pos, size_t pos_len,
                                 const uint8_t *fieldnorm, size_t fn_len, uint8_t record_option,
                                 bool from_device, tq_segment **out) {
  if (!ctx || !out || !idx) return fail(TQ_ERR_INVALID, "tq_segment_upload: null argument");
  if (idx_len < 8) return fail(TQ_ERR_FORMAT, "idx sub-file shorter than its 8-byte header");
  if (record_option > TQ_WITH_FREQS_AND_POSITIONS)
    return fail(TQ_ERR_INVALID, "bad record_option %u", record_option);
  if (fieldnorm && fn_len < max_doc)
    return fail(TQ_ERR_FORMAT, "fieldnorm file has %zu bytes for max_doc %u", fn_len, max_doc);
  if (std::find(ctx->devices.begin(), ctx->devices.end(), device) == ctx->devices.end())
    return fail(TQ_ERR_INVALID, "device %d not part of this context", device);
  HIP_TRY(hipSetDevice(device));
  tq_segment *s = new tq_segment();
  s->submit = tq_new_submit_queue();  // (exists for the segment's whole life: tq_submit takes no creation lock)
  if (!s->submit) {
    delete s;
    return fail(TQ_ERR_INVALID, "tq_segment_upload: out of memory");
  }
  s->ctx = ctx;
  s->device = device;
  s->dscratch = ctx->scratch_for(device);
  s->max_doc = max_doc;
  s->record_option = record_option;
  s->idx_len = idx_len;
  s->pos_len = (pos && pos_len) ? pos_len : 0;
  if (!from_device) {
    s->h_idx.assign(idx, idx + idx_len);
    if (pos && pos_len) s->h_pos.assign(pos, pos + pos_len);
  }
  const hipMemcpyKind kind = from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  auto up = [&](uint8_t **dst, const uint8_t *src, size_t n) -> int {
    HIP_TRY(hipMalloc((void **)dst, n + PAD));
    HIP_TRY(hipMemset(*dst + n, 0, PAD));
    HIP_TRY(hipMemcpy(*dst, src, n, kind));
    return TQ_OK;
  };
  int rc = up(&s->d_idx, idx, idx_len);
  if (rc == TQ_OK && pos && pos_len) rc = up(&s->d_pos, pos, pos_len);
  if (rc == TQ_OK && fieldnorm) rc = up(&s->d_fn, fieldnorm, max_doc);
  if (rc == TQ_OK) {
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_stage_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_batch_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->side_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
      e = hipEventCreateWithFlags(&s->ev_copy_done[i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_buf_free[i], hipEventDisableTiming);
    }
    for (int i = 0; i < tq_segment::kTimingRing; ++i) {
      if (e == hipSuccess) e = hipEventCreate(&s->ev_t0[i]);
      if (e == hipSuccess) e = hipEventCreate(&s->ev_t1[i]);
      if (e == hipSuccess) e = hipEventCreate(&s->ev_k0[i]);
      if (e == hipSuccess) e = hipEventCreate(&s->ev_k1[i]);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_match_counter, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_tp_info, 2 * sizeof(TqpInfo));
    if (e != hipSuccess) rc = fail(TQ_ERR_HIP, "segment setup: %s", hipGetErrorString(e));
  }
  if (rc != TQ_OK) {
    tq_segment_free(s);
    return rc;
  }
  s->dseg.idx = s->d_idx;
  s->dseg.pos = s->d_pos;
  s->dseg.fieldnorm = s->d_fn;
  s->dseg.max_doc = max_doc;
  s->dseg.const_fieldnorm_id = 1;  // FieldNormReader::constant(max_doc, 1)
  s->dseg.min_fieldnorm_id = 1;
  if (fieldnorm && !from_device) {
    uint8_t mn = 255;
    for (uint32_t d = 0; d < max_doc; ++d) mn = fieldnorm[d] < mn ? fieldnorm[d] : mn;
    s->dseg.min_fieldnorm_id = max_doc ? mn : 0;
  } else if (fieldnorm) {  // smallest fieldnorm id present: a device reduction, 4 bytes back
    uint32_t *slot = (uint32_t *)s->d_tp_info;
    uint32_t mn = 255;
    hipError_t e = hipMemcpy(slot, &mn, 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = tqp_launch_min_fieldnorm(s->d_fn, max_doc, slot, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&mn, slot, 4, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) {
      tq_segment_free(s);
      return fail(TQ_ERR_HIP, "fieldnorm scan: %s", hipGetErrorString(e));
    }
    s->dseg.min_fieldnorm_id = max_doc ? mn : 0;
  }
  // Bm25Weight.cache under the segment's own average fieldnorm (bm25.rs:62-69 with avg = total_num_tokens /
  // max_doc, the 8-byte header of the .idx sub-file): the range maxima of lists with bitmaps are built under it
  {
    uint64_t total_tokens = 0;
    hipError_t e = hipSuccess;
    if (!from_device)
      memcpy(&total_tokens, idx, 8);
    else
      e = hipMemcpy(&total_tokens, s->d_idx, 8, hipMemcpyDeviceToHost);
    const float avg = max_doc ? (float)total_tokens / (float)max_doc : 0.0f;
    if (e == hipSuccess && total_tokens && std::isfinite(avg) && avg > 0.0f) {
      const tantivy_amd::Bm25Weight w = tantivy_amd::Bm25Weight::from_idf(1.0f, avg);
      e = hipMalloc((void **)&s->d_local_cache, 256 * sizeof(float));
      if (e == hipSuccess) e = hipMemcpy(s->d_local_cache, w.cache, 256 * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
      tq_segment_free(s);
      return fail(TQ_ERR_HIP, "segment cache: %s", hipGetErrorString(e));
    }
  }
  *out = s;
  return TQ_OK;
}

int tq_segment_upload(tq_ctx *ctx, int device, uint32_t max_doc, const uint8_t *idx,
                      size_t idx_len, const uint8_t *pos, size_t pos_len, const uint8_t *fieldnorm,
                      size_t fn_len, uint8_t record_option, tq_segment **out) {
  return segment_upload_common(ctx, device, max_doc, idx, idx_len, pos, pos_len, fieldnorm, fn_len,
                               record_option, false, out);
}

int tq_segment_upload_device(tq_ctx *ctx, int device, uint32_t max_doc, const uint8_t *d_idx,
                             size_t idx_len, const uint8_t *d_pos, size_t pos_len,
                             const uint8_t *d_fieldnorm, size_t fn_len, uint8_t record_option,
                             tq_segment **out) {
  return segment_upload_common(ctx, device, max_doc, d_idx, idx_len, d_pos, pos_len, d_fieldnorm,
                               fn_len, record_option, true, out);
}

void tq_segment_free(tq_segment *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->batch_in_flight && s->ev_batch_done) (void)hipEventSynchronize(s->ev_batch_done);
  for (void *slab : s->term_slabs) (void)hipFree(slab);  // (the terms' table blobs: tq_terms.cpp term_alloc)
  // (bitmaps, byte-wide tfs, position directories, plain lists: the arena and its overflow)
  dense_arena_free(s);
  for (void *ptr : s->dense_extra) (void)hipFree(ptr);
  if (s->d_terms) (void)hipFree(s->d_terms);
  if (s->d_idx) (void)hipFree(s->d_idx);
  if (s->d_pos) (void)hipFree(s->d_pos);
  if (s->d_fn) (void)hipFree(s->d_fn);
  if (s->d_alive) (void)hipFree(s->d_alive);
  if (s->d_docmat) (void)hipFree(s->d_docmat);
  if (s->d_doccls) (void)hipFree(s->d_doccls);
  if (s->d_tp_info) (void)hipFree(s->d_tp_info);
  if (s->d_local_cache) (void)hipFree(s->d_local_cache);
  if (s->d_match_counter) (void)hipFree(s->d_match_counter);
  s->d_stage.release();
  s->d_out_scores.release();
  s->d_out_docs.release();
  s->d_out_counts.release();
  s->d_misc.release();
  for (int i = 0; i < 2; ++i) {
    s->h_prep_stage[i].release();
    if (s->ev_prep[i]) (void)hipEventDestroy(s->ev_prep[i]);
  }
  if (s->ev_prep_order) (void)hipEventDestroy(s->ev_prep_order);
  s->d_thr.release();
  tq_free_plan_scratch(s->plan);
  s->plan = nullptr;
  tq_free_submit_queue(s->submit);
  s->submit = nullptr;
  s->d_qmatches.release();
  s->d_share_words.release();
  s->d_ashare_words.release();
  s->d_bshare_words.release();
  s->d_count_queries.release();
  s->d_count_out.release();
  s->d_count_bits.release();
  s->d_count_wgs.release();
  s->h_stage.release();
  s->h_out.release();
  if (s->side_stream) (void)hipStreamSynchronize(s->side_stream);
  if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
  for (hipEvent_t ev : {s->ev_stage_done, s->ev_fork, s->ev_join, s->ev_batch_done, s->ev_copy_done[0],
                        s->ev_copy_done[1], s->ev_buf_free[0], s->ev_buf_free[1]})
    if (ev) (void)hipEventDestroy(ev);
  if (s->side_stream) (void)hipStreamDestroy(s->side_stream);
  if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
  s->d_stage_alt.release();
  for (int i = 0; i < tq_segment::kTimingRing; ++i)
    for (hipEvent_t ev : {s->ev_t0[i], s->ev_t1[i], s->ev_k0[i], s->ev_k1[i]})
      if (ev) (void)hipEventDestroy(ev);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

}  // extern "C"

extern "C" {

int tq_segment_set_alive_bitset(tq_segment *s, const uint8_t *bytes, size_t len) {
  if (!s) return fail(TQ_ERR_INVALID, "tq_segment_set_alive_bitset: null segment");
  TQ_SEGMENT_LOCK(s);
  HIP_TRY(hipSetDevice(s->device));
  {
    const int wrc = wait_segment_idle(s);  // batches may run on a caller's stream
    if (wrc != TQ_OK) return wrc;
  }
  if (s->d_alive) (void)hipFree(s->d_alive);
  s->d_alive = nullptr;
  s->bytes_alive = 0;
  s->dseg.alive = nullptr;
  if (!bytes) return TQ_OK;  // no deletes
  // BitSet::serialize (common/src/bitset.rs:215-223): u32 LE max_value, then 64-bit tiny sets
  if (len < 4) return fail(TQ_ERR_FORMAT, "alive bitset shorter than its header");
  const uint32_t max_value = rd32(bytes);
  const size_t need = ((size_t)s->max_doc + 63) / 64 * 8;
  if (max_value != s->max_doc || len - 4 < need || (len - 4) % 8 != 0)
    return fail(TQ_ERR_FORMAT, "alive bitset for %u docs / %zu bytes does not fit max_doc %u",
                max_value, len - 4, s->max_doc);
  HIP_TRY(hipMalloc((void **)&s->d_alive, len - 4 + PAD));
  HIP_TRY(hipMemset(s->d_alive + (len - 4), 0, PAD));
  HIP_TRY(hipMemcpy(s->d_alive, bytes + 4, len - 4, hipMemcpyHostToDevice));
  s->bytes_alive = len - 4;
  s->dseg.alive = s->d_alive;
  return TQ_OK;
}

int tq_last_batch_match_counts(tq_segment *s, uint32_t *out, uint32_t n) {
  if (!s || !out) return fail(TQ_ERR_INVALID, "tq_last_batch_match_counts: null argument");
  TQ_SEGMENT_LOCK(s);
  if (n > s->last_batch_queries) return fail(TQ_ERR_INVALID, "the last batch had %u queries", s->last_batch_queries);
  HIP_TRY(hipSetDevice(s->device));
  {
    const int wrc = wait_segment_idle(s);
    if (wrc != TQ_OK) return wrc;
  }
  HIP_TRY(hipMemcpy(out, s->d_qmatches.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return TQ_OK;
}

int tq_last_batch_query_kernels(tq_segment *s, uint32_t *out, uint32_t n) {
  if (!s || !out) return fail(TQ_ERR_INVALID, "tq_last_batch_query_kernels: null argument");
  TQ_SEGMENT_LOCK(s);
  if (!s->opt.record_query_kernels || s->last_query_kernel.size() < n)
    return fail(TQ_ERR_INVALID, "the last batch recorded %zu queries (option \"record_query_kernels\")", s->last_query_kernel.size());
  memcpy(out, s->last_query_kernel.data(), (size_t)n * sizeof(uint32_t));
  return TQ_OK;
}

int tq_count_batch(tq_segment *s, const tq_query *queries, uint32_t n_queries,
                   uint32_t *out_counts) {
  if (!s || (!queries && n_queries) || !out_counts)
    return fail(TQ_ERR_INVALID, "tq_count_batch: null argument");
  TQ_SEGMENT_LOCK(s);
  if (n_queries == 0) return TQ_OK;
  return count_batch(s, queries, n_queries, out_counts);
}

int tq_last_batch_stats(tq_segment *s, tq_batch_stats *out) {
  if (!s || !out) return fail(TQ_ERR_INVALID, "tq_last_batch_stats: null argument");
  TQ_SEGMENT_LOCK(s);
  HIP_TRY(hipSetDevice(s->device));
  if (s->stats_pending) {
    {
      const int wrc = wait_segment_idle(s);  // batches may run on a caller's stream
      if (wrc != TQ_OK) return wrc;
    }
    unsigned long long m = 0;
    HIP_TRY(hipMemcpy(&m, s->d_match_counter, sizeof m, hipMemcpyDeviceToHost));
    s->stats.matches = m;
    s->stats.algorithmic_bytes += m;  // 1 fieldnorm byte per scored doc
    if (s->opt.timing) {
      uint64_t first = s->batches_reported;
      if (s->batches_timed - first > (uint64_t)tq_segment::kTimingRing)
        first = s->batches_timed - tq_segment::kTimingRing;
      double k_sum = 0, t_sum = 0;
      uint32_t n = 0;
      for (uint64_t b = first; b < s->batches_timed; ++b) {
        const int i = (int)(b % tq_segment::kTimingRing);
        float km = 0, tm = 0;
        if (hipEventElapsedTime(&km, s->ev_k0[i], s->ev_k1[i]) == hipSuccess &&
            hipEventElapsedTime(&tm, s->ev_t0[i], s->ev_t1[i]) == hipSuccess) {
          k_sum += km;
          t_sum += tm;
          ++n;
        }
      }
      s->batches_reported = s->batches_timed;
      if (n) {
        s->stats.kernel_ms = (float)(k_sum / n);
        s->stats.total_ms = (float)(t_sum / n);
      }
      s->stats.batches_averaged = n;
    }
    s->stats.host_plan_ms = s->host_ms_n ? (float)(s->host_ms_sum / s->host_ms_n) : 0.0f;
    s->host_ms_sum = 0;
    s->host_ms_n = 0;
    s->stats_pending = false;
  }
  *out = s->stats;
  return TQ_OK;
}

int tq_segment_reserve_columns(tq_segment *s, const uint64_t *postings_offs, uint32_t n) {
  if (!s || (!postings_offs && n)) return fail(TQ_ERR_INVALID, "tq_segment_reserve_columns: null argument");
  TQ_SEGMENT_LOCK(s);
  if (!s->terms.empty())
    return fail(TQ_ERR_INVALID, "tq_segment_reserve_columns: call it before the first tq_term_prepare");
  s->reserved_cols.clear();
  for (uint32_t i = 0; i < n && i < TQD_MAT_SLOTS; ++i) s->reserved_cols[postings_offs[i]] = true;
  s->cols_reserved = true;
  return TQ_OK;
}

int tq_segment_get_stats(tq_segment *s, tq_segment_stats *out) {
  if (!s || !out) return fail(TQ_ERR_INVALID, "tq_segment_get_stats: null argument");
  TQ_SEGMENT_LOCK(s);
  tq_segment_stats r{};
  r.index_bytes = s->idx_len;
  r.positions_bytes = s->pos_len;
  r.fieldnorm_bytes = s->d_fn ? s->max_doc : 0;
  r.alive_bytes = s->bytes_alive;
  r.term_table_bytes = s->bytes_term_tables + s->d_terms_cap * sizeof(TqdTerm);
  r.bitmap_bytes = s->bytes_bitmaps;
  r.docmat_bytes = s->bytes_docmat;
  r.posdir_bytes = s->bytes_posdir;
  r.scratch_bytes = s->d_stage.cap + s->d_stage_alt.cap + s->d_out_scores.cap + s->d_out_docs.cap +
                    s->d_out_counts.cap + s->d_misc.cap + s->d_thr.cap + s->d_qmatches.cap + s->d_share_words.cap +
                    s->d_ashare_words.cap + s->d_bshare_words.cap + s->d_count_queries.cap + s->d_count_out.cap +
                    s->d_count_bits.cap + s->d_count_wgs.cap;
  {
    std::lock_guard<std::mutex> lk(s->dscratch->m);
    r.device_scratch_bytes = s->dscratch->partials.cap + s->dscratch->share_stage.cap + s->dscratch->ashare_stage.cap +
                             s->dscratch->bshare_stage.cap;
  }
  r.n_terms = (uint32_t)s->terms.size();
  r.n_dense_lists = s->n_dense_lists;
  r.n_docmat_columns = s->n_mat_slots;
  r.probe_evictions = (uint32_t)std::min<uint64_t>(s->probe_evictions, 0xFFFFFFFFull);
  r.dense_budget_bytes = s->dense_budget();
  *out = r;
  return TQ_OK;
}

int tq_set_option(tq_segment *s, const char *name, int64_t value) {
  if (!s || !name) return fail(TQ_ERR_INVALID, "tq_set_option: null argument");
  TQ_SEGMENT_LOCK(s);
  if (!strcmp(name, "exhaustive"))  // (the two words tq_submit reads without the segment lock: atomic stores)
    __atomic_store_n(&s->opt.exhaustive, value != 0 ? 1 : 0, __ATOMIC_RELAXED);
  else if (!strcmp(name, "timing"))
    s->opt.timing = value != 0;
  else if (!strcmp(name, "use_dpp"))
    s->opt.use_dpp = value != 0;
  else if (!strcmp(name, "use_dense"))
    s->opt.use_dense = value != 0;
  else if (!strcmp(name, "or_windows"))
    s->opt.or_windows = value < 0 ? -1 : (value != 0);
  else if (!strcmp(name, "dense_ratio") && value >= 1)  // affects terms prepared afterwards
    s->opt.dense_ratio = (int)value;
  else if (!strcmp(name, "rdir_budget_x") && value >= 0 && value <= 0x7FFFFFFF)
    s->opt.rdir_budget_x = (int)value;
  else if (!strcmp(name, "probe_budget_x") && value >= 0 && value <= 0x7FFFFFFF)
    s->opt.probe_budget_x = (int)value, s->probe_full = false;
  else if (!strcmp(name, "dense_budget_x") && value >= 0)
    s->opt.dense_budget_x = (int)value;
  else if (!strcmp(name, "bound_slack_ppm") && value >= 0 && value <= 1000000000)
    __atomic_store_n(&s->opt.bound_slack_ppm, (int)value, __ATOMIC_RELAXED);
  else if (!strcmp(name, "dense"))  // affects terms prepared afterwards
    s->opt.dense = value != 0;
  else if (!strcmp(name, "docmat"))  // affects terms prepared afterwards
    s->opt.docmat = value != 0;
  else if (!strcmp(name, "docsig"))  // affects terms prepared afterwards
    s->opt.docsig = value != 0;
  else if (!strcmp(name, "device_prepare"))  // affects terms prepared afterwards
    s->opt.device_prepare = value != 0;
  else if (!strcmp(name, "xunion_ratio") && value >= 0 && value <= 0x7FFFFFFF)
    s->opt.xunion_ratio = (int)value;
  else if (!strcmp(name, "count_bitmap_ratio") && value >= 0 && value <= 0x7FFFFFFF)
    s->opt.count_bitmap_ratio = (int)value;
  else if (!strcmp(name, "ashare_min_batch") && value >= 0 && value <= 0x7FFFFFFF)
    s->opt.ashare_min_batch = (int)value;
  else if (!strcmp(name, "xunion_min_queries") && value >= 1 && value <= 0x7FFFFFFF)
    s->opt.xunion_min_queries = (int)value;
  else if (!strcmp(name, "record_query_kernels"))
    s->opt.record_query_kernels = value != 0;
  else if (!strcmp(name, "debug") && value >= -1 && value <= 0x7FFFFFFF)  // (diagnosis: the kernels' TQ_DEBUG word)
    s->opt.debug = (int)value;
  else if (!strcmp(name, "submit_window_us") && value >= 0 && value <= 1000000)
    s->opt.submit_window_us = (int)value;
  else
    return fail(TQ_ERR_INVALID, "unknown option '%s'", name);
  return TQ_OK;
}

int tq_decode_postings(tq_segment *s, tq_term_handle term, uint32_t *docs, uint32_t *tfs) {
  if (!s || !docs || !tfs) return fail(TQ_ERR_INVALID, "tq_decode_postings: null argument");
  TQ_SEGMENT_LOCK(s);
  if (term >= s->terms.size()) return fail(TQ_ERR_INVALID, "unknown term handle %u", term);
  HIP_TRY(hipSetDevice(s->device));
  int rc = sync_terms(s, s->stream);
  if (rc != TQ_OK) return rc;
  const TermHost &t = s->terms[term];
  const size_t bytes = (size_t)t.doc_freq * sizeof(uint32_t);
  rc = s->d_misc.ensure(2 * bytes + 64);
  if (rc != TQ_OK) return rc;
  uint32_t *dd = (uint32_t *)s->d_misc.p, *dt = dd + t.doc_freq;
  hipError_t e = tqk_launch_decode_list(s->dseg, s->d_terms, term, t.n_blocks, dd, dt,
                                        s->opt.use_dpp != 0, s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "decode launch: %s", hipGetErrorString(e));
  HIP_TRY(hipMemcpyAsync(docs, dd, bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(tfs, dt, bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TQ_OK;
}

int tq_decode_position_deltas(tq_segment *s, tq_term_handle term, uint32_t *out, uint64_t cap,
                              uint64_t *n_out) {
  if (!s || !n_out) return fail(TQ_ERR_INVALID, "tq_decode_position_deltas: null argument");
  TQ_SEGMENT_LOCK(s);
  if (term >= s->terms.size()) return fail(TQ_ERR_INVALID, "unknown term handle %u", term);
  const TermHost &t = s->terms[term];
  if (t.positions_len == 0) return fail(TQ_ERR_UNSUPPORTED, "term has no positions on the device");
  *n_out = t.n_positions;
  const uint64_t n = std::min<uint64_t>(cap, t.n_positions);
  if (n == 0 || !out) return TQ_OK;
  HIP_TRY(hipSetDevice(s->device));
  int rc = sync_terms(s, s->stream);
  if (rc != TQ_OK) return rc;
  rc = s->d_misc.ensure(n * sizeof(uint32_t));
  if (rc != TQ_OK) return rc;
  hipError_t e = tqk_launch_decode_positions(s->dseg, s->d_terms, term, (uint32_t *)s->d_misc.p, n,
                                             s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "decode launch: %s", hipGetErrorString(e));
  HIP_TRY(hipMemcpyAsync(out, s->d_misc.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TQ_OK;
}

}  // extern "C"

extern "C" {
// ---- cross-segment merge (merge_top_k)
int tq_merge_topk(const float *scores, const uint32_t *docs, const uint32_t *counts,
                  uint32_t n_segments, uint32_t n_queries, uint32_t stride, uint32_t offset,
                  uint32_t limit, float *out_scores, uint32_t *out_segment_ords,
                  uint32_t *out_docs, uint32_t *out_counts) {
  if (!scores || !docs || !counts || !out_scores || !out_segment_ords || !out_docs || !out_counts)
    return fail(TQ_ERR_INVALID, "tq_merge_topk: null argument");
  struct H {
    float s;
    uint32_t o, d;
  };
  std::vector<H> all;
  for (uint32_t q = 0; q < n_queries; ++q) {
    all.clear();
    for (uint32_t sg = 0; sg < n_segments; ++sg) {
      const uint32_t c = std::min(counts[(size_t)sg * n_queries + q], stride);
      const size_t b = ((size_t)sg * n_queries + q) * stride;
      for (uint32_t i = 0; i < c; ++i) all.push_back({scores[b + i], sg, docs[b + i]});
    }
    // top_score_collector.rs:590-600: sort key desc, then DocAddress asc.  Scores are compared
    // through the order-preserving u32 transform the device keys use (a strict weak order even
    // for NaN inputs, which partial_cmp().unwrap_or(Equal) in the reference is not)
    auto sortable = [](float x) {
      uint32_t u;
      memcpy(&u, &x, 4);
      if ((u & 0x7FFFFFFFu) == 0u) u = 0u;  // -0.0 == +0.0
      return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
    };
    std::sort(all.begin(), all.end(), [&](const H &a, const H &b) {
      const uint32_t ka = sortable(a.s), kb = sortable(b.s);
      if (ka != kb) return ka > kb;
      if (a.o != b.o) return a.o < b.o;
      return a.d < b.d;
    });
    uint32_t w = 0;
    for (size_t i = offset; i < all.size() && w < limit; ++i, ++w) {
      out_scores[(size_t)q * limit + w] = all[i].s;
      out_segment_ords[(size_t)q * limit + w] = all[i].o;
      out_docs[(size_t)q * limit + w] = all[i].d;
    }
    out_counts[q] = w;
    for (; w < limit; ++w) {
      out_scores[(size_t)q * limit + w] = 0.0f;
      out_segment_ords[(size_t)q * limit + w] = 0xFFFFFFFFu;
      out_docs[(size_t)q * limit + w] = TQ_TERMINATED;
    }
  }
  return TQ_OK;
}

int tq_merge_topk_device(tq_ctx *ctx, int device, const float *d_scores, const uint32_t *d_docs,
                         const uint32_t *d_counts, const uint32_t *segment_ords,
                         uint32_t n_segments, uint32_t n_queries, uint32_t stride,
                         uint32_t offset, uint32_t limit, float *d_out_scores,
                         uint32_t *d_out_segment_ords, uint32_t *d_out_docs,
                         uint32_t *d_out_counts, void *hip_stream) {
  if (!ctx || !d_scores || !d_docs || !d_counts || !d_out_scores || !d_out_segment_ords ||
      !d_out_docs || !d_out_counts)
    return fail(TQ_ERR_INVALID, "tq_merge_topk_device: null argument");
  if (limit == 0) return fail(TQ_ERR_INVALID, "tq_merge_topk_device: limit 0");
  HIP_TRY(hipSetDevice(device));
  TqkSegMergeParams p{};
  p.scores = d_scores;
  p.docs = d_docs;
  p.counts = d_counts;
  p.segment_ords = segment_ords;
  p.out_scores = d_out_scores;
  p.out_segment_ords = d_out_segment_ords;
  p.out_docs = d_out_docs;
  p.out_counts = d_out_counts;
  p.n_segments = n_segments;
  p.n_queries = n_queries;
  p.stride = stride;
  p.offset = offset;
  p.limit = limit;
  hipError_t e = tqk_launch_merge_segments(p, (hipStream_t)hip_stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "merge launch: %s", hipGetErrorString(e));
  return TQ_OK;
}

int tq_copy_to_host_async(tq_ctx *ctx, int device, void *dst_pinned_host, const void *src_device, size_t bytes,
                          void *hip_stream) {
  if (!ctx || !dst_pinned_host || !src_device) return fail(TQ_ERR_INVALID, "tq_copy_to_host_async: null argument");
  if (!bytes) return TQ_OK;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpyAsync(dst_pinned_host, src_device, bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
  return TQ_OK;
}

}  // extern "C"
