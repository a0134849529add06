// tq_terms.cpp — tq_term_prepare: the skip list of a posting list unrolled into random-access tables (host walk or
// device walk), the dense lists' side tables (bitmap + rank directory, byte-wide tfs, position directory, plain
// lists) in the segment's arena, the doc matrix and its signature bits.
//
// Host-side format walkers restate (file:line under the tantivy checkout):
//   skip entries        src/postings/skip.rs:205-253,275-302
//   list framing        src/postings/block_segment_postings.rs:78-88,107-116
//   vint tail           src/postings/compression/vint.rs:44-108
//   positions framing   src/positions/reader.rs:43-56,84-101
// Part of the C ABI library of include/tantivy_amd.h (internal declarations: tq_internal.hpp).
#include "tq_internal.hpp"

namespace tqi {
int term_prepare_device(tq_segment *s, uint64_t postings_off, uint32_t postings_len,
                        uint64_t positions_off, uint32_t positions_len, uint32_t doc_freq,
                        tq_term_handle *out);
int build_dense_device(tq_segment *s, uint32_t handle);
int register_term(tq_segment *s, const TqdTerm &dt, const TermHost &th, uint64_t postings_off,
                  tq_term_handle *out);
int add_to_doc_signatures(tq_segment *s, uint32_t handle);
int ensure_docmat(tq_segment *s);

// Dense lists also get their term freqs as one byte per posting (255 = "255 or more: read the
// packed value"): with the posting index from the bitmap's rank the tf of a candidate is ONE load,
// where block record -> packed tf bits are two dependent ones (the shared-union kernel's scoring
// stage is a chain of dependent gathers, 1.6 us each under load).  d_tfs = the decoded tfs.

// A term's table blob (block records, coarse table, tail, position-block table): carved out of 4 MB slabs — one
// hipMalloc per prepared term was most of the 40 us a sparse term cost to prepare, i.e. most of a batch that names
// two thousand new terms (the 65 536-term stream of bench.py).  Slabs live as long as the segment; a blob handed
// back by a failed preparation is simply not reused.  Large blobs get an allocation of their own.
int term_alloc(tq_segment *s, size_t bytes, uint8_t **out) {
  constexpr size_t kSlab = (size_t)4 << 20;
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (need > kSlab / 4) {
    HIP_TRY(hipMalloc((void **)out, bytes));
    s->term_slabs.push_back(*out);
    return TQ_OK;
  }
  if (need > s->term_slab_left) {
    void *slab = nullptr;
    HIP_TRY(hipMalloc(&slab, kSlab));
    s->term_slabs.push_back(slab);
    s->term_slab_cur = (uint8_t *)slab;
    s->term_slab_left = kSlab;
  }
  *out = s->term_slab_cur;
  s->term_slab_cur += need;
  s->term_slab_left -= need;
  return TQ_OK;
}

// A failed preparation hands its blob back (a caller retrying a term that keeps failing — a corrupt list —
// allocated again on every attempt until tq_segment_free).  The slab's LAST allocation is rolled back (failures are
// reported before anything else is allocated, so that is the common case); a blob with an allocation of its own is
// freed; anything else stays in its slab.  The device may still be writing to it: the caller has synchronised.
void term_release(tq_segment *s, uint8_t *blob, size_t bytes) {
  if (!blob) return;
  constexpr size_t kSlab = (size_t)4 << 20;
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (s->bytes_term_tables >= bytes) s->bytes_term_tables -= bytes;
  if (need > kSlab / 4) {
    for (size_t i = s->term_slabs.size(); i-- > 0;)
      if (s->term_slabs[i] == blob) {
        s->term_slabs.erase(s->term_slabs.begin() + (ptrdiff_t)i);
        (void)hipFree(blob);
        break;
      }
    return;
  }
  if (s->term_slab_cur == blob + need) {
    s->term_slab_cur = blob;
    s->term_slab_left += need;
  }
}

// ---- the arena as a reserved address range (tq_internal.hpp: dense_arena_vmm)
constexpr size_t kVmmChunk = (size_t)32 << 20;
static bool vmm_map_more(tq_segment *s, size_t bytes) {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = s->device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) {
    (void)hipGetLastError();
    gran = (size_t)2 << 20;
  }
  // (the driver reported 4 KB and then refused hipMemSetAccess on a chunk that started 47 MB into the range: chunks are
  // multiples of 2 MB — the device's large page — whatever it reports)
  // (and hipMemSetAccess refused 32 MB chunks at 2-, 4- and 16-MB-aligned addresses while it took one at a 64-MB-aligned
  // address: every chunk is a multiple of kVmmChunk at a multiple of kVmmChunk from a kVmmChunk-aligned base, and access
  // is set over the whole mapped prefix)
  gran = std::max<size_t>(gran, kVmmChunk);
  bytes = (bytes + gran - 1) / gran * gran;
  if (s->dense_arena_mapped + bytes > s->dense_arena_cap) return false;
  static const bool trace = getenv("TQ_TRACE") != nullptr;
  auto why = [&](const char *what, hipError_t e) {
    if (trace) fprintf(stderr, "[tq] arena: %s of %zu MB at +%zu MB failed: %s\n", what, bytes >> 20, s->dense_arena_mapped >> 20, hipGetErrorString(e));
    (void)hipGetLastError();
  };
  hipMemGenericAllocationHandle_t h{};
  hipError_t e = hipMemCreate(&h, bytes, &prop, 0);
  if (e != hipSuccess) {
    why("hipMemCreate", e);
    return false;
  }
  uint8_t *at = s->dense_arena + s->dense_arena_mapped;
  e = hipMemMap(at, bytes, 0, h, 0);
  if (e != hipSuccess) {
    why("hipMemMap", e);
    (void)hipMemRelease(h);
    return false;
  }
  hipMemAccessDesc ad{};
  ad.location = prop.location;
  ad.flags = hipMemAccessFlagsProtReadWrite;
  // (the whole mapped prefix, not the new chunk alone: the driver refused the third chunk's own range — "invalid argument")
  e = hipMemSetAccess(s->dense_arena, s->dense_arena_mapped + bytes, &ad, 1);
  if (e != hipSuccess) e = hipMemSetAccess(at, bytes, &ad, 1);
  if (e != hipSuccess) {
    why("hipMemSetAccess", e);
    (void)hipMemUnmap(at, bytes);
    (void)hipMemRelease(h);
    return false;
  }
  s->dense_arena_chunks.emplace_back(h, bytes);
  s->dense_arena_mapped += bytes;
  return true;
}
static bool vmm_reserve(tq_segment *s, size_t first_bytes) {
  static const bool kVmm = tune_u32("TQ_VMM_ARENA", 1) != 0;  // 0: one hipMalloc of the budgets' size + overflow allocations (rounds 3-5)
  if (!kVmm) return false;
  constexpr size_t kReserve = (size_t)24 << 30;  // (the table offsets reach 32 GB)
  void *va = nullptr;
  // (the alignment argument is not honoured — 2 MB-aligned addresses came back for a 1 GB request: the range is a
  // chunk longer and the arena starts at its first multiple of kVmmChunk)
  const hipError_t re = hipMemAddressReserve(&va, kReserve + kVmmChunk, 0, nullptr, 0);
  if (re != hipSuccess || !va) {
    if (getenv("TQ_TRACE")) fprintf(stderr, "[tq] arena: hipMemAddressReserve of %zu GB: %s, address %p\n", kReserve >> 30, hipGetErrorString(re), va);
    (void)hipGetLastError();
    return false;
  }
  s->dense_arena_va = va;
  va = (void *)(((uintptr_t)va + kVmmChunk - 1) & ~(uintptr_t)(kVmmChunk - 1));
  s->dense_arena = (uint8_t *)va;
  s->dense_arena_cap = kReserve;
  s->dense_arena_mapped = 0;
  s->dense_arena_first = first_bytes;
  s->dense_arena_vmm = true;
  if (!vmm_map_more(s, first_bytes)) {  // (nothing mapped: hand the range back, the old arena takes over)
    if (getenv("TQ_TRACE")) fprintf(stderr, "[tq] arena: first chunk of %zu MB not mapped\n", first_bytes >> 20);
    (void)hipMemAddressFree(s->dense_arena_va, kReserve + kVmmChunk);
    s->dense_arena_va = nullptr;
    s->dense_arena = nullptr;
    s->dense_arena_cap = 0;
    s->dense_arena_vmm = false;
    return false;
  }
  return true;
}
void dense_arena_free(tq_segment *s) {
  if (!s->dense_arena) return;
  if (!s->dense_arena_vmm) {
    (void)hipFree(s->dense_arena);
  } else {
    size_t off = 0;
    for (auto &ch : s->dense_arena_chunks) {
      (void)hipMemUnmap(s->dense_arena + off, ch.second);
      (void)hipMemRelease(ch.first);
      off += ch.second;
    }
    s->dense_arena_chunks.clear();
    (void)hipMemAddressFree(s->dense_arena_va, s->dense_arena_cap + kVmmChunk);
    s->dense_arena_va = nullptr;
  }
  s->dense_arena = nullptr;
  s->dense_arena_cap = s->dense_arena_used = s->dense_arena_mapped = 0;
  s->dense_arena_vmm = false;
}

// a side table of a dense list: from the segment's arena, else a device allocation of its own
int dense_alloc(tq_segment *s, size_t bytes, void **out) {
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (!s->dense_arena && s->dense_arena_cap == 0) {
    const size_t cap = std::max<size_t>(s->dense_budget() + s->probe_budget() + s->rdir_budget(), (size_t)1 << 20) + PAD;
    void *base = nullptr;
    if (vmm_reserve(s, cap)) {
      // (the first chunk is what the old arena was: the budgets' worth in one piece)
    } else if (hipMalloc(&base, cap) == hipSuccess) {
      s->dense_arena = (uint8_t *)base;
      s->dense_arena_cap = cap;
    } else {
      (void)hipGetLastError();
      s->dense_arena_cap = 1;  // (tried once: every table gets its own allocation)
    }
  }
  if (s->dense_arena && s->dense_arena_vmm) {
    // growth beyond the budgets (a probe pool grown by a batch that names more lists than it holds, tables of a
    // segment whose options were raised): in multiples of 32 MB
    const size_t want = s->dense_arena_used + need + PAD;
    bool ok = want <= s->dense_arena_cap;
    // (a quarter of what is mapped at a time: a stream discovering a large vocabulary adds hundreds of MB of range
    // directories, and a mapping is three driver calls)
    while (ok && want > s->dense_arena_mapped)
      ok = vmm_map_more(s, std::max<size_t>({want - s->dense_arena_mapped, s->dense_arena_mapped / 4, kVmmChunk}));
    if (ok) {
      *out = s->dense_arena + s->dense_arena_used;
      s->dense_arena_used += need;
      return TQ_OK;
    }
  } else if (s->dense_arena && s->dense_arena_used + need + PAD <= s->dense_arena_cap) {
    *out = s->dense_arena + s->dense_arena_used;
    s->dense_arena_used += need;
    return TQ_OK;
  }
  void *ptr = nullptr;
  HIP_TRY(hipMalloc(&ptr, bytes));
  s->dense_extra.push_back(ptr);
  *out = ptr;
  return TQ_OK;
}
// (tables are only ever released with the segment; a failed build leaves its bytes unused)
void dense_release(tq_segment *s, void *ptr) {
  for (size_t i = 0; i < s->dense_extra.size(); ++i)
    if (s->dense_extra[i] == ptr) {
      (void)hipFree(ptr);
      s->dense_extra.erase(s->dense_extra.begin() + (long)i);
      return;
    }
}

// Range directories (TermHost::rdir_blob; rdir_lookup in tq_common.hpp): carved out of 32 MB chunks of the segment's
// arena, kept until the segment is closed.  rdir_plan: the list's shift S — ranges of 2^S docs, the smallest S in 2..16 that leaves at most
// df ranges (one to two postings per range; entries keep 16 bits of the doc id) — or 0: no directory (the option is
// off, the segment or the list is tiny, or the directories have reached "rdir_budget_x").  rdir_bytes: entries (one
// u32 per posting) behind (max_doc >> S) + 2 directory slots (padded to four).
size_t rdir_dir_words(const tq_segment *s, uint32_t S) { return (((size_t)(s->max_doc >> S) + 2u) + 3u) & ~(size_t)3u; }
size_t rdir_bytes(const tq_segment *s, uint32_t doc_freq, uint32_t S) {
  return (rdir_dir_words(s, S) + (size_t)doc_freq) * sizeof(uint32_t);
}
uint32_t rdir_plan(tq_segment *s, uint32_t doc_freq) {
  static const uint32_t kRatio = tune_u32("TQ_RDIR_RATIO", 0);  // (experiments: only lists below max_doc / ratio)
  static const uint32_t kMinDf = std::max<uint32_t>(1u, tune_u32("TQ_RDIR_MIN_DF", 256));  // (below: the directory would outweigh the list)
  if (!s->opt.dense || s->opt.rdir_budget_x <= 0 || s->max_doc < 4096u || doc_freq < kMinDf) return 0u;
  if (kRatio && (uint64_t)doc_freq * kRatio >= s->max_doc) return 0u;
  static const uint32_t kPerRange = std::max<uint32_t>(1u, tune_u32("TQ_RDIR_PER_RANGE", 1));  // (postings per range, at least: 1 / 2 / 4 / 8 — and2_distinct 1.32 / 1.36 / 1.42 / 1.54 ms)
  uint32_t S = 2;
  while (S < 16u && (s->max_doc >> S) + 1u > doc_freq / kPerRange) ++S;
  if (s->rdir_bytes_total + rdir_bytes(s, doc_freq, S) > s->rdir_budget()) return 0u;
  return S;
}
int rdir_alloc(tq_segment *s, size_t bytes, void **out) {
  constexpr size_t kChunk = (size_t)32 << 20;
  const size_t need = (bytes + 255) & ~(size_t)255;
  // (chunks come out of the segment's arena of side tables — dense_alloc — so that they lie within the 32 GB the
  // shared launches' table offsets span; a process with several indexes open hands out distant addresses otherwise)
  if (need > kChunk / 2) {
    void *own = nullptr;
    const int arc = dense_alloc(s, need, &own);
    if (arc != TQ_OK) return arc;
    s->rdir_chunks.push_back({own, need});
    *out = own;
  } else {
    if (need > s->rdir_chunk_left) {
      void *chunk = nullptr;
      const int arc = dense_alloc(s, kChunk, &chunk);
      if (arc != TQ_OK) return arc;
      s->rdir_chunks.push_back({chunk, kChunk});
      s->rdir_chunk_cur = (uint8_t *)chunk;
      s->rdir_chunk_left = kChunk;
    }
    *out = s->rdir_chunk_cur;
    s->rdir_chunk_cur += need;
    s->rdir_chunk_left -= need;
  }
  s->rdir_bytes_total += bytes;
  s->bytes_term_tables += bytes;
  s->share_span_terms = ~(size_t)0;  // (the tables' address span is taken again: tq_search.cpp)
  return TQ_OK;
}
// rmax_list (the list's largest tf/(tf + norm), what the shared intersection launch bounds a probed list with) of the
// lists whose directories an earlier tq_term_prepare_batch call built, once that call's launch has finished
void prep_apply_lmax(tq_segment *s, bool wait) {
  for (int bx = 0; bx < 2; ++bx) {
    std::vector<uint32_t> &hs = s->prep_lmax_handles[bx];
    if (hs.empty() || !s->ev_prep[bx]) continue;
    if (wait) {
      if (hipEventSynchronize(s->ev_prep[bx]) != hipSuccess) continue;
    } else if (hipEventQuery(s->ev_prep[bx]) != hipSuccess) {
      (void)hipGetLastError();
      continue;
    }
    for (size_t i = 0; i < hs.size(); ++i) {
      TermHost &t = s->terms[hs[i]];
      const uint32_t lmax = s->prep_lmax_host[bx][i];
      if (t.rdir_blob && !t.rmax_blob && lmax) t.rmax_list = std::min<uint32_t>(lmax, 255u);
    }
    hs.clear();
  }
}

int build_tf8(tq_segment *s, uint32_t handle, const uint32_t *d_tfs) {
  TermHost &t = s->terms[handle];
  const size_t bytes = ((size_t)t.doc_freq + 7) & ~(size_t)7;
  if (s->dense_bytes_total + bytes > s->dense_budget()) return TQ_OK;
  void *blob = nullptr;
  {
    const int arc = dense_alloc(s, bytes + PAD, &blob);
    if (arc != TQ_OK) return arc;
  }
  hipError_t e = tqk_launch_tf8_pack(d_tfs, t.doc_freq, (uint8_t *)blob, s->stream);
  if (e != hipSuccess) {
    dense_release(s, blob);
    return fail(TQ_ERR_HIP, "tf8 pack: %s", hipGetErrorString(e));
  }
  t.tf8_blob = blob;
  s->h_dterms[handle].tf8 = (const uint8_t *)blob;
  s->dense_bytes_total += bytes;
  s->bytes_bitmaps += bytes;
  mark_term_dirty(s, handle);
  return TQ_OK;
}

// Range maxima of a list that gets a bitmap (its own or the probe tables'): tq_ashare.hip bounds the non-leader
// lists of an intersection per TQD_RM_SHIFT-doc range instead of by their weight (block_wand_intersection.rs:59-85
// uses the block-max of the secondaries' current blocks).  dd / dt = the decoded list; acc = n_ranges + 1 u32 of
// scratch.  13 KB per list of a 10M-doc segment (five levels, one byte per 1024 docs at the finest).
int build_rmax(tq_segment *s, uint32_t handle, const uint32_t *dd, const uint32_t *dt, uint32_t *acc) {
  TermHost &t = s->terms[handle];
  if (t.rmax_blob || !s->d_local_cache || !t.doc_freq) return TQ_OK;
  const uint32_t n_ranges = (s->max_doc >> TQD_RM_SHIFT) + 1u;
  const uint32_t n_out = tqd_rm_level_off(s->max_doc, TQD_RM_LEVELS);  // every level (tq_device.h)
  void *blob = nullptr;
  const int arc = dense_alloc(s, n_out, &blob);
  if (arc != TQ_OK) return arc;
  hipError_t e = hipMemsetAsync(acc, 0, ((size_t)n_ranges + 1u) * sizeof(uint32_t), s->stream);
  if (e == hipSuccess)
    e = tqp_launch_rmax(dd, dt, t.doc_freq, s->d_fn, s->dseg.const_fieldnorm_id, s->d_local_cache, acc, s->max_doc,
                        (uint8_t *)blob, acc + n_ranges, s->stream);
  uint32_t lmax = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&lmax, acc + n_ranges, 4, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess) {
    dense_release(s, blob);
    return fail(TQ_ERR_HIP, "range maxima: %s", hipGetErrorString(e));
  }
  t.rmax_blob = blob;
  t.rmax_list = lmax ? std::min<uint32_t>(lmax, 255u) : 255u;
  s->bytes_bitmaps += n_out;
  return TQ_OK;
}

// A list WITHOUT a bitmap as plain arrays — doc ids, then min(tf, 255) per posting — for the
// doc-major union launch (tq_xunion.hip), which scatters such a list into its tile row with a
// cursor instead of decoding blocks.  Built on the batch's stream the first time an unpruned
// union needs the list; counts against the side tables' budget (false = over it: *ok stays false).
int build_flat(tq_segment *s, uint32_t handle, hipStream_t st, bool *ok) {
  TermHost &t = s->terms[handle];
  *ok = t.flat_blob != nullptr;
  if (*ok || t.doc_freq == 0) return TQ_OK;
  const size_t doc_bytes = ((size_t)t.doc_freq * sizeof(uint32_t) + 15) & ~(size_t)15;
  const size_t bytes = doc_bytes + (((size_t)t.doc_freq + 15) & ~(size_t)15);
  if (s->dense_bytes_total + bytes > s->dense_budget()) return TQ_OK;
  void *blob = nullptr;
  {
    const int arc = dense_alloc(s, bytes + PAD, &blob);
    if (arc != TQ_OK) return arc;
  }
  const hipError_t e = tqk_launch_flat_list(s->dseg, t.d_self, 0u, t.n_blocks, (uint32_t *)blob,
                                            (uint8_t *)blob + doc_bytes, st);
  if (e != hipSuccess) {
    dense_release(s, blob);
    return fail(TQ_ERR_HIP, "flat list: %s", hipGetErrorString(e));
  }
  t.flat_blob = blob;
  s->dense_bytes_total += bytes;
  s->bytes_bitmaps += bytes;
  *ok = true;
  return TQ_OK;
}

// Orders work about to be enqueued on `st` after the segment's previous batch, whatever stream
// that batch ran on (no-op when it is the same stream: stream order already holds).
int order_after_last_batch(tq_segment *s, hipStream_t st) {
  if (s->batch_in_flight && s->last_stream != st)
    HIP_TRY(hipStreamWaitEvent(st, s->ev_batch_done, 0));
  return TQ_OK;
}
// Host-side wait for everything the segment has in flight (its own stream and the last batch).
int wait_segment_idle(tq_segment *s) {
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (s->batch_in_flight) {
    HIP_TRY(hipEventSynchronize(s->ev_batch_done));
    s->batch_in_flight = false;
  }
  return TQ_OK;
}

// The doc matrix (TqdSegment::docmat): allocated with the first list that needs it.
int ensure_docmat(tq_segment *s) {
  if (s->d_docmat) return TQ_OK;
  const size_t mat_bytes = (size_t)s->max_doc * sizeof(uint64_t);
  if (s->dense_bytes_total + mat_bytes > s->dense_budget()) return TQ_OK;  // (stays null: over budget)
  HIP_TRY(hipMalloc((void **)&s->d_docmat, mat_bytes + PAD));
  HIP_TRY(hipMemsetAsync((uint8_t *)s->d_docmat + mat_bytes, 0, PAD, s->stream));
  const hipError_t e = tqk_launch_docmat_init(s->d_docmat, s->d_fn, s->dseg.const_fieldnorm_id, s->max_doc, s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "docmat init: %s", hipGetErrorString(e));
  s->dense_bytes_total += mat_bytes;
  s->bytes_docmat = mat_bytes;
  s->dseg.docmat = s->d_docmat;
  return TQ_OK;
}

// Lists WITHOUT a column in the doc matrix (the sparse, high-weight lists; dense lists beyond the
// 40 columns) share the top 16 bits of the doc-matrix words: every such list sets bit
// 48 + hash(handle) of the docs it holds.  A clear bit proves "not in the list"; a set bit means
// "maybe" (another list with the same bit, or this one).  The union kernels test it where they
// used to assume the list holds every candidate — a rare list holds a fraction of a percent of
// them, and each wrong guess cost a seek and a block search.  The same gather that brings a
// candidate's fieldnorm id and column bits brings its signature.  Only prepared (queried) lists
// set bits; built by one decode of the list.
int add_to_doc_signatures(tq_segment *s, uint32_t handle) {
  TermHost &t = s->terms[handle];
  if (t.doc_freq == 0) return TQ_OK;
  const bool has_col = ((s->h_dterms[handle].has_freq >> 8) & 0xFFu) != 0u;
  bool want_sig = s->opt.docsig && s->opt.docmat && s->opt.dense && s->max_doc >= 4096u && !has_col;
  // (a list without tables of its own also gets its range directory from the same decode)
  const uint32_t rd_shift = !(t.dense_blob && t.tf8_blob) && !t.rdir_blob ? rdir_plan(s, t.doc_freq) : 0u;
  if (want_sig) {
    const int rc = ensure_docmat(s);
    if (rc != TQ_OK) return rc;
    want_sig = s->d_docmat != nullptr;
  }
  if (!want_sig && !rd_shift) return TQ_OK;
  const size_t bytes = (size_t)t.doc_freq * sizeof(uint32_t);
  int rc = s->d_misc.ensure(2 * bytes + 128);
  if (rc != TQ_OK) return rc;
  uint32_t *dd = (uint32_t *)s->d_misc.p, *dt = dd + t.doc_freq;
  hipError_t e = tqk_launch_decode_list(s->dseg, t.d_self, 0u, t.n_blocks, dd, dt,
                                        s->opt.use_dpp != 0, s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "decode launch: %s", hipGetErrorString(e));
  if (want_sig) {
    const uint32_t bit = (handle * 0x9E3779B1u) >> (32 - 4);  // 0 .. TQD_SIG_BITS - 1
    e = tqk_launch_docmat_set(s->d_docmat, dd, t.doc_freq, (TQD_SIG_SHIFT - 8u) + bit, s->max_doc, s->stream);
    if (e != hipSuccess) return fail(TQ_ERR_HIP, "docmat signature: %s", hipGetErrorString(e));
    s->h_dterms[handle].has_freq |= (bit + 1u) << 16;
    mark_term_dirty(s, handle);
  }
  if (rd_shift) {
    void *tab = nullptr;
    rc = rdir_alloc(s, rdir_bytes(s, t.doc_freq, rd_shift), &tab);
    if (rc != TQ_OK) return rc;
    e = tqk_launch_rdir_fill(dd, dt, t.doc_freq, s->max_doc, (uint32_t *)tab, rd_shift, s->stream);
    if (e != hipSuccess) return fail(TQ_ERR_HIP, "range directory: %s", hipGetErrorString(e));
    t.rdir_blob = tab;
    t.rdir_ent = (uint32_t *)tab + rdir_dir_words(s, rd_shift);
    t.rdir_shift = rd_shift;
    if (s->d_local_cache) {  // the list's largest tf/(tf + norm): td_rmax_scatter_kernel's value, one range
      uint32_t *acc = dt + t.doc_freq + 8u, lmax = 0;
      e = hipMemsetAsync(acc, 0, sizeof(uint32_t), s->stream);
      if (e == hipSuccess) e = tqp_launch_list_max(dd, dt, t.doc_freq, s->d_fn, s->dseg.const_fieldnorm_id, s->d_local_cache, acc, s->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(&lmax, acc, 4, hipMemcpyDeviceToHost, s->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
      if (e != hipSuccess) return fail(TQ_ERR_HIP, "list maximum: %s", hipGetErrorString(e));
      if (lmax) t.rmax_list = std::min<uint32_t>(lmax, 255u);
    }
  }
  return TQ_OK;
}

}  // namespace tqi

namespace tqi {
// What the host walk of one posting list leaves: the bytes of the term's blob (records, coarse table, tails, position
// tables, room for its own record) and everything of its records but the device pointers.
struct WalkedTerm {
  std::vector<uint8_t> hb;
  size_t total = 0, o_rec = 0, o_coarse = 0, o_tdocs = 0, o_ttfs = 0, o_pboff = 0, o_ptail = 0, o_self = 0;
  TqdTerm dt{};
  TermHost th;
  uint64_t postings_off = 0;
};
static int host_walk_term(tq_segment *s, uint64_t postings_off, uint32_t postings_len, uint64_t positions_off,
                          uint32_t positions_len, uint32_t doc_freq, WalkedTerm &w) {
  const uint8_t *data = s->h_idx.data() + 8 + postings_off;
  const size_t len = postings_len;
  const uint64_t abs0 = 8 + postings_off;  // offset of `data` inside the uploaded sub-file

  int record = s->record_option;
  const uint32_t n_full = doc_freq / 128u, n_tail = doc_freq % 128u;
  size_t at = 0;
  const uint8_t *skip = nullptr;
  size_t skip_len = 0;
  if (doc_freq >= 128u) {  // block_segment_postings.rs:78-88
    uint64_t sl;
    if (!read_vint(data, len, at, sl) || sl > len - at)
      return fail(TQ_ERR_FORMAT, "bad skip_len for term at %llu", (unsigned long long)postings_off);
    skip = data + at;
    skip_len = (size_t)sl;
    at += skip_len;
    if (skip_len < 8ull * n_full) record = TQ_BASIC;  // :107-116 (JSON terms without freqs)
  }
  const size_t entry = record == TQ_BASIC ? 5 : (record == TQ_WITH_FREQS ? 8 : 12);
  if (skip_len < entry * n_full)
    return fail(TQ_ERR_FORMAT, "skip data too short: %zu < %zu", skip_len, entry * n_full);
  const bool has_freq = record != TQ_BASIC;
  const size_t payload = at;

  const uint32_t n_blocks = n_full + (n_tail ? 1u : 0u);
  // (scratch kept per thread: eight allocations per term were a third of a sparse term's walk)
  static thread_local std::vector<uint32_t> b_last, b_meta, b_off, block_pos, tail_docs, tail_tfs, coarse;
  b_last.assign(n_blocks, 0u);
  b_meta.assign(n_blocks, 0u);
  b_off.assign(n_blocks, 0u);
  block_pos.assign(n_blocks + 1, 0u);
  size_t running = 0;
  uint64_t running_pos = 0;
  uint32_t last_doc = 0;
  for (uint32_t i = 0; i < n_full; ++i) {  // skip.rs:205-253,275-302
    const uint8_t *e = skip + entry * i;
    const uint32_t ld = rd32(e);
    const uint32_t doc_bits = e[4] & 0x1Fu, strict = (e[4] >> 6) & 1u;
    uint32_t tf_bits = 0, tf_sum = 0, bm_fn = 0, bm_tf = 0;
    if (record == TQ_WITH_FREQS) {
      tf_bits = e[5];
      bm_fn = e[6];
      bm_tf = e[7];
    } else if (record == TQ_WITH_FREQS_AND_POSITIONS) {
      tf_bits = e[5];
      tf_sum = rd32(e + 6);
      bm_fn = e[10];
      bm_tf = e[11];
    }
    if (tf_bits > 32u) return fail(TQ_ERR_FORMAT, "tf bit width %u > 32", tf_bits);
    if (i && ld <= last_doc) return fail(TQ_ERR_FORMAT, "skip last_doc not increasing");
    if (running_pos > 0xFFFFFFFFull)
      return fail(TQ_ERR_UNSUPPORTED, "term with more than 2^32 positions");
    b_last[i] = ld;
    b_meta[i] = doc_bits | (strict << 6) | (tf_bits << 8) | (bm_fn << 16) | (bm_tf << 24);
    b_off[i] = (uint32_t)running;  // < postings_len, a u32 (term_info.rs:10-17)
    block_pos[i] = (uint32_t)running_pos;
    running += 16u * (size_t)(doc_bits + tf_bits);
    running_pos += tf_sum;
    last_doc = ld;
  }
  if (payload + running > len) return fail(TQ_ERR_FORMAT, "bitpacked payload exceeds the list");
  tail_docs.assign(n_tail, 0u);
  tail_tfs.assign(n_tail, 1u);
  if (n_tail) {  // vint.rs:44-108; docs delta from the last full block (0 if none)
    size_t t = payload + running;
    uint32_t prev = n_full ? last_doc : 0u;
    for (uint32_t i = 0; i < n_tail; ++i) {
      uint32_t d;
      if (!read_vint32_block(data, len, t, d)) return fail(TQ_ERR_FORMAT, "truncated vint docs");
      prev += d;
      tail_docs[i] = prev;
    }
    if (has_freq && t < len) {
      for (uint32_t i = 0; i < n_tail; ++i)
        if (!read_vint32_block(data, len, t, tail_tfs[i]))
          return fail(TQ_ERR_FORMAT, "truncated vint term freqs");
    }
    if (running_pos > 0xFFFFFFFFull)
      return fail(TQ_ERR_UNSUPPORTED, "term with more than 2^32 positions");
    b_last[n_full] = tail_docs[n_tail - 1];
    b_meta[n_full] = 0xFFFFFFFFu;
    b_off[n_full] = 0;
    block_pos[n_full] = (uint32_t)running_pos;
    if (record == TQ_WITH_FREQS_AND_POSITIONS)  // tf sums only index a positions stream
      for (uint32_t i = 0; i < n_tail; ++i) running_pos += tail_tfs[i];
    last_doc = tail_docs[n_tail - 1];
  }
  if (running_pos > 0xFFFFFFFFull)
    return fail(TQ_ERR_UNSUPPORTED, "term with more than 2^32 positions");
  block_pos[n_blocks] = (uint32_t)running_pos;
  if (last_doc >= TQ_TERMINATED) return fail(TQ_ERR_FORMAT, "doc id >= TERMINATED");
  if (last_doc >= s->max_doc)
    return fail(TQ_ERR_FORMAT, "doc id %u >= max_doc %u", last_doc, s->max_doc);

  // coarse[b] = first block j with last_doc[j] >= b << shift, about one block per bucket
  uint32_t shift = 7;
  while (shift < 31 && ((uint64_t)(s->max_doc - 1) >> shift) + 1 > 2ull * n_blocks + 2) ++shift;
  const uint32_t n_buckets = (uint32_t)(((uint64_t)(s->max_doc - 1)) >> shift) + 1;
  coarse.assign(n_buckets + 1, 0u);
  {
    uint32_t j = 0;
    for (uint32_t b = 0; b <= n_buckets; ++b) {
      const uint64_t lo = (uint64_t)b << shift;
      while (j < n_blocks && (uint64_t)b_last[j] < lo) ++j;
      coarse[b] = j;
    }
  }

  // positions stream (positions/reader.rs:43-56,84-101)
  std::vector<uint64_t> pos_block_off;
  std::vector<uint8_t> pos_widths;
  std::vector<uint32_t> pos_tail;
  const bool want_pos = s->record_option == TQ_WITH_FREQS_AND_POSITIONS && !s->h_pos.empty() &&
                        record == TQ_WITH_FREQS_AND_POSITIONS;
  if (want_pos) {
    if (positions_off > s->h_pos.size() || (uint64_t)positions_len > s->h_pos.size() - positions_off)
      return fail(TQ_ERR_FORMAT, "positions_range outside the pos file");
    const uint8_t *pd = s->h_pos.data() + positions_off;
    size_t pa = 0;
    uint64_t nb;
    if (!read_vint(pd, positions_len, pa, nb) || nb > positions_len - pa)
      return fail(TQ_ERR_FORMAT, "bad positions header");
    pos_widths.assign(pd + pa, pd + pa + nb);
    pa += (size_t)nb;
    size_t prun = 0;
    pos_block_off.resize((size_t)nb);
    for (size_t i = 0; i < nb; ++i) {
      if (pos_widths[i] > 32) return fail(TQ_ERR_FORMAT, "position bit width > 32");
      pos_block_off[i] = (uint64_t)(positions_off + pa + prun) | ((uint64_t)pos_widths[i] << 56);
      prun += 16u * (size_t)pos_widths[i];
    }
    size_t t = pa + prun;
    if (t > positions_len) return fail(TQ_ERR_FORMAT, "bitpacked positions exceed the range");
    while (t < positions_len) {  // uncompress_vint_unsorted_until_end
      uint32_t v;
      if (!read_vint32_block(pd, positions_len, t, v))
        return fail(TQ_ERR_FORMAT, "truncated vint positions");
      pos_tail.push_back(v);
    }
    const uint64_t n_pos = (uint64_t)nb * 128u + pos_tail.size();
    if (n_pos != running_pos)
      return fail(TQ_ERR_FORMAT, "positions stream holds %llu values, postings say %llu",
                  (unsigned long long)n_pos, (unsigned long long)running_pos);
  }

  // one blob holding every per-term array
  auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  size_t total = 0;
  auto place = [&](size_t bytes) {
    const size_t o = total;
    total = align16(total + bytes);
    return o;
  };
  const size_t o_rec = place(16 * (size_t)(n_blocks + 1));
  const size_t o_coarse = place(4 * coarse.size());
  const size_t o_tdocs = place(4 * (size_t)n_tail);
  const size_t o_ttfs = place(4 * (size_t)n_tail);
  const size_t o_pboff = place(8 * pos_block_off.size());
  const size_t o_ptail = place(4 * pos_tail.size());
  const size_t o_self = place(sizeof(TqdTerm));  // the term's own record: what the table-building kernels of its
  total += PAD;                                  // preparation read (the segment's term table is synced per batch)
  std::vector<uint8_t> &hb = w.hb;
  hb.assign(total, 0);
  for (uint32_t i = 0; i <= n_blocks; ++i) {
    const uint32_t r[4] = {i < n_blocks ? b_last[i] : TQ_TERMINATED, i < n_blocks ? b_meta[i] : 0u,
                           i < n_blocks ? b_off[i] : 0u, block_pos[i]};
    memcpy(hb.data() + o_rec + 16 * (size_t)i, r, 16);
  }
  memcpy(hb.data() + o_coarse, coarse.data(), 4 * coarse.size());
  if (n_tail) {
    memcpy(hb.data() + o_tdocs, tail_docs.data(), 4 * (size_t)n_tail);
    memcpy(hb.data() + o_ttfs, tail_tfs.data(), 4 * (size_t)n_tail);
  }
  if (!pos_block_off.empty()) memcpy(hb.data() + o_pboff, pos_block_off.data(), 8 * pos_block_off.size());
  if (!pos_tail.empty()) memcpy(hb.data() + o_ptail, pos_tail.data(), 4 * pos_tail.size());
  w.total = total;
  w.o_rec = o_rec, w.o_coarse = o_coarse, w.o_tdocs = o_tdocs, w.o_ttfs = o_ttfs, w.o_pboff = o_pboff, w.o_ptail = o_ptail,
  w.o_self = o_self;
  w.postings_off = postings_off;
  TqdTerm &dt = w.dt;
  dt = TqdTerm{};
  dt.payload_base = abs0 + payload;
  dt.n_full = n_full;
  dt.n_tail = n_tail;
  dt.n_blocks = n_blocks;
  dt.doc_freq = doc_freq;
  dt.n_pos_blocks = (uint32_t)pos_block_off.size();
  dt.n_pos_tail = (uint32_t)pos_tail.size();
  dt.has_freq = has_freq ? 1u : 0u;
  dt.coarse_shift = shift;
  TermHost &th = w.th;
  th = TermHost{};
  th.doc_freq = doc_freq;
  th.n_blocks = n_blocks;
  th.n_full = n_full;
  th.n_tail = n_tail;
  th.last_doc = last_doc;
  th.postings_len = postings_len;
  th.positions_len = want_pos ? positions_len : 0;
  th.n_positions = want_pos ? running_pos : 0;
  return TQ_OK;
}
// the device pointers of a walked term whose blob lives at `blob` (its own record goes into the blob's last slot)
static void place_walked_term(WalkedTerm &w, uint8_t *blob) {
  TqdTerm &dt = w.dt;
  dt.rec = (const uint4 *)(blob + w.o_rec);
  dt.coarse = (const uint32_t *)(blob + w.o_coarse);
  dt.tail_docs = (const uint32_t *)(blob + w.o_tdocs);
  dt.tail_tfs = (const uint32_t *)(blob + w.o_ttfs);
  dt.pos_blk = (const uint64_t *)(blob + w.o_pboff);
  dt.pos_tail = (const uint32_t *)(blob + w.o_ptail);
  memcpy(w.hb.data() + w.o_self, &dt, sizeof dt);
  w.th.blob = blob;
  w.th.d_self = (const TqdTerm *)(blob + w.o_self);
}
}  // namespace tqi

extern "C" {

int tq_term_prepare(tq_segment *s, uint64_t postings_off, uint32_t postings_len,
                    uint64_t positions_off, uint32_t positions_len, uint32_t doc_freq,
                    tq_term_handle *out) {
  if (!s || !out) return fail(TQ_ERR_INVALID, "tq_term_prepare: null argument");
  TQ_SEGMENT_LOCK(s);
  auto it = s->term_by_off.find(postings_off);
  if (it != s->term_by_off.end()) {
    *out = it->second;
    return TQ_OK;
  }
  if (doc_freq == 0) return fail(TQ_ERR_INVALID, "tq_term_prepare: doc_freq 0 (term absent)");
  const size_t body_len = s->idx_len - 8;
  if (postings_off > body_len || (uint64_t)postings_len > body_len - postings_off)
    return fail(TQ_ERR_FORMAT, "postings_range [%llu,+%u) outside the idx body (%zu)",
                (unsigned long long)postings_off, postings_len, body_len);
  HIP_TRY(hipSetDevice(s->device));
  if (s->device_prepare())
    return term_prepare_device(s, postings_off, postings_len, positions_off, positions_len, doc_freq,
                               out);
  WalkedTerm w;
  {
    const int wrc = host_walk_term(s, postings_off, postings_len, positions_off, positions_len, doc_freq, w);
    if (wrc != TQ_OK) return wrc;
  }
  uint8_t *blob = nullptr;
  {
    const int arc = term_alloc(s, w.total, &blob);
    if (arc != TQ_OK) return arc;
  }
  s->bytes_term_tables += w.total;
  place_walked_term(w, blob);
  hipError_t ce = hipMemcpy(blob, w.hb.data(), w.total, hipMemcpyHostToDevice);
  if (ce != hipSuccess) {
    term_release(s, blob, w.total);
    return fail(TQ_ERR_HIP, "term upload: %s", hipGetErrorString(ce));
  }
  return register_term(s, w.dt, w.th, postings_off, out);
}

// The new terms of a batch together: every list is walked on the host as tq_term_prepare walks
// it, the blobs go up from ONE pinned staging buffer with asynchronous copies and one wait, and the signature bits of
// all the lists without a column are set by ONE launch (tq_term_prepare costs a blocking copy and two launches per
// term: two thousand new terms per 10 000-query batch while a 65 536-term vocabulary is being discovered).  Lists
// dense enough for tables of their own are prepared one by one as before (there are at most a few hundred per
// segment).  out[i] = the handle of infos[i] (known terms: the one they have).
int tq_term_prepare_batch(tq_segment *s, const tq_term_info *infos, uint32_t n, tq_term_handle *out) {
  if (!s || (!infos && n) || (!out && n)) return fail(TQ_ERR_INVALID, "tq_term_prepare_batch: null argument");
  std::vector<uint32_t> fresh;  // indices of infos that need preparing
  {
    TQ_SEGMENT_LOCK(s);
    for (uint32_t i = 0; i < n; ++i) {
      auto it = s->term_by_off.find(infos[i].postings_off);
      if (it != s->term_by_off.end()) {
        out[i] = it->second;
        continue;
      }
      out[i] = TQ_TERM_ABSENT;
      fresh.push_back(i);
    }
  }
  if (fresh.empty()) return TQ_OK;
  static const bool trace = getenv("TQ_TRACE") != nullptr;
  const auto tp0 = std::chrono::steady_clock::now();
  auto tp1 = tp0, tp2 = tp0, tp3 = tp0, tp4 = tp0;
  struct Report {
    bool on;
    const std::chrono::steady_clock::time_point &a, &b, &c, &d, &e;
    size_t n;
    ~Report() {
      if (!on) return;
      auto us = [](auto x, auto y) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(y - x).count(); };
      fprintf(stderr, "[tq] prepare_batch %zu new terms: walk %ld us, lock %ld us, place %ld us, register + tables %ld us, copy + launch %ld us\n", n,
              us(a, b), us(b, c), us(c, d), us(d, e), us(e, std::chrono::steady_clock::now()));
    }
  } report{trace, tp0, tp1, tp2, tp3, tp4, fresh.size()};
  // The host walk of the new lists runs WITHOUT the segment's lock (it reads the host copy of the index and a few
  // options): a second host thread can walk the next batch's terms while the first plans and enqueues this one.
  std::vector<WalkedTerm> prewalked;
  std::vector<char> prewalked_ok;
  if (!s->device_prepare()) {
    const size_t body = s->idx_len - 8;
    prewalked.resize(fresh.size());
    prewalked_ok.assign(fresh.size(), 0);
    auto walk_some = [&](size_t k0, size_t step) {
      for (size_t k = k0; k < fresh.size(); k += step) {
        const tq_term_info &ti = infos[fresh[k]];
        if (ti.doc_freq == 0 || ti.postings_off > body || (uint64_t)ti.postings_len > body - ti.postings_off) continue;  // (reported below)
        if (s->opt.dense && s->max_doc >= 4096u && (uint64_t)ti.doc_freq * (uint64_t)s->opt.dense_ratio >= s->max_doc) continue;
        if (host_walk_term(s, ti.postings_off, ti.postings_len, ti.positions_off, ti.positions_len, ti.doc_freq, prewalked[k]) == TQ_OK)
          prewalked_ok[k] = 1;  // (a malformed list: walked again under the lock, where its error is reported)
      }
    };
    // (a batch that discovers a large vocabulary names a couple of thousand new lists, 0.8 us of walking each: a few
    // short-lived helper threads take their share — TQ_PREP_THREADS, default 4 walkers from 512 lists on)
    static const uint32_t kWalkers = std::max<uint32_t>(1u, tune_u32("TQ_PREP_THREADS", 4));
    const size_t n_walk = std::min<size_t>(kWalkers, fresh.size() / 256u);
    std::vector<std::thread> helpers;
    for (size_t w = 1; w < n_walk; ++w) {
      try {
        helpers.emplace_back(walk_some, w, n_walk);
      } catch (...) {  // (a thread limit: the caller walks that share too)
        walk_some(w, n_walk);
      }
    }
    walk_some(0, std::max<size_t>(1, n_walk));
    for (std::thread &h : helpers) h.join();
  }
  tp1 = std::chrono::steady_clock::now();
  TQ_SEGMENT_LOCK(s);
  tp2 = tp3 = tp4 = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(s->device));
  const size_t dense_bytes = (((size_t)s->max_doc + 31) / 32 + 1) * sizeof(uint2);
  auto one_by_one = [&](uint32_t i) {
    return tq_term_prepare(s, infos[i].postings_off, infos[i].postings_len, infos[i].positions_off, infos[i].positions_len,
                           infos[i].doc_freq, &out[i]);
  };
  if (s->device_prepare()) {  // (no host copy of the index: the device walk, term by term)
    for (uint32_t i : fresh) {
      const int rc = one_by_one(i);
      if (rc != TQ_OK) return rc;
    }
    return TQ_OK;
  }
  const size_t body_len = s->idx_len - 8;
  std::vector<WalkedTerm> walked;
  std::vector<uint32_t> walked_of;
  std::unordered_set<uint64_t> walked_offs;  // (postings offsets seen in this call: a term named twice is prepared once)
  walked.reserve(fresh.size());
  walked_offs.reserve(fresh.size() * 2);
  size_t stage_bytes = 0;
  for (const uint32_t &i : fresh) {
    const tq_term_info &ti = infos[i];
    if (s->term_by_off.count(ti.postings_off)) {  // (named twice in this call, or prepared by another thread meanwhile)
      out[i] = s->term_by_off[ti.postings_off];
      continue;
    }
    if (ti.doc_freq == 0) return fail(TQ_ERR_INVALID, "tq_term_prepare_batch: doc_freq 0 (term absent)");
    if (ti.postings_off > body_len || (uint64_t)ti.postings_len > body_len - ti.postings_off)
      return fail(TQ_ERR_FORMAT, "postings_range [%llu,+%u) outside the idx body (%zu)", (unsigned long long)ti.postings_off,
                  ti.postings_len, body_len);
    const bool dense = s->opt.dense && s->max_doc >= 4096u && (uint64_t)ti.doc_freq * (uint64_t)s->opt.dense_ratio >= s->max_doc &&
                       s->dense_bytes_total + dense_bytes <= s->dense_budget();
    const bool dup = !walked_offs.insert(ti.postings_off).second;
    if (dense || dup) {
      if (!dup) {
        const int rc = one_by_one(i);
        if (rc != TQ_OK) return rc;
      }
      continue;
    }
    const size_t fk = (size_t)(&i - fresh.data());
    if (fk < prewalked_ok.size() && prewalked_ok[fk]) {
      walked.push_back(std::move(prewalked[fk]));
    } else {
      walked.emplace_back();
      const int wrc = host_walk_term(s, ti.postings_off, ti.postings_len, ti.positions_off, ti.positions_len, ti.doc_freq, walked.back());
      if (wrc != TQ_OK) return wrc;
    }
    walked_of.push_back(i);
    stage_bytes += (walked.back().total + 255) & ~(size_t)255;
  }
  if (!walked.empty()) {
    // blobs: ONE device region for the lot, the bytes through one of two pinned buffers, one asynchronous copy on the
    // segment's copy stream — and no wait: an event orders every later reader (the next batch's kernels, table
    // builders on the segment's stream) behind the copy and the signature launch.  (A copy per term was 3 us of host
    // time each; a wait for the segment's stream stalled the batches in flight for a millisecond per call.)
    hipStream_t cs = s->copy_stream ? s->copy_stream : s->stream;
    const size_t ptr_bytes = (walked.size() * (2 * sizeof(const TqdTerm *) + 4 * sizeof(uint32_t)) + 255) & ~(size_t)255;
    prep_apply_lmax(s, false);
    const uint32_t bx = s->prep_calls++ & 1u;
    if (!s->ev_prep[bx]) HIP_TRY(hipEventCreateWithFlags(&s->ev_prep[bx], hipEventDisableTiming));
    if (s->prep_used[bx]) HIP_TRY(hipEventSynchronize(s->ev_prep[bx]));  // (the copy of two calls ago: long done)
    prep_apply_lmax(s, false);  // (what that call left in the buffer, before it is written again)
    s->prep_lmax_handles[bx].clear();
    int rc = s->h_prep_stage[bx].ensure(stage_bytes + 2 * ptr_bytes);
    if (rc != TQ_OK) return rc;
    uint8_t *hs = (uint8_t *)s->h_prep_stage[bx].p;
    size_t at = 0;
    uint8_t *region = nullptr;
    rc = term_alloc(s, stage_bytes + ptr_bytes, &region);
    if (rc != TQ_OK) return rc;
    s->bytes_term_tables += stage_bytes + ptr_bytes;
    for (WalkedTerm &w : walked) {
      place_walked_term(w, region + at);
      memcpy(hs + at, w.hb.data(), w.total);
      at += (w.total + 255) & ~(size_t)255;
    }
    // handles; which of them get signature bits (their record pointers and bits ride behind the blobs)
    const bool sigs = s->opt.docsig && s->opt.docmat && s->opt.dense && s->max_doc >= 4096u;
    if (sigs) {
      rc = ensure_docmat(s);
      if (rc != TQ_OK) return rc;
    }
    tp3 = tp4 = std::chrono::steady_clock::now();
    // per list that gets either: record pointer, range directory, signature bit, shift, doc freq, (out) list maximum
    // — six arrays behind the blobs
    const TqdTerm **h_selfs = (const TqdTerm **)(hs + stage_bytes);
    std::vector<uint32_t *> tabs;
    std::vector<uint32_t> bits, shifts, dfs;
    uint32_t n_sig = 0;
    bool any_tab = false;
    for (size_t k = 0; k < walked.size(); ++k) {
      WalkedTerm &w = walked[k];
      const uint32_t handle = (uint32_t)s->terms.size();
      s->terms.push_back(w.th);
      s->terms.back().wants_col = !s->cols_reserved || s->reserved_cols.count(w.postings_off) != 0;
      s->h_dterms.push_back(w.dt);
      mark_term_dirty(s, handle);
      s->term_by_off.emplace(w.postings_off, handle);
      out[walked_of[k]] = handle;
      if (!w.th.doc_freq) continue;
      const bool sig = sigs && s->d_docmat;
      const uint32_t S = rdir_plan(s, w.th.doc_freq);
      void *tab = nullptr;
      if (S) {
        if (rdir_alloc(s, rdir_bytes(s, w.th.doc_freq, S), &tab) != TQ_OK) {  // (no memory for it: the list does without)
          (void)hipGetLastError();
          tab = nullptr;
        } else {
          s->terms[handle].rdir_blob = tab;
          s->terms[handle].rdir_ent = (uint32_t *)tab + rdir_dir_words(s, S);
          s->terms[handle].rdir_shift = S;
          any_tab = true;
        }
      }
      if (!sig && !tab) continue;
      const uint32_t bit = (handle * 0x9E3779B1u) >> (32 - 4);  // (add_to_doc_signatures' bit)
      h_selfs[n_sig++] = w.th.d_self;
      tabs.push_back((uint32_t *)tab);
      bits.push_back(sig ? bit : 0xFFFFFFFFu);
      shifts.push_back(tab ? S : 0u);
      dfs.push_back(w.th.doc_freq);
      s->prep_lmax_handles[bx].push_back(handle);
      if (sig) s->h_dterms[handle].has_freq |= (bit + 1u) << 16;
    }
    if (n_sig) {
      uint8_t *at2 = (uint8_t *)(h_selfs + n_sig);
      memcpy(at2, tabs.data(), (size_t)n_sig * sizeof(uint32_t *));
      at2 += (size_t)n_sig * sizeof(uint32_t *);
      for (const std::vector<uint32_t> *arr : {&bits, &shifts, &dfs}) {
        memcpy(at2, arr->data(), (size_t)n_sig * sizeof(uint32_t));
        at2 += (size_t)n_sig * sizeof(uint32_t);
      }
      memset(at2, 0, (size_t)n_sig * sizeof(uint32_t));  // (the lists' maxima: written by the launch)
    }
    tp4 = std::chrono::steady_clock::now();
    // (the doc matrix may just have been created on the segment's stream: the copy stream follows it)
    hipError_t e = hipSuccess;
    if (cs != s->stream) {
      if (!s->ev_prep_order) e = hipEventCreateWithFlags(&s->ev_prep_order, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventRecord(s->ev_prep_order, s->stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(cs, s->ev_prep_order, 0);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(region, hs, stage_bytes + ptr_bytes, hipMemcpyHostToDevice, cs);
    if (e == hipSuccess && n_sig) {
      const TqdTerm **d_selfs = (const TqdTerm **)(region + stage_bytes);
      uint32_t *const *d_tabs = (uint32_t *const *)(d_selfs + n_sig);
      const uint32_t *d_bits = (const uint32_t *)(d_tabs + n_sig);
      uint32_t *d_lmax = (uint32_t *)(d_bits + 3 * n_sig);
      e = tqk_launch_docsig_batch(s->dseg, d_selfs, d_bits, n_sig, s->d_docmat, any_tab ? d_tabs : nullptr, d_bits + n_sig,
                                  d_bits + 2 * n_sig, s->d_local_cache, d_lmax, s->opt.use_dpp != 0, cs);
      if (e == hipSuccess && any_tab && s->d_local_cache) {
        uint32_t *h_lmax = (uint32_t *)(hs + stage_bytes + ptr_bytes);
        e = hipMemcpyAsync(h_lmax, d_lmax, (size_t)n_sig * sizeof(uint32_t), hipMemcpyDeviceToHost, cs);
        s->prep_lmax_host[bx] = h_lmax;
      } else {
        s->prep_lmax_handles[bx].clear();
      }
    } else {
      s->prep_lmax_handles[bx].clear();
    }
    if (e == hipSuccess) e = hipEventRecord(s->ev_prep[bx], cs);
    if (e == hipSuccess && cs != s->stream) e = hipStreamWaitEvent(s->stream, s->ev_prep[bx], 0);
    if (e != hipSuccess) return fail(TQ_ERR_HIP, "term upload: %s", hipGetErrorString(e));
    s->prep_used[bx] = true;
    s->prep_pending = (int)bx;  // (the next batch on a stream of the caller's waits for the event too: tq_search.cpp)
  }
  for (uint32_t i : fresh)  // (a term named twice in one call)
    if (out[i] == TQ_TERM_ABSENT) {
      auto it = s->term_by_off.find(infos[i].postings_off);
      if (it != s->term_by_off.end()) out[i] = it->second;
    }
  return TQ_OK;
}

}  // extern "C"

namespace tqi {


const char *tqp_message(uint32_t st) {
  switch (st) {
    case TQP_BAD_SKIP_LEN: return "bad skip_len";
    case TQP_SKIP_TOO_SHORT: return "skip data too short";
    case TQP_NOT_INCREASING: return "skip last_doc not increasing";
    case TQP_BAD_TF_WIDTH: return "tf bit width > 32";
    case TQP_TOO_MANY_POSITIONS: return "term with more than 2^32 positions";
    case TQP_PAYLOAD_TOO_LONG: return "bitpacked payload exceeds the list";
    case TQP_TRUNCATED_TAIL: return "truncated vint tail";
    case TQP_DOC_OUT_OF_RANGE: return "doc id >= max_doc / TERMINATED";
    case TQP_BAD_POS_HEADER: return "bad positions header";
    case TQP_POS_COUNT_MISMATCH: return "positions stream and postings disagree on the number of positions";
    case TQP_BAD_POS_WIDTH: return "position bit width > 32";
    case TQP_POS_PAYLOAD_TOO_LONG: return "bitpacked positions exceed the range";
    default: return "unknown";
  }
}

// the part of tq_term_prepare both paths share: the handle, and the dense-list structures
int register_term(tq_segment *s, const TqdTerm &dt, const TermHost &th, uint64_t postings_off,
                  tq_term_handle *out) {
  const uint32_t handle = (uint32_t)s->terms.size();
  s->terms.push_back(th);
  s->terms.back().wants_col = !s->cols_reserved || s->reserved_cols.count(postings_off) != 0;
  s->h_dterms.push_back(dt);
  mark_term_dirty(s, handle);
  s->term_by_off.emplace(postings_off, handle);
  *out = handle;
  // 0.25 B/doc per bitmap: worth it for lists whose 128-doc blocks span few docs, and only while
  // the bitmaps together stay within a fixed multiple of the segment's own size
  const size_t dense_bytes = (((size_t)s->max_doc + 31) / 32 + 1) * sizeof(uint2);
  const size_t budget = s->dense_budget();
  if (s->opt.dense && s->max_doc >= 4096u &&
      (uint64_t)th.doc_freq * (uint64_t)s->opt.dense_ratio >= s->max_doc &&
      s->dense_bytes_total + dense_bytes <= budget) {
    s->dense_bytes_total += dense_bytes;
    // (device scans whether or not a host copy of the index exists: round 4's host builder pulled the decoded
    // list back, built bitmap and rank directory in a loop and uploaded 2.5 MB per list)
    const int rc = build_dense_device(s, handle);
    if (rc != TQ_OK) return rc;
  }
  return add_to_doc_signatures(s, handle);
}

// tq_term_prepare without a host copy of the index: two small kernels walk the list's skip data
// and positions header where they lie in HBM (tq_prepare.hip); the host sizes the tables from
// TermInfo, and reads back 48 bytes of facts in between (no index bytes).
int term_prepare_device(tq_segment *s, uint64_t postings_off, uint32_t postings_len,
                        uint64_t positions_off, uint32_t positions_len, uint32_t doc_freq,
                        tq_term_handle *out) {
  const uint32_t n_full = doc_freq / 128u, n_tail = doc_freq % 128u;
  const uint32_t n_blocks = n_full + (n_tail ? 1u : 0u);
  uint32_t shift = 7;
  while (shift < 31 && ((uint64_t)(s->max_doc - 1) >> shift) + 1 > 2ull * n_blocks + 2) ++shift;
  const uint32_t n_buckets = (uint32_t)(((uint64_t)(s->max_doc - 1)) >> shift) + 1;
  const bool maybe_pos = s->record_option == TQ_WITH_FREQS_AND_POSITIONS && s->d_pos != nullptr;
  if (maybe_pos && (positions_off > s->pos_len || (uint64_t)positions_len > s->pos_len - positions_off))
    return fail(TQ_ERR_FORMAT, "positions_range outside the pos file");
  auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  size_t total = 0;
  auto place = [&](size_t bytes) {
    const size_t o = total;
    total = align16(total + bytes);
    return o;
  };
  const size_t o_rec = place(16 * (size_t)(n_blocks + 1));
  const size_t o_coarse = place(4 * (size_t)(n_buckets + 1));
  const size_t o_tdocs = place(4 * (size_t)n_tail);
  const size_t o_ttfs = place(4 * (size_t)n_tail);
  const size_t o_self = place(sizeof(TqdTerm));
  total += PAD;
  uint8_t *blob = nullptr;
  {
    const int arc = term_alloc(s, total, &blob);
    if (arc != TQ_OK) return arc;
  }
  s->bytes_term_tables += total;
  uint8_t *pblob = nullptr;
  size_t ptotal_alloc = 0;
  // a failed preparation hands its blobs back (positions blob first: the slab's last allocation), once the stream is idle
  auto bail = [&](int rc) {
    (void)hipStreamSynchronize(s->stream);
    term_release(s, pblob, ptotal_alloc);
    term_release(s, blob, total);
    return rc;
  };
  hipError_t e = hipMemsetAsync(blob, 0, total, s->stream);
  TqpPostingsParams pp{};
  pp.idx = s->d_idx;
  pp.pos = s->d_pos;
  pp.postings_off = postings_off;
  pp.positions_off = positions_off;
  pp.postings_len = postings_len;
  pp.positions_len = positions_len;
  pp.doc_freq = doc_freq;
  pp.record_option = s->record_option;
  pp.max_doc = s->max_doc;
  pp.want_pos = maybe_pos ? 1u : 0u;
  pp.rec = (uint4 *)(blob + o_rec);
  pp.tail_docs = (uint32_t *)(blob + o_tdocs);
  pp.tail_tfs = (uint32_t *)(blob + o_ttfs);
  pp.info = s->d_tp_info;
  if (e == hipSuccess) e = tqp_launch_postings(pp, s->stream);
  TqpInfo info{};
  if (e == hipSuccess) e = hipMemcpyAsync(&info, s->d_tp_info, sizeof info, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess) return bail(fail(TQ_ERR_HIP, "device term prepare: %s", hipGetErrorString(e)));
  if (info.status != TQP_OK)
    return bail(fail(info.status == TQP_TOO_MANY_POSITIONS ? TQ_ERR_UNSUPPORTED : TQ_ERR_FORMAT,
                     "term at %llu: %s", (unsigned long long)postings_off, tqp_message(info.status)));
  e = tqp_launch_coarse((const uint4 *)(blob + o_rec), n_blocks, shift, n_buckets,
                        (uint32_t *)(blob + o_coarse), s->stream);
  if (e != hipSuccess) return bail(fail(TQ_ERR_HIP, "coarse table: %s", hipGetErrorString(e)));
  // positions tables, sized from the walk
  const bool want_pos = maybe_pos && info.record == TQ_WITH_FREQS_AND_POSITIONS;
  uint32_t n_pos_tail = 0;
  size_t o_pboff = 0, o_ptail = 0;
  if (want_pos) {
    const uint64_t tail_cap = info.n_positions - info.n_pos_blocks * 128ull;
    if (tail_cap > 127ull)
      return bail(fail(TQ_ERR_FORMAT, "positions stream and postings disagree on the number of positions"));
    size_t ptotal = 0;
    o_pboff = 0;
    ptotal = align16(8 * (size_t)info.n_pos_blocks);
    o_ptail = ptotal;
    ptotal = align16(ptotal + 4 * (size_t)tail_cap) + PAD;
    if (term_alloc(s, ptotal, &pblob) != TQ_OK) return bail(TQ_ERR_HIP);
    ptotal_alloc = ptotal;
    s->bytes_term_tables += ptotal;
    if (e == hipSuccess) e = hipMemsetAsync(pblob, 0, ptotal, s->stream);
    TqpPositionsParams qp{};
    qp.pos = s->d_pos;
    qp.positions_off = positions_off;
    qp.pos_hdr = info.pos_hdr;
    qp.n_pos_blocks = info.n_pos_blocks;
    qp.n_positions = info.n_positions;
    qp.positions_len = positions_len;
    qp.pos_tail_cap = (uint32_t)tail_cap;
    qp.pos_blk = (uint64_t *)(pblob + o_pboff);
    qp.pos_tail = (uint32_t *)(pblob + o_ptail);
    qp.result = (uint32_t *)(s->d_tp_info + 1);
    uint32_t res[2] = {0, 0};
    if (e == hipSuccess) e = tqp_launch_positions(qp, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(res, s->d_tp_info + 1, sizeof res, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess || res[0] != TQP_OK) {
      return bail(e != hipSuccess ? fail(TQ_ERR_HIP, "device positions prepare: %s", hipGetErrorString(e))
                                  : fail(TQ_ERR_FORMAT, "term at %llu: %s", (unsigned long long)postings_off,
                                         tqp_message(res[0])));
    }
    n_pos_tail = res[1];
  } else {
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  TqdTerm dt{};
  dt.rec = (const uint4 *)(blob + o_rec);
  dt.coarse = (const uint32_t *)(blob + o_coarse);
  dt.tail_docs = (const uint32_t *)(blob + o_tdocs);
  dt.tail_tfs = (const uint32_t *)(blob + o_ttfs);
  dt.pos_blk = (const uint64_t *)(pblob ? pblob + o_pboff : blob + o_rec);
  dt.pos_tail = (const uint32_t *)(pblob ? pblob + o_ptail : blob + o_rec);
  dt.payload_base = 8 + postings_off + info.payload;
  dt.n_full = n_full;
  dt.n_tail = n_tail;
  dt.n_blocks = n_blocks;
  dt.doc_freq = doc_freq;
  dt.n_pos_blocks = want_pos ? (uint32_t)info.n_pos_blocks : 0u;
  dt.n_pos_tail = n_pos_tail;
  dt.has_freq = info.record != TQ_BASIC ? 1u : 0u;
  dt.coarse_shift = shift;
  {
    const hipError_t se = hipMemcpy(blob + o_self, &dt, sizeof dt, hipMemcpyHostToDevice);
    if (se != hipSuccess) return bail(fail(TQ_ERR_HIP, "term record upload: %s", hipGetErrorString(se)));
  }
  TermHost th;
  th.blob = blob;
  th.d_self = (const TqdTerm *)(blob + o_self);
  th.pos_blob = pblob;
  th.doc_freq = doc_freq;
  th.n_blocks = n_blocks;
  th.n_full = n_full;
  th.n_tail = n_tail;
  th.last_doc = info.last_doc;
  th.postings_len = postings_len;
  th.positions_len = want_pos ? positions_len : 0;
  th.n_positions = want_pos ? info.n_positions : 0;
  return register_term(s, dt, th, postings_off, out);
}

// Dense lists (doc_freq >= max_doc/TQD_DENSE_RATIO) also get a membership bitmap with a rank directory,
// {32 doc bits, number of postings before them} per 32 docs: a probe of doc d costs one 8-byte load instead
// of a block decode; the posting index (=> block, slot, tf) falls out of the rank.  Derived data like the
// unrolled skip table; the index bytes stay untouched.  The list is decoded once on the device (the kernel of
// tq_decode_postings), bitmap bits set by atomic OR, rank directory and position directory by grid-wide
// scans (tq_prepare.hip); 4 bytes (the validity flag) come back.
int build_dense_device(tq_segment *s, uint32_t handle) {
  int rc = TQ_OK;
  TermHost &t = s->terms[handle];
  const size_t bytes = (size_t)t.doc_freq * sizeof(uint32_t);
  const size_t n_words = ((size_t)s->max_doc + 31) / 32 + 1;
  const size_t scan_words = tqp_scan_scratch_words((uint32_t)std::max<size_t>(n_words, (size_t)t.doc_freq / 4 + 2));
  const size_t rm_words = ((size_t)s->max_doc >> TQD_RM_SHIFT) + 8;
  rc = s->d_misc.ensure(2 * bytes + 64 + (scan_words + rm_words) * sizeof(uint32_t));
  if (rc != TQ_OK) return rc;
  uint32_t *dd = (uint32_t *)s->d_misc.p, *dt = dd + t.doc_freq;
  uint32_t *scan_scratch = dt + t.doc_freq + 16;  // (tile sums of the scans below)
  uint32_t *rm_acc = scan_scratch + scan_words;
  hipError_t e = tqk_launch_decode_list(s->dseg, t.d_self, 0u, t.n_blocks, dd, dt,
                                        s->opt.use_dpp != 0, s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "decode launch: %s", hipGetErrorString(e));
  rc = build_tf8(s, handle, dt);
  if (rc != TQ_OK) return rc;
  if (t.tf8_blob) {  // (the shared launches probe a list through bitmap + tf bytes: only then is it bounded by ranges)
    rc = build_rmax(s, handle, dd, dt, rm_acc);
    if (rc != TQ_OK) return rc;
  }
  void *blob = nullptr;
  {
    const int arc = dense_alloc(s, n_words * sizeof(uint2), &blob);
    if (arc != TQ_OK) return arc;
  }
  s->bytes_bitmaps += n_words * sizeof(uint2);
  ++s->n_dense_lists;
  uint32_t *bad = (uint32_t *)s->d_tp_info;
  e = hipMemsetAsync(blob, 0, n_words * sizeof(uint2), s->stream);
  if (e == hipSuccess) e = hipMemsetAsync(bad, 0, 4, s->stream);
  if (e == hipSuccess)
    e = tqp_launch_dense(dd, t.doc_freq, s->max_doc, (uint2 *)blob, (uint32_t)n_words, bad, scan_scratch, s->stream);
  uint32_t h_bad = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess || h_bad) {
    dense_release(s, blob);
    return e != hipSuccess ? fail(TQ_ERR_HIP, "dense tables: %s", hipGetErrorString(e))
                           : fail(TQ_ERR_FORMAT, "posting list not strictly increasing below max_doc");
  }
  t.dense_blob = blob;
  s->h_dterms[handle].dense = (const uint2 *)blob;
  mark_term_dirty(s, handle);
  if (s->n_mat_slots < TQD_MAT_SLOTS && s->opt.docmat && t.wants_col) {  // the list's column of the doc matrix
    {
      const int mrc = ensure_docmat(s);
      if (mrc != TQ_OK) return mrc;
    }
    if (s->d_docmat) {
      const uint32_t slot = s->n_mat_slots++;
      e = tqk_launch_docmat_set(s->d_docmat, dd, t.doc_freq, slot, s->max_doc, s->stream);
      if (e != hipSuccess) return fail(TQ_ERR_HIP, "docmat set: %s", hipGetErrorString(e));
      // the list's tf classes (the first TQD_CLS_SLOTS columns; 8 B per doc once, inside the dense budget)
      static const bool kDocCls = tune_u32("TQ_DOCCLS", 1) != 0;
      if (kDocCls && slot < TQD_CLS_SLOTS && t.tf8_blob) {
        const size_t cls_bytes = (size_t)s->max_doc * sizeof(uint64_t);
        if (!s->d_doccls && s->dense_bytes_total + cls_bytes <= s->dense_budget()) {
          HIP_TRY(hipMalloc((void **)&s->d_doccls, cls_bytes + PAD));
          HIP_TRY(hipMemsetAsync(s->d_doccls, 0, cls_bytes + PAD, s->stream));
          s->dense_bytes_total += cls_bytes;
          s->bytes_docmat += cls_bytes;
          s->dseg.doccls = s->d_doccls;
        }
        if (s->d_doccls) {
          e = tqk_launch_doccls_set(s->d_doccls, dd, dt, t.doc_freq, slot, s->max_doc, s->stream);
          if (e != hipSuccess) return fail(TQ_ERR_HIP, "doccls set: %s", hipGetErrorString(e));
          s->h_dterms[handle].has_freq |= 1u << 24;  // (TqdTermHead::has_freq bit 24: the list's classes are in doccls)
        }
      }
      s->h_dterms[handle].has_freq |= (slot + 1u) << 8;
    }
  }
  if (t.positions_len > 0) {  // position directory: positions before every fourth posting
    const size_t n_dir = ((size_t)t.doc_freq + 3) / 4 + 1;
    // ... followed by the bitmap's doc bits alone (TqdTerm::bits: the phrase sweep's stream), 16-byte aligned
    const size_t dir_bytes = (n_dir * sizeof(uint32_t) + PAD + 15) & ~(size_t)15;
    const size_t n_bits = (n_words + 255) & ~(size_t)255;
    void *db = nullptr;
    {
      const int arc = dense_alloc(s, dir_bytes + n_bits * sizeof(uint32_t) + PAD, &db);
      if (arc != TQ_OK) return arc;
    }
    e = tqp_launch_posdir(dt, t.doc_freq, (uint32_t *)db, (uint32_t)n_dir, scan_scratch, s->stream);
    if (e == hipSuccess)
      e = tqp_launch_bits((const uint2 *)blob, (uint32_t)n_words, (uint32_t *)((uint8_t *)db + dir_bytes), (uint32_t)n_bits, s->stream);
    uint32_t total = 0;
    if (e == hipSuccess)
      e = hipMemcpyAsync(&total, (uint32_t *)db + (n_dir - 1), 4, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess || total != (uint32_t)t.n_positions) {
      dense_release(s, db);
      return e != hipSuccess ? fail(TQ_ERR_HIP, "position directory: %s", hipGetErrorString(e))
                             : fail(TQ_ERR_FORMAT, "term freqs sum to %u positions, the stream holds %llu",
                                    total, (unsigned long long)t.n_positions);
    }
    t.posdir_blob = db;
    s->h_dterms[handle].pos_dir = (const uint32_t *)db;
    s->h_dterms[handle].bits = (const uint32_t *)((uint8_t *)db + dir_bytes);
    s->dense_bytes_total += n_dir * sizeof(uint32_t) + n_bits * sizeof(uint32_t);
    s->bytes_posdir += n_dir * sizeof(uint32_t) + n_bits * sizeof(uint32_t);
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TQ_OK;
}


// Bitmap + rank directory and byte-wide tfs of a list BELOW "dense_ratio", for the boolean leads of the
// shared launch only (tq_ashare.hip probes every list but the leader through them): built the first time a
// boolean query names the list, inside "probe_budget_x"; no doc-matrix column, no position directory, and
// TermHost::dense_blob stays null — the other kernels and planners keep treating the list as sparse.
// *ok = the list can be probed (it has its own tables, or these).
void probe_begin_batch(tq_segment *s) {
  ++s->probe_batch;
  s->probe_waited = false;
}
void probe_touch(tq_segment *s, uint32_t handle) {
  const int32_t sl = s->terms[handle].probe_slot;
  if (sl >= 0) s->probe_slots[(size_t)sl].last_batch = s->probe_batch;
}
namespace {
// a slot for `handle`: a free one, a new one while the pool is below its budget, else the least recently used one
// (its owner loses its tables: the batches in flight are waited for first), else — every slot belongs to the batch
// being planned — one more.  *slot = -1: the list does not fit a slot (more postings than max_doc / 32: such a list
// gets tables of its own long before the budget of the dense lists is used up).
int probe_slot_acquire(tq_segment *s, uint32_t handle, bool must, int32_t *slot, bool *no_room) {
  *no_room = false;
  *slot = -1;
  TermHost &t = s->terms[handle];
  if (!s->probe_slot_bytes) {
    const size_t n_words = ((size_t)s->max_doc + 31) / 32 + 1;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    s->probe_bm_bytes = up(n_words * sizeof(uint2));
    // (lists below "dense_ratio" are the ones that need probe tables: max_doc / dense_ratio postings at most; a longer
    // list that did not get tables of its own takes the oversize road of build_probe_tables)
    s->probe_tf_cap = up(std::max<size_t>((size_t)s->max_doc / (size_t)std::max(1, std::min(s->opt.dense_ratio, 1 << 20)), 4096u) + PAD);
    s->probe_dir_cap = up(s->probe_tf_cap + 16 + PAD);  // ((df + 3) / 4 + 1) * 4 bytes
    s->probe_rm_bytes = up(tqd_rm_level_off(s->max_doc, TQD_RM_LEVELS));
    s->probe_slot_bytes = s->probe_bm_bytes + s->probe_tf_cap + s->probe_dir_cap + s->probe_rm_bytes;
  }
  if ((size_t)t.doc_freq + 8 > s->probe_tf_cap - PAD) return TQ_OK;
  const size_t budget_slots = std::max<size_t>(TQ_MAX_TERMS, s->probe_budget() / s->probe_slot_bytes);
  int32_t pick = -1;
  for (size_t i = 0; i < s->probe_slots.size() && pick < 0; ++i)
    if (s->probe_slots[i].owner == 0xFFFFFFFFu) pick = (int32_t)i;
  if (pick < 0 && s->probe_slots.size() >= budget_slots) {  // least recently used, not by this batch
    // (a query that can run without the tables — a shared-launch candidate — only takes a slot nobody has used for
    // kIdle batches: with a working set above the budget plain LRU rebuilt hundreds of tables per batch, 47 ms of
    // host time at 4 096 terms; a nested query MUST have its bitmaps and takes the least recently used slot)
    static const uint64_t kIdle = std::max<uint32_t>(1u, tune_u32("TQ_PROBE_IDLE_BATCHES", 64));
    uint64_t oldest = must ? s->probe_batch : (s->probe_batch > kIdle ? s->probe_batch - kIdle : 0);
    for (size_t i = 0; i < s->probe_slots.size(); ++i)
      if (s->probe_slots[i].last_batch < oldest) {
        oldest = s->probe_slots[i].last_batch;
        pick = (int32_t)i;
      }
    if (pick < 0 && !must) {
      *no_room = true;
      s->probe_no_room_batch = s->probe_batch;  // (the rest of this batch does not scan the slots again)
      return TQ_OK;
    }
    if (pick >= 0 && !must) {  // (a full pool takes few new lists per batch from callers that can do without)
      static const uint32_t kPerBatch = tune_u32("TQ_PROBE_REPLACE_PER_BATCH", 1);
      if (s->probe_replaced_batch != s->probe_batch) {
        s->probe_replaced_batch = s->probe_batch;
        s->probe_replaced_n = 0;
      }
      if (s->probe_replaced_n++ >= kPerBatch) {
        *no_room = true;
        s->probe_no_room_batch = s->probe_batch;
        return TQ_OK;
      }
    }
    if (pick >= 0) {
      // a batch in flight may still read the slot — unless nobody has used it for three batches (two are in flight at most)
      if (!s->probe_waited && s->probe_slots[(size_t)pick].last_batch + 3 > s->probe_batch) {
        const int wrc = wait_segment_idle(s);
        if (wrc != TQ_OK) return wrc;
        s->probe_waited = true;
      }
      TermHost &old = s->terms[s->probe_slots[(size_t)pick].owner];
      old.probe_dense_blob = old.probe_tf8_blob = old.probe_posdir_blob = nullptr;
      if (old.rmax_blob && !old.dense_blob) {  // (the range maxima lived in the slot)
        old.rmax_blob = nullptr;
        if (!old.rdir_blob) old.rmax_list = 255;  // (a list with a range directory keeps its maximum)
      }
      old.probe_slot = -1;
      s->probe_slots[(size_t)pick].owner = 0xFFFFFFFFu;
      ++s->probe_evictions;
      s->share_span_terms = ~(size_t)0;
    }
  }
  if (pick < 0) {  // a new slot (below the budget, or every slot is this batch's)
    void *base = nullptr;
    const int arc = dense_alloc(s, s->probe_slot_bytes, &base);
    if (arc != TQ_OK) return arc;
    tq_segment::ProbeSlot ps;
    ps.base = (uint8_t *)base;
    s->probe_slots.push_back(ps);
    s->probe_bytes_total += s->probe_slot_bytes;
    s->bytes_bitmaps += s->probe_slot_bytes;
    pick = (int32_t)s->probe_slots.size() - 1;
  }
  s->probe_slots[(size_t)pick].owner = handle;
  s->probe_slots[(size_t)pick].last_batch = s->probe_batch;
  t.probe_slot = pick;
  *slot = pick;
  return TQ_OK;
}
void probe_slot_release(tq_segment *s, uint32_t handle) {  // (a failed build)
  TermHost &t = s->terms[handle];
  if (t.probe_slot < 0) return;
  s->probe_slots[(size_t)t.probe_slot].owner = 0xFFFFFFFFu;
  t.probe_slot = -1;
}
}  // namespace

int build_probe_tables(tq_segment *s, uint32_t handle, bool *ok, bool must) {
  const bool any_size = must;
  TermHost &t = s->terms[handle];
  *ok = (t.dense_blob && t.tf8_blob) || (t.probe_dense_blob && t.probe_tf8_blob);
  if (*ok) {
    probe_touch(s, handle);
    return TQ_OK;
  }
  if (!s->opt.dense || !s->opt.use_dense || !t.doc_freq || s->opt.probe_budget_x <= 0) return TQ_OK;
  if (!must && s->probe_no_room_batch == s->probe_batch) return TQ_OK;
  // (segments below 4096 docs: the shared launches are not used there — only nested boolean queries, which reach every
  // list through a bitmap whatever the segment's size, ask with any_size)
  if (s->max_doc < 4096u && !any_size) return TQ_OK;
  const size_t n_words = ((size_t)s->max_doc + 31) / 32 + 1;
  HIP_TRY(hipSetDevice(s->device));
  int32_t slot = -1;
  bool no_room = false;
  int rc = probe_slot_acquire(s, handle, must, &slot, &no_room);
  if (rc != TQ_OK || no_room) return rc;
  uint8_t *base = nullptr;
  if (slot >= 0) {
    base = s->probe_slots[(size_t)slot].base;
  } else {
    // a list of more postings than a slot holds ("dense_ratio" below 32 leaves such lists without tables of their
    // own): tables of its own size, kept for good — there are at most 32 of them
    const size_t tf_room = (((size_t)t.doc_freq + 8 + PAD) + 255) & ~(size_t)255;
    void *own = nullptr;
    rc = dense_alloc(s, s->probe_bm_bytes + 2 * tf_room + 256 + s->probe_rm_bytes, &own);
    if (rc != TQ_OK) return rc;
    base = (uint8_t *)own;
    s->probe_bytes_total += s->probe_bm_bytes + 2 * tf_room + 256 + s->probe_rm_bytes;
    s->bytes_bitmaps += s->probe_bm_bytes + 2 * tf_room + 256 + s->probe_rm_bytes;
  }
  // (layout of a slot: bitmap | tf bytes | position directory | range maxima; an oversize list: the same, its own sizes)
  const size_t tf_cap = slot >= 0 ? s->probe_tf_cap : ((((size_t)t.doc_freq + 8 + PAD) + 255) & ~(size_t)255);
  const size_t dir_cap = slot >= 0 ? s->probe_dir_cap : tf_cap + 256;
  const size_t bytes = (size_t)t.doc_freq * sizeof(uint32_t);
  const size_t scan_words = tqp_scan_scratch_words((uint32_t)n_words);
  rc = s->d_misc.ensure(2 * bytes + 64 + (scan_words + ((size_t)s->max_doc >> TQD_RM_SHIFT) + 8) * sizeof(uint32_t));
  if (rc != TQ_OK) {
    probe_slot_release(s, handle);
    return rc;
  }
  uint32_t *dd = (uint32_t *)s->d_misc.p, *dt = dd + t.doc_freq;
  uint32_t *scan_scratch = dt + t.doc_freq + 16;
  uint32_t *rm_acc = scan_scratch + scan_words;
  hipError_t e = tqk_launch_decode_list(s->dseg, t.d_self, 0u, t.n_blocks, dd, dt, s->opt.use_dpp != 0, s->stream);
  void *blob = base, *tfb = base + s->probe_bm_bytes;
  uint32_t *bad = (uint32_t *)s->d_tp_info;
  if (e == hipSuccess) e = tqk_launch_tf8_pack(dt, t.doc_freq, (uint8_t *)tfb, s->stream);
  if (e == hipSuccess) e = hipMemsetAsync(blob, 0, n_words * sizeof(uint2), s->stream);
  if (e == hipSuccess) e = hipMemsetAsync(bad, 0, 4, s->stream);
  if (e == hipSuccess) e = tqp_launch_dense(dd, t.doc_freq, s->max_doc, (uint2 *)blob, (uint32_t)n_words, bad, scan_scratch, s->stream);
  // range maxima of the list (tq_ashare.hip's bound on a probed list), into the slot
  const uint32_t n_ranges = (s->max_doc >> TQD_RM_SHIFT) + 1u;
  uint8_t *rmb = base + s->probe_bm_bytes + tf_cap + dir_cap;
  const bool want_rm = !t.rmax_blob && s->d_local_cache;
  if (want_rm) {
    if (e == hipSuccess) e = hipMemsetAsync(rm_acc, 0, ((size_t)n_ranges + 1u) * sizeof(uint32_t), s->stream);
    if (e == hipSuccess)
      e = tqp_launch_rmax(dd, dt, t.doc_freq, s->d_fn, s->dseg.const_fieldnorm_id, s->d_local_cache, rm_acc, s->max_doc,
                          rmb, rm_acc + n_ranges, s->stream);
  }
  uint32_t h_bad = 0, lmax = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess && want_rm) e = hipMemcpyAsync(&lmax, rm_acc + n_ranges, 4, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess || h_bad) {
    probe_slot_release(s, handle);
    return e != hipSuccess ? fail(TQ_ERR_HIP, "probe tables: %s", hipGetErrorString(e))
                           : fail(TQ_ERR_FORMAT, "posting list not strictly increasing below max_doc");
  }
  t.probe_dense_blob = blob;
  t.probe_tf8_blob = tfb;
  if (slot < 0) t.probe_own_dir = base + s->probe_bm_bytes + tf_cap;  // (where its position directory goes)
  if (want_rm) {
    t.rmax_blob = rmb;
    t.rmax_list = lmax ? std::min<uint32_t>(lmax, 255u) : 255u;
  }
  *ok = true;
  return TQ_OK;
}

// The position directory of a list whose tables came from build_probe_tables: entry j = positions before posting
// 4 j, what a phrase inside a boolean query needs to find a doc's positions from the bitmap's rank (tq_tree.hip).
// Lives in the list's slot of the probe pool.
int build_probe_posdir(tq_segment *s, uint32_t handle, bool *ok) {
  TermHost &t = s->terms[handle];
  *ok = t.posdir_blob || t.probe_posdir_blob;
  if (*ok || !t.doc_freq || t.positions_len == 0) return TQ_OK;
  const bool own = t.dense_blob && t.tf8_blob;  // (a dense list whose directory did not fit when its tables were built)
  if (!own && !(t.probe_dense_blob && t.probe_tf8_blob)) return TQ_OK;
  if (!own && t.probe_slot < 0 && !t.probe_own_dir) return TQ_OK;
  const size_t n_dir = ((size_t)t.doc_freq + 3) / 4 + 1;
  const size_t need = n_dir * sizeof(uint32_t);
  if (!own && t.probe_slot >= 0 && need + PAD > s->probe_dir_cap) return TQ_OK;
  HIP_TRY(hipSetDevice(s->device));
  const size_t bytes = (size_t)t.doc_freq * sizeof(uint32_t);
  const size_t scan_words = tqp_scan_scratch_words((uint32_t)n_dir);
  int rc = s->d_misc.ensure(2 * bytes + 64 + (scan_words + 8) * sizeof(uint32_t));
  if (rc != TQ_OK) return rc;
  uint32_t *dd = (uint32_t *)s->d_misc.p, *dt = dd + t.doc_freq;
  uint32_t *scan_scratch = dt + t.doc_freq + 16;
  hipError_t e = tqk_launch_decode_list(s->dseg, t.d_self, 0u, t.n_blocks, dd, dt, s->opt.use_dpp != 0, s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "decode launch: %s", hipGetErrorString(e));
  void *db = nullptr;
  if (own) {
    rc = dense_alloc(s, need + PAD, &db);
    if (rc != TQ_OK) return rc;
    s->bytes_posdir += need;
  } else if (t.probe_slot >= 0) {
    db = s->probe_slots[(size_t)t.probe_slot].base + s->probe_bm_bytes + s->probe_tf_cap;
  } else {
    db = t.probe_own_dir;
  }
  e = tqp_launch_posdir(dt, t.doc_freq, (uint32_t *)db, (uint32_t)n_dir, scan_scratch, s->stream);
  uint32_t total = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&total, (uint32_t *)db + (n_dir - 1), 4, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess || total != (uint32_t)t.n_positions)
    return e != hipSuccess ? fail(TQ_ERR_HIP, "position directory: %s", hipGetErrorString(e))
                           : fail(TQ_ERR_FORMAT, "term freqs sum to %u positions, the stream holds %llu", total,
                                  (unsigned long long)t.n_positions);
  t.probe_posdir_blob = db;
  *ok = true;
  return TQ_OK;
}

void mark_term_dirty(tq_segment *s, uint32_t handle) {
  s->d_terms_dirty = true;
  s->d_terms_dirty_from = std::min<size_t>(s->d_terms_dirty_from, handle);
}

// The device copy of the term table, brought up to date (blocking form: codec access, Count, a grown table).
// tq_search_batch* takes the cheap road when it can: the records of the terms prepared since the last batch are
// appended through the batch's own staging blob (terms_pending_sync / tq_search.cpp) — no wait for the batches in
// flight, which never read beyond the entries they were planned with.
int sync_terms(tq_segment *s, hipStream_t st) {
  if (!s->d_terms_dirty) return TQ_OK;
  const size_t n = s->h_dterms.size();
  // nothing in flight may still read the table while it is rewritten
  {
    const int wrc = wait_segment_idle(s);
    if (wrc != TQ_OK) return wrc;
  }
  if (n > s->d_terms_cap) {
    HIP_TRY(hipStreamSynchronize(st));
    if (s->d_terms) (void)hipFree(s->d_terms);
    s->d_terms = nullptr;
    s->d_terms_cap = 0;
    size_t cap = std::max<size_t>(1024, n * 2);
    HIP_TRY(hipMalloc((void **)&s->d_terms, cap * sizeof(TqdTerm)));
    s->d_terms_cap = cap;
    s->d_terms_dirty_from = 0;
  }
  const size_t from = std::min(s->d_terms_dirty_from, n);
  if (n > from)
    HIP_TRY(hipMemcpy(s->d_terms + from, s->h_dterms.data() + from, (n - from) * sizeof(TqdTerm), hipMemcpyHostToDevice));
  s->d_terms_dirty = false;
  s->d_terms_dirty_from = ~(size_t)0;
  s->d_terms_synced = n;
  return TQ_OK;
}

}  // namespace tqi
