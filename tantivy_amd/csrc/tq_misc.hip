// tq_misc.hip — top-k merge kernels and whole-list decode (codec parity, bitmap construction).
// Shared device helpers: tq_common.hpp.
#include "tq_common.hpp"

namespace {

// =================================================================== merge kernel
// One wavefront per query: reduce its partial lists to the final top-k, sorted.
// PRE: the first level of a two-level merge (TqkMergeParams::pre_slices): wavefront (q, s) reduces slice s of query
// q's lists into the slice's first list; the final launch reads one list per slice.
template <int KPL, bool PRE>
__global__ __launch_bounds__(64) void merge_kernel(TqkMergeParams p) {
  const int lane = (int)__lane_id();
  const uint32_t q = PRE ? blockIdx.x % p.n_queries : blockIdx.x;
  if (q >= p.n_queries) return;
  const TqdQuery *Q = uni_ptr(p.queries + q);
  const uint32_t k = uni(Q->k);
  uint32_t part_start = uni(Q->part_start), n_parts = uni(Q->n_parts);
  const uint32_t per = tqk_merge_slice_lists(n_parts, p.pre_slices);
  uint32_t stride = 1u;  // (in lists)
  if constexpr (PRE) {
    if (per == 1u) return;  // few lists: the final launch reads them all
    const uint32_t sl = blockIdx.x / p.n_queries;
    if (sl * per >= n_parts) return;
    part_start += sl * per;
    n_parts = n_parts - sl * per < per ? n_parts - sl * per : per;
    if (n_parts == 1u) return;  // (already "merged")
  } else {
    stride = per;
    n_parts = (n_parts + per - 1u) / per;
  }
  TopK<KPL> tk;
  tk.reset(k);
  // a full partial list's k-th key is a lower bound of the final k-th key: the largest of them
  // (one 8-byte load per list, 64 lists at a time) keeps almost every other key out of the
  // serial insertions below
  uint64_t floor_key = 0;
  for (uint32_t pi = (uint32_t)lane; pi < n_parts; pi += WAVE) {
    const uint64_t kth = p.partials[(uint64_t)(part_start + pi * stride) * (uint64_t)(KPL * 64) + (k - 1u)];
    floor_key = kth > floor_key ? kth : floor_key;
  }
  for (int o = 32; o; o >>= 1) {
    const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(floor_key >> 32), o, WAVE) << 32) |
                           (uint32_t)__shfl_xor((int)(uint32_t)floor_key, o, WAVE);
    floor_key = other > floor_key ? other : floor_key;
  }
  // the loads of a group of lists are issued together (one wavefront walks hundreds of lists:
  // one round trip per list was the whole cost of this kernel)
  constexpr uint32_t GROUP = KPL <= 2 ? 8u : (KPL <= 4 ? 4u : 1u);
  for (uint32_t pi0 = 0; pi0 < n_parts; pi0 += GROUP) {
    uint64_t keys[GROUP][KPL];
#pragma unroll
    for (uint32_t g = 0; g < GROUP; ++g) {
      const uint32_t pi = pi0 + g < n_parts ? pi0 + g : n_parts - 1u;  // clamped: unconditional loads
      const uint64_t *src = p.partials + (uint64_t)(part_start + pi * stride) * (uint64_t)(KPL * 64);
#pragma unroll
      for (int r = 0; r < KPL; ++r) keys[g][r] = src[(uint32_t)r * 64u + (uint32_t)lane];
    }
#pragma unroll
    for (uint32_t g = 0; g < GROUP; ++g) {
      if (pi0 + g >= n_parts) break;
#pragma unroll
      for (int r = 0; r < KPL; ++r)
        tk.offer(keys[g][r] != 0ull && keys[g][r] >= floor_key, keys[g][r], lane);
    }
  }
  if constexpr (PRE) {  // the slice's top-k replaces its first list (nothing else reads this slice's lists meanwhile)
    flush_partial<KPL>(tk, const_cast<uint64_t *>(p.partials), part_start, lane);
    return;
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

// =================================================================== batch-start zeroing
__global__ __launch_bounds__(256) void zero_regions_kernel(TqkZeroParams p) {
  const uint32_t r = blockIdx.y;
  uint32_t *dst = p.ptr[r];
  const uint32_t n = p.words[r];
  // 16-byte stores over the aligned middle, single words at the ends
  const uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u) >> 2;
  const uint32_t h = head < n ? head : n;
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
  if (tid < h) dst[tid] = 0u;
  const uint32_t n4 = (n - h) >> 2;
  uint4 *d4 = reinterpret_cast<uint4 *>(dst + h);
  for (uint32_t i = tid; i < n4; i += stride) d4[i] = make_uint4(0u, 0u, 0u, 0u);
  const uint32_t tail0 = h + (n4 << 2);
  if (tid < n - tail0) dst[tail0 + tid] = 0u;
}

// =================================================================== whole-list decode (codec parity)
template <bool USE_DPP>
__global__ __launch_bounds__(256) void decode_list_kernel(TqdSegment seg, const TqdTerm *terms,
                                                          uint32_t handle, uint32_t *docs,
                                                          uint32_t *tfs) {
  const int lane = (int)__lane_id();
  const uint32_t wave = uni(threadIdx.x >> 6);
  const TermRef t = load_term(terms, handle);
  const uint32_t j = blockIdx.x * 4u + wave;
  if (j >= t.n_blocks) return;
  const Dec d = decode_block<USE_DPP, false>(uni_ptr(seg.idx), t, j, lane);
  const uint32_t i0 = j * 128u + 2u * (uint32_t)lane;
  const uint32_t df = uni(terms[handle].doc_freq);
  if (i0 < df) {
    docs[i0] = d.d0;
    tfs[i0] = d.t0;
  }
  if (i0 + 1u < df) {
    docs[i0 + 1u] = d.d1;
    tfs[i0 + 1u] = d.t1;
  }
}

__global__ void decode_positions_kernel(TqdSegment seg, const TqdTerm *terms, uint32_t handle,
                                        uint32_t *out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = position_delta(seg.pos, terms + handle, i);
}

// =================================================================== cross-segment merge
// merge_top_k (sort_key_top_collector.rs:76-95): key = (score desc, segment_ord asc, doc asc).
// One wavefront per query; S*stride candidates; selection by repeated max (k <= 1024, tiny).
__global__ __launch_bounds__(64) void merge_segments_kernel(TqkSegMergeParams p) {
  const int lane = (int)__lane_id();
  const uint32_t q = blockIdx.x;
  if (q >= p.n_queries) return;
  const uint32_t total = p.n_segments * p.stride;
  const uint32_t want = p.offset + p.limit;
  // Each output rank r: the candidate with exactly r candidates ordered before it.
  // O(total^2 / 64) compares per query; total is S*k (e.g. 8*10).
  uint32_t n_valid = 0;
  for (uint32_t s = 0; s < p.n_segments; ++s) n_valid += p.counts[(uint64_t)s * p.n_queries + q];
  for (uint32_t c = (uint32_t)lane; c < total; c += 64u) {
    const uint32_t s = c / p.stride, i = c % p.stride;
    const uint32_t cnt = p.counts[(uint64_t)s * p.n_queries + q];
    if (i >= cnt) continue;
    const uint64_t at = ((uint64_t)s * p.n_queries + q) * p.stride + i;
    const float sc = p.scores[at];
    const uint32_t dc = p.docs[at];
    const uint32_t so = p.segment_ords ? p.segment_ords[s] : s;
    uint32_t before = 0;
    for (uint32_t s2 = 0; s2 < p.n_segments; ++s2) {
      const uint32_t cnt2 = p.counts[(uint64_t)s2 * p.n_queries + q];
      const uint32_t so2 = p.segment_ords ? p.segment_ords[s2] : s2;
      const uint64_t b2 = ((uint64_t)s2 * p.n_queries + q) * p.stride;
      for (uint32_t i2 = 0; i2 < cnt2; ++i2) {
        const float sc2 = p.scores[b2 + i2];
        const uint32_t dc2 = p.docs[b2 + i2];
        const bool lt = (sc2 > sc) || (sc2 == sc && (so2 < so || (so2 == so && dc2 < dc)));
        before += lt ? 1u : 0u;
      }
    }
    if (before >= p.offset && before < want) {
      const uint64_t o = (uint64_t)q * p.limit + (before - p.offset);
      p.out_scores[o] = sc;
      p.out_segment_ords[o] = so;
      p.out_docs[o] = dc;
    }
  }
  const uint32_t got = n_valid > p.offset ? (n_valid - p.offset < p.limit ? n_valid - p.offset : p.limit) : 0u;
  for (uint32_t r = got + (uint32_t)lane; r < p.limit; r += 64u) {
    const uint64_t o = (uint64_t)q * p.limit + r;
    p.out_scores[o] = 0.0f;
    p.out_segment_ords[o] = 0xFFFFFFFFu;
    p.out_docs[o] = TQD_TERMINATED;
  }
  if (lane == 0) p.out_counts[q] = got;
}

// =================================================================== doc matrix
__global__ void docmat_init_kernel(uint64_t *mat, const uint8_t *fieldnorm, uint32_t const_id,
                                   uint32_t max_doc) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < max_doc) mat[d] = fieldnorm ? (uint64_t)fieldnorm[d] : (uint64_t)const_id;
}
__global__ void docmat_set_kernel(uint64_t *mat, const uint32_t *docs, uint32_t n, uint32_t slot,
                                  uint32_t max_doc) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && docs[i] < max_doc)  // (a malformed list is rejected by the host right after)
    atomicOr((unsigned long long *)(mat + docs[i]), 1ull << (8u + slot));
}

// The signature bits of MANY lists in one launch (tq_term_prepare_batch: a batch's new sparse terms): one workgroup of
// four wavefronts per list, a wavefront per 128-doc block, the docs' doc-matrix words get bit items[i].y.
// ... and, where tabs[i] is not null, the list's range directory (rdir_lookup, tq_common.hpp): tabs[i] = the directory
// ((max_doc >> S) + 2 slots, padded to a multiple of four), the dfs[i] entries right behind it, shifts[i] = the list's S.  Entries are
// written where their posting's index says; a posting fills the directory slots of the ranges that begin after its
// predecessor's doc, the last posting the slots after its own.  With a Bm25 cache: lmax_out[i] (zeroed) gets the list's
// largest tf/(tf + norm) as a byte rounded up (td_rmax_scatter_kernel's value: what build_rmax calls the list maximum).
template <bool USE_DPP>
__global__ __launch_bounds__(256) void docsig_batch_kernel(TqdSegment seg, const TqdTerm *const *selfs, const uint32_t *bits,
                                                           uint64_t *mat, uint32_t *const *tabs, const uint32_t *shifts,
                                                           const uint32_t *dfs, const float *cache, uint32_t *lmax_out) {
  const int lane = (int)__lane_id();
  const uint32_t wave = uni(threadIdx.x >> 6);
  const TqdTerm *self = selfs[blockIdx.x];
  const TermRef t = load_term(self, 0u);
  const uint32_t b = bits[blockIdx.x];
  const uint64_t bit = 1ull << (TQD_SIG_SHIFT + (b & 15u));
  const bool sig = mat != nullptr && b != 0xFFFFFFFFu;
  uint32_t *dir = tabs ? tabs[blockIdx.x] : nullptr;
  const uint32_t S = dir ? shifts[blockIdx.x] : 0u, df = dir ? dfs[blockIdx.x] : 0u;
  const uint32_t n_ranges = (seg.max_doc >> S) + 1u;
  uint32_t *ent = dir ? dir + ((n_ranges + 1u + 3u) & ~3u) : nullptr;
  uint32_t lmax = 0;
  for (uint32_t j = wave; j < t.n_blocks; j += 4u) {
    const Dec d = decode_block<USE_DPP, false>(uni_ptr(seg.idx), t, j, lane);
    if (sig) {
      if (d.d0 < seg.max_doc) atomicOr((unsigned long long *)(mat + d.d0), (unsigned long long)bit);
      if (d.d1 < seg.max_doc) atomicOr((unsigned long long *)(mat + d.d1), (unsigned long long)bit);
    }
    if (ent) {
      const uint32_t i0 = 128u * j + 2u * (uint32_t)lane;
      const uint32_t up = (uint32_t)__shfl_up((int)d.d1, 1, WAVE);
      // (the range after the predecessor's; the list's first posting starts at range 0)
      uint32_t r_from = lane ? (up >> S) + 1u : (j ? (block_prev_last(t, j) >> S) + 1u : 0u);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t doc = e ? d.d1 : d.d0, tf = e ? d.t1 : d.t0, i = i0 + (uint32_t)e;
        if (i < df && doc < seg.max_doc) {
          ent[i] = rdir_entry(doc, tf, S);
          const uint32_t r_to = doc >> S;
          for (uint32_t r = r_from; r <= r_to; ++r) dir[r] = i;
          r_from = r_to + 1u;
          if (i + 1u == df)
            for (uint32_t r = r_from; r <= n_ranges; ++r) dir[r] = df;
          if (cache) {
            const float f = (float)tf;
            const float tfn = f / (f + cache[seg.fieldnorm ? (uint32_t)seg.fieldnorm[doc] : seg.const_fieldnorm_id]);
            uint32_t q = (uint32_t)(tfn * 255.0f) + 1u;
            q = q > 255u ? 255u : q;
            lmax = q > lmax ? q : lmax;
          }
        }
      }
    }
  }
  if (!ent || !cache) return;  // (uniform per workgroup)
  for (int o = 32; o; o >>= 1) {
    const uint32_t other = (uint32_t)__shfl_xor((int)lmax, o, WAVE);
    lmax = other > lmax ? other : lmax;
  }
  if (lane == 0 && lmax) atomicMax(lmax_out + blockIdx.x, lmax);
}
// the same from a decoded list (docs ascending, n of them): tq_term_prepare's path
__global__ void rdir_fill_kernel(const uint32_t *docs, const uint32_t *tfs, uint32_t n, uint32_t max_doc, uint32_t *dir,
                                 uint32_t S) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t *ent = dir + (((max_doc >> S) + 2u + 3u) & ~3u);
  const uint32_t doc = docs[i];
  if (doc >= max_doc) return;
  ent[i] = rdir_entry(doc, tfs[i], S);
  const uint32_t r_to = doc >> S;
  for (uint32_t r = i ? (docs[i - 1u] >> S) + 1u : 0u; r <= r_to; ++r) dir[r] = i;
  if (i + 1u == n)
    for (uint32_t r = r_to + 1u; r <= (max_doc >> S) + 1u; ++r) dir[r] = n;
}

__global__ void doccls_set_kernel(uint64_t *cls, const uint32_t *docs, const uint32_t *tfs, uint32_t n, uint32_t slot,
                                  uint32_t max_doc) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && docs[i] < max_doc) {
    const uint32_t tf = tfs[i];
    const uint64_t c = tf >= 3u ? 3ull : (tf ? (uint64_t)tf : 3ull);  // (tf 0 — a corrupt index — reads the tf byte)
    atomicOr((unsigned long long *)(cls + docs[i]), (unsigned long long)(c << (2u * slot)));
  }
}

__global__ void tf8_pack_kernel(const uint32_t *tfs, uint32_t n, uint8_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint8_t)(tfs[i] < 255u ? tfs[i] : 255u);
}

}  // namespace

// =================================================================== launch wrappers
hipError_t tqk_launch_tf8_pack(const uint32_t *tfs, uint32_t n, uint8_t *out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(tf8_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tfs, n, out);
  return hipGetLastError();
}
hipError_t tqk_launch_docmat_init(uint64_t *mat, const uint8_t *fieldnorm, uint32_t const_id,
                                  uint32_t max_doc, hipStream_t st) {
  if (max_doc == 0) return hipSuccess;
  hipLaunchKernelGGL(docmat_init_kernel, dim3((max_doc + 255) / 256), dim3(256), 0, st, mat,
                     fieldnorm, const_id, max_doc);
  return hipGetLastError();
}
hipError_t tqk_launch_docmat_set(uint64_t *mat, const uint32_t *docs, uint32_t n, uint32_t slot,
                                 uint32_t max_doc, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(docmat_set_kernel, dim3((n + 255) / 256), dim3(256), 0, st, mat, docs, n, slot,
                     max_doc);
  return hipGetLastError();
}
hipError_t tqk_launch_docsig_batch(const TqdSegment &seg, const TqdTerm *const *selfs, const uint32_t *bits, uint32_t n,
                                   uint64_t *mat, uint32_t *const *tabs, const uint32_t *shifts, const uint32_t *dfs,
                                   const float *cache, uint32_t *lmax_out, bool use_dpp, hipStream_t st) {
  if (n == 0) return hipSuccess;
  if (use_dpp)
    hipLaunchKernelGGL((docsig_batch_kernel<true>), dim3(n), dim3(256), 0, st, seg, selfs, bits, mat, tabs, shifts, dfs, cache, lmax_out);
  else
    hipLaunchKernelGGL((docsig_batch_kernel<false>), dim3(n), dim3(256), 0, st, seg, selfs, bits, mat, tabs, shifts, dfs, cache, lmax_out);
  return hipGetLastError();
}
hipError_t tqk_launch_rdir_fill(const uint32_t *docs, const uint32_t *tfs, uint32_t n, uint32_t max_doc, uint32_t *dir,
                                uint32_t S, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(rdir_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, docs, tfs, n, max_doc, dir, S);
  return hipGetLastError();
}
hipError_t tqk_launch_doccls_set(uint64_t *cls, const uint32_t *docs, const uint32_t *tfs, uint32_t n, uint32_t slot,
                                 uint32_t max_doc, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(doccls_set_kernel, dim3((n + 255) / 256), dim3(256), 0, st, cls, docs, tfs, n, slot, max_doc);
  return hipGetLastError();
}
hipError_t tqk_launch_zero(const TqkZeroParams &p, hipStream_t st) {
  if (p.n == 0) return hipSuccess;
  uint32_t max_words = 0;
  for (uint32_t i = 0; i < p.n; ++i) max_words = max_words > p.words[i] ? max_words : p.words[i];
  if (max_words == 0) return hipSuccess;
  const uint32_t bx = (max_words / 4u + 2047u) / 2048u;  // (<= 8 16-byte stores per thread)
  hipLaunchKernelGGL(zero_regions_kernel, dim3(bx ? (bx > 256u ? 256u : bx) : 1u, p.n), dim3(256), 0, st, p);
  return hipGetLastError();
}
template <bool PRE>
static void launch_merge_t(const TqkMergeParams &p, int kpl, dim3 grid, dim3 block, hipStream_t st) {
  switch (kpl) {
    case 1: merge_kernel<1, PRE><<<grid, block, 0, st>>>(p); break;
    case 2: merge_kernel<2, PRE><<<grid, block, 0, st>>>(p); break;
    case 4: merge_kernel<4, PRE><<<grid, block, 0, st>>>(p); break;
    default: merge_kernel<16, PRE><<<grid, block, 0, st>>>(p); break;
  }
}
hipError_t tqk_launch_merge(const TqkMergeParams &p, int kpl, hipStream_t st) {
  if (p.n_queries == 0) return hipSuccess;
  const dim3 block(64);
  if (p.pre_slices) {
    launch_merge_t<true>(p, kpl, dim3(p.n_queries * p.pre_slices), block, st);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  launch_merge_t<false>(p, kpl, dim3(p.n_queries), block, st);
  return hipGetLastError();
}
hipError_t tqk_launch_decode_list(const TqdSegment &seg, const TqdTerm *terms, uint32_t handle,
                                  uint32_t n_blocks, uint32_t *docs, uint32_t *tfs, bool use_dpp,
                                  hipStream_t st) {
  if (n_blocks == 0) return hipSuccess;
  const dim3 grid((n_blocks + 3) / 4), block(256);
  if (use_dpp)
    hipLaunchKernelGGL((decode_list_kernel<true>), grid, block, 0, st, seg, terms, handle, docs, tfs);
  else
    hipLaunchKernelGGL((decode_list_kernel<false>), grid, block, 0, st, seg, terms, handle, docs, tfs);
  return hipGetLastError();
}
hipError_t tqk_launch_decode_positions(const TqdSegment &seg, const TqdTerm *terms,
                                       uint32_t handle, uint32_t *out, uint64_t n,
                                       hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(decode_positions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     seg, terms, handle, out, n);
  return hipGetLastError();
}
hipError_t tqk_launch_merge_segments(const TqkSegMergeParams &p, hipStream_t st) {
  if (p.n_queries == 0) return hipSuccess;
  hipLaunchKernelGGL(merge_segments_kernel, dim3(p.n_queries), dim3(64), 0, st, p);
  return hipGetLastError();
}
