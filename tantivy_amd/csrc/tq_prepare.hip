// tq_prepare.hip — tq_term_prepare on the device: the sequential skip list of a posting list
// (src/postings/skip.rs:205-302) unrolled into the random-access tables of tq_device.h by one
// wavefront, straight from the index bytes in HBM — no host copy of the `.idx` / `.pos` files, so
// a segment encoded on the device (tq_encode_*_device) becomes searchable without its bytes
// visiting the host.  Also here: the derived structures of a dense list (bitmap + rank directory,
// position directory) built from the decoded list on the device.
//
// Format walkers restated (file:line under the tantivy checkout):
//   list framing        src/postings/block_segment_postings.rs:78-88,107-116
//   skip entries        src/postings/skip.rs:16-43,205-253,275-302
//   vint tail           src/postings/compression/vint.rs:44-108
//   positions framing   src/positions/reader.rs:43-56,84-101
#include "tq_common.hpp"
#include "tq_prepare.h"

namespace {

// common VInt (common/src/vint.rs:61-112): 7-bit groups, stop bit on the LAST byte
__device__ __forceinline__ bool rd_vint(const uint8_t *d, uint64_t len, uint64_t &at, uint64_t &out) {
  uint64_t r = 0;
  uint32_t shift = 0;
  while (at < len) {
    const uint8_t b = d[at++];
    r |= (uint64_t)(b & 127u) << shift;
    if (b & 128u) {
      out = r;
      return true;
    }
    shift += 7;
    if (shift > 63) return false;
  }
  return false;
}
__device__ __forceinline__ bool rd_vint32(const uint8_t *d, uint64_t len, uint64_t &at, uint32_t &out) {
  uint32_t r = 0, shift = 0;
  while (at < len) {
    const uint8_t b = d[at++];
    r += (uint32_t)(b & 127u) << shift;
    if (b & 128u) {
      out = r;
      return true;
    }
    shift += 7;
  }
  return false;
}
__device__ __forceinline__ uint32_t rd32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t x, int lane) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint64_t y = ((uint64_t)(uint32_t)__shfl_up((int)(x >> 32), d, WAVE) << 32) |
                       (uint32_t)__shfl_up((int)(uint32_t)x, d, WAVE);
    if (lane >= d) x += y;
  }
  return x;
}

// One wavefront per term.  Writes rec[0..n_blocks] (the extra record carries the total position
// count), the pre-decoded vint tail, and the facts the host needs to finish (info).
__global__ __launch_bounds__(64) void tp_postings_kernel(TqpPostingsParams p) {
  const int lane = (int)__lane_id();
  const uint8_t *data = p.idx + 8 + p.postings_off;
  const uint64_t len = p.postings_len;
  const uint32_t n_full = p.doc_freq / 128u, n_tail = p.doc_freq % 128u;
  TqpInfo info{};
  info.status = TQP_OK;
  uint32_t record = p.record_option;
  uint64_t at = 0, skip_len = 0;
  if (p.doc_freq >= 128u) {  // block_segment_postings.rs:78-88
    if (!rd_vint(data, len, at, skip_len) || skip_len > len - at) info.status = TQP_BAD_SKIP_LEN;
    if (info.status == TQP_OK && skip_len < 8ull * n_full) record = 0;  // :107-116 (JSON terms without freqs)
  }
  const uint32_t entry = record == 0 ? 5u : (record == 1 ? 8u : 12u);
  if (info.status == TQP_OK && skip_len < (uint64_t)entry * n_full) info.status = TQP_SKIP_TOO_SHORT;
  const uint8_t *skip = data + at;
  const uint64_t payload = at + skip_len;
  uint64_t running = 0, running_pos = 0;
  uint32_t last_doc = 0;
  if (info.status == TQP_OK) {
    uint32_t st = TQP_OK;  // per lane; the worst one wins after the loop
    for (uint32_t i0 = 0; i0 < n_full; i0 += WAVE) {  // skip.rs:205-253,275-302, 64 entries a round
      const uint32_t i = i0 + (uint32_t)lane;
      const bool on = i < n_full;
      uint32_t ld = 0, doc_bits = 0, strict = 0, tf_bits = 0, tf_sum = 0, bm_fn = 0, bm_tf = 0;
      if (on) {
        const uint8_t *e = skip + (uint64_t)entry * i;
        ld = rd32(e);
        doc_bits = e[4] & 0x1Fu;
        strict = (e[4] >> 6) & 1u;
        if (record == 1) {
          tf_bits = e[5];
          bm_fn = e[6];
          bm_tf = e[7];
        } else if (record == 2) {
          tf_bits = e[5];
          tf_sum = rd32(e + 6);
          bm_fn = e[10];
          bm_tf = e[11];
        }
      }
      const uint32_t prev_ld = __shfl_up(ld, 1, WAVE);
      const uint32_t before = lane == 0 ? last_doc : prev_ld;
      if (on && i && ld <= before && st < TQP_NOT_INCREASING) st = TQP_NOT_INCREASING;
      if (on && tf_bits > 32u) {
        st = TQP_BAD_TF_WIDTH;
        tf_bits = 0;
      }
      const uint64_t bytes = on ? 16ull * (doc_bits + tf_bits) : 0ull;
      const uint64_t incl_b = wave_incl_scan64(bytes, lane);
      const uint64_t incl_p = wave_incl_scan64((uint64_t)tf_sum, lane);
      if (on) {
        const uint64_t off = running + incl_b - bytes, ppos = running_pos + incl_p - tf_sum;
        if (ppos > 0xFFFFFFFFull || off > 0xFFFFFFFFull) st = TQP_TOO_MANY_POSITIONS;
        p.rec[i] = make_uint4(ld, doc_bits | (strict << 6) | (tf_bits << 8) | (bm_fn << 16) | (bm_tf << 24),
                              (uint32_t)off, (uint32_t)ppos);
      }
      running += readlane64(incl_b, 63);
      running_pos += readlane64(incl_p, 63);
      const uint32_t n_here = n_full - i0 < (uint32_t)WAVE ? n_full - i0 : (uint32_t)WAVE;
      last_doc = (uint32_t)__builtin_amdgcn_readlane((int)ld, (int)(n_here - 1u));
    }
    for (int o = 32; o; o >>= 1) {
      const uint32_t other = (uint32_t)__shfl_xor((int)st, o, WAVE);
      st = other > st ? other : st;
    }
    info.status = uni(st);
  }
  if (info.status == TQP_OK && payload + running > len) info.status = TQP_PAYLOAD_TOO_LONG;
  // vint tail (vint.rs:44-108): byte-serial, <= 127 postings — one lane
  if (info.status == TQP_OK && n_tail) {
    uint32_t st = TQP_OK;
    uint32_t tail_last = 0;
    uint64_t tail_pos = 0;
    if (lane == 0) {
      uint64_t t = payload + running;
      uint32_t prev = n_full ? last_doc : 0u;
      for (uint32_t i = 0; i < n_tail && st == TQP_OK; ++i) {
        uint32_t d;
        if (!rd_vint32(data, len, t, d)) {
          st = TQP_TRUNCATED_TAIL;
          break;
        }
        prev += d;
        p.tail_docs[i] = prev;
      }
      tail_last = prev;
      const bool has_tfs = record != 0 && t < len;
      for (uint32_t i = 0; i < n_tail && st == TQP_OK; ++i) {
        uint32_t f = 1u;
        if (has_tfs && !rd_vint32(data, len, t, f)) st = TQP_TRUNCATED_TAIL;
        p.tail_tfs[i] = f;
        if (record == 2) tail_pos += f;  // tf sums only index a positions stream
      }
    }
    st = uni(st);
    tail_last = uni(tail_last);
    tail_pos = uni64(tail_pos);
    if (st != TQP_OK) info.status = st;
    if (info.status == TQP_OK) {
      if (running_pos > 0xFFFFFFFFull) info.status = TQP_TOO_MANY_POSITIONS;
      if (lane == 0) p.rec[n_full] = make_uint4(tail_last, 0xFFFFFFFFu, 0u, (uint32_t)running_pos);
      running_pos += tail_pos;
      last_doc = tail_last;
    }
  }
  if (info.status == TQP_OK && running_pos > 0xFFFFFFFFull) info.status = TQP_TOO_MANY_POSITIONS;
  const uint32_t n_blocks = n_full + (n_tail ? 1u : 0u);
  if (info.status == TQP_OK && lane == 0)
    p.rec[n_blocks] = make_uint4(TQD_TERMINATED, 0u, 0u, (uint32_t)running_pos);
  if (info.status == TQP_OK && (last_doc >= TQD_TERMINATED || last_doc >= p.max_doc)) info.status = TQP_DOC_OUT_OF_RANGE;
  info.record = record;
  info.payload = payload;
  info.n_positions = running_pos;
  info.last_doc = last_doc;
  // positions stream header (positions/reader.rs:43-56): VInt n_blocks, then that many width bytes
  info.n_pos_blocks = 0;
  info.pos_hdr = 0;
  if (info.status == TQP_OK && p.want_pos) {
    uint64_t pa = 0, nb = 0;
    const uint8_t *pd = p.pos + p.positions_off;
    if (!rd_vint(pd, p.positions_len, pa, nb) || nb > p.positions_len - pa) {
      info.status = TQP_BAD_POS_HEADER;
    } else {
      info.n_pos_blocks = nb;
      info.pos_hdr = pa;
      if (nb * 128ull > running_pos) info.status = TQP_POS_COUNT_MISMATCH;
    }
  }
  if (lane == 0) *p.info = info;
}

// coarse[b] = first block j with last_doc[j] >= b << shift (one bucket per thread)
__global__ void tp_coarse_kernel(const uint4 *rec, uint32_t n_blocks, uint32_t shift, uint32_t n_buckets,
                                 uint32_t *coarse) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > n_buckets) return;
  const uint64_t lo_doc = (uint64_t)b << shift;
  uint32_t lo = 0, hi = n_blocks;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((uint64_t)rec[mid].x >= lo_doc)
      hi = mid;
    else
      lo = mid + 1u;
  }
  coarse[b] = lo;
}

// positions stream tables: pos_blk[i] = absolute byte offset of position block i | width << 56
// (running sum of 16 * width), then the vint tail until the end of the range.  One wavefront.
__global__ __launch_bounds__(64) void tp_positions_kernel(TqpPositionsParams p) {
  const int lane = (int)__lane_id();
  const uint8_t *pd = p.pos + p.positions_off;
  const uint8_t *widths = pd + p.pos_hdr;
  const uint64_t nb = p.n_pos_blocks;
  uint64_t running = 0;
  uint32_t status = TQP_OK;
  for (uint64_t i0 = 0; i0 < nb; i0 += WAVE) {
    const uint64_t i = i0 + (uint64_t)lane;
    const uint32_t w = i < nb ? widths[i] : 0u;
    if (__ballot(w > 32u)) status = TQP_BAD_POS_WIDTH;
    const uint64_t bytes = 16ull * w;
    const uint64_t incl = wave_incl_scan64(bytes, lane);
    if (i < nb)
      p.pos_blk[i] = (p.positions_off + p.pos_hdr + nb + running + incl - bytes) | ((uint64_t)w << 56);
    running += readlane64(incl, 63);
  }
  uint64_t t = p.pos_hdr + nb + running;
  if (status == TQP_OK && t > p.positions_len) status = TQP_POS_PAYLOAD_TOO_LONG;
  uint32_t n_tail = 0;
  if (status == TQP_OK && lane == 0) {
    while (t < p.positions_len) {  // uncompress_vint_unsorted_until_end
      uint32_t v;
      if (!rd_vint32(pd, p.positions_len, t, v)) {
        status = TQP_TRUNCATED_TAIL;
        break;
      }
      if (n_tail < p.pos_tail_cap) p.pos_tail[n_tail] = v;
      ++n_tail;
    }
  }
  status = uni(status);
  n_tail = uni(n_tail);
  if (status == TQP_OK && nb * 128ull + n_tail != p.n_positions) status = TQP_POS_COUNT_MISMATCH;
  if (lane == 0) {
    p.result[0] = status;
    p.result[1] = n_tail;
  }
}

// ---- derived structures of a dense list, from the decoded doc ids / tfs
__global__ void td_check_bits_kernel(const uint32_t *docs, uint32_t n, uint32_t max_doc, uint2 *tab,
                                     uint32_t *bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t d = docs[i];
  if (d >= max_doc || (i && d <= docs[i - 1u])) {
    atomicOr(bad, 1u);
    return;
  }
  atomicOr(&tab[d >> 5].x, 1u << (d & 31u));
}
// rank directory: tab[w].y = number of postings before word w.  One workgroup walks the words
// in chunks with a carry (n_words = max_doc / 32: a few hundred thousand).
__global__ __launch_bounds__(1024) void td_rank_kernel(uint2 *tab, uint32_t n_words) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_words; base += 1024u) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_words ? (uint32_t)__popc(tab[i].x) : 0u;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
      const uint32_t u = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
      __syncthreads();
      sh[threadIdx.x] += u;
      __syncthreads();
    }
    const uint32_t carry = carry_s;
    if (i < n_words) tab[i].y = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023u) carry_s = carry + sh[1023];
    __syncthreads();
  }
}
// position directory: dir[j] = sum of the tfs of postings 0..4j-1; dir[n_dir-1] = the total
__global__ __launch_bounds__(1024) void td_posdir_kernel(const uint32_t *tfs, uint32_t n, uint32_t *dir,
                                                         uint32_t n_dir) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_dir; base += 1024u) {  // one group of four postings per thread
    const uint32_t j = base + threadIdx.x;
    uint32_t v = 0;
    for (uint32_t e = 0; e < 4u; ++e) {
      const uint64_t i = 4ull * j + e;
      if (i < n) v += tfs[i];
    }
    sh[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
      const uint32_t u = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
      __syncthreads();
      sh[threadIdx.x] += u;
      __syncthreads();
    }
    const uint32_t carry = carry_s;
    if (j < n_dir) dir[j] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023u) carry_s = carry + sh[1023];
    __syncthreads();
  }
}
__global__ void td_min_fieldnorm_kernel(const uint8_t *fieldnorm, uint32_t max_doc, uint32_t *out) {
  uint32_t mn = 255u;
  for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < max_doc; d += gridDim.x * blockDim.x)
    mn = fieldnorm[d] < mn ? fieldnorm[d] : mn;
  for (int o = 32; o; o >>= 1) {
    const uint32_t other = (uint32_t)__shfl_xor((int)mn, o, WAVE);
    mn = other < mn ? other : mn;
  }
  if ((threadIdx.x & 63u) == 0u) atomicMin(out, mn);
}

}  // namespace

hipError_t tqp_launch_postings(const TqpPostingsParams &p, hipStream_t st) {
  hipLaunchKernelGGL(tp_postings_kernel, dim3(1), dim3(64), 0, st, p);
  return hipGetLastError();
}
hipError_t tqp_launch_coarse(const uint4 *rec, uint32_t n_blocks, uint32_t shift, uint32_t n_buckets,
                             uint32_t *coarse, hipStream_t st) {
  hipLaunchKernelGGL(tp_coarse_kernel, dim3((n_buckets + 1 + 255) / 256), dim3(256), 0, st, rec, n_blocks,
                     shift, n_buckets, coarse);
  return hipGetLastError();
}
hipError_t tqp_launch_positions(const TqpPositionsParams &p, hipStream_t st) {
  hipLaunchKernelGGL(tp_positions_kernel, dim3(1), dim3(64), 0, st, p);
  return hipGetLastError();
}
hipError_t tqp_launch_dense(const uint32_t *docs, uint32_t n, uint32_t max_doc, uint2 *tab,
                            uint32_t n_words, uint32_t *bad, hipStream_t st) {
  if (n) hipLaunchKernelGGL(td_check_bits_kernel, dim3((n + 255) / 256), dim3(256), 0, st, docs, n, max_doc, tab, bad);
  hipLaunchKernelGGL(td_rank_kernel, dim3(1), dim3(1024), 0, st, tab, n_words);
  return hipGetLastError();
}
hipError_t tqp_launch_posdir(const uint32_t *tfs, uint32_t n, uint32_t *dir, uint32_t n_dir,
                             hipStream_t st) {
  hipLaunchKernelGGL(td_posdir_kernel, dim3(1), dim3(1024), 0, st, tfs, n, dir, n_dir);
  return hipGetLastError();
}
hipError_t tqp_launch_min_fieldnorm(const uint8_t *fieldnorm, uint32_t max_doc, uint32_t *out,
                                    hipStream_t st) {
  hipLaunchKernelGGL(td_min_fieldnorm_kernel, dim3(256), dim3(256), 0, st, fieldnorm, max_doc, out);
  return hipGetLastError();
}
