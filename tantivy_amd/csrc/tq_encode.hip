// tq_encode.hip — device-side writers of tantivy's postings codec: what
// PostingsSerializer::{new_term, write_doc, write_block, close_term} (src/postings/serializer.rs:
// 342-481, skip entries src/postings/skip.rs:55-88) and PositionSerializer (src/positions/
// serializer.rs:46-91) produce on the CPU, for a whole batch of terms at once, bit-identical.
//
// Work items are laid out in output order — per term [header | full blocks ... | tail] — sized in
// a first pass, placed by one exclusive scan and written by a second pass that re-reads the
// inputs (8 B per posting, twice) instead of parking packed blocks in a temporary:
//   measure  one wavefront per 128-value block: strict deltas, bit widths (OR-reduce), tf sum,
//            block-max (fieldnorm id, tf) pair = arg max of tf/(tf+cache[fieldnorm id]) with the
//            LAST maximum winning, as Iterator::max_by does (serializer.rs:404-428); one wavefront
//            per term sizes the vint tail and the header;
//   scan     three small kernels (partials / top / apply), u64 offsets;
//   write    BitPacker4x layout (4 interleaved little-endian bit streams, 16*b bytes) assembled
//            in LDS with ds_or, then copied to its (unaligned) place with dword stores inside and
//            byte stores at the edges; skip entries and vint tails by byte stores.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/tantivy_amd.h"
#include "tq_launch.h"

namespace {

constexpr int WAVE = 64;
constexpr int ENC_WAVES = 4;  // wavefronts per workgroup

// where block b lives: its term (| first block of the term << 31), its index inside the term and
// the index of its first value
struct BlkInfo {
  uint32_t term_first, j;
  uint64_t base;
};

struct EncParams {
  const uint64_t *term_starts;  // n_terms + 1, into values / tfs
  const uint32_t *blk_first;    // n_terms + 1: number of full blocks before term t
  BlkInfo *blk_info;            // n_blocks, written by enc_plan_kernel
  const uint32_t *values;       // doc ids (postings) or position deltas (positions)
  const uint32_t *tfs;          // postings with freqs, else null
  const uint8_t *fieldnorm_ids; // postings: block-max needs them, else null
  const float *cache;           // 256 f32: K1 * (1 - B + B * fieldnorm / avg)
  uint32_t *blk_meta;           // n_blocks: doc_bits | tf_bits << 8 | bw_fn << 16 | bw_tf << 24
  uint32_t *blk_tfsum;          // n_blocks (positions recorded)
  uint32_t *item_size;          // n_blocks + 2 * n_terms
  const uint64_t *item_off;     // same + 1 (after the scan)
  uint8_t *out;
  uint64_t *out_term_starts;    // n_terms + 1
  uint32_t n_terms, n_blocks;
  uint32_t fn_limit;            // num_docs - 1: doc ids are clamped for the fieldnorm gather
  uint32_t has_freq, has_pos, has_bm25, positions_file;
};

// wave-wide reductions on the DPP network (row_shr 1/2/4/8, row_bcast 15/31: six VALU ops, no LDS
// round trips); lanes without a source read the identity 0; the total lands in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t last_lane(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
  v |= dpp0<0x111, 0xF>(v);
  v |= dpp0<0x112, 0xF>(v);
  v |= dpp0<0x114, 0xF>(v);
  v |= dpp0<0x118, 0xF>(v);
  v |= dpp0<0x142, 0xA>(v);
  v |= dpp0<0x143, 0xC>(v);
  return last_lane(v);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int /*lane*/) {
  v += dpp0<0x111, 0xF>(v);
  v += dpp0<0x112, 0xF>(v);
  v += dpp0<0x114, 0xF>(v);
  v += dpp0<0x118, 0xF>(v);
  v += dpp0<0x142, 0xA>(v);
  v += dpp0<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ uint32_t wave_add(uint32_t v) { return last_lane(wave_incl_scan(v, 0)); }
// max of (hi, lo) pairs compared as one 64-bit key
__device__ __forceinline__ uint64_t wave_max64(uint64_t v) {
  uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
#define TQ_MAX_STEP(CTRL, MASK)                                       \
  {                                                                   \
    const uint32_t h2 = dpp0<CTRL, MASK>(hi), l2 = dpp0<CTRL, MASK>(lo); \
    const bool take = h2 > hi || (h2 == hi && l2 > lo);               \
    hi = take ? h2 : hi;                                              \
    lo = take ? l2 : lo;                                              \
  }
  TQ_MAX_STEP(0x111, 0xF)
  TQ_MAX_STEP(0x112, 0xF)
  TQ_MAX_STEP(0x114, 0xF)
  TQ_MAX_STEP(0x118, 0xF)
  TQ_MAX_STEP(0x142, 0xA)
  TQ_MAX_STEP(0x143, 0xC)
#undef TQ_MAX_STEP
  return ((uint64_t)last_lane(hi) << 32) | last_lane(lo);
}
__device__ __forceinline__ uint32_t bit_len(uint32_t v) { return v ? 32u - (uint32_t)__clz(v) : 0u; }
__device__ __forceinline__ uint32_t vint_len(uint32_t v) {  // compression/vint.rs: 7 bits a byte
  return v < (1u << 7) ? 1u : v < (1u << 14) ? 2u : v < (1u << 21) ? 3u : v < (1u << 28) ? 4u : 5u;
}
__device__ __forceinline__ uint32_t vint_len64(uint64_t v) {  // common VInt
  uint32_t n = 1;
  while (v >= 128u) {
    v >>= 7;
    ++n;
  }
  return n;
}
__device__ __forceinline__ uint32_t skip_entry_size(const EncParams &p) {  // skip.rs:55-88
  return 5u + (p.has_freq ? (p.has_pos ? 7u : 3u) : 0u);
}
// header of a term: postings = [VInt(skip_len) | skip entries] when doc_freq >= 128
// (serializer.rs:466-470); positions = VInt(n_full) | n_full width bytes (positions/serializer.rs:78-84)
__device__ __forceinline__ uint32_t header_size(const EncParams &p, uint32_t n_full) {
  if (p.positions_file) return vint_len64(n_full) + n_full;
  if (!n_full) return 0u;
  const uint64_t skip_len = (uint64_t)n_full * skip_entry_size(p);
  return vint_len64(skip_len) + (uint32_t)skip_len;
}

// the two values of this lane (i = lane, lane + 64) of block b, their stored form and the widths
struct BlockVals {
  uint32_t v0, v1;  // raw values (doc ids / deltas)
  uint32_t d0, d1;  // what gets bit-packed
  uint32_t t0, t1;  // tfs
};
// Both passes are latency-bound if a wavefront walks "block record -> values -> fieldnorms ->
// results" one block at a time (3.3 us per block measured).  So a wavefront takes ENC_CHUNK
// consecutive blocks: lane i loads block i's record once (coalesced), the per-block loop gets it
// through v_readlane, the values of block i+2 and the fieldnorm gather of block i+1 are in flight
// while block i is reduced, and the per-block results leave through lane i at the end.
constexpr uint32_t ENC_CHUNK = 32;
struct RawBlock {
  uint32_t v0, v1, t0, t1, prev, term;  // term: | first << 31
};
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t i) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)i);
}
__device__ __forceinline__ RawBlock load_raw(const EncParams &p, const BlkInfo &mine, uint32_t i,
                                             int lane) {
  RawBlock r;
  r.term = rl(mine.term_first, i);
  const uint64_t base = ((uint64_t)rl((uint32_t)(mine.base >> 32), i) << 32) | rl((uint32_t)mine.base, i);
  // every load is issued unconditionally (clamped addresses, dummy sources): conditional loads in
  // the pipelined loops make the compiler fall back to s_waitcnt vmcnt(0)
  const uint32_t *tfp = p.has_freq ? p.tfs : p.values;
  r.v0 = p.values[base + lane];
  r.v1 = p.values[base + 64 + lane];
  r.prev = p.values[base ? base - 1 : 0];
  r.t0 = tfp[base + lane];
  r.t1 = tfp[base + 64 + lane];
  if (!p.has_freq) r.t0 = r.t1 = 1u;
  if (r.term >> 31) r.prev = 0u;
  return r;
}
__device__ __forceinline__ void finish_block(const EncParams &p, const RawBlock &r, int lane,
                                             BlockVals &x) {
  x.v0 = r.v0;
  x.v1 = r.v1;
  x.t0 = r.t0;
  x.t1 = r.t1;
  if (p.positions_file) {
    x.d0 = x.v0;
    x.d1 = x.v1;
    return;
  }
  // compress_block_sorted(block, last_doc_id_encoded): strict delta, the first block of a term
  // (offset 0 <-> None) stores its first doc raw (compression/mod.rs:36-45)
  uint32_t p0 = __shfl_up(x.v0, 1, WAVE);
  uint32_t p1 = __shfl_up(x.v1, 1, WAVE);
  const uint32_t v0_last = last_lane(x.v0);
  if (lane == 0) {
    p1 = v0_last;
    p0 = r.prev;
  }
  x.d0 = x.v0 - p0 - 1u;
  if (lane == 0 && (r.term >> 31)) x.d0 = x.v0;
  x.d1 = x.v1 - p1 - 1u;
}

// block b -> its term (largest t with blk_first[t] <= b), one thread per block
__global__ __launch_bounds__(256) void enc_plan_kernel(EncParams p) {
  const uint32_t b = blockIdx.x * 256u + threadIdx.x;
  if (b >= p.n_blocks) return;
  uint32_t lo = 0, hi = p.n_terms;
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (p.blk_first[mid] <= b)
      lo = mid;
    else
      hi = mid;
  }
  const uint32_t j = b - p.blk_first[lo];
  BlkInfo o;
  o.term_first = lo | (j == 0u ? 0x80000000u : 0u);
  o.j = j;
  o.base = p.term_starts[lo] + 128ull * j;
  p.blk_info[b] = o;
}

__global__ __launch_bounds__(WAVE *ENC_WAVES) void enc_measure_kernel(EncParams p) {
  __shared__ float cache[256];
  const int lane = (int)__lane_id();
  const uint32_t wave = blockIdx.x * ENC_WAVES + threadIdx.x / WAVE;
  const uint32_t n_waves = gridDim.x * ENC_WAVES;
  if (p.has_bm25) {
    cache[threadIdx.x] = p.cache[threadIdx.x];
    __syncthreads();
  }
  const uint8_t *fnp = p.has_bm25 ? p.fieldnorm_ids : reinterpret_cast<const uint8_t *>(p.values);
  const uint32_t fn_mask = p.has_bm25 ? 0xFFFFFFFFu : 0u;  // no fieldnorms: every lane reads byte 0
  auto gather = [&](const RawBlock &r, uint32_t &f0, uint32_t &f1) __attribute__((always_inline)) {
    // (a doc id >= num_docs is the caller's bug; it must not become an out-of-bounds read)
    const uint32_t a0 = r.v0 & fn_mask, a1 = r.v1 & fn_mask;
    f0 = fnp[a0 < p.fn_limit ? a0 : p.fn_limit];
    f1 = fnp[a1 < p.fn_limit ? a1 : p.fn_limit];
  };
  for (uint32_t b0 = wave * ENC_CHUNK; b0 < p.n_blocks; b0 += n_waves * ENC_CHUNK) {
    const uint32_t nb = p.n_blocks - b0 < ENC_CHUNK ? p.n_blocks - b0 : ENC_CHUNK;
    BlkInfo mine{};
    if ((uint32_t)lane < nb) mine = p.blk_info[b0 + lane];
    uint32_t meta_mine = 0, tfsum_mine = 0;
    RawBlock r0 = load_raw(p, mine, 0, lane);
    RawBlock r1 = nb > 1u ? load_raw(p, mine, 1, lane) : r0;
    uint32_t nf0, nf1;
    gather(r0, nf0, nf1);
    for (uint32_t i = 0; i < nb; ++i) {
      const RawBlock cur = r0;
      const uint32_t f0 = nf0, f1 = nf1;
      r0 = r1;
      r1 = load_raw(p, mine, i + 2u < nb ? i + 2u : nb - 1u, lane);
      gather(r0, nf0, nf1);
      BlockVals x;
      finish_block(p, cur, lane, x);
      const uint32_t doc_bits = bit_len(wave_or(x.d0 | x.d1));
      uint32_t tf_bits = 0, bw_fn = 0, bw_tf = 0, tfsum = 0;
      if (p.has_freq) {
        tf_bits = bit_len(wave_or((x.t0 - 1u) | (x.t1 - 1u)));  // minus-one encoded (mod.rs:54-75)
        if (p.has_pos) tfsum = wave_add(x.t0 + x.t1);
        if (p.has_bm25) {
          const float tf0 = (float)x.t0, tf1 = (float)x.t1;
          const float s0 = tf0 / (tf0 + cache[f0]), s1 = tf1 / (tf1 + cache[f1]);
          // non-negative floats order like their bits; the index breaks ties towards the last
          const uint64_t k0 = ((uint64_t)__float_as_uint(s0) << 32) | (uint32_t)lane;
          const uint64_t k1 = ((uint64_t)__float_as_uint(s1) << 32) | (uint32_t)(lane + 64);
          const uint64_t best = wave_max64(k0 > k1 ? k0 : k1);
          const uint32_t bi = (uint32_t)best & 127u;
          const uint32_t bl = bi & 63u;  // wave-uniform (came through readlane)
          bw_fn = bi < 64u ? rl(f0, bl) : rl(f1, bl);
          bw_tf = bi < 64u ? rl(x.t0, bl) : rl(x.t1, bl);
          bw_tf = bw_tf > 255u ? 255u : bw_tf;  // encode_block_wand_max_tf (skip.rs:31-34)
        }
      }
      if ((uint32_t)lane == i) {
        meta_mine = doc_bits | (tf_bits << 8) | (bw_fn << 16) | (bw_tf << 24);
        tfsum_mine = tfsum;
      }
    }
    if ((uint32_t)lane < nb) {
      const uint32_t b = b0 + (uint32_t)lane, t = mine.term_first & 0x7FFFFFFFu;
      p.blk_meta[b] = meta_mine;
      if (p.has_pos) p.blk_tfsum[b] = tfsum_mine;
      p.item_size[b + 2u * t + 1u] = 16u * ((meta_mine & 255u) + ((meta_mine >> 8) & 255u));
    }
  }
  // headers and tails: one wavefront per term
  for (uint32_t t = wave; t < p.n_terms; t += n_waves) {
    const uint64_t lo = p.term_starts[t], hi = p.term_starts[t + 1u];
    const uint32_t n_full = (uint32_t)((hi - lo) >> 7), n_tail = (uint32_t)((hi - lo) & 127u);
    const uint64_t base = lo + (uint64_t)n_full * 128u;
    uint32_t bytes = 0;
    for (uint32_t i = (uint32_t)lane; i < n_tail; i += WAVE) {
      uint32_t v = p.values[base + i];
      if (!p.positions_file) {
        const uint32_t prev = i ? p.values[base + i - 1u] : (n_full ? p.values[base - 1u] : 0u);
        v -= prev;  // compress_vint_sorted: plain deltas (vint.rs:3-25)
      }
      bytes += vint_len(v);
      if (p.has_freq) bytes += vint_len(p.tfs[base + i]);
    }
    bytes = wave_add(bytes);
    if (lane == 0) {
      p.item_size[p.blk_first[t] + 2u * t] = header_size(p, n_full);
      p.item_size[p.blk_first[t + 1u] + 2u * t + 1u] = bytes;
    }
  }
}

// ---- exclusive scan of item_size (u32) into item_off (u64), n + 1 outputs
constexpr uint32_t SCAN_WG = 256, SCAN_PER_THREAD = 16, SCAN_TILE = SCAN_WG * SCAN_PER_THREAD;
__global__ __launch_bounds__(SCAN_WG) void scan_partials_kernel(const uint32_t *in, uint64_t *partials,
                                                                uint32_t n) {
  __shared__ uint64_t red[SCAN_WG];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
  uint64_t s = 0;
  for (uint32_t i = 0; i < SCAN_PER_THREAD; ++i)
    if (base + i < n) s += in[base + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t o = SCAN_WG / 2; o; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(SCAN_WG) void scan_top_kernel(uint64_t *partials, uint32_t n_partials) {
  __shared__ uint64_t sh[SCAN_WG];
  uint64_t carry = 0;
  for (uint32_t base = 0; base < n_partials; base += SCAN_WG) {
    const uint32_t i = base + threadIdx.x;
    const uint64_t v = i < n_partials ? partials[i] : 0ull;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t o = 1; o < SCAN_WG; o <<= 1) {
      const uint64_t u = threadIdx.x >= o ? sh[threadIdx.x - o] : 0ull;
      __syncthreads();
      sh[threadIdx.x] += u;
      __syncthreads();
    }
    if (i < n_partials) partials[i] = carry + sh[threadIdx.x] - v;  // exclusive
    carry += sh[SCAN_WG - 1];
    __syncthreads();
  }
}
__global__ __launch_bounds__(SCAN_WG) void scan_apply_kernel(const uint32_t *in, const uint64_t *partials,
                                                             uint64_t *out, uint32_t n) {
  __shared__ uint64_t sh[SCAN_WG];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
  uint64_t s = 0;
  for (uint32_t i = 0; i < SCAN_PER_THREAD; ++i)
    if (base + i < n) s += in[base + i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t o = 1; o < SCAN_WG; o <<= 1) {
    const uint64_t u = threadIdx.x >= o ? sh[threadIdx.x - o] : 0ull;
    __syncthreads();
    sh[threadIdx.x] += u;
    __syncthreads();
  }
  uint64_t run = partials[blockIdx.x] + sh[threadIdx.x] - s;
  for (uint32_t i = 0; i < SCAN_PER_THREAD; ++i) {
    if (base + i <= n) {  // out[n] = the total
      out[base + i] = run;
      if (base + i < n) run += in[base + i];
    }
  }
}

// ---- write pass
__device__ __forceinline__ void put_bytes(uint8_t *dst, uint64_t v, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) dst[i] = (uint8_t)(v >> (8u * i));
}
__device__ __forceinline__ uint32_t put_vint(uint8_t *dst, uint32_t v) {  // stop bit on the last byte
  uint32_t n = 0;
  for (;;) {
    const uint8_t b = (uint8_t)(v & 127u);
    v >>= 7;
    if (!v) {
      dst[n++] = b | 128u;
      return n;
    }
    dst[n++] = b;
  }
}
__device__ __forceinline__ uint32_t put_vint64(uint8_t *dst, uint64_t v) {
  uint32_t n = 0;
  for (;;) {
    const uint8_t b = (uint8_t)(v & 127u);
    v >>= 7;
    if (!v) {
      dst[n++] = b | 128u;
      return n;
    }
    dst[n++] = b;
  }
}
// BitPacker4x: value i lives in stream i & 3 at bit (i >> 2) * b; word w of stream l is u32 4*w + l
__device__ __forceinline__ void pack_value(uint32_t *words, uint32_t i, uint32_t b, uint32_t v) {
  if (!b) return;
  const uint32_t l = i & 3u, pos = (i >> 2) * b, w = pos >> 5, sh = pos & 31u;
  atomicOr(&words[4u * w + l], v << sh);
  if (sh + b > 32u) atomicOr(&words[4u * (w + 1u) + l], v >> (32u - sh));
}

__global__ __launch_bounds__(WAVE *ENC_WAVES) void enc_write_kernel(EncParams p) {
  __shared__ uint32_t pk_all[ENC_WAVES][260];
  const int lane = (int)__lane_id();
  const uint32_t wslot = threadIdx.x / WAVE;
  uint32_t *pk = pk_all[wslot];
  const uint32_t wave = blockIdx.x * ENC_WAVES + wslot;
  const uint32_t n_waves = gridDim.x * ENC_WAVES;
  const uint32_t entry = skip_entry_size(p);
  for (uint32_t b0 = wave * ENC_CHUNK; b0 < p.n_blocks; b0 += n_waves * ENC_CHUNK) {
    const uint32_t nb = p.n_blocks - b0 < ENC_CHUNK ? p.n_blocks - b0 : ENC_CHUNK;
    BlkInfo mine{};
    uint32_t meta_mine = 0;
    uint64_t off_mine = 0;
    if ((uint32_t)lane < nb) {
      mine = p.blk_info[b0 + lane];
      meta_mine = p.blk_meta[b0 + lane];
      off_mine = p.item_off[b0 + (uint32_t)lane + 2u * (mine.term_first & 0x7FFFFFFFu) + 1u];
    }
    RawBlock r0 = load_raw(p, mine, 0, lane);
    for (uint32_t i = 0; i < nb; ++i) {
      const RawBlock cur = r0;
      r0 = load_raw(p, mine, i + 1u < nb ? i + 1u : nb - 1u, lane);
      BlockVals x;
      finish_block(p, cur, lane, x);
      const uint32_t meta = rl(meta_mine, i);
      const uint32_t doc_bits = meta & 255u, tf_bits = (meta >> 8) & 255u;
      const uint32_t n_words = 4u * (doc_bits + tf_bits);
      for (uint32_t k = (uint32_t)lane; k < n_words + 1u; k += WAVE) pk[k] = 0u;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      pack_value(pk, (uint32_t)lane, doc_bits, x.d0);
      pack_value(pk, (uint32_t)lane + 64u, doc_bits, x.d1);
      if (p.has_freq) {
        pack_value(pk + 4u * doc_bits, (uint32_t)lane, tf_bits, x.t0 - 1u);
        pack_value(pk + 4u * doc_bits, (uint32_t)lane + 64u, tf_bits, x.t1 - 1u);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // payload -> out[off, off + n): dword stores where a dword lies inside, bytes at the edges
      const uint64_t off = ((uint64_t)rl((uint32_t)(off_mine >> 32), i) << 32) | rl((uint32_t)off_mine, i);
      const uint32_t n = 4u * n_words;
      const uint64_t a0 = off & ~3ull;
      const uint32_t lead = (uint32_t)(off - a0);  // bytes of the first dword that are not ours
      const uint32_t n_dw = (lead + n + 3u) >> 2;
      for (uint32_t k = (uint32_t)lane; k < n_dw; k += WAVE) {
        const int32_t s = (int32_t)(4u * k) - (int32_t)lead;  // source byte of this dword
        if (s >= 0 && (uint32_t)s + 4u <= n) {
          const uint32_t wi = (uint32_t)s >> 2, sh = ((uint32_t)s & 3u) * 8u;
          const uint32_t lo = pk[wi], hi = pk[wi + 1u];
          const uint32_t v = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
          *reinterpret_cast<uint32_t *>(p.out + a0 + 4ull * k) = v;
        } else {
          for (int q = 0; q < 4; ++q) {
            const int32_t sb = s + q;
            if (sb >= 0 && (uint32_t)sb < n)
              p.out[a0 + 4ull * k + (uint32_t)q] =
                  (uint8_t)(pk[(uint32_t)sb >> 2] >> (((uint32_t)sb & 3u) * 8u));
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    // skip entries (skip.rs:55-88) / width bytes into the terms' headers: lane i writes block i's
    if ((uint32_t)lane < nb) {
      const uint32_t b = b0 + (uint32_t)lane, t = mine.term_first & 0x7FFFFFFFu, j = mine.j;
      const uint32_t bf = p.blk_first[t], n_full = p.blk_first[t + 1u] - bf;
      const uint64_t hoff = p.item_off[bf + 2u * t];
      const uint32_t doc_bits = meta_mine & 255u, tf_bits = (meta_mine >> 8) & 255u;
      if (p.positions_file) {
        p.out[hoff + vint_len64(n_full) + j] = (uint8_t)doc_bits;
      } else {
        uint8_t *e = p.out + hoff + vint_len64((uint64_t)n_full * entry) + (uint64_t)j * entry;
        put_bytes(e, p.values[mine.base + 127u], 4);
        e[4] = (uint8_t)(doc_bits | 0x40u);  // strict-delta flag, always set (skip.rs:64-67)
        if (p.has_freq) {
          e[5] = (uint8_t)tf_bits;
          uint32_t at = 6;
          if (p.has_pos) {
            put_bytes(e + at, p.blk_tfsum[b], 4);
            at += 4;
          }
          e[at] = (uint8_t)(meta_mine >> 16);
          e[at + 1] = (uint8_t)(meta_mine >> 24);
        }
      }
    }
  }
  // headers' length prefixes and the vint tails: one wavefront per term
  for (uint32_t t = wave; t < p.n_terms; t += n_waves) {
    const uint64_t lo = p.term_starts[t], hi = p.term_starts[t + 1u];
    const uint32_t n_full = (uint32_t)((hi - lo) >> 7), n_tail = (uint32_t)((hi - lo) & 127u);
    const uint64_t base = lo + (uint64_t)n_full * 128u;
    const uint64_t hoff = p.item_off[p.blk_first[t] + 2u * t];
    if (lane == 0) {
      p.out_term_starts[t] = hoff;
      if (t + 1u == p.n_terms) p.out_term_starts[p.n_terms] = p.item_off[p.n_blocks + 2u * p.n_terms];
      if (p.positions_file)
        put_vint64(p.out + hoff, n_full);
      else if (n_full)
        put_vint64(p.out + hoff, (uint64_t)n_full * entry);
    }
    if (!n_tail) continue;
    uint8_t *dst = p.out + p.item_off[p.blk_first[t + 1u] + 2u * t + 1u];
    // values first (2 per lane: i = lane, lane + 64), then the tfs
    uint32_t v[2] = {0u, 0u}, f[2] = {0u, 0u}, lv[2] = {0u, 0u}, lf[2] = {0u, 0u};
    for (int h = 0; h < 2; ++h) {
      const uint32_t i = (uint32_t)lane + 64u * (uint32_t)h;
      if (i < n_tail) {
        v[h] = p.values[base + i];
        if (!p.positions_file) {
          const uint32_t prev = i ? p.values[base + i - 1u] : (n_full ? p.values[base - 1u] : 0u);
          v[h] -= prev;
        }
        lv[h] = vint_len(v[h]);
        if (p.has_freq) {
          f[h] = p.tfs[base + i];
          lf[h] = vint_len(f[h]);
        }
      }
    }
    const uint32_t sv0 = wave_incl_scan(lv[0], lane), tv0 = last_lane(sv0);
    const uint32_t sv1 = wave_incl_scan(lv[1], lane), tv1 = last_lane(sv1);
    const uint32_t sf0 = wave_incl_scan(lf[0], lane), tf0 = last_lane(sf0);
    const uint32_t sf1 = wave_incl_scan(lf[1], lane);
    const uint32_t docs_total = tv0 + tv1;
    if (lv[0]) put_vint(dst + (sv0 - lv[0]), v[0]);
    if (lv[1]) put_vint(dst + tv0 + (sv1 - lv[1]), v[1]);
    if (lf[0]) put_vint(dst + docs_total + (sf0 - lf[0]), f[0]);
    if (lf[1]) put_vint(dst + docs_total + tf0 + (sf1 - lf[1]), f[1]);
  }
}

}  // namespace

// ------------------------------------------------------------------ host side
struct tq_encoder {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_kernel_ms = 0.0f;
  // grow-only scratch
  void *d_scratch = nullptr;
  size_t scratch_cap = 0;
  void *d_inputs = nullptr;  // host-pointer entry points stage their inputs / output here
  size_t inputs_cap = 0;
};

namespace {
#define ENC_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e__ = (expr);                                                             \
    if (e__ != hipSuccess)                                                               \
      return tq_internal_fail(TQ_ERR_HIP, #expr, hipGetErrorString(e__));                \
  } while (0)

int grow(void **p, size_t *cap, size_t n) {
  if (n <= *cap) return TQ_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  ENC_TRY(hipMalloc(p, n));
  *cap = n;
  return TQ_OK;
}
size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// tf cache of Bm25Weight::for_one_term_without_explain(_, _, avg) (bm25.rs:62-69,132-146): it
// does not depend on the term, only on the segment's average fieldnorm
void tf_cache(float avg, float *cache) {
  for (int id = 0; id < 256; ++id) {
    uint32_t fn = (uint32_t)id;  // fieldnorm/code.rs: ids < 24 exact, then 3-bit mantissa steps
    if (id >= 24) {
      const uint32_t b = (uint32_t)id - 24u, bits = b & 7u, shift = b >> 3;
      fn = 24u + (shift == 0 ? bits : ((bits | 8u) << (shift - 1u)));
    }
    cache[id] = 1.2f * (1.0f - 0.75f + 0.75f * (float)fn / avg);
  }
}

int encode_device(tq_encoder *enc, bool positions_file, uint32_t n_terms,
                  const uint64_t *h_term_starts, const uint64_t *d_term_starts,
                  const uint32_t *d_values, const uint32_t *d_tfs, const uint8_t *d_fieldnorm_ids,
                  uint32_t num_docs, float avg_fieldnorm, uint8_t record_option, uint8_t *d_out,
                  uint64_t out_cap, uint64_t *d_out_term_starts, uint64_t *total_out,
                  hipStream_t st) {
  if (n_terms == 0) {  // nothing to write
    const uint64_t zero = 0;
    ENC_TRY(hipMemcpyAsync(d_out_term_starts, &zero, 8, hipMemcpyHostToDevice, st));
    ENC_TRY(hipEventRecord(enc->ev0, st));
    ENC_TRY(hipEventRecord(enc->ev1, st));
    ENC_TRY(hipStreamSynchronize(st));
    *total_out = 0;
    return TQ_OK;
  }
  // full blocks per term (host: O(n_terms + n_blocks))
  std::vector<uint32_t> blk_first(n_terms + 1);
  uint64_t nb = 0;
  for (uint32_t t = 0; t < n_terms; ++t) {
    if (h_term_starts[t + 1] < h_term_starts[t])
      return tq_internal_fail(TQ_ERR_INVALID, "tq_encode", "term_starts must not decrease");
    blk_first[t] = (uint32_t)nb;
    nb += (h_term_starts[t + 1] - h_term_starts[t]) >> 7;
    if (nb > 0x7FFFFFFFull) return tq_internal_fail(TQ_ERR_UNSUPPORTED, "tq_encode", "too many blocks");
  }
  blk_first[n_terms] = (uint32_t)nb;
  const uint32_t n_blocks = (uint32_t)nb;
  const uint32_t n_items = n_blocks + 2u * n_terms;
  const uint32_t n_partials = (n_items + 1u + SCAN_TILE - 1) / SCAN_TILE;  // + the total's slot

  size_t o = 0;
  auto take = [&](size_t n) {
    const size_t at = o;
    o = align256(o + n);
    return at;
  };
  const size_t o_first = take(4ull * (n_terms + 1)), o_info = take(16ull * n_blocks);
  const size_t o_meta = take(4ull * n_blocks), o_tfsum = take(4ull * n_blocks);
  const size_t o_size = take(4ull * n_items), o_off = take(8ull * (n_items + 1));
  const size_t o_part = take(8ull * (n_partials + 1)), o_cache = take(1024);
  if (int rc = grow(&enc->d_scratch, &enc->scratch_cap, o)) return rc;
  uint8_t *sc = (uint8_t *)enc->d_scratch;

  float cache[256];
  const bool has_freq = !positions_file && record_option != TQ_BASIC;
  const bool has_bm25 = has_freq && d_fieldnorm_ids && num_docs > 0;  // serializer.rs:353-377
  if (has_bm25) tf_cache(avg_fieldnorm, cache);
  ENC_TRY(hipMemcpyAsync(sc + o_first, blk_first.data(), 4ull * (n_terms + 1), hipMemcpyHostToDevice, st));
  if (has_bm25) ENC_TRY(hipMemcpyAsync(sc + o_cache, cache, 1024, hipMemcpyHostToDevice, st));

  EncParams p{};
  p.term_starts = d_term_starts;
  p.blk_first = (const uint32_t *)(sc + o_first);
  p.blk_info = (BlkInfo *)(sc + o_info);
  p.values = d_values;
  p.tfs = has_freq ? d_tfs : nullptr;
  p.fieldnorm_ids = d_fieldnorm_ids;
  p.cache = (const float *)(sc + o_cache);
  p.blk_meta = (uint32_t *)(sc + o_meta);
  p.blk_tfsum = (uint32_t *)(sc + o_tfsum);
  p.item_size = (uint32_t *)(sc + o_size);
  p.item_off = (const uint64_t *)(sc + o_off);
  p.out = d_out;
  p.out_term_starts = d_out_term_starts;
  p.n_terms = n_terms;
  p.n_blocks = n_blocks;
  p.fn_limit = has_bm25 ? num_docs - 1u : 0u;
  p.has_freq = has_freq;
  p.has_pos = !positions_file && record_option == TQ_WITH_FREQS_AND_POSITIONS;
  p.has_bm25 = has_bm25;
  p.positions_file = positions_file;

  const uint32_t work = std::max((n_blocks + ENC_CHUNK - 1) / ENC_CHUNK, n_terms);
  // grid cap (the loops are grid-stride over chunks of ENC_CHUNK blocks)
  static const uint32_t kMaxWgs = [] {
    const char *e = getenv("TQ_ENC_WGS");
    return e ? (uint32_t)atoi(e) : 8192u;
  }();
  const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>((work + ENC_WAVES - 1) / ENC_WAVES, kMaxWgs));
  ENC_TRY(hipEventRecord(enc->ev0, st));
  if (n_blocks) enc_plan_kernel<<<(n_blocks + 255) / 256, 256, 0, st>>>(p);
  enc_measure_kernel<<<grid, WAVE * ENC_WAVES, 0, st>>>(p);
  scan_partials_kernel<<<n_partials, SCAN_WG, 0, st>>>(p.item_size, (uint64_t *)(sc + o_part), n_items);
  scan_top_kernel<<<1, SCAN_WG, 0, st>>>((uint64_t *)(sc + o_part), n_partials);
  scan_apply_kernel<<<n_partials, SCAN_WG, 0, st>>>(p.item_size, (const uint64_t *)(sc + o_part),
                                                    (uint64_t *)(sc + o_off), n_items);
  // the total decides whether the output fits: 8 bytes back to the host
  uint64_t total = 0;
  ENC_TRY(hipMemcpyAsync(&total, sc + o_off + 8ull * n_items, 8, hipMemcpyDeviceToHost, st));
  ENC_TRY(hipStreamSynchronize(st));
  *total_out = total;
  if (total > out_cap)
    return tq_internal_fail(TQ_ERR_INVALID, "tq_encode", "output buffer too small (see *out_len)");
  enc_write_kernel<<<grid, WAVE * ENC_WAVES, 0, st>>>(p);
  ENC_TRY(hipEventRecord(enc->ev1, st));
  ENC_TRY(hipGetLastError());
  return TQ_OK;
}

int encode_host(tq_encoder *enc, bool positions_file, uint32_t n_terms, const uint64_t *term_starts,
                const uint32_t *values, const uint32_t *tfs, const uint8_t *fieldnorm_ids,
                uint32_t num_docs, float avg_fieldnorm, uint8_t record_option, uint8_t *out,
                uint64_t out_cap, uint64_t *out_term_starts, uint64_t *out_len) {
  if (!enc || !term_starts || !out_term_starts || !out_len || (!out && out_cap))
    return tq_internal_fail(TQ_ERR_INVALID, "tq_encode", "null argument");
  if (record_option > TQ_WITH_FREQS_AND_POSITIONS)
    return tq_internal_fail(TQ_ERR_INVALID, "tq_encode", "bad record_option");
  const uint64_t n_vals = term_starts[n_terms] - term_starts[0];
  if (term_starts[0] != 0) return tq_internal_fail(TQ_ERR_INVALID, "tq_encode", "term_starts[0] must be 0");
  if (n_vals && !values) return tq_internal_fail(TQ_ERR_INVALID, "tq_encode", "null values");
  const bool want_tfs = !positions_file && record_option != TQ_BASIC;
  if (want_tfs && n_vals && !tfs) return tq_internal_fail(TQ_ERR_INVALID, "tq_encode", "null tfs");
  ENC_TRY(hipSetDevice(enc->device));
  hipStream_t st = enc->stream;
  size_t o = 0;
  auto take = [&](size_t n) {
    const size_t at = o;
    o = align256(o + n + 16);
    return at;
  };
  const size_t o_ts = take(8ull * (n_terms + 1)), o_vals = take(4ull * n_vals);
  const size_t o_tfs = take(want_tfs ? 4ull * n_vals : 0);
  const size_t o_fn = take(fieldnorm_ids ? num_docs : 0), o_ots = take(8ull * (n_terms + 1));
  const size_t o_out = take(out_cap);
  if (int rc = grow(&enc->d_inputs, &enc->inputs_cap, o)) return rc;
  uint8_t *d = (uint8_t *)enc->d_inputs;
  ENC_TRY(hipMemcpyAsync(d + o_ts, term_starts, 8ull * (n_terms + 1), hipMemcpyHostToDevice, st));
  if (n_vals) ENC_TRY(hipMemcpyAsync(d + o_vals, values, 4ull * n_vals, hipMemcpyHostToDevice, st));
  if (want_tfs && n_vals) ENC_TRY(hipMemcpyAsync(d + o_tfs, tfs, 4ull * n_vals, hipMemcpyHostToDevice, st));
  if (fieldnorm_ids && num_docs)
    ENC_TRY(hipMemcpyAsync(d + o_fn, fieldnorm_ids, num_docs, hipMemcpyHostToDevice, st));
  int rc = encode_device(enc, positions_file, n_terms, term_starts, (const uint64_t *)(d + o_ts),
                         (const uint32_t *)(d + o_vals), want_tfs ? (const uint32_t *)(d + o_tfs) : nullptr,
                         fieldnorm_ids ? d + o_fn : nullptr, num_docs, avg_fieldnorm, record_option,
                         d + o_out, out_cap, (uint64_t *)(d + o_ots), out_len, st);
  if (rc != TQ_OK) return rc;
  if (*out_len) ENC_TRY(hipMemcpyAsync(out, d + o_out, *out_len, hipMemcpyDeviceToHost, st));
  ENC_TRY(hipMemcpyAsync(out_term_starts, d + o_ots, 8ull * (n_terms + 1), hipMemcpyDeviceToHost, st));
  ENC_TRY(hipStreamSynchronize(st));
  ENC_TRY(hipEventElapsedTime(&enc->last_kernel_ms, enc->ev0, enc->ev1));
  return TQ_OK;
}
}  // namespace

extern "C" {

int tq_encoder_create(tq_ctx *ctx, int device, tq_encoder **out) {
  if (!ctx || !out) return tq_internal_fail(TQ_ERR_INVALID, "tq_encoder_create", "null argument");
  if (!tq_internal_ctx_has_device(ctx, device))
    return tq_internal_fail(TQ_ERR_INVALID, "tq_encoder_create", "device not part of this context");
  ENC_TRY(hipSetDevice(device));
  tq_encoder *e = new tq_encoder();
  e->device = device;
  hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipEventCreate(&e->ev0);
  if (err == hipSuccess) err = hipEventCreate(&e->ev1);
  if (err != hipSuccess) {
    tq_encoder_free(e);
    return tq_internal_fail(TQ_ERR_HIP, "tq_encoder_create", hipGetErrorString(err));
  }
  *out = e;
  return TQ_OK;
}

void tq_encoder_free(tq_encoder *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->d_scratch) (void)hipFree(e->d_scratch);
  if (e->d_inputs) (void)hipFree(e->d_inputs);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int tq_encode_postings(tq_encoder *enc, uint32_t n_terms, const uint64_t *term_starts,
                       const uint32_t *docs, const uint32_t *tfs, const uint8_t *fieldnorm_ids,
                       uint32_t num_docs, float avg_fieldnorm, uint8_t record_option, uint8_t *out,
                       uint64_t out_cap, uint64_t *out_term_starts, uint64_t *out_len) {
  return encode_host(enc, false, n_terms, term_starts, docs, tfs, fieldnorm_ids, num_docs,
                     avg_fieldnorm, record_option, out, out_cap, out_term_starts, out_len);
}

int tq_encode_positions(tq_encoder *enc, uint32_t n_terms, const uint64_t *term_starts,
                        const uint32_t *position_deltas, uint8_t *out, uint64_t out_cap,
                        uint64_t *out_term_starts, uint64_t *out_len) {
  return encode_host(enc, true, n_terms, term_starts, position_deltas, nullptr, nullptr, 0, 0.0f,
                     TQ_BASIC, out, out_cap, out_term_starts, out_len);
}

int tq_encode_postings_device(tq_encoder *enc, uint32_t n_terms, const uint64_t *term_starts,
                              const uint64_t *d_term_starts, const uint32_t *d_docs,
                              const uint32_t *d_tfs, const uint8_t *d_fieldnorm_ids,
                              uint32_t num_docs, float avg_fieldnorm, uint8_t record_option,
                              uint8_t *d_out, uint64_t out_cap, uint64_t *d_out_term_starts,
                              uint64_t *out_len, void *hip_stream) {
  if (!enc || !term_starts || !d_term_starts || !d_out_term_starts || !out_len)
    return tq_internal_fail(TQ_ERR_INVALID, "tq_encode_postings_device", "null argument");
  if (record_option > TQ_WITH_FREQS_AND_POSITIONS)
    return tq_internal_fail(TQ_ERR_INVALID, "tq_encode_postings_device", "bad record_option");
  ENC_TRY(hipSetDevice(enc->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : enc->stream;
  return encode_device(enc, false, n_terms, term_starts, d_term_starts, d_docs, d_tfs,
                       d_fieldnorm_ids, num_docs, avg_fieldnorm, record_option, d_out, out_cap,
                       d_out_term_starts, out_len, st);
}

int tq_encode_positions_device(tq_encoder *enc, uint32_t n_terms, const uint64_t *term_starts,
                               const uint64_t *d_term_starts, const uint32_t *d_position_deltas,
                               uint8_t *d_out, uint64_t out_cap, uint64_t *d_out_term_starts,
                               uint64_t *out_len, void *hip_stream) {
  if (!enc || !term_starts || !d_term_starts || !d_out_term_starts || !out_len)
    return tq_internal_fail(TQ_ERR_INVALID, "tq_encode_positions_device", "null argument");
  ENC_TRY(hipSetDevice(enc->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : enc->stream;
  return encode_device(enc, true, n_terms, term_starts, d_term_starts, d_position_deltas, nullptr,
                       nullptr, 0, 0.0f, TQ_BASIC, d_out, out_cap, d_out_term_starts, out_len, st);
}

int tq_encoder_last_kernel_ms(tq_encoder *enc, float *ms) {
  if (!enc || !ms) return tq_internal_fail(TQ_ERR_INVALID, "tq_encoder_last_kernel_ms", "null argument");
  ENC_TRY(hipSetDevice(enc->device));
  ENC_TRY(hipEventSynchronize(enc->ev1));
  ENC_TRY(hipEventElapsedTime(ms, enc->ev0, enc->ev1));
  enc->last_kernel_ms = *ms;
  return TQ_OK;
}

}  // extern "C"
