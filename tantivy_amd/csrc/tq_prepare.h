// tq_prepare.h — parameter blocks of the device-side tq_term_prepare (tq_prepare.hip <-> tq_api.cpp)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum TqpStatus : uint32_t {
  TQP_OK = 0,
  TQP_BAD_SKIP_LEN,
  TQP_SKIP_TOO_SHORT,
  TQP_NOT_INCREASING,
  TQP_BAD_TF_WIDTH,
  TQP_TOO_MANY_POSITIONS,
  TQP_PAYLOAD_TOO_LONG,
  TQP_TRUNCATED_TAIL,
  TQP_DOC_OUT_OF_RANGE,
  TQP_BAD_POS_HEADER,
  TQP_POS_COUNT_MISMATCH,
  TQP_BAD_POS_WIDTH,
  TQP_POS_PAYLOAD_TOO_LONG,
};

// what the host needs back from the walk of one posting list: 48 bytes of facts, no index bytes
struct TqpInfo {
  uint32_t status;      // TqpStatus
  uint32_t record;      // effective record option (a JSON term may lack the freqs of its field)
  uint64_t payload;     // offset of block 0's payload inside the list
  uint64_t n_positions; // sum of the term freqs (fields with positions)
  uint64_t n_pos_blocks;
  uint64_t pos_hdr;     // bytes of the positions header's VInt
  uint32_t last_doc;
  uint32_t pad;
};

struct TqpPostingsParams {
  const uint8_t *idx;   // the field's .idx sub-file on the device (8-byte header included)
  const uint8_t *pos;   // .pos sub-file or null
  uint64_t postings_off, positions_off;
  uint32_t postings_len, positions_len;
  uint32_t doc_freq, record_option, max_doc, want_pos;
  uint4 *rec;           // n_blocks + 1
  uint32_t *tail_docs, *tail_tfs;
  TqpInfo *info;
};

struct TqpPositionsParams {
  const uint8_t *pos;
  uint64_t positions_off, pos_hdr, n_pos_blocks, n_positions;
  uint32_t positions_len, pos_tail_cap;
  uint64_t *pos_blk;
  uint32_t *pos_tail;
  uint32_t *result;     // [0] TqpStatus, [1] number of tail values
};

hipError_t tqp_launch_postings(const TqpPostingsParams &p, hipStream_t st);
hipError_t tqp_launch_coarse(const uint4 *rec, uint32_t n_blocks, uint32_t shift, uint32_t n_buckets,
                             uint32_t *coarse, hipStream_t st);
hipError_t tqp_launch_positions(const TqpPositionsParams &p, hipStream_t st);
// bitmap + rank directory of a dense list from its decoded doc ids (tab zeroed by the caller);
// *bad != 0 afterwards: the list was not strictly increasing below max_doc
// scan_scratch: tqp_scan_scratch_words(n_words) u32 of device memory (tile sums of the grid-wide scan)
uint32_t tqp_scan_scratch_words(uint32_t n_items);
hipError_t tqp_launch_dense(const uint32_t *docs, uint32_t n, uint32_t max_doc, uint2 *tab,
                            uint32_t n_words, uint32_t *bad, uint32_t *scan_scratch, hipStream_t st);
// position directory of a dense list: dir[j] = positions before posting 4j (n_dir = (n + 3) / 4 + 1 entries);
// scan_scratch: tqp_scan_scratch_words(n_dir) u32
hipError_t tqp_launch_posdir(const uint32_t *tfs, uint32_t n, uint32_t *dir, uint32_t n_dir,
                             uint32_t *scan_scratch, hipStream_t st);
// the doc bits of a bitmap + rank directory alone: bits[w] = tab[w].x for w < n_words, 0 up to n_padded
hipError_t tqp_launch_bits(const uint2 *tab, uint32_t n_words, uint32_t *bits, uint32_t n_padded, hipStream_t st);
// range maxima of a list with a bitmap (tq_device.h TQD_RM_*: tq_ashare.hip's bound on the non-leader lists);
// acc = (max_doc >> TQD_RM_SHIFT) + 1 ZEROED u32 of scratch, out = tqd_rm_level_off(max_doc, TQD_RM_LEVELS) bytes,
// *list_max zeroed
// *out (zeroed) = the list's largest tf/(tf + norm) under `cache`, as td_rmax_* round it (a byte, rounded up)
hipError_t tqp_launch_list_max(const uint32_t *docs, const uint32_t *tfs, uint32_t n, const uint8_t *fieldnorm,
                               uint32_t const_id, const float *cache, uint32_t *out, hipStream_t st);
hipError_t tqp_launch_rmax(const uint32_t *docs, const uint32_t *tfs, uint32_t n, const uint8_t *fieldnorm,
                           uint32_t const_id, const float *cache, uint32_t *acc, uint32_t max_doc, uint8_t *out,
                           uint32_t *list_max, hipStream_t st);
hipError_t tqp_launch_min_fieldnorm(const uint8_t *fieldnorm, uint32_t max_doc, uint32_t *out,
                                    hipStream_t st);
