// tq_device.h — structures shared by the host planner (tq_api.cpp) and the gfx950 kernels
// (tq_kernels.hip).  Layouts are fixed (uploaded as raw bytes).
#pragma once
#include <stdint.h>

#define TQD_TERMINATED 0x7FFFFFFFu
#define TQD_MAX_TERMS 16

// Tunables of the scan kernels (see DESIGN.md "AND kernel").
#define TQD_WAVES_PER_WG 4    // independent wavefronts per workgroup
#define TQD_AND_M 4           // driver-list blocks per tile
#define TQD_AND_CH 4          // leader-list blocks per hash-table fill
#define TQD_AND_SLOTS 1024    // hash slots per wavefront (load factor <= 0.5)
#define TQD_OR_WINDOW 4096    // docs per OR tile (one workgroup)

// One 128-doc block of a posting list, the skip entry of src/postings/skip.rs:205-253 unrolled
// from its sequential form (running byte/position offsets made absolute).
struct TqdBlock {
  uint32_t last_doc;   // last doc id in the block
  uint32_t bits;       // doc_bits | strict<<6 | tf_bits<<8 | bm_fieldnorm_id<<16 | bm_tf_code<<24
                       // 0xFFFFFFFF => the vint tail (pre-decoded in tail_docs/tail_tfs)
  uint64_t byte_off;   // absolute offset of the bitpacked doc payload inside the .idx sub-file
};

struct TqdTerm {
  const TqdBlock *blocks;     // n_blocks entries (full blocks, then the tail pseudo-block)
  const uint32_t *tail_docs;  // n_tail
  const uint32_t *tail_tfs;   // n_tail
  const uint64_t *block_pos;  // n_blocks+1: index (in positions) of the first position of a block
  // positions stream (src/positions/reader.rs): per position-block absolute byte offset / width
  const uint64_t *pos_block_off;  // n_pos_blocks
  const uint8_t *pos_widths;      // n_pos_blocks
  const uint32_t *pos_tail;       // vint tail, pre-decoded deltas
  uint32_t n_full, n_tail, n_blocks, doc_freq;
  uint32_t n_pos_blocks, n_pos_tail;
  uint32_t has_freq;  // 0 => every tf reads as 1
  uint32_t max_bm_tf_code;
};

struct TqdQuery {
  uint32_t term[TQD_MAX_TERMS];  // term handles in execution order
  float weight[TQD_MAX_TERMS];
  uint32_t phrase_off[TQD_MAX_TERMS];  // max_offset - term_offset (phrase_scorer.rs:372-385)
  uint32_t n_terms;
  uint32_t mode;
  uint32_t k;
  uint32_t cache_idx;   // which 256-float tf cache
  uint32_t tile_start;  // first global tile of this query
  uint32_t n_tiles;
  uint32_t part_start;  // first partial top-k list of this query
  uint32_t n_parts;
};

struct TqdSegment {
  const uint8_t *idx;        // .idx sub-file (8-byte header included), padded
  const uint8_t *pos;        // .pos sub-file or null
  const uint8_t *fieldnorm;  // max_doc bytes or null
  uint32_t max_doc;
  uint32_t const_fieldnorm_id;
};
