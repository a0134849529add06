// term_info_store.cpp — see term_info_store.hpp.  File:line references are to
// src/termdict/fst_termdict/term_info_store.rs unless noted.
#include "term_info_store.hpp"

#include <algorithm>
#include <cstring>

namespace tantivy_amd {
namespace {

constexpr size_t kTermInfoBytes = 3 * 4 + 2 * 8;          // postings/term_info.rs:37
constexpr size_t kBlockMetaBytes = 8 + kTermInfoBytes + 3;  // :51-53

uint64_t rd(const uint8_t *p, size_t n) {
  uint64_t v = 0;
  for (size_t i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i);
  return v;
}
void wr(std::vector<uint8_t> &out, uint64_t v, size_t n) {
  for (size_t i = 0; i < n; ++i) out.push_back((uint8_t)(v >> (8 * i)));
}
// bitpacker/src/lib.rs:34-37
uint8_t compute_num_bits(uint64_t n) {
  uint8_t a = 0;
  while (n) {
    ++a;
    n >>= 1;
  }
  return a <= 56 ? a : 64;
}
// :108-128 — bits [addr, addr + num_bits) of a little-endian bit stream, zero padded at the end
uint64_t extract_bits(const uint8_t *data, size_t len, size_t addr_bits, uint8_t num_bits) {
  if (num_bits > 56) throw TantivyError(TantivyError::DataCorruption, "term info wider than 56 bits");
  const size_t addr = addr_bits / 8;
  uint8_t buf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (addr < len) std::memcpy(buf, data + addr, std::min<size_t>(8, len - addr));
  const uint64_t v = rd(buf, 8) >> (addr_bits % 8);
  return v & ((1ull << num_bits) - 1);
}
// bitpacker/src/bitpacker.rs:24-58: values appended LSB first, flushed to a byte boundary
struct BitWriter {
  uint64_t buf = 0;
  unsigned n = 0;
  void write(uint64_t val, uint8_t bits, std::vector<uint8_t> &out) {
    if (bits == 0) return;
    if (n + bits > 64) {
      buf |= val << n;  // n < 64 here
      wr(out, buf, 8);
      buf = val >> (64 - n);
      n = n + bits - 64;
    } else {
      buf |= val << n;
      n += bits;
      if (n == 64) {
        wr(out, buf, 8);
        buf = 0;
        n = 0;
      }
    }
  }
  void flush(std::vector<uint8_t> &out) {
    if (n) wr(out, buf, (n + 7) / 8);
    buf = 0;
    n = 0;
  }
};

}  // namespace

void term_dictionary_values(const uint8_t *file, size_t len, size_t *store_off, size_t *store_len) {
  if (!file || len < 16) throw TantivyError(TantivyError::DataCorruption, "term dictionary shorter than its footer");
  if (rd(file + len - 4, 4) != 1)  // DictionaryType::Fst (src/termdict/mod.rs:52-56,82-98)
    throw TantivyError(TantivyError::Unsupported, "only fst term dictionaries are supported");
  if (rd(file + len - 8, 4) != 1)  // FST_VERSION (termdict.rs:19,130-134)
    throw TantivyError(TantivyError::DataCorruption, "unsupported fst dictionary version");
  const uint64_t slen = rd(file + len - 16, 8);
  if (slen > len - 16) throw TantivyError(TantivyError::DataCorruption, "term info store larger than the file");
  *store_off = len - 16 - (size_t)slen;
  *store_len = (size_t)slen;
}

TermInfoStore TermInfoStore::open(const uint8_t *bytes, size_t len) {
  if (!bytes || len < 16) throw TantivyError(TantivyError::DataCorruption, "term info store shorter than its header");
  const uint64_t meta_len = rd(bytes, 8);
  TermInfoStore s;
  s.num_terms_ = (size_t)rd(bytes + 8, 8);
  if (meta_len > len - 16) throw TantivyError(TantivyError::DataCorruption, "term info block metas exceed the store");
  const size_t n_blocks = (s.num_terms_ + TERM_INFO_BLOCK_LEN - 1) / TERM_INFO_BLOCK_LEN;
  if (meta_len < n_blocks * kBlockMetaBytes)
    throw TantivyError(TantivyError::DataCorruption, "too few term info block metas");
  s.metas_.assign(bytes + 16, bytes + 16 + meta_len);
  s.infos_.assign(bytes + 16 + meta_len, bytes + len);
  return s;
}

TermInfo TermInfoStore::get(uint64_t term_ord) const {
  if (term_ord >= num_terms_) throw TantivyError(TantivyError::InvalidArgument, "term ordinal out of range");
  const uint8_t *m = metas_.data() + (term_ord / TERM_INFO_BLOCK_LEN) * kBlockMetaBytes;
  const uint64_t offset = rd(m, 8);
  TermInfo ref;  // TermInfo::deserialize, postings/term_info.rs:50-63
  ref.doc_freq = (uint32_t)rd(m + 8, 4);
  ref.postings_start = rd(m + 12, 8);
  ref.postings_end = ref.postings_start + rd(m + 20, 4);
  ref.positions_start = rd(m + 24, 8);
  ref.positions_end = ref.positions_start + rd(m + 32, 4);
  const uint8_t df_bits = m[36], post_bits = m[37], pos_bits = m[38];
  const size_t inner = term_ord % TERM_INFO_BLOCK_LEN;
  if (inner == 0) return ref;
  if (offset > infos_.size()) throw TantivyError(TantivyError::DataCorruption, "term info block offset");
  // :64-101: entry k holds (postings start, positions start, doc freq) of term k+1 relative to
  // the block's first term; the ends are the next entry's starts
  const uint8_t *data = infos_.data() + offset;
  const size_t dlen = infos_.size() - (size_t)offset;
  const size_t nb = (size_t)df_bits + post_bits + pos_bits, a = nb * (inner - 1);
  TermInfo ti;
  ti.postings_start = ref.postings_start + extract_bits(data, dlen, a, post_bits);
  ti.postings_end = ref.postings_start + extract_bits(data, dlen, a + nb, post_bits);
  ti.positions_start = ref.positions_start + extract_bits(data, dlen, a + post_bits, pos_bits);
  ti.positions_end = ref.positions_start + extract_bits(data, dlen, a + post_bits + nb, pos_bits);
  ti.doc_freq = (uint32_t)extract_bits(data, dlen, a + post_bits + pos_bits, df_bits);
  return ti;
}

void TermInfoStoreWriter::write_term_info(const TermInfo &ti) {  // :263-270
  ++num_terms_;
  block_.push_back(ti);
  if (block_.size() >= TERM_INFO_BLOCK_LEN) flush_block();
}

void TermInfoStoreWriter::flush_block() {  // :208-261
  if (block_.empty()) return;
  const TermInfo ref = block_.front(), last = block_.back();
  const uint64_t post_end = last.postings_end - ref.postings_start;
  const uint64_t pos_end = last.positions_end - ref.positions_start;
  uint32_t max_df = 0;
  for (size_t i = 1; i < block_.size(); ++i) max_df = std::max(max_df, block_[i].doc_freq);
  const uint8_t df_bits = compute_num_bits(max_df);
  const uint8_t post_bits = compute_num_bits(post_end), pos_bits = compute_num_bits(pos_end);
  if (post_bits > 56 || pos_bits > 56)
    throw TantivyError(TantivyError::InvalidArgument, "term info offsets beyond 2^56");
  wr(metas_, infos_.size(), 8);
  wr(metas_, ref.doc_freq, 4);  // TermInfo::serialize, postings/term_info.rs:41-48
  wr(metas_, ref.postings_start, 8);
  wr(metas_, ref.postings_end - ref.postings_start, 4);
  wr(metas_, ref.positions_start, 8);
  wr(metas_, ref.positions_end - ref.positions_start, 4);
  metas_.push_back(df_bits);
  metas_.push_back(post_bits);
  metas_.push_back(pos_bits);
  BitWriter bw;
  for (size_t i = 1; i < block_.size(); ++i) {
    bw.write(block_[i].postings_start - ref.postings_start, post_bits, infos_);
    bw.write(block_[i].positions_start - ref.positions_start, pos_bits, infos_);
    bw.write(block_[i].doc_freq, df_bits, infos_);
  }
  bw.write(post_end, post_bits, infos_);  // the ends of the last term
  bw.write(pos_end, pos_bits, infos_);
  bw.flush(infos_);
  block_.clear();
}

void TermInfoStoreWriter::serialize(std::vector<uint8_t> &out) {  // :272-283
  flush_block();
  wr(out, metas_.size(), 8);
  wr(out, num_terms_, 8);
  out.insert(out.end(), metas_.begin(), metas_.end());
  out.insert(out.end(), infos_.begin(), infos_.end());
}

}  // namespace tantivy_amd
