// term_info_store.hpp — host-side mirror of tantivy's TermInfoStore (the value half of the term
// dictionary, src/termdict/fst_termdict/term_info_store.rs): term ordinal -> TermInfo in O(1), and
// the writer that segment finalisation needs next to the device-side codec writers
// (SURVEY.md §8f.3).  The FST that maps term bytes to ordinals stays with tantivy-fst.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "searcher.hpp"

namespace tantivy_amd {

constexpr size_t TERM_INFO_BLOCK_LEN = 256;  // term_info_store.rs:12

// src/termdict/fst_termdict/termdict.rs:123-140: a field's term dictionary file is
// fst | term info store | u64 store_len | u32 fst version | u32 dictionary type (1 = Fst,
// src/termdict/mod.rs:82-98); returns the store's byte range.
void term_dictionary_values(const uint8_t *file, size_t len, size_t *store_off, size_t *store_len);

class TermInfoStore {
 public:
  // TermInfoStore::open (:131-143): u64 block-meta bytes | u64 num_terms | metas | bitpacked infos.
  // Keeps a copy of the bytes.
  static TermInfoStore open(const uint8_t *bytes, size_t len);
  TermInfo get(uint64_t term_ord) const;  // :145-159
  size_t num_terms() const { return num_terms_; }

 private:
  size_t num_terms_ = 0;
  std::vector<uint8_t> metas_, infos_;
};

class TermInfoStoreWriter {  // :166-290
 public:
  void write_term_info(const TermInfo &ti);
  void serialize(std::vector<uint8_t> &out);

 private:
  void flush_block();
  std::vector<uint8_t> metas_, infos_;
  std::vector<TermInfo> block_;
  uint64_t num_terms_ = 0;
};

}  // namespace tantivy_amd
