// bm25.hpp — host-side mirror of tantivy's Bm25Weight (src/query/bm25.rs) and the fieldnorm
// code table (src/fieldnorm/code.rs).  Product code: the weights and the 256-entry tf cache
// handed to the device in tq_query are computed here, in f32, exactly as the reference does.
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace tantivy_amd {

using Score = float;
using DocId = uint32_t;
constexpr DocId TERMINATED = 0x7FFFFFFFu;  // src/docset.rs:12

// src/fieldnorm/code.rs: id -> fieldnorm.  Ids < 24 are exact; above that a 3-bit mantissa /
// 5-bit exponent code (the generating rule the reference checks its literal table against).
inline uint32_t id_to_fieldnorm(uint8_t id) {
  if (id < 24) return id;
  const uint8_t b = (uint8_t)(id - 24);
  const uint32_t bits = b & 7u, shift = b >> 3;
  const uint32_t exp_part = shift == 0 ? bits : ((bits | 8u) << (shift - 1u));
  return 24u + exp_part;
}
// code.rs:7-11: largest id whose fieldnorm <= value
inline uint8_t fieldnorm_to_id(uint32_t fieldnorm) {
  int lo = 0, hi = 256;
  while (lo < hi) {
    const int mid = (lo + hi) / 2;
    if (id_to_fieldnorm((uint8_t)mid) <= fieldnorm)
      lo = mid + 1;
    else
      hi = mid;
  }
  return (uint8_t)(lo - 1);
}

// bm25.rs:8-9
constexpr Score K1 = 1.2f;
constexpr Score B = 0.75f;

// bm25.rs:52-56
inline Score idf(uint64_t doc_freq, uint64_t doc_count) {
  // the reference asserts doc_count >= doc_freq (bm25.rs:53); inconsistent (e.g. remote)
  // statistics must not wrap the subtraction
  if (doc_freq > doc_count) throw std::invalid_argument("idf: doc_freq > doc_count");
  const Score x = ((Score)(doc_count - doc_freq) + 0.5f) / ((Score)doc_freq + 0.5f);
  return std::log(1.0f + x);  // f32 ln, as Rust's f32::ln
}

struct Bm25Weight {
  Score weight = 0.0f;
  Score cache[256];
  Score average_fieldnorm = 0.0f;

  // bm25.rs:158-166 (new) + :62-69 (compute_tf_cache)
  static Bm25Weight from_idf(Score idf_value, Score average_fieldnorm) {
    Bm25Weight w;
    w.weight = idf_value * (1.0f + K1);
    w.average_fieldnorm = average_fieldnorm;
    for (int id = 0; id < 256; ++id)
      w.cache[id] = K1 * (1.0f - B + B * (Score)id_to_fieldnorm((uint8_t)id) / average_fieldnorm);
    return w;
  }
  // bm25.rs:132-146
  static Bm25Weight for_one_term(uint64_t term_doc_freq, uint64_t total_num_docs,
                                 Score avg_fieldnorm) {
    return from_idf(idf(term_doc_freq, total_num_docs), avg_fieldnorm);
  }
  // bm25.rs:95-129: one term => its idf; several (a phrase) => idf summed in term order
  static Bm25Weight for_terms(const std::vector<uint64_t> &term_doc_freqs,
                              uint64_t total_num_docs, uint64_t total_num_tokens) {
    const Score avg = (Score)total_num_tokens / (Score)total_num_docs;
    if (term_doc_freqs.size() == 1) return for_one_term(term_doc_freqs[0], total_num_docs, avg);
    Score idf_sum = 0.0f;
    for (uint64_t df : term_doc_freqs) idf_sum += idf(df, total_num_docs);
    return from_idf(idf_sum, avg);
  }
  // bm25.rs:77-88
  Bm25Weight boost_by(Score boost) const {
    Bm25Weight w = *this;
    if (boost != 1.0f) w.weight = weight * boost;
    return w;
  }
  Score tf_factor(uint8_t fieldnorm_id, uint32_t term_freq) const {
    const Score tf = (Score)term_freq;
    return tf / (tf + cache[fieldnorm_id]);
  }
  Score score(uint8_t fieldnorm_id, uint32_t term_freq) const {
    return weight * tf_factor(fieldnorm_id, term_freq);
  }
  Score max_score() const { return score(255u, 2013265944u); }
};

}  // namespace tantivy_amd
