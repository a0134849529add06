// tq_plan_misc.cpp — build_dense_plan (unpruned unions, doc-major: rows and queries of tq_xunion.hip) and
// plan_bool_query (clause layout of the union kernel's boolean instantiation)
// Part of the C ABI library of include/tantivy_amd.h (internal declarations: tq_internal.hpp).
#include "tq_internal.hpp"

namespace tqi {

// The doc-major union group (tq_xunion.hip): rows = the distinct (list, weight) pairs of its queries, those with a
// bitmap first; tasks = runs of 128-doc tiles handed out by an atomic counter to one workgroup per
// CU; every query gets a result list of grid * k entries (a workgroup appends at most k).
int build_dense_plan(tq_segment *s, Group &g, PlanScratch &ps, uint32_t cus) {
  const uint32_t n_rows = (uint32_t)ps.xrow_term.size();
  std::vector<uint32_t> new_row(n_rows, 0u);
  ps.xrows.assign(n_rows, TqkDenseRow{});
  uint32_t n_a = 0;
  for (int pass = 0; pass < 2; ++pass)  // bitmap rows, then the others; first use order inside each
    for (uint32_t r = 0, at = pass ? n_a : 0u; r < n_rows; ++r) {
      const uint32_t h = (uint32_t)(ps.xrow_term[r] >> 32);
      const TermHost &th = s->terms[h];
      const bool bitmap = th.dense_blob && th.tf8_blob;
      if (bitmap != (pass == 0)) continue;
      TqkDenseRow row{};
      row.handle = h;
      row.doc_freq = th.doc_freq;
      {
        const uint32_t wb = (uint32_t)ps.xrow_term[r];
        memcpy(&row.w, &wb, sizeof wb);
      }
      if (bitmap) {
        row.dense = (const uint2 *)th.dense_blob;
        row.tf8 = (const uint8_t *)th.tf8_blob;
        ++n_a;
      } else {
        const size_t doc_bytes = ((size_t)th.doc_freq * sizeof(uint32_t) + 15) & ~(size_t)15;
        row.flat_docs = (const uint32_t *)th.flat_blob;
        row.tf8 = (const uint8_t *)th.flat_blob + doc_bytes;
      }
      new_row[r] = at;
      ps.xrows[at++] = row;
    }
  ps.x_bitmap_rows = n_a;
  const uint32_t n_tiles = (s->max_doc + TQK_XU_TILE - 1) / TQK_XU_TILE;
  static const uint32_t kTaskDiv = std::max<uint32_t>(1u, tune_u32("TQ_XU_TASKS_PER_CU", 16));
  ps.x_tiles_per_task = std::min<uint32_t>(32u, std::max<uint32_t>(1u, n_tiles / (cus * kTaskDiv)));
  const uint32_t n_tasks = (n_tiles + ps.x_tiles_per_task - 1) / ps.x_tiles_per_task;
  ps.xgrid = std::min<uint32_t>(n_tasks, cus);
  g.kpl = g.max_k <= 64 ? 1 : 2;
  {  // a staging list per (workgroup, query): fewer workgroups rather than gigabytes of staging (TQ_XU_STAGE_MB)
    static const uint64_t kStageBudget = (uint64_t)std::max<uint32_t>(1u, tune_u32("TQ_XU_STAGE_MB", 1024)) << 20;
    const uint64_t per_wg = (uint64_t)g.queries.size() * (uint64_t)(g.kpl + 1) * 64u * sizeof(uint64_t);
    ps.xgrid = (uint32_t)std::max<uint64_t>(1u, std::min<uint64_t>(ps.xgrid, kStageBudget / std::max<uint64_t>(1u, per_wg)));
  }
  g.n_chunks = n_tasks;
  g.total_tiles = n_tiles;
  ps.x_list_stride = ps.xgrid * g.max_k;
  ps.xqueries.assign(g.queries.size(), TqkDenseQuery{});
  ps.x_max_terms = 1;
  for (size_t qi = 0; qi < g.queries.size(); ++qi) {
    TqdQuery &dq = g.queries[qi];
    TqkDenseQuery &xq = ps.xqueries[qi];
    for (uint32_t i = 0; i < 8u; ++i) {  // (beyond n_terms: the all-zero row)
      const uint32_t row = i < dq.n_terms ? new_row[ps.xrow_of[xrow_key(dq.term[i], dq.weight[i])]] : n_rows;
      (i < 4 ? xq.rows_lo : xq.rows_hi) |= row << (8u * (i & 3u));
    }
    ps.x_max_terms = std::max(ps.x_max_terms, dq.n_terms);
    xq.nt_k = dq.n_terms | (dq.k << 8);
    xq.thr_row = dq.thr_index;
    dq.part_start = (uint32_t)(qi * ps.x_list_stride);
    dq.n_parts = ps.x_list_stride;
    dq.chunk_first = (uint32_t)qi;  // (merge_lists_kernel: the query whose list_count word counts this list)
  }
  return TQ_OK;
}

// Planning of one TQ_MODE_BOOL query: clause layout of the union kernel (tq_union.hip), pruning
// flags, tile sizes.  An empty result leaves dq.n_terms == 0 and n_tiles == 0.
int plan_bool_query(tq_segment *s, const tq_query &q, uint32_t qi, TqdQuery &dq, uint64_t &qbytes,
                    uint32_t &n_tiles, uint32_t &tile_cost, uint32_t &n_thr_rows, bool exhaustive) {
  // BooleanQuery whose clauses are terms or unions of terms (`+a b -c`, `+a +(b OR c)`),
  // BooleanWeight::complex_scorer (boolean_weight.rs:236-431).  A clause = the terms sharing
  // one clause_of value.  Absent terms are EmptyScorers: they drop out of unions, an empty
  // Must clause empties the query (:249-251), empty Should / MustNot clauses are removed.
  struct Clause {
    uint8_t occur;
    uint32_t id, n = 0, terms[TQ_MAX_TERMS];
    uint64_t cost = 0;  // BufferedUnionScorer::cost = sum of the lists' costs (doc freqs)
  };
  Clause cl[TQ_MAX_TERMS];
  uint32_t n_cl = 0;
  bool empty = false;
  for (uint32_t i = 0; i < q.n_terms; ++i) {
    if (q.occurs[i] > TQ_MUST_NOT) return fail(TQ_ERR_INVALID, "query %u: bad occur", qi);
    const uint32_t id = q.clause_of ? q.clause_of[i] : i;
    uint32_t c = 0;
    while (c < n_cl && cl[c].id != id) ++c;
    if (c == n_cl) {
      cl[n_cl].id = id;
      cl[n_cl].occur = q.occurs[i];
      ++n_cl;
    } else if (cl[c].occur != q.occurs[i]) {
      return fail(TQ_ERR_INVALID, "query %u: clause %u mixes occurs", qi, id);
    }
    if (q.terms[i] == TQ_TERM_ABSENT) continue;
    cl[c].terms[cl[c].n++] = i;
    cl[c].cost += s->terms[q.terms[i]].doc_freq;
  }
  uint32_t must[TQ_MAX_TERMS], should[TQ_MAX_TERMS], mustnot[TQ_MAX_TERMS];
  uint32_t n_must = 0, n_should = 0, n_not = 0;
  for (uint32_t c = 0; c < n_cl; ++c) {
    if (cl[c].occur == TQ_MUST) {
      if (cl[c].n == 0) empty = true;
      must[n_must++] = c;
    } else if (cl[c].n) {
      if (cl[c].occur == TQ_SHOULD)
        should[n_should++] = c;
      else
        mustnot[n_not++] = c;
    }
  }
  // minimum_number_should_match (:272-305): more than there are Should clauses matches
  // nothing; all of them turns them into Must clauses; 1 makes the union required
  uint32_t msm = q.min_should_match;
  if (msm > n_should) empty = true;
  if (!empty && msm >= 2 && msm == n_should) {
    for (uint32_t i = 0; i < n_should; ++i) must[n_must++] = should[i];
    n_should = 0;
    msm = 0;
  }
  if (msm >= 2)
    for (uint32_t i = 0; i < n_should; ++i)
      if (cl[should[i]].n > 1)
        return fail(TQ_ERR_UNSUPPORTED,
                    "query %u: min_should_match > 1 over nested unions stays on the CPU", qi);
  // MustNot clauses only: no include scorer, EmptyScorer (boolean_weight.rs:340-349)
  if (n_must == 0 && n_should == 0) empty = true;
  if (!empty) {
    uint32_t n = 0;
    auto put = [&](uint32_t i, uint32_t role) {
      dq.term[n] = q.terms[i];
      dq.weight[n] = role == TQD_ROLE_MUST_NOT ? 0.0f : q.weights[i];
      dq.roles |= role << (2u * n);
      qbytes += s->terms[q.terms[i]].postings_len;
      ++n;
    };
    auto put_by_weight = [&](uint32_t *idx, uint32_t cnt, uint32_t role) {
      small_stable_sort(idx, idx + cnt,
                       [&](uint32_t a, uint32_t b) { return q.weights[a] > q.weights[b]; });
      for (uint32_t i = 0; i < cnt; ++i) put(idx[i], role);
    };
    uint32_t flat[TQ_MAX_TERMS], n_flat = 0;
    if (n_must) {
      // Must clauses by cost ascending (intersect_scorers, intersection.rs:31): the cheapest
      // leads; then the MustNot terms (they only exclude: densest first), then the Should
      // terms in clause order
      small_stable_sort(must, must + n_must,
                       [&](uint32_t a, uint32_t b) { return cl[a].cost < cl[b].cost; });
      // optional Should lists lead too (MaxScore for RequiredOptionalScorer, see union_body)
      // (only when pruning: with every match scored the extra ownership probes cost 60 %)
      static const bool kOptLead = tune_u32("TQ_BOOL_OPT_LEAD", 1) != 0;
      const bool opt_lead = kOptLead && n_should > 0 && msm == 0 && !exhaustive;
      if (opt_lead) {
        for (uint32_t c = 0; c < n_should; ++c)
          for (uint32_t i = 0; i < cl[should[c]].n; ++i) flat[n_flat++] = cl[should[c]].terms[i];
        put_by_weight(flat, n_flat, TQD_ROLE_SHOULD);
        dq.n_opt_lead = n;
        n_flat = 0;
        n_should = 0;
      }
      Clause &lead = cl[must[0]];
      put_by_weight(lead.terms, lead.n, TQD_ROLE_MUST);
      dq.n_lead = n;
      for (uint32_t c = 1; c < n_must; ++c) {
        for (uint32_t i = 0; i < cl[must[c]].n; ++i) put(cl[must[c]].terms[i], TQD_ROLE_MUST);
        dq.clause_end |= 1u << (n - 1u);
      }
      for (uint32_t c = 0; c < n_not; ++c)
        for (uint32_t i = 0; i < cl[mustnot[c]].n; ++i) flat[n_flat++] = cl[mustnot[c]].terms[i];
      small_stable_sort(flat, flat + n_flat, [&](uint32_t a, uint32_t b) {
        return s->terms[q.terms[a]].doc_freq > s->terms[q.terms[b]].doc_freq;
      });
      for (uint32_t i = 0; i < n_flat; ++i) put(flat[i], TQD_ROLE_MUST_NOT);
      for (uint32_t c = 0; c < n_should; ++c)
        for (uint32_t i = 0; i < cl[should[c]].n; ++i) put(cl[should[c]].terms[i], TQD_ROLE_SHOULD);
    } else {
      // no Must: the Should terms form the leading union (by weight descending, as the pure
      // union does), MustNot terms exclude
      for (uint32_t c = 0; c < n_should; ++c)
        for (uint32_t i = 0; i < cl[should[c]].n; ++i) flat[n_flat++] = cl[should[c]].terms[i];
      put_by_weight(flat, n_flat, TQD_ROLE_SHOULD);
      dq.n_lead = n;
      for (uint32_t c = 0; c < n_not; ++c)
        for (uint32_t i = 0; i < cl[mustnot[c]].n; ++i) put(cl[mustnot[c]].terms[i], TQD_ROLE_MUST_NOT);
    }
    dq.min_should = msm;
    dq.n_terms = n;
    bool nonneg = true;
    uint32_t sparse = 0;
    for (uint32_t i = 0; i < n; ++i) {
      nonneg = nonneg && dq.weight[i] >= 0.0f;
      if (!(s->terms[dq.term[i]].dense_blob && s->opt.use_dense)) ++sparse;
    }
    if (!exhaustive && nonneg) {
      dq.flags |= TQD_QF_PRUNE;
      if (q.k <= 2 * TQD_THR_SLOTS) {
        dq.thr_index = n_thr_rows;
        n_thr_rows += 4u;  // union kernel: 64 slots for k <= 16, 256 above; the window kernel 64 / 128
      }
    }
    const uint32_t c_lb = 1u + n + 8u * sparse;
    static const uint32_t kBoolTileNum = std::max<uint32_t>(1u, tune_u32("TQ_BOOL_TILE_NUM", TQD_AND_TILE * 2u));
    dq.tile_blocks = std::min<uint32_t>(TQD_AND_TILE, std::max<uint32_t>(1u, kBoolTileNum / c_lb));
    tile_cost = dq.tile_blocks * c_lb;
    uint32_t acc_tiles = 0;
    for (uint32_t i = 0; i <= TQ_MAX_TERMS; ++i) {
      dq.lead_tile_start[i] = acc_tiles;
      if (i < dq.n_lead)
        acc_tiles += (s->terms[dq.term[i]].n_blocks + dq.tile_blocks - 1) / dq.tile_blocks;
    }
    n_tiles = acc_tiles;
  }
  return TQ_OK;
}


}  // namespace tqi

namespace tqi {

// Does a TQ_MODE_BOOL query need the nested-tree evaluator (tq_tree.hip)?  A clause of several terms whose terms
// are not all Should inside it (an intersection / exclusion / optional terms inside the nested query), a nested
// minimum_number_should_match above 1, or a top-level minimum above 1 next to a nested clause.
bool bool_query_is_tree(const tq_query &q) {
  if (q.mode != TQ_MODE_BOOL || !q.occurs || !q.terms) return false;
  if (q.nested_occurs)  // a PhraseQuery inside the boolean query
    for (uint32_t i = 0; i < q.n_terms && i < TQ_MAX_TERMS; ++i)
      if (q.nested_occurs[i] != 255u && (q.nested_occurs[i] & (TQ_NESTED_PHRASE | TQ_NESTED_ANY))) return true;
  uint32_t size_of[TQ_MAX_TERMS] = {0};
  bool multi = false;
  for (uint32_t i = 0; i < q.n_terms && i < TQ_MAX_TERMS; ++i) {
    const uint32_t id = q.clause_of ? q.clause_of[i] : i;
    if (id >= TQ_MAX_TERMS) return false;  // (reported by the planners)
    multi = multi || ++size_of[id] > 1;
  }
  if (!multi) {  // every clause is one term: a nested occur other than Must / Should makes it a tree (a lone MustNot matches nothing)
    if (q.nested_occurs)
      for (uint32_t i = 0; i < q.n_terms; ++i)
        if (q.nested_occurs[i] == TQ_MUST_NOT) return true;
    return false;
  }
  for (uint32_t i = 0; i < q.n_terms; ++i) {
    const uint32_t id = q.clause_of ? q.clause_of[i] : i;
    if (q.nested_occurs && q.nested_occurs[i] != TQ_SHOULD && (size_of[id] > 1 || q.nested_occurs[i] == TQ_MUST_NOT)) return true;
    if (q.clause_min_should && q.clause_min_should[id] >= 2) return true;
  }
  if (q.atom_of)  // a conjunction of several terms inside a clause
    for (uint32_t i = 0; i < q.n_terms; ++i)
      for (uint32_t j = 0; j < i; ++j)
        if (q.atom_of[i] == q.atom_of[j] && (q.clause_of ? q.clause_of[i] == q.clause_of[j] : false)) return true;
  // "at least m of n" over nested unions (disjunction.rs:113-139): what plan_bool_query does not take — m == n turns
  // the Should clauses into Must clauses there, m > n matches nothing
  if (q.min_should_match >= 2) {
    uint32_t present[TQ_MAX_TERMS] = {0}, n_should = 0;
    bool multi_should = false;
    for (uint32_t i = 0; i < q.n_terms; ++i) {
      if (q.occurs[i] != TQ_SHOULD || q.terms[i] == TQ_TERM_ABSENT) continue;
      const uint32_t id = q.clause_of ? q.clause_of[i] : i;
      if (++present[id] == 1u) ++n_should;
      multi_should = multi_should || present[id] > 1u;
    }
    return multi_should && q.min_should_match < n_should;
  }
  return false;
}

// One nested boolean query -> its descriptor for tq_tree.hip.  BooleanWeight::complex_scorer on every level
// (boolean_weight.rs:236-431): absent terms are EmptyScorers — an absent term empties its conjunction, an empty Must
// member empties its (nested) query, empty Should / MustNot members are removed; minimum_number_should_match above the
// number of Should scorers left matches nothing; a nested query that cannot match empties the query when it is a Must
// clause and is removed otherwise; a query without Must and Should clauses matches nothing (:340-349).  Clause order =
// score-sum order: Must clauses by cost ascending (intersect_scorers, intersection.rs:31: an intersection costs its
// cheapest member, a union the sum of its members), then Should, then MustNot; inside a clause Must members by cost
// ascending, then Should, then MustNot; inside a conjunction terms by doc freq ascending.  tq.n_terms == 0: the
// query matches nothing.
int plan_tree_query(tq_segment *s, const tq_query &q, uint32_t qi, TqdTreeQuery &tq, uint64_t &qbytes, uint64_t table_base) {
  tq = TqdTreeQuery{};
  tq.k = q.k;
  struct Atom {  // a member of a nested query: one term, a conjunction of terms, or a phrase
    uint32_t id, inner, n = 0, handle[TQ_MAX_TERMS], off[TQ_MAX_TERMS];
    float w[TQ_MAX_TERMS];
    bool empty = false, phrase = false;
    bool any = false;       // a UNION of its terms (TQ_NESTED_ANY) instead of a conjunction
    uint32_t n_named = 0;   // terms the caller named (absent ones included)
    uint64_t cost = ~0ull;  // its rarest list; a phrase: PhraseScorer::cost (phrase_scorer.rs:566-573)
  };
  struct Clause {
    uint32_t id, outer, n = 0, msm = 0;
    uint32_t atom[TQ_MAX_TERMS];  // indices into atoms[]
    bool empty = false;
    uint64_t cost = 0;
  };
  Atom atoms[TQ_MAX_TERMS];
  Clause cl[TQ_MAX_TERMS];
  uint32_t n_cl = 0, n_at = 0;
  for (uint32_t i = 0; i < q.n_terms; ++i) {
    if (q.occurs[i] > TQ_MUST_NOT) return fail(TQ_ERR_INVALID, "query %u: bad occur", qi);
    uint32_t inner = q.nested_occurs ? q.nested_occurs[i] : (uint32_t)TQ_SHOULD;
    const bool in_phrase = inner != 255u && (inner & TQ_NESTED_PHRASE);
    const bool in_any = inner != 255u && (inner & TQ_NESTED_ANY);
    if (in_phrase && in_any) return fail(TQ_ERR_INVALID, "query %u: a member is a phrase or a union, not both", qi);
    if (inner != 255u) inner &= ~(uint32_t)(TQ_NESTED_PHRASE | TQ_NESTED_ANY);
    if (in_any && !q.atom_of) return fail(TQ_ERR_INVALID, "query %u: a union inside a nested boolean query needs atom_of", qi);
    if (inner > TQ_MUST_NOT) return fail(TQ_ERR_INVALID, "query %u: bad nested occur", qi);
    if (in_phrase && (!q.atom_of || !q.phrase_offsets))
      return fail(TQ_ERR_INVALID, "query %u: a phrase inside a boolean query needs atom_of and phrase_offsets", qi);
    const uint32_t id = q.clause_of ? q.clause_of[i] : i;
    if (id >= TQ_MAX_TERMS) return fail(TQ_ERR_INVALID, "query %u: clause_of value %u above %u", qi, id, TQ_MAX_TERMS - 1u);
    uint32_t c = 0;
    while (c < n_cl && cl[c].id != id) ++c;
    if (c == n_cl) {
      cl[n_cl].id = id;
      cl[n_cl].outer = q.occurs[i];
      cl[n_cl].msm = q.clause_min_should ? q.clause_min_should[id] : 0u;
      ++n_cl;
    } else if (cl[c].outer != q.occurs[i]) {
      return fail(TQ_ERR_INVALID, "query %u: clause %u mixes occurs", qi, id);
    }
    // the term's atom inside its clause (atom_of NULL: every term its own)
    const uint32_t aid = q.atom_of ? (uint32_t)q.atom_of[i] : 0x100u + i;
    uint32_t a = TQ_MAX_TERMS;
    for (uint32_t x = 0; x < cl[c].n; ++x)
      if (atoms[cl[c].atom[x]].id == aid) a = cl[c].atom[x];
    if (a == TQ_MAX_TERMS) {
      a = n_at++;
      atoms[a].id = aid;
      atoms[a].inner = inner;
      atoms[a].phrase = in_phrase;
      atoms[a].any = in_any;
      cl[c].atom[cl[c].n++] = a;
    } else if (atoms[a].inner != inner || atoms[a].phrase != in_phrase || atoms[a].any != in_any) {
      return fail(TQ_ERR_INVALID, "query %u: a conjunction (atom_of %u) mixes nested occurs", qi, aid);
    }
    ++atoms[a].n_named;
    if (q.terms[i] == TQ_TERM_ABSENT) {
      if (!in_any) atoms[a].empty = true;  // an EmptyScorer inside the conjunction (a union just loses the term)
      continue;
    }
    if (q.terms[i] >= s->terms.size()) return fail(TQ_ERR_INVALID, "query %u: unknown term handle %u", qi, q.terms[i]);
    if (!(q.weights[i] >= 0.0f)) return fail(TQ_ERR_UNSUPPORTED, "query %u: negative boost inside a nested boolean query", qi);
    if (in_phrase && s->terms[q.terms[i]].positions_len == 0)
      return fail(TQ_ERR_UNSUPPORTED, "query %u: phrase on a field without positions", qi);
    atoms[a].handle[atoms[a].n] = q.terms[i];
    atoms[a].off[atoms[a].n] = in_phrase ? q.phrase_offsets[i] : 0u;
    atoms[a].w[atoms[a].n++] = q.weights[i];
    if (in_any)  // BufferedUnionScorer::cost = the sum of its scorers' (buffered_union.rs:326-328)
      atoms[a].cost = (atoms[a].cost == ~0ull ? 0ull : atoms[a].cost) + s->terms[q.terms[i]].doc_freq;
    else
      atoms[a].cost = std::min<uint64_t>(atoms[a].cost, s->terms[q.terms[i]].doc_freq);
    qbytes += s->terms[q.terms[i]].postings_len;
  }
  for (uint32_t a = 0; a < n_at; ++a) {  // conjunctions: rarest list first (Intersection::score sums in that order)
    Atom &A = atoms[a];
    if (A.phrase) {
      // PhraseQuery::new needs two terms (phrase_query.rs:51-55); one cursor per term in registers
      if (A.n_named < 2) return fail(TQ_ERR_INVALID, "query %u: a phrase needs >= 2 terms", qi);
      if (A.n_named > TQK_TREE_PHRASE_TERMS)
        return fail(TQ_ERR_UNSUPPORTED, "query %u: a phrase inside a boolean query takes at most %u terms", qi, TQK_TREE_PHRASE_TERMS);
      if (!A.empty) {
        // PhraseScorer::cost = size_hint of the intersection of its lists * 10 * terms (phrase_scorer.rs:566-573;
        // estimate_intersection, size_hint.rs:11-36: co-location factor 1.3 shrinking by 0.1 per list)
        double est = 0.0, smallest = 0.0, f = 1.3;
        for (uint32_t i = 0; i < A.n; ++i) {
          const double df = (double)s->terms[A.handle[i]].doc_freq;
          if (i == 0) {
            est = smallest = df;
          } else {
            f = std::max(1.0, f - 0.1);
            est *= df / (double)std::max<uint32_t>(1u, s->max_doc) * f;
            smallest = std::min(smallest, df);
          }
        }
        A.cost = (uint64_t)std::min(std::round(est), smallest) * 10u * A.n;
      }
      continue;  // (the terms keep the caller's order: their offsets say where they stand)
    }
    for (uint32_t i = 1; i < A.n && !A.any; ++i)
      for (uint32_t j = i; j > 0 && s->terms[A.handle[j]].doc_freq < s->terms[A.handle[j - 1]].doc_freq; --j) {
        std::swap(A.handle[j], A.handle[j - 1]);
        std::swap(A.w[j], A.w[j - 1]);
      }
    if (A.n == 0) A.empty = true;
  }
  auto rank_of = [](uint32_t occur) { return occur == TQ_MUST ? 0u : (occur == TQ_SHOULD ? 1u : 2u); };
  // every clause on its own level
  for (uint32_t c = 0; c < n_cl; ++c) {
    Clause &C = cl[c];
    uint32_t keep = 0;
    for (uint32_t x = 0; x < C.n; ++x) {  // empty members: a Must one empties the clause, the others are removed
      const Atom &A = atoms[C.atom[x]];
      if (A.empty) {
        if (A.inner == TQ_MUST) C.empty = true;
        continue;
      }
      C.atom[keep++] = C.atom[x];
    }
    C.n = keep;
    std::stable_sort(C.atom, C.atom + C.n, [&](uint32_t x, uint32_t y) {
      const uint32_t rx = rank_of(atoms[x].inner), ry = rank_of(atoms[y].inner);
      if (rx != ry) return rx < ry;
      return rx == 0u && atoms[x].cost < atoms[y].cost;
    });
    uint32_t n_m = 0, n_s = 0;
    uint64_t min_must = ~0ull, sum_should = 0;
    for (uint32_t x = 0; x < C.n; ++x) {
      const Atom &A = atoms[C.atom[x]];
      if (A.inner == TQ_MUST) {
        ++n_m;
        min_must = std::min(min_must, A.cost);
      } else if (A.inner == TQ_SHOULD) {
        ++n_s;
        sum_should += A.cost;
      }
    }
    if (C.msm > n_s || (n_m == 0 && n_s == 0)) C.empty = true;
    C.msm = n_m ? C.msm : std::max<uint32_t>(1u, C.msm);  // without a Must member the union of the Should members is the doc set
    C.cost = n_m ? min_must : sum_should;
  }
  uint32_t order[TQ_MAX_TERMS], n_ord = 0, n_must = 0, n_should = 0;
  for (uint32_t pass = 0; pass < 3; ++pass)
    for (uint32_t c = 0; c < n_cl; ++c) {
      const uint32_t want = pass == 0 ? (uint32_t)TQ_MUST : (pass == 1 ? (uint32_t)TQ_SHOULD : (uint32_t)TQ_MUST_NOT);
      if (cl[c].outer != want) continue;
      if (cl[c].empty) {
        if (want == TQ_MUST) return TQ_OK;  // (tq.n_terms == 0: nothing matches)
        continue;
      }
      order[n_ord++] = c;
      n_must += want == TQ_MUST ? 1u : 0u;
      n_should += want == TQ_SHOULD ? 1u : 0u;
    }
  std::stable_sort(order, order + n_must, [&](uint32_t a, uint32_t b) { return cl[a].cost < cl[b].cost; });
  if (q.min_should_match > n_should || (n_must == 0 && n_should == 0)) return TQ_OK;
  tq.top_has_must = n_must ? 1u : 0u;
  tq.top_need = n_must ? q.min_should_match : std::max<uint32_t>(1u, q.min_should_match);
  if (tq.top_need > 15u) return fail(TQ_ERR_UNSUPPORTED, "query %u: minimum_number_should_match above 15", qi);
  uint32_t n = 0;
  for (uint32_t o = 0; o < n_ord; ++o) {
    const Clause &C = cl[order[o]];
    if (C.msm > 15u) return fail(TQ_ERR_UNSUPPORTED, "query %u: nested minimum_number_should_match above 15", qi);
    tq.first_term[o] = n;
    tq.outer[o] = C.outer;
    tq.inner_need[o] = C.msm;
    for (uint32_t x = 0; x < C.n; ++x) {
      const Atom &A = atoms[C.atom[x]];
      uint32_t max_off = 0;
      for (uint32_t i = 0; i < A.n; ++i) max_off = std::max(max_off, A.off[i]);
      for (uint32_t i = 0; i < A.n; ++i) {
        const TermHost &th = s->terms[A.handle[i]];
        const bool own = th.dense_blob && th.tf8_blob;
        const void *bm = own ? th.dense_blob : th.probe_dense_blob, *t8 = own ? th.tf8_blob : th.probe_tf8_blob;
        if (!bm || !t8)
          return fail(TQ_ERR_UNSUPPORTED, "query %u: a nested boolean query names a list without a bitmap (options \"dense\" / \"use_dense\" / \"probe_budget_x\" off, or a list of more than max_doc / 32 postings whose own tables did not fit \"dense_budget_x\")", qi);
        tq.dense_off[n] = (uint32_t)(((uint64_t)bm - table_base) >> 3);
        tq.tf8_off[n] = (uint32_t)(((uint64_t)t8 - table_base) >> 3);
        memcpy(&tq.weight_bits[n], &A.w[A.phrase ? 0u : i], sizeof(float));  // (a phrase scores with ONE weight: the sum of its idfs)
        tq.handle[n] = A.handle[i];
        tq.inner[n] = A.inner;
        tq.atom_end[n] = (i + 1u == A.n ? 1u : 0u) | (A.phrase ? 2u : 0u) | (A.any ? 4u : 0u);
        if (A.phrase) {
          const void *dir = own && th.posdir_blob ? th.posdir_blob : th.probe_posdir_blob;
          if (!dir)
            return fail(TQ_ERR_UNSUPPORTED, "query %u: a phrase inside a boolean query names a list without a position directory (the list has no positions table that fits its probe slot)", qi);
          tq.dir_off[n] = (uint32_t)(((uint64_t)dir - table_base) >> 3);
          tq.phrase_off[n] = max_off - A.off[i];
          tq.has_phrase = 1u;
        }
        ++n;
      }
    }
  }
  tq.first_term[n_ord] = n;
  tq.n_terms = n;
  tq.n_clauses = n_ord;
  return TQ_OK;
}

}  // namespace tqi
