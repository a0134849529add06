"""Shared test helpers: tiny tokenised corpora -> tantivy-format segments through the ORACLE's
serializers (test infrastructure), mirroring Index::create_in_ram + SimpleTokenizer/LowerCaser."""
import re

import numpy as np

from oracle import oracle as O


def tokenize(text):
    # SimpleTokenizer (split on non-alphanumeric) + LowerCaser (tokenizer defaults of TEXT)
    return [t.lower() for t in re.findall(r"[A-Za-z0-9]+", text)]


def corpus_segment(docs, with_positions=True):
    """Returns (segment, vocab: term -> term_id).  fieldnorm(doc) = token count."""
    toks = [tokenize(d) for d in docs]
    vocab = sorted({t for d in toks for t in d})
    term_id = {t: i for i, t in enumerate(vocab)}
    postings = [[] for _ in vocab]
    positions = [[] for _ in vocab]
    for doc, d in enumerate(toks):
        per = {}
        for p, t in enumerate(d):
            per.setdefault(t, []).append(p)
        for t, ps in per.items():
            postings[term_id[t]].append((doc, len(ps)))
            positions[term_id[t]].append(ps)
    fieldnorms = [len(d) for d in toks]
    seg = O.build_segment(
        len(docs), postings, fieldnorms,
        record_option=O.WITH_FREQS_AND_POSITIONS if with_positions else O.WITH_FREQS,
        positions=positions if with_positions else None)
    return seg, term_id


def nearly_equals(a, b, eps=5e-4):
    # assert_nearly_equals! (src/lib.rs:406-425): abs epsilon 5e-4
    return abs(a - b) < eps


def rel_close(a, b, rel=1e-5):
    return abs(a - b) <= rel * max(abs(a), abs(b), 1e-30)


def random_postings(rng, max_doc, df, max_tf=10):
    docs = np.sort(rng.choice(max_doc, size=df, replace=False)).astype(np.uint32)
    tfs = rng.integers(1, max_tf + 1, size=df).astype(np.uint32)
    return list(zip(docs.tolist(), tfs.tolist()))
