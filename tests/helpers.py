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


def pack4x(vals, num_bits):
    """BitPacker4x layout, written independently of the oracle's C packer: 128
    values -> 16*num_bits bytes; SIMD lane l = values l, l+4, ... as one little-endian bit stream
    whose 32-bit word w lands at byte 16*w + 4*l."""
    assert len(vals) == 128
    out = bytearray(16 * num_bits)
    for lane in range(4):
        acc = 0
        for k, v in enumerate(vals[lane::4]):
            assert 0 <= v < (1 << num_bits) or (num_bits == 0 and v == 0)
            acc |= int(v) << (k * num_bits)
        stream = acc.to_bytes(4 * num_bits, "little") if num_bits else b""
        for w in range(num_bits):
            out[16 * w + 4 * lane: 16 * w + 4 * lane + 4] = stream[4 * w: 4 * w + 4]
    return bytes(out)


def vint_stop_last(v):
    """tantivy's VInt (common/src/vint.rs): 7-bit groups, little-endian, 0x80 on the LAST byte."""
    out = bytearray()
    while True:
        b = v & 127
        v >>= 7
        if v == 0:
            out.append(b | 128)
            return bytes(out)
        out.append(b)


def legacy_posting_list(postings):
    """One WithFreqs posting list as PRE-strict-delta tantivy wrote it (readers still accept it,
    compression/mod.rs:105-125, block_segment_postings.rs:36-54): width byte without bit 6, doc
    deltas v[i]-v[i-1] seeded with the previous block's last doc (0 for the first block), term
    freqs stored raw.  8-byte skip entries (last_doc, doc_bits, tf_bits, block-max fieldnorm id,
    block-max tf = 0: unknown).  Returns the list's bytes (VInt skip_len | skip | blocks | tail)."""
    docs = [d for d, _ in postings]
    tfs = [t for _, t in postings]
    n_full = len(docs) // 128
    skip, payload = bytearray(), bytearray()
    prev = 0
    for b in range(n_full):
        bd = docs[128 * b: 128 * b + 128]
        bt = tfs[128 * b: 128 * b + 128]
        deltas = [bd[0] - prev] + [bd[i] - bd[i - 1] for i in range(1, 128)]
        db = max(deltas).bit_length()
        tb = max(bt).bit_length()
        skip += int(bd[-1]).to_bytes(4, "little") + bytes([db, tb, 0, 0])
        payload += pack4x(deltas, db) + pack4x(bt, tb)
        prev = bd[-1]
    tail = bytearray()
    for d in docs[128 * n_full:]:
        tail += vint_stop_last(d - prev)
        prev = d
    for t in tfs[128 * n_full:]:
        tail += vint_stop_last(t)
    head = (vint_stop_last(len(skip)) + bytes(skip)) if len(docs) >= 128 else b""
    return head + bytes(payload) + bytes(tail)


def exhaustive_by_default(module):
    """The library executes the reference's way by default: block-max pruned top-k ("exhaustive" =
    0).  The parity tests assert match counts and full match sets next to the top-k, which only
    the exhaustive scan reports, so their DeviceIndex starts every segment with "exhaustive" = 1;
    the tests of the pruned mode switch it off explicitly.  Returns a namespace with the module's
    names and that DeviceIndex."""
    import types

    class DeviceIndex(module.DeviceIndex):
        def add_segment(self, *a, **kw):
            super().add_segment(*a, **kw)
            self.set_option("exhaustive", 1, self.n_segments - 1)

        def add_segment_device(self, *a, **kw):
            super().add_segment_device(*a, **kw)
            self.set_option("exhaustive", 1, self.n_segments - 1)

    ns = types.SimpleNamespace(**{k: getattr(module, k) for k in dir(module) if not k.startswith("__")})
    ns.DeviceIndex = DeviceIndex
    return ns
