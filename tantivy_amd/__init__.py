"""tantivy_amd — MI355X-native query-execution path for tantivy (posting decode, AND/OR/phrase,
BM25, top-k) behind a C ABI (include/tantivy_amd.h).  This package is only the Python binding of
that library; there is no CPU fallback: importing `binding` without the built HIP library raises.
"""
from .binding import (DeviceIndex, Encoder, TermInfoStore, term_dictionary_values, TantivyAmdError, lib, bm25_for_terms, MODE_AND, MODE_OR,  # noqa: F401
                      MODE_PHRASE, MODE_TERM, MODE_BOOL, SHOULD, MUST, MUST_NOT, TERMINATED)
