"""MI355X-native drop-in for probly-search's `Index::query` -> `ScoreCalculator` hot path.

Host-side mirror of the reference's public surface (src/lib.rs:3-14 of probly-search 2.0.1):
`Index`, `QueryResult`, `score.bm25`, `score.zero_to_one`, over the C ABI of
include/probly_search_amd.h (libprobly_search_amd.so: C++ host index/flattener/planner + HIP
kernels for gfx950).  There is no CPU scoring path in this package: queries need the HIP
extension and a GPU, and fail loudly otherwise.
"""
from ._lib import LibraryNotBuilt, PsError, lib_path, load  # noqa: F401
from .index import FieldDetails, Index, QueryResult, Snapshot, whitespace_tokenizer  # noqa: F401
from .keys import KeyedIndex, KeyedSnapshot, KeyTable  # noqa: F401
from . import score  # noqa: F401
from .score import ScoreCalculator, bm25, zero_to_one  # noqa: F401

__all__ = ["Index", "Snapshot", "QueryResult", "FieldDetails", "score", "bm25", "zero_to_one", "ScoreCalculator",
           "whitespace_tokenizer", "KeyTable", "KeyedIndex", "KeyedSnapshot", "PsError", "LibraryNotBuilt", "load", "lib_path"]
