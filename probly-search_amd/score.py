"""`probly_search::score` (src/score/mod.rs:1-5): the two shipped ScoreCalculator
implementations, as descriptors the GPU engine understands.

The reference's trait (src/score/calculator.rs:33-70) takes host references into the index
(HashMap, arena indices), so the callbacks themselves cannot cross to the device; what crosses
the C ABI is `ps_scorer_desc {kind, bm25_k1, bm25_b}` and the kernels implement before_each /
score / finalize of exactly these two calculators.
"""


class BM25:
    """score::bm25::BM25 (src/score/default/bm25.rs:14-20): public, mutable k1 / b."""
    kind = 1

    def __init__(self, bm25k1=1.2, bm25b=0.75):
        self.bm25k1 = bm25k1
        self.bm25b = bm25b

    def __repr__(self):
        return "BM25(bm25k1=%r, bm25b=%r)" % (self.bm25k1, self.bm25b)


class ZeroToOne:
    """score::zero_to_one::ZeroToOne (src/score/default/zero_to_one.rs:24-26).  Its per-query
    state (score_by_document_and_field) lives in LDS for the duration of one kernel."""
    kind = 2
    bm25k1 = 0.0
    bm25b = 0.0

    def __repr__(self):
        return "ZeroToOne()"


class _Bm25Module:
    BM25 = BM25

    @staticmethod
    def new():
        """score::bm25::new() — k1 = 1.2, b = 0.75 (src/score/default/bm25.rs:21-26)."""
        return BM25()


class _ZeroToOneModule:
    ZeroToOne = ZeroToOne

    @staticmethod
    def new():
        """score::zero_to_one::new() (src/score/default/zero_to_one.rs:35-39)."""
        return ZeroToOne()


bm25 = _Bm25Module()
zero_to_one = _ZeroToOneModule()
