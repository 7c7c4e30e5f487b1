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


class TermData:
    """TermData (src/score/calculator.rs:9-19)."""
    __slots__ = ("query_term_index", "query_term", "query_term_expanded", "query_terms_len")

    def __init__(self, query_term_index, query_term, query_term_expanded, query_terms_len):
        self.query_term_index = query_term_index
        self.query_term = query_term
        self.query_term_expanded = query_term_expanded
        self.query_terms_len = query_terms_len


class FieldData:
    """FieldData (src/score/calculator.rs:21-26): fields_boost + per-field FieldDetails."""
    __slots__ = ("fields_boost", "fields")

    def __init__(self, fields_boost, fields):
        self.fields_boost = fields_boost
        self.fields = fields


class DocumentPointer:
    """DocumentPointer (src/index.rs:354-361): details_key + per-field term_frequency."""
    __slots__ = ("details_key", "term_frequency")

    def __init__(self, details_key, term_frequency):
        self.details_key = details_key
        self.term_frequency = term_frequency


class DocumentDetails:
    """DocumentDetails (src/index.rs:342-349): key + per-field field_length."""
    __slots__ = ("key", "field_length")

    def __init__(self, key, field_length):
        self.key = key
        self.field_length = field_length


class ScoreCalculator:
    """trait ScoreCalculator<T, M> (src/score/calculator.rs:33-70).  Subclass it for a custom
    scorer: `Index.query` then hands the three methods to the library as C callbacks
    (PS_SCORER_HOST_CALLBACKS) and the library runs the reference's driver loop on the host,
    calling them in the reference's order.  The two shipped calculators (bm25, zero_to_one) are
    NOT routed this way: they run on the GPU."""
    kind = 3
    bm25k1 = 0.0
    bm25b = 0.0

    def before_each(self, term_expansion, document_frequency, documents):
        """-> M or None.  `documents` supports len() (documents.len())."""
        return None

    def score(self, before_output, document_pointer, document_details, index_node, field_data, term_expansion):
        """-> float or None (required)."""
        raise NotImplementedError

    def finalize(self, scores):
        """scores: list[QueryResult], mutable in place (score rewrite, deletion)."""
        return None


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
