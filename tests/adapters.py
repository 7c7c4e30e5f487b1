"""Adapters giving the oracle Index and the product Index the same test-facing surface."""
import probly_search_amd as psa
from oracle import oracle as orc


class ProductIndex:
    """kat_runner-facing view of probly_search_amd.Index; `query` goes through the GPU."""

    def __init__(self, fields_num):
        self.idx = psa.Index(fields_num)
        self.fields_num = fields_num

    def add_document(self, key, fields, tokenizer=None):
        self.idx.add_field_values(key, fields, tokenizer)

    def remove_document(self, key):
        self.idx.remove_document(key)

    def vacuum(self):
        self.idx.vacuum()

    def query(self, q, scorer, boosts, tokenizer=None):
        return [(r.key, r.score) for r in self.idx.query(q, scorer, tokenizer, boosts)]

    def expand_term(self, t):
        return self.idx.expand_term(t)

    def field_details(self, i):
        f = self.idx.fields[i]
        return f.sum, f.avg

    def docs_len(self):
        return self.idx.docs_len()

    def doc_field_length(self, key):
        return self.idx.doc_field_length(key)

    def children(self, term=""):
        return self.idx.children(term)

    def count_nodes(self):
        return self.idx.count_nodes()

    def arena_doc_live(self):
        return self.idx.live_pointers()

    def count_documents(self, term):
        return self.idx.count_documents(term)


def product_scorer(name, **kw):
    if name == "bm25":
        s = psa.bm25.new()
        if "k1" in kw:
            s.bm25k1 = kw["k1"]
        if "b" in kw:
            s.bm25b = kw["b"]
        return s
    return psa.zero_to_one.new()


def oracle_scorer(name, **kw):
    return orc.bm25(**kw) if name == "bm25" else orc.zero_to_one()


def replay(steps, fields_num, *indexes):
    for st in steps:
        for ix in indexes:
            if "add" in st:
                ix.add_document(st["add"][0], st["add"][1])
            elif "remove" in st:
                ix.remove_document(st["remove"])
            elif "vacuum" in st:
                ix.vacuum()


def run_device_planned(snap, queries, boosts, top_k, scorer=None):
    """A batch through ps_snapshot_query_batch_device_planned_flat (planner + K1d preparation on the
    device): per query [(key, score), ...]."""
    import probly_search_amd as psa
    from probly_search_amd import dist as psd, synth
    text, offsets = synth.pack_queries(list(queries))
    B = len(queries)
    buf = psd._DeviceBuffer(psd.block_bytes(B, top_k))
    base = buf.ptr.value
    snap.query_batch_device_planned_flat(text, offsets, scorer or psa.bm25.new(), boosts, top_k, base, base + 8 * B * top_k,
                                         base + 16 * B * top_k, stream=None)
    return psd.unpack_blocks(buf.to_host(), 1, B, top_k, [B])
