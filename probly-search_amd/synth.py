"""Deterministic synthetic corpora / query batches restating BASELINE.json's configs
(SURVEY.md App. C).  Used by bench.py and the large parity tests; numpy only.

Generator: splitmix64 counter streams from `seed`; u01 = (x >> 11) * 2^-53; Zipf(s, V) by
inverse-CDF binary search over a cumulative table computed on the box that runs both the CPU
oracle and the GPU path (so libm differences cannot desynchronise them).  Vocabulary: rank r ->
fixed-width 6-letter lowercase stem (bijective scramble of r in base 26), so no stem is a prefix
of another; variant k of a stem = stem + ["", "s", "ed", "ing"][k].  Doc i has key i, field0
length ~U{3..9}, field1 length ~U{16..48}; tokens are Zipf-drawn stems with a uniform variant.
Queries: Q Zipf-drawn stems, space-joined.
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
GOLDEN = np.uint64(0x9E3779B97F4A7C15)
SUFFIXES = [b"", b"s", b"ed", b"ing"]

#            N docs     F  V        s    X  B     Q  scorer          K
CONFIGS = {
    "C1": dict(n_docs=50_000, fields=1, vocab=20_000, zipf_s=1.0, variants=1, batch=1, q_terms=2,
               scorer="bm25", top_k=10, seed=0x5EED0001),
    "C2": dict(n_docs=1_000_000, fields=2, vocab=100_000, zipf_s=1.0, variants=1, batch=1024, q_terms=3,
               scorer="bm25", top_k=10, seed=0x5EED0002),
    "C3": dict(n_docs=1_000_000, fields=2, vocab=100_000, zipf_s=1.0, variants=1, batch=1024, q_terms=3,
               scorer="zero_to_one", top_k=10, seed=0x5EED0002),  # (C2's documents, the other scorer: one corpus serves both)
    "C4": dict(n_docs=5_000_000, fields=2, vocab=100_000, zipf_s=1.0, variants=1, batch=8192, q_terms=3,
               scorer="bm25", top_k=10, seed=0x5EED0004),
    "C5": dict(n_docs=1_000_000, fields=2, vocab=100_000, zipf_s=1.2, variants=4, batch=1024, q_terms=2,
               scorer="bm25", top_k=10, seed=0x5EED0005),
}


def splitmix64(seed, n, stream=0):
    """First n outputs of the splitmix64 sequence seeded with mix(seed, stream)."""
    with np.errstate(over="ignore"):
        s0 = np.uint64(seed) ^ (np.uint64(stream + 1) * np.uint64(0xD1B54A32D192ED03))
        x = s0 + GOLDEN * np.arange(1, n + 1, dtype=np.uint64)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def u01(x):
    return (x >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def zipf_cdf(vocab, s):
    w = np.arange(1, vocab + 1, dtype=np.float64) ** (-float(s))
    c = np.cumsum(w)
    return c / c[-1]


def stems(vocab):
    """(vocab, 6) uint8 matrix of lowercase stems; rank r -> scramble(r) written in base 26."""
    r = np.arange(vocab, dtype=np.uint64)
    x = (r * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(26 ** 6)
    out = np.empty((vocab, 6), dtype=np.uint8)
    for k in range(6):
        out[:, 5 - k] = (x % np.uint64(26)).astype(np.uint8) + ord("a")
        x //= np.uint64(26)
    return out


class Corpus:
    def __init__(self, n_docs, fields, vocab, zipf_s, variants, seed, **_):
        self.n_docs, self.fields, self.vocab, self.zipf_s, self.variants, self.seed = (
            n_docs, fields, vocab, zipf_s, variants, seed)
        self.cdf = zipf_cdf(vocab, zipf_s)
        self.stems = stems(vocab)

    def chunks(self, chunk_docs=100_000):
        """Yields (keys u64[n], text uint8[...], offsets u64[n*F+1]) for add_documents_flat.
        Every field value ends with one space (an empty trailing token, skipped at index time)."""
        F = self.fields
        lo = [3, 16, 16, 16, 16, 16, 16, 16][:F]
        hi = [9, 48, 48, 48, 48, 48, 48, 48][:F]
        for c0 in range(0, self.n_docs, chunk_docs):
            n = min(chunk_docs, self.n_docs - c0)
            cid = c0 // chunk_docs
            lens = np.empty((n, F), dtype=np.int64)
            for f in range(F):
                r = splitmix64(self.seed, n, stream=1000 * cid + 10 + f)
                lens[:, f] = lo[f] + (r % np.uint64(hi[f] - lo[f] + 1)).astype(np.int64)
            n_tok = int(lens.sum())
            ranks = np.searchsorted(self.cdf, u01(splitmix64(self.seed, n_tok, stream=1000 * cid + 1)), side="right")
            np.minimum(ranks, self.vocab - 1, out=ranks)
            if self.variants > 1:
                var = (splitmix64(self.seed, n_tok, stream=1000 * cid + 2) % np.uint64(self.variants)).astype(np.int64)
            else:
                var = np.zeros(n_tok, dtype=np.int64)
            suf_len = np.array([len(s) for s in SUFFIXES], dtype=np.int64)[var]
            tok_len = 6 + suf_len + 1  # stem + suffix + separator
            tok_off = np.zeros(n_tok + 1, dtype=np.int64)
            np.cumsum(tok_len, out=tok_off[1:])
            text = np.full(int(tok_off[-1]), ord(" "), dtype=np.uint8)
            base = tok_off[:-1]
            st = self.stems[ranks]
            for k in range(6):
                text[base + k] = st[:, k]
            for v in range(1, self.variants):
                sel = np.nonzero(var == v)[0]
                for k, ch in enumerate(SUFFIXES[v]):
                    text[base[sel] + 6 + k] = ch
            # field boundaries in tokens -> bytes
            tok_end = np.cumsum(lens.reshape(-1))
            offsets = np.zeros(n * F + 1, dtype=np.uint64)
            offsets[1:] = tok_off[tok_end].astype(np.uint64)
            keys = np.arange(c0, c0 + n, dtype=np.uint64)
            yield keys, text, offsets

    def queries(self, n, q_terms, salt=0):
        """n query strings of q_terms Zipf-drawn stems each (salt selects an independent batch)."""
        ranks = np.searchsorted(self.cdf, u01(splitmix64(self.seed, n * q_terms, stream=900_000 + salt)),
                                side="right")
        np.minimum(ranks, self.vocab - 1, out=ranks)
        st = self.stems[ranks].reshape(n, q_terms, 6)
        out = []
        for i in range(n):
            out.append(" ".join(bytes(st[i, j]).decode("ascii") for j in range(q_terms)))
        return out


def pack_queries(queries):
    """list[str] -> (uint8 text, uint64 offsets[n+1]) for the *_flat batch entry points."""
    enc = [q.encode("utf-8") for q in queries]
    offsets = np.zeros(len(enc) + 1, dtype=np.uint64)
    np.cumsum([len(e) for e in enc], out=offsets[1:])
    text = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy()
    return text, offsets


def fill(index, corpus, chunk_docs=100_000):
    """index: anything with add_documents_flat(keys, text, offsets) (product Index or the oracle)."""
    for keys, text, offsets in corpus.chunks(chunk_docs):
        index.add_documents_flat(keys, text, offsets)
    return index
