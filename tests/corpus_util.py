"""Small random corpora / mutation scripts shared by the parity tests (seeded, deterministic)."""
import random

SYLL = ["a", "ab", "abc", "abd", "b", "ba", "bab", "c", "ca", "cab", "d", "de", "x", "xy", "xyz", "é", "éa", "日本", "日"]


def random_term(rng):
    return "".join(rng.choice(SYLL) for _ in range(rng.randint(1, 3)))


def random_field(rng, vocab, lo, hi):
    n = rng.randint(lo, hi)
    toks = [rng.choice(vocab) for _ in range(n)]
    if rng.random() < 0.15:
        toks.insert(rng.randint(0, len(toks)), "")  # doubled space -> empty token
    return " ".join(toks)


def build_script(seed, n_docs=40, fields=2, vocab_size=30, mutate=True, shuffle_keys=False, multi_valued=False):
    """-> (fields_num, steps) where steps use kat_runner's vocabulary (add / remove / vacuum)."""
    rng = random.Random(seed)
    vocab = sorted({random_term(rng) for _ in range(vocab_size)})
    keys = list(range(n_docs))
    if shuffle_keys:
        keys = [k * 7919 % 100003 + (1 << 40) * (k % 3) for k in keys]
        rng.shuffle(keys)
    steps = []
    live = []
    for k in keys:
        vals = []
        for f in range(fields):
            if multi_valued and rng.random() < 0.3:
                vals.append([random_field(rng, vocab, 0, 4) for _ in range(rng.randint(0, 3))])
            else:
                vals.append(random_field(rng, vocab, 0 if f else 1, 3 + 5 * f))
        steps.append({"add": [k, vals]})
        live.append(k)
        if mutate and live and rng.random() < 0.12:
            victim = rng.choice(live)
            live.remove(victim)
            steps.append({"remove": victim})
            if rng.random() < 0.4:
                steps.append({"vacuum": True})
        if mutate and live and rng.random() < 0.06:  # re-add an existing key WITHOUT removing it
            again = rng.choice(live)
            steps.append({"add": [again, [random_field(rng, vocab, 1, 4) for _ in range(fields)]]})
    return fields, steps, vocab


def random_queries(seed, vocab, n=25):
    rng = random.Random(seed * 7 + 1)
    qs = []
    for _ in range(n):
        terms = []
        for _ in range(rng.randint(1, 4)):
            t = rng.choice(vocab)
            r = rng.random()
            if r < 0.35 and len(t) > 1:
                t = t[:rng.randint(1, len(t) - 1)]  # prefix -> expansions
            elif r < 0.45:
                t = t + "q"  # miss
            terms.append(t)
        if rng.random() < 0.2:
            terms.append(terms[0])  # repeated query term
        if rng.random() < 0.15:
            terms.insert(1, "")  # empty token still counts in query_terms_len
        qs.append(" ".join(terms))
    return qs
