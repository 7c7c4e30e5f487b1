"""Test-side emulation of what the HIP kernels do with (query plan, CSR planes), in plain Python
floats (IEEE f64, no FMA).  It lets the `-m "not gpu"` suite check the host half of the product
(flattener + planner: doc ids, layers, tile tables, idf / boosts, entry order) against the oracle
without a GPU.  It lives in tests/ on purpose: the product has no CPU scoring path.
"""
import struct


def _tile_ranges(csr, e, n_tiles):
    """Yield (tile, indices) exactly as the kernels walk a list: table slot -> [rb, re), filter by tile."""
    T = csr["tile_docs"]
    shift = e["shift"] & 0xFF
    for t in range(n_tiles):
        slot = t >> shift
        rb = int(csr["table"][e["tbl_off"] + slot])
        re = int(csr["table"][e["tbl_off"] + slot + 1])
        idx = []
        for i in range(rb, re):
            local = int(csr["doc"][e["post_off"] + i]) - t * T
            if shift != 0 and not (0 <= local < T):
                continue
            assert 0 <= local < T, "table slot with shift 0 holds a posting outside its tile"
            idx.append(e["post_off"] + i)
        yield t, idx


def emulate(snap, scorer, query, boosts, tokenizer=None):
    """-> canonical [(key, score)] the GPU path must produce for this query."""
    csr = snap.host_csr()
    info = snap.info()
    entries, qtl = snap.plan(query, scorer, tokenizer)
    F, T = info["fields_num"], info["tile_docs"]
    n_tiles = max(1, (info["n_ids"] + T - 1) // T)
    alive = lambda d: bool((int(csr["alive"][d >> 5]) >> (d & 31)) & 1)  # delta removals: dropped when emitted
    seen = 0
    if scorer.kind == 1:
        k1, b = scorer.bm25k1, scorer.bm25b
        acc, tag = {}, {}
        for e in entries:
            n_seen = 0
            for _t, idx in _tile_ranges(csr, e, n_tiles):
                for pi in idx:
                    n_seen += 1
                    d = int(csr["doc"][pi])
                    s = 0.0
                    for x in range(F):
                        tf = int(csr["tf"][x][pi])
                        if tf > 0:
                            tfd = float(tf)
                            fl = float(int(csr["fl"][x][pi]))
                            tfn = ((k1 + 1.0) * tfd) / (k1 * ((1.0 - b) + b * (fl / float(csr["avg"][x]))) + tfd)
                            s += tfn * e["idf"] * boosts[x] * e["boost"]
                    visited = tag.get(d) == e["qterm"]
                    if s > 0.0:
                        if d in acc:
                            acc[d] = max(acc[d], s) if visited else acc[d] + s
                        else:
                            acc[d] = s
                    tag[d] = e["qterm"]
            assert n_seen == e["len"], "tile tables do not cover the list exactly once"
            seen += n_seen
        res = [(int(csr["keys"][d]), s) for d, s in acc.items() if alive(d)]
    else:
        rec, fls = {}, {}
        for e in entries:
            layer = e["shift"] >> 8
            n_seen = 0
            for _t, idx in _tile_ranges(csr, e, n_tiles):
                for pi in idx:
                    n_seen += 1
                    d = int(csr["doc"][pi])
                    for x in range(F):
                        tf = int(csr["tf"][x][pi])
                        key = (d, e["node"], x)
                        if tf > 0 and (layer == 0 or rec.get(key, 0) == 0):
                            rec[key] = tf
                        fls[(d, x)] = int(csr["fl"][x][pi])
            assert n_seen == e["len"]
        order = sorted(range(len(entries)), key=lambda i: -entries[i]["boost"])  # stable
        docs = sorted({k[0] for k in rec if alive(k[0])})
        res = []
        for d in docs:
            best = 0.0
            for x in range(F):
                consumed_q, used, pool = set(), {}, 0.0
                for i in order:
                    e = entries[i]
                    tf = rec.get((d, e["node"], x), 0)
                    if tf == 0:
                        continue
                    if e["qterm"] in consumed_q:
                        continue
                    if used.get(e["node"], 0) >= tf:
                        continue
                    used[e["node"]] = used.get(e["node"], 0) + 1
                    consumed_q.add(e["qterm"])
                    df = float(tf)
                    pool += min(e["boost"] / df, 1.0) * df / float(max(fls[(d, x)], qtl))
                best = max(pool, best)
            res.append((int(csr["keys"][d]), best))
    return sorted(res, key=lambda r: (-r[1], r[0]))


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]
