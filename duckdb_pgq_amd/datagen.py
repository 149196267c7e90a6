"""Seeded synthetic inputs with the reference's table shapes.

The reference ships no LDBC data (benchmark/ldbc/download-benchmark-data.py:9 fetches it from the
network), so every LDBC-sized config runs on tables produced here: `person(p_personid)` and
`knows(k_person1id, k_person2id)` with both edge directions, as benchmark/ldbc/snb-load.sql:23-24
loads them.  Everything is a pure function of (V, rows, seed): counter-based splitmix64 hashing, no
library PRNG state, and every sort key is made unique so the result does not depend on the sort
algorithm.  The same bytes are fed to the HIP path, the C oracle and the compiled reference.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

# (V, directed knows rows, seed) per LDBC scale factor — SURVEY.md §8d table.
LDBC_SIZES = {
    "sf0.1": (1_528, 28_532, 0x5EED0000),
    "sf1": (9_892, 361_246, 0x5EED0001),
    "sf10": (65_645, 3_877_032, 0x5EED000A),
    "sf100": (448_626, 39_882_396, 0x5EED0064),
}


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser over uint64 (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _stream(seed: int, stream: int, n: int) -> np.ndarray:
    base = splitmix64(np.array([(seed << 8) ^ stream], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        return splitmix64(np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base)


def _perm(seed: int, stream: int, n: int) -> np.ndarray:
    """Deterministic permutation of range(n): argsort of unique 64-bit keys."""
    bits = max(1, int(n - 1).bit_length())
    keys = (_stream(seed, stream, n) >> np.uint64(bits) << np.uint64(bits)) | np.arange(n, dtype=np.uint64)
    return np.argsort(keys).astype(np.int64)


def person_ids(V: int, seed: int) -> np.ndarray:
    """Sparse 64-bit ids of LDBC magnitude (~1e13), in a shuffled table order."""
    i = np.arange(V, dtype=np.int64)
    ids = i * np.int64(4398046511) + (i % 7) * np.int64(1 << 40)
    return ids[_perm(seed, 1, V)]


def _degrees(V: int, total: int, seed: int, alpha: float, cap: int) -> np.ndarray:
    u = (_stream(seed, 2, V) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    u = np.maximum(u, 1.0 / float(1 << 53))
    shape = u ** (-1.0 / (alpha - 1.0))
    lo, hi = 1e-3, float(cap)
    for _ in range(60):  # bisection on the scale so that sum(deg) ~ total
        mid = 0.5 * (lo + hi)
        d = np.clip(np.floor(mid * shape), 1, cap)
        if d.sum() < total:
            lo = mid
        else:
            hi = mid
    d = np.clip(np.floor(hi * shape), 1, cap).astype(np.int64)
    return d


def ldbc_knows(V: int, rows: int, seed: int, alpha: float = 2.2, cap: int = 1000):
    """Return (person_id[V], k_person1id[R], k_person2id[R]) int64 arrays, R ~ rows (even).

    Degree sequence: discrete power law (alpha, min 1, cap) scaled to sum ~ rows; endpoints by the
    configuration model; self-loops and duplicate pairs removed; then mirrored (second half of the
    table is the first half with the columns swapped, like the double COPY in snb-load.sql:23-24).
    """
    vid = person_ids(V, seed)
    deg = _degrees(V, rows, seed, alpha, cap)
    stubs = np.repeat(np.arange(V, dtype=np.int64), deg)
    if stubs.size % 2:
        stubs = stubs[:-1]
    stubs = stubs[_perm(seed, 3, stubs.size)]
    a, b = stubs[0::2], stubs[1::2]
    keep = a != b
    a, b = a[keep], b[keep]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    und = np.unique(lo * np.int64(V) + hi)
    und = und[_perm(seed, 4, und.size)]  # table order is not sorted by endpoint
    lo, hi = und // V, und % V
    flip = (_stream(seed, 5, und.size) & np.uint64(1)).astype(bool)
    p1 = np.where(flip, hi, lo)
    p2 = np.where(flip, lo, hi)
    src = np.concatenate([vid[p1], vid[p2]])
    dst = np.concatenate([vid[p2], vid[p1]])
    return vid, src, dst


def ldbc(scale: str):
    V, rows, seed = LDBC_SIZES[scale]
    return ldbc_knows(V, rows, seed)


def small_graph(V: int, E: int, seed: int, dangling: int = 0, dup_edges: int = 0):
    """Small directed multigraph for edge-case tests: random endpoints (self-loops allowed),
    `dup_edges` repeated rows, `dangling` rows whose endpoint is not a vertex."""
    vid = person_ids(V, seed) if V else np.zeros(0, np.int64)
    if V == 0 or E == 0:
        src = np.zeros(0, np.int64)
        dst = np.zeros(0, np.int64)
    else:
        s = (_stream(seed, 10, E) % np.uint64(V)).astype(np.int64)
        d = (_stream(seed, 11, E) % np.uint64(V)).astype(np.int64)
        src, dst = vid[s], vid[d]
        if dup_edges:
            k = min(dup_edges, E)
            src = np.concatenate([src, src[:k]])
            dst = np.concatenate([dst, dst[:k]])
    if dangling:
        bad = np.int64(-7) - np.arange(dangling, dtype=np.int64)
        other = vid[(np.arange(dangling) % max(V, 1))] if V else bad
        src = np.concatenate([src, bad[: dangling // 2], other[dangling // 2 :]])
        dst = np.concatenate([dst, other[: dangling // 2], bad[dangling // 2 :]])
    return vid, src.astype(np.int64), dst.astype(np.int64)


def pick_sources(vid: np.ndarray, n: int, seed: int, batch: int = 0) -> np.ndarray:
    """n source ids drawn uniformly (seeded, without replacement when possible)."""
    V = vid.size
    p = _perm(seed ^ 0xB0F5, 20 + batch, V)
    return vid[p[: min(n, V)]] if V else np.zeros(0, np.int64)


def replicate_tables(tables: dict, copies: int, stride: int | None = None) -> dict:
    """Scale a small graph fixture by id-shifted replication (Train Benchmark SF1 -> SF<copies>): copy c
    adds c*stride to every id, so the result of a pattern query is the union of the shifted SF1 results."""
    if stride is None:
        stride = int(max(int(t.max()) for t in tables.values() if t.size)) + 1
    out = {}
    shift = (np.arange(copies, dtype=np.int64) * stride)
    for name, t in tables.items():
        t = np.asarray(t, np.int64)
        rep = np.repeat(shift, t.shape[0])
        if t.ndim == 1:
            out[name] = np.tile(t, copies) + rep
        else:
            out[name] = np.tile(t, (copies, 1)) + rep[:, None]
    out["_stride"] = stride
    return out
