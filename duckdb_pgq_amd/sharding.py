"""Host-side logic of the multi-GPU path (one process per GPU, torch.distributed).

The data path needs no collective: every rank holds the same staged base tables and builds/expands
only the vertices it owns (gg_csr_build_shard: owner = hash(vertex id) mod N).  What is left on the
host is (a) the ownership function, mirrored here so tests can shard the oracle the same way, and
(b) combining the per-rank results of one query with ONE small all-reduce.
"""
from __future__ import annotations

import numpy as np

GOLD = np.uint64(0x9E3779B97F4A7C15)
MASK64 = (1 << 64) - 1
MASK32 = 0xFFFFFFFF

# order of the per-query result vector that ranks exchange
FIELDS = ("rows1", "rows2", "digest1", "digest2", "traversed_edges", "frontier_entries")
LANEWISE = (2, 3)  # digests: u32 sums mod 2^32 carried in the low half (DESIGN.md "Row digest")


def owner_of(ids: np.ndarray, n_parts: int) -> np.ndarray:
    """Shard that owns each vertex id — bit-identical to `owns()` in csrc/gg_csr.hip."""
    ids = np.ascontiguousarray(ids, dtype=np.int64).view(np.uint64)
    if n_parts <= 1:
        return np.zeros(ids.shape, np.int64)
    with np.errstate(over="ignore"):
        h = (ids * GOLD) >> np.uint64(32)
    return ((h * np.uint64(n_parts)) >> np.uint64(32)).astype(np.int64)  # floor(h32 * n / 2^32)


def stats_to_vec(st: dict) -> list:
    return [st["rows"][1], st["rows"][2], st["digest"][1], st["digest"][2], st["traversed_edges"], st["frontier_entries"]]


def split_halves(vec) -> list:
    """u64 values -> 32-bit halves in int64 slots, so an int64 SUM all-reduce can never wrap."""
    out = []
    for x in vec:
        out += [int(x) & MASK32, int(x) >> 32]
    return out


def join_halves(parts) -> list:
    vec = []
    for i in range(len(parts) // 2):
        lo, hi = int(parts[2 * i]), int(parts[2 * i + 1])
        if i in LANEWISE:
            vec.append(lo & MASK32)  # 32-bit digest: wraps mod 2^32, never carries into the high half
        else:
            vec.append((lo + (hi << 32)) & MASK64)
    return vec


def combine(vec, dist=None, device=None) -> list:
    """All-reduce one query's per-rank result vector (SUM over ranks)."""
    if dist is None or not dist.is_initialized():
        return [int(x) for x in vec]
    # (a group of one rank still makes the call: that is how a one-GPU box exercises the RCCL path)
    import torch

    halves = split_halves(vec)
    if device in (None, "cpu"):
        t = torch.tensor(halves, dtype=torch.int64)
        dist.all_reduce(t)
        return join_halves(t.tolist())
    # device tensors: one page-locked staging tensor and one device tensor per vector length, kept across queries —
    # up, all-reduce and down are queued on torch's stream without a host synchronisation in between (building a
    # tensor from a list and reading it back with tolist() were two of them per query)
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (len(halves), str(device))
    bufs = _combine_buffers.get(key)
    if bufs is None:
        host = torch.empty(len(halves), dtype=torch.int64).pin_memory()
        bufs = (host, host.numpy(), torch.empty(len(halves), dtype=torch.int64, device=device))
        _combine_buffers[key] = bufs
    host, host_np, dev = bufs
    host_np[:] = halves
    dev.copy_(host, non_blocking=True)
    dist.all_reduce(dev)
    host.copy_(dev, non_blocking=True)
    torch.cuda.current_stream(dev.device).synchronize()  # (the stream the copies and the collective were queued on)
    return join_halves(host_np.tolist())


def combine_dev(gg, words, dist=None, device=None) -> list:
    """The same combine with the rank's result vector already ON THE DEVICE (gg.expand_khop_dev: six uint64 words in
    FIELDS order, left in library memory without a host synchronisation): torch's stream is ordered behind the library's
    by an event (gg.stream_wait), the words are all-reduced IN PLACE (SUM; one collective on a view of that memory) and
    cross to the host once — one round trip per query instead of three (down from the expansion, up as a tensor, down
    again).  Counts stay far below 2^63; the two digests are 32-bit sums, masked here."""
    import torch

    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream(device)
    gg.stream_wait(stream.cuda_stream, 0)
    t = torch.as_tensor(words, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t)
    key = ("dev", t.numel(), str(device))
    host = _combine_buffers.get(key)
    if host is None:
        host = torch.empty(t.numel(), dtype=torch.int64).pin_memory()
        _combine_buffers[key] = host
    host.copy_(t, non_blocking=True)
    stream.synchronize()
    vec = [int(x) & MASK64 for x in host.tolist()]
    for i in LANEWISE:
        vec[i] &= MASK32
    return vec


_combine_buffers = {}


def dsum(a: int, b: int) -> int:
    return ((a & MASK32) + (b & MASK32)) & MASK32


def source_batches(rank: int, world: int, per_rank: int):
    """Source-batch sharding of the 64-lane BFS: rank r runs batches r, r+world, ... —
    `per_rank` of them, disjoint across ranks, together the first world*per_rank batches."""
    return [rank + i * world for i in range(per_rank)]


def local_edge_rows(src: np.ndarray, dst: np.ndarray, part: int, n_parts: int):
    """The edge rows rank `part` keeps when the edge table is hash-partitioned across GPUs by endpoint
    ownership: a row lives on the owner of its source (it feeds that rank's forward CSR rows) and on the
    owner of its destination (reverse CSR rows) — at most two ranks, 2/N - 1/N^2 of the table per rank.
    gg_csr_build_shard drops every other row anyway, so building from the local rows gives the same shard."""
    if n_parts <= 1:
        return src, dst
    keep = (owner_of(src, n_parts) == part) | (owner_of(dst, n_parts) == part)
    return np.ascontiguousarray(src[keep]), np.ascontiguousarray(dst[keep])
