"""What the direct-address id dictionary buys: the SF100-shaped graph with its vertex ids renumbered
0..V-1 (dense) against the LDBC-like sparse ids, same edges otherwise.  Prints per-step times and the
densification kernel's time for both."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import duckdb_pgq_amd as pkg  # noqa: E402


def run(vid, src, dst, steps=10):
    gg = pkg.GG(0)
    gg.set_edge_rowid(False)
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    for _ in range(2):
        c = gg.build_csr()
        st = gg.expand_khop(c, 1, 2)
        c.close()
    gg.profile_reset()
    gg.profile(True)
    t = time.perf_counter()
    for _ in range(steps):
        c = gg.build_csr()
        st = gg.expand_khop(c, 1, 2)
        c.close()
    dt = (time.perf_counter() - t) / steps
    gg.profile(False)
    prof = gg.profile_get()
    gg.close()
    return {"ms_per_step": dt * 1e3, "densify_hist_us": prof["densify_hist"][1] * 1e3 / steps,
            "rows": st["rows"][1:3], "digest": st["digest"][1:3]}


def main():
    scale = sys.argv[1] if len(sys.argv) > 1 else "sf100"
    vid, src, dst = pkg.datagen.ldbc(scale)
    sparse = run(vid, src, dst)
    order = np.argsort(vid)
    dense_of = np.empty(vid.size, np.int64)
    dense_of[order] = np.arange(vid.size)
    pos = np.searchsorted(vid[order], src)
    src_d = order[pos]  # position in the vertex table == new id
    pos = np.searchsorted(vid[order], dst)
    dst_d = order[pos]
    dense = run(np.arange(vid.size, dtype=np.int64), src_d.astype(np.int64), dst_d.astype(np.int64))
    # same dense numbering (id == table position) => identical counts and digests
    print(json.dumps({"scale": scale, "sparse_ids": sparse, "dense_ids": dense,
                      "same_result": sparse["rows"] == dense["rows"] and sparse["digest"] == dense["digest"]}))


if __name__ == "__main__":
    main()
