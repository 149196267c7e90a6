"""The N > 1 code paths end to end on ONE GPU.  Two fresh child ranks (torch.distributed.run, gloo backend, both on
cuda:0) run the bench programs exactly as the driver launches them on a multi-GPU node, except for the backend:
rank/shard plumbing, the ownership-sharded build + expansion, the combine all-reduce, the graph-sharded BFS
exchange.  And ONE rank under the same launcher with the nccl backend: RCCL itself carries every collective of
those programs — init_process_group("nccl", device_id=...), the combine on a device tensor, the in-place SUM on the
library's own frontier words (__cuda_array_interface__ view), barriers — on the only GPU there is.  Not proven
here: scaling (one GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script, *args, ranks=2, timeout=600, backend="gloo"):
    env = dict(os.environ, GG_BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, script), *args]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_two_ranks_on_one_gpu_combine_to_the_oracle_result():
    line = _torchrun("bench.py", "--gpus", "2", "--workload", "sf1", "--steps", "3", "--warmup", "1", "--no-extras")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["parity_vs_oracle"] is True
    assert line["config"]["rows_1hop"] > 0 and line["config"]["rows_2hop"] > line["config"]["rows_1hop"]
    assert line["roofline"] is not None and 0 < line["roofline"]["frac"] <= 1


def test_graph_sharded_bfs_two_ranks_on_one_gpu_equal_the_whole_graph_bfs():
    line = _torchrun("bench_bfs.py", "--graph-sharded", "--workload", "sf1", "--batches", "3")
    assert line["n_gpus"] == 2 and line["rows_match_whole_graph_bfs"] is True
    assert line["levels_per_batch"] > 1


def test_source_batch_sharded_bfs_two_ranks_on_one_gpu():
    line = _torchrun("bench_bfs.py", "--workload", "sf1", "--batches", "4")
    assert line["n_gpus"] == 2 and line["parity_vs_oracle"] is True and line["value"] > 0


def test_rccl_carries_the_collectives_of_bench_py_at_world_size_one():
    line = _torchrun("bench.py", "--gpus", "1", "--workload", "sf1", "--steps", "3", "--warmup", "1", "--no-extras",
                     ranks=1, backend="nccl")
    assert line["n_gpus"] == 1 and line["parity_vs_oracle"] is True
    assert line["config"].get("collective_backend") == "nccl"


def test_rccl_adds_the_frontier_words_of_the_graph_sharded_bfs_in_place():
    line = _torchrun("bench_bfs.py", "--graph-sharded", "--workload", "sf1", "--batches", "3", ranks=1, backend="nccl")
    assert line["n_gpus"] == 1 and line["rows_match_whole_graph_bfs"] is True
    assert line["collective_backend"] == "nccl" and line["levels_per_batch"] > 1


def test_rccl_reduces_the_statistics_of_the_source_batch_bfs():
    line = _torchrun("bench_bfs.py", "--workload", "sf1", "--batches", "4", ranks=1, backend="nccl")
    assert line["n_gpus"] == 1 and line["parity_vs_oracle"] is True and line["collective_backend"] == "nccl"
