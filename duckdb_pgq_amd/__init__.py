"""duckdb_pgq_amd — MI355X-native graph pattern-matching hot path (CSR build, k-hop MATCH, 64-lane BFS).

The product is the C-ABI library `libgg.so` (hand-written HIP for gfx950, include/gg.h) and the C++
DuckDB operators above it (duckdb_pgq_amd/host/).  This Python package is only the harness used by
tests/ and bench.py: a ctypes binding of that C-ABI (`gg`) and the seeded data generator (`datagen`).
There is no CPU fallback anywhere in this package: without the built extension and a HIP device
every compute call raises.
"""
from . import datagen  # noqa: F401
from .gg import GG, GGError, Csr, load_library, LIB_PATH  # noqa: F401
