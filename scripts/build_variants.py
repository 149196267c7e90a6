#!/usr/bin/env python3
"""Build libgg variants with extra -D flags into build/libgg_<name>.so (diagnostic A/B builds).
usage: build_variants.py name1:-DX=1,-DY=2 name2:-DZ=3 ..."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)


def build(spec):
    name, _, flags = spec.partition(":")
    out = os.path.join(ROOT, "build", f"libgg_{name}.so")
    srcs = [os.path.join(ge.CSRC, s) for s in ge.HIP_SOURCES]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function",
           "-I" + os.path.join(ROOT, "include"), "-I" + ge.CSRC, "-o", out] + [f for f in flags.split(",") if f] + srcs
    subprocess.check_call(cmd)
    return out


with ThreadPoolExecutor(4) as ex:
    for o in ex.map(build, sys.argv[1:]):
        print("built", o)
