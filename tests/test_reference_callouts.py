"""The maintainers' route of the planner rules: a reference built with oracle/callout.patch —
call-outs at the top of three CreatePlan overloads, the source-over-sinks case in Executor::BuildPipelines, the
write observation — hosts the extension built with -DGG_REFERENCE_CALLOUTS.  No interposition shim is loaded and
no file of that extension build touches a private member of the reference.  The SAME test files that run against
the stock reference + shim run here in a child process with GG_REF_VARIANT=patched (oracle/ref_duckdb.py picks the
library, the extension and leaves the shim out)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHED = os.path.join(ROOT, "oracle", "_ref_patched", "libduckdb.so")
EXT = os.path.join(ROOT, "duckdb_pgq_amd", "callouts", "gg_duckdb.duckdb_extension")

pytestmark = pytest.mark.skipif(not (os.path.exists(PATCHED) and os.path.exists(EXT)),
                                reason="oracle/_ref_patched / callouts extension not built (needs /root/reference)")


def _run(args, timeout):
    env = dict(os.environ, GG_REF_VARIANT="patched")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *args], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    return r.stdout


def test_the_rules_register_through_the_call_outs_without_the_shim():
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); from oracle import ref_duckdb as R; "
            "assert R.rules_route() == 'callouts'; d = R.RefDuckDB(threads=2); assert d.hook is None; "
            "d.execute(\"LOAD '\" + R.EXTENSION + \"'\"); ext = ctypes.CDLL(R.EXTENSION); "
            "assert ext.gg_plan_rules_available() == 1 and ext.gg_plan_rules_by_callout() == 1; "
            "maps = open('/proc/self/maps').read(); assert 'libgg_plan_hook' not in maps and '_ref_patched' in maps; "
            "d.close(); print('callouts ok')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, GG_REF_VARIANT="patched"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "callouts ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_no_private_access_in_the_call_out_build_of_the_extension():
    """The -DGG_REFERENCE_CALLOUTS build compiles neither `#define private public` nor the interposed rules: checked
    on the preprocessed text of ALL FIVE host sources where the reference's headers are present.  (The planner rules
    get the connection they plan for from the call-out's first argument, oracle/callout.patch: PlanCallouts::plan_fn.)"""
    ref = "/root/reference/src/include"
    patched = os.path.join(ROOT, "oracle", "_ref_patched", "src", "include")
    if not os.path.isdir(ref):
        pytest.skip("reference headers not present")
    host = os.path.join(ROOT, "duckdb_pgq_amd", "host")
    for name in ("gg_operators.cpp", "gg_duckdb_extension.cpp", "gg_plan_rule.cpp", "gg_ingest.cpp", "gg_pipeline.cpp"):
        out = subprocess.run(["g++", "-std=c++11", "-E", "-dD", "-DGG_REFERENCE_CALLOUTS", "-DNDEBUG", "-I" + patched, "-I" + ref,
                              "-I" + os.path.join(ROOT, "include"), os.path.join(host, name)], capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "#define private public" not in out.stdout, name
        assert "#define protected public" not in out.stdout, name
        # (the shim's header may still declare its entry points; nothing looks them up or defines a rule for them)
        assert "GGBuildPipelinesRule" not in out.stdout and '"gg_plan_hook_register"' not in out.stdout, name
    # ... and the plain build (for a stock reference behind the shim) is the one that does widen the access
    plain = subprocess.run(["g++", "-std=c++11", "-E", "-dD", "-DNDEBUG", "-I" + ref, "-I" + os.path.join(ROOT, "include"),
                            os.path.join(host, "gg_plan_rule.cpp")], capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0 and "#define private public" in plain.stdout


def test_planner_rule_tests_pass_on_the_patched_reference():
    out = _run(["tests/test_plan_rule.py"], timeout=1200)
    assert "53 passed" in out or " passed" in out


@pytest.mark.gpu
def test_extension_tests_pass_on_the_patched_reference():
    _run(["tests/test_duckdb_extension.py", "-m", "gpu"], timeout=1800)
