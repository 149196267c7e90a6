#!/usr/bin/env python3
"""Extract the reference's own operator test vectors for the hot path — the (statement, expected result) pairs of
`test/sql/join/inner/*.test` and `test/sql/cte/*.test` — into tests/golden/ref_sql_vectors.json.

The vectors are DATA: every statement the files run, in order, with the outcome the reference's test runner demands
(ok / error / the expected values, their sort mode, or the "N values hashing to <md5>" form), loops and foreach blocks
unrolled the way test/sqlite/test_sqllogictest.cpp does it (:1440-1530).  tests/test_reference_vectors.py replays them
inside the compiled reference with the planner rules off (pins this reader and the replay) and on (the substituted
engine must neither break what it declines nor mis-answer what it takes).

    python3 tests/golden/make_sqllogic_vectors.py            (needs /root/reference; rewrites the json)
"""
import glob
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_sql_vectors.json")
DIRS = ["test/sql/join/inner", "test/sql/cte"]

NUMERIC = ["tinyint", "smallint", "integer", "bigint", "hugeint", "utinyint", "usmallint", "uinteger", "ubigint", "float",
           "double"]
COLLECTIONS = {"<numeric>": NUMERIC, "<integral>": NUMERIC[:9], "<signed>": NUMERIC[:5], "<unsigned>": NUMERIC[5:9],
               "<alltypes>": NUMERIC + ["bool", "interval", "varchar"]}


def parse(path):
    """-> list of records; loops unrolled with ${name} replaced."""
    lines = open(path).read().split("\n")
    i, n = 0, len(lines)

    def block():
        """records until endloop / EOF"""
        nonlocal i
        out = []
        while i < n:
            line = lines[i].rstrip("\r")
            tok = line.split()
            if not tok or line.startswith("#"):
                i += 1
                continue
            if tok[0] == "endloop":
                i += 1
                return out
            if tok[0] in ("loop", "foreach"):
                i += 1
                if tok[0] == "loop":
                    name, values = tok[1], [str(v) for v in range(int(tok[2]), int(tok[3]))]
                else:
                    name, values = tok[1], []
                    for t in tok[2:]:
                        values += COLLECTIONS.get(t.lower(), [t])
                body = block()
                for v in values:
                    for rec in body:
                        r = dict(rec)
                        r["sql"] = r["sql"].replace("${%s}" % name, v)
                        out.append(r)
                continue
            if tok[0] == "statement":
                i += 1
                sql = []
                while i < n and lines[i].strip():
                    sql.append(lines[i])
                    i += 1
                out.append({"kind": "statement", "expect": tok[1], "sql": "\n".join(sql), "line": i})
                continue
            if tok[0] == "query":
                cols = tok[1]
                sort = tok[2] if len(tok) > 2 and tok[2] in ("nosort", "rowsort", "valuesort") else "nosort"
                i += 1
                sql = []
                while i < n and lines[i].strip() != "----" and lines[i].strip():
                    sql.append(lines[i])
                    i += 1
                expected = []
                if i < n and lines[i].strip() == "----":
                    i += 1
                    while i < n and lines[i] != "":
                        expected.append(lines[i])
                        i += 1
                out.append({"kind": "query", "columns": cols, "sort": sort, "sql": "\n".join(sql), "expected": expected,
                            "line": i})
                continue
            if tok[0] in ("require", "mode", "load", "restart", "hash-threshold", "halt"):
                out.append({"kind": "directive", "sql": line, "line": i})
                i += 1
                continue
            raise SystemExit(f"{path}:{i + 1}: unknown directive {tok[0]!r}")
        return out

    return block()


def main():
    files = {}
    for d in DIRS:
        for path in sorted(glob.glob(os.path.join(REF, d, "*.test"))):
            files[os.path.relpath(path, REF)] = parse(path)
    doc = {"source": "statements and expected results of the reference's own sqllogictest files (test/sql/join/inner, "
                     "test/sql/cte; .test_slow files left out), loops unrolled; made by tests/golden/make_sqllogic_vectors.py",
           "files": files}
    json.dump(doc, open(OUT, "w"), indent=0, sort_keys=True)
    print({k: len(v) for k, v in files.items()}, sum(len(v) for v in files.values()))


if __name__ == "__main__":
    main()
