"""Train Benchmark SF1 fixture (data files the reference ships under benchmark/trainbenchmark/sf1/,
copied to tests/golden/trainbenchmark_sf1/) and the ConnectedSegments golden rows
(benchmark/trainbenchmark/connectedsegments.benchmark:34-38)."""
import os

import numpy as np

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trainbenchmark_sf1")

CONNECTEDSEGMENTS_GOLDEN = np.array(
    [
        [6, 7, 8, 9, 10, 11, 12],
        [121, 122, 123, 124, 125, 126, 127],
        [688, 689, 690, 691, 692, 693, 694],
        [128, 129, 130, 131, 132, 133, 134],
    ],
    dtype=np.int64,
)


def load(name: str) -> np.ndarray:
    return np.loadtxt(os.path.join(DIR, name + ".csv"), delimiter=",", dtype=np.int64, ndmin=2)


def tables():
    return {"Segment": load("Segment"), "connectsTo": load("connectsTo"), "monitoredBy": load("monitoredBy")}


def connectedsegments_via_joins(orc, t):
    """benchmark/trainbenchmark/queries/connectedsegments.sql evaluated as the reference does: a chain
    of inner hash joins (11 of them), each through the oracle's JoinHashTable restatement."""
    seg = t["Segment"][:, 0]
    ct = t["connectsTo"]
    mb = t["monitoredBy"]
    rows = seg.reshape(-1, 1)  # (segment1)
    for _ in range(5):  # ct1..ct5: last segment = ctN.TrackElement1_id, append TrackElement2_id
        m = orc.hash_join(ct[:, 0], rows[:, -1])
        rows = np.hstack([rows[m[:, 0]], ct[m[:, 1], 1:2]])
    # mb1 on segment1 -> sensor
    m = orc.hash_join(mb[:, 0], rows[:, 0])
    rows = np.hstack([mb[m[:, 1], 1:2], rows[m[:, 0]]])  # (sensor, s1..s6)
    for i in range(2, 7):  # mb_i.TrackElement_id = segment_i AND mb_i.Sensor_id = mb1.Sensor_id
        m = orc.hash_join(mb[:, 0], rows[:, i])
        keep = mb[m[:, 1], 1] == rows[m[:, 0], 0]
        rows = rows[m[keep, 0]]
    return rows


def connectedsegments_sql(hops=5, segment="INNER JOIN", extra=""):
    """benchmark/trainbenchmark/queries/connectedsegments.sql:1-25, generalised over the walk length."""
    cols = ", ".join(f"ct{i}.TrackElement1_id AS segment{i}" for i in range(1, hops + 1))
    sql = f"SELECT mb1.Sensor_id AS sensor, {cols}, ct{hops}.TrackElement2_id AS segment{hops + 1}\nFROM Segment\n"
    sql += f"{segment} connectsTo as ct1 ON Segment.id = ct1.TrackElement1_id\n"
    for i in range(2, hops + 1):
        sql += f"INNER JOIN connectsTo as ct{i} ON ct{i-1}.TrackElement2_id = ct{i}.TrackElement1_id\n"
    for i in range(1, hops + 1):
        sql += f"INNER JOIN monitoredBy as mb{i} ON mb{i}.TrackElement_id = ct{i}.TrackElement1_id\n"
    sql += f"INNER JOIN monitoredBy as mb{hops + 1} ON mb{hops + 1}.TrackElement_id = ct{hops}.TrackElement2_id\n"
    sql += "WHERE " + " AND ".join(f"mb1.Sensor_id = mb{i}.Sensor_id" for i in range(2, hops + 2)) + extra
    return sql
