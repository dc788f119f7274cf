"""A performance floor for the hot path (y += A*x, src/matrix/csr-matrix-spmv.cpp:21-33 and its COO / ELLPACK /
hybrid siblings): eleven workloads, one per kernel family a BASELINE configuration runs through, each timed with
HIP events and compared with the committed table tests/golden/perf_floor.json (tools/perf_floor.py --write
regenerates it).  A launch more than 15 % slower than the table fails -- the ELLPACK regression of round 2
(L = 33: 0.96 -> 0.53 of the roofline) was found by hand on the last morning; this turns such a change red.

The table holds the SLOWER of all boxes measured so far and the test takes the fastest of five rounds, so the
margin is for box-to-box spread (a few per cent on this pool), not for noise inside a run.
"""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _table():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "perf_floor.json")))


def test_every_workload_has_a_floor():
    import perf_floor
    assert set(_table()["workloads"]) == set(perf_floor.WORKLOADS)
    for name, (spec, fmt, flags) in perf_floor.WORKLOADS.items():
        row = _table()["workloads"][name]
        assert (row["matrix"], row["format"], row["flags"]) == (spec, fmt, flags) and row["us"] > 0


@pytest.mark.parametrize("name", sorted(json.load(open(os.path.join(ROOT, "tests", "golden", "perf_floor.json")))["workloads"])
                         if os.path.exists(os.path.join(ROOT, "tests", "golden", "perf_floor.json")) else [])
def test_launch_time_within_the_floor(name):
    import perf_floor
    table = _table()
    us, info = perf_floor.measure(name)
    floor = table["workloads"][name]["us"]
    assert us <= table["tolerance"] * floor, "%s: %.1f us per launch, the table has %.1f us (x %.2f allowed): %r" % (
        name, us, floor, table["tolerance"], info)
