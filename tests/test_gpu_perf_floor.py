"""A performance floor for the hot path (y += A*x, src/matrix/csr-matrix-spmv.cpp:21-33 and its COO / ELLPACK /
hybrid siblings): one workload per kernel family a BASELINE configuration runs through, each timed with HIP events
and compared with the committed table tests/golden/perf_floor.json (tools/perf_floor.py --write regenerates it).

Round 5: calibrated per box.  The table keeps every launch time together with the STREAM triad of the box it was
measured on; the test measures this box's triad right before the workload and compares us * triad_gbs -- the launch
time in units of what the box's memory delivers.  Bandwidth-bound workloads fail when more than 10 % slower than the
table in those units (the old gate was x 1.15 over the slowest box ever seen in microseconds; x 1.07, tried first in
round 5, failed twice on boxes of EQUAL triad whose launches were 7-8 % slower with identical code -- the table's
"what" quotes the runs); the latency-bound ones (a web graph: 24 us) keep x 1.15; rows whose launch differs by
8-10 % between such boxes (a wave per long row) carry their own gate of 1.12.  Round 6: the queen-like and kkt-like rows
are uploaded three times in the process and the fastest copy counts (their launches move by 3-6 % with the physical pages
the arrays get, tools/placement_probe.py) -- gate 1.06; Poisson 4096^2, which does not move, 1.04; and no row may be more
than 20 % slower in microseconds than the table's box whatever the triads say.  The test takes the fastest of five rounds.
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
        assert (row["matrix"], row["format"], row["flags"]) == (spec, fmt, flags) and row["us"] > 0 and row["triad_gbs"] > 1000
        assert row["bound"] == ("latency" if name in perf_floor.LATENCY_BOUND else "bandwidth")


@pytest.mark.parametrize("name", sorted(json.load(open(os.path.join(ROOT, "tests", "golden", "perf_floor.json")))["workloads"])
                         if os.path.exists(os.path.join(ROOT, "tests", "golden", "perf_floor.json")) else [])
def test_launch_time_within_the_floor(name):
    import perf_floor
    table = _table()
    us, info = perf_floor.measure(name)
    row = table["workloads"][name]
    allowed = row.get("tolerance", table["tolerance"][row["bound"]])  # (a row may carry its own: see the table's "what")
    ratio = us * info["triad_gbs"] / (row["us"] * row["triad_gbs"])
    if us <= 1.02 * row["us"]:
        return  # not slower in microseconds than the table's box: no regression, whatever this box's triad says
    #  beside the triad-normalised ratio an absolute ceiling in microseconds: a box with a low triad does not get to be
    # 16 % slower
    assert us <= perf_floor.ABSOLUTE_CEILING * row["us"], "%s: %.1f us per launch, the table has %.1f us: more than x %.2f slower in microseconds: %r" % (
        name, us, row["us"], perf_floor.ABSOLUTE_CEILING, info)
    assert ratio <= allowed, "%s: %.1f us per launch at a triad of %.0f GB/s, the table has %.1f us at %.0f GB/s: x %.3f in units of the box's triad (x %.2f allowed): %r" % (
        name, us, info["triad_gbs"], row["us"], row["triad_gbs"], ratio, allowed, info)
