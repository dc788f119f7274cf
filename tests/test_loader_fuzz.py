"""Property tests of the Matrix Market loader and converters (hypothesis): random matrices written
with random layout noise (header case, comments, '+' signs, tokens split over lines) must load to the
same entries as the reference library does, and convert to the layouts the oracle produces."""
import gzip

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import hostlib
from helpers import assert_bitexact


@pytest.fixture(scope="module")
def host():
    return hostlib.Host()


def _case():
    return st.integers(1, 30).flatmap(lambda rows: st.integers(1, 30).flatmap(lambda cols: st.tuples(
        st.just(rows), st.just(cols),
        st.lists(st.tuples(st.integers(1, rows), st.integers(1, cols),
                           st.floats(-1e6, 1e6, allow_nan=False, allow_infinity=False, width=64)),
                 min_size=0, max_size=60),
        st.sampled_from(["real", "integer", "pattern", "complex"]),
        st.sampled_from(["general", "symmetric"]),
        st.integers(0, 2 ** 31 - 1))))


def _text(rows, cols, entries, field, symmetry, seed):
    rng = np.random.default_rng(seed)
    case = [str.lower, str.upper, str.title][seed % 3]
    lines = ["%%MatrixMarket " + " ".join(case(w) for w in ("matrix", "coordinate", field, symmetry))]
    for k in range(seed % 3):
        lines.append("% comment " + str(k))
    lines.append("%d %d %d" % (rows, cols, len(entries)))
    toks = []
    for (i, j, a) in entries:
        toks += [("+" if rng.integers(4) == 0 else "") + str(i), str(j)]
        if field == "real":
            # (-0.0 >= 0 is true, and "+-0.0" is not a number)
            toks.append(("+" if np.copysign(1.0, a) > 0 and rng.integers(4) == 0 else "") + repr(float(a)))
        elif field == "integer":
            toks.append(str(int(a) % 1000))
        elif field == "complex":
            toks += [repr(float(a)), repr(float(-a))]
    # tokens separated by random whitespace, records not aligned with lines
    body = ""
    for t in toks:
        body += t + [" ", "\n", "\t", "  \n"][int(rng.integers(4))]
    return "\n".join(lines) + "\n" + body + "\n"


@settings(max_examples=120, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(_case())
def test_loader_and_converters_against_reference(host, oracle, reflib, tmp_path_factory, case):
    rows, cols, entries, field, symmetry, seed = case
    text = _text(rows, cols, entries, field, symmetry, seed)
    h = host.mm_from_text(text)
    r = reflib.mm_from_string(text)
    hi, hj, ha = host.mm_entries(h)
    ri, rj, ra = reflib.mm_entries(r)
    assert hi.tolist() == ri.tolist() and hj.tolist() == rj.tolist()
    assert_bitexact(ha, ra, "values")
    info, rinfo = host.mm_info(h), reflib.mm_info(r)
    assert (info["rows"], info["columns"], info["num_entries"], info["field"], info["symmetry"]) == (
        rinfo["rows"], rinfo["columns"], rinfo["num_entries"], rinfo["field"], rinfo["symmetry"])
    # converters against the oracle (stable order; the reference's own order of duplicate (i, j)
    # entries is unspecified, so only duplicate-free inputs are compared with it)
    A, ci, p, j, a = host.csr(h)
    op, oc, ov = oracle.csr_from_coordinate(rows, hi, hj, ha)
    assert p.tolist() == op.tolist() and j.tolist() == oc.tolist()
    assert_bitexact(a, ov, "csr values")
    keys = hi.astype(np.int64) * 1000 + hj
    if len(np.unique(keys)) == len(keys):
        RA = reflib.csr_from_mm(r)
        rp, rc, rv = reflib.csr_arrays(RA)
        assert p.tolist() == rp.tolist() and j.tolist() == rc.tolist()
        assert_bitexact(a, rv, "csr values vs reference")
        reflib.csr_free(RA)
    x = np.random.default_rng(seed).uniform(-1, 1, cols)
    assert_bitexact(host.csr_spmv(A, rows, x, threads=2), oracle.csr_spmv(rows, op, oc, ov, x), "csr y")
    Hc, hinfo, ej, ea, cr, cc, cv = host.hybrid(h)
    OH = oracle.hybrid_from_coordinate(rows, hi, hj, ha)
    assert hinfo["row_length"] == OH["row_length"] and ej.tolist() == OH["ell_col"].tolist()
    assert cr.tolist() == OH["coo_row"].tolist() and cc.tolist() == OH["coo_col"].tolist()
    host.mm_free(h)
    reflib.mm_free(r)


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(_case())
def test_gzip_round_trip(host, tmp_path_factory, case):
    rows, cols, entries, field, symmetry, seed = case
    text = _text(rows, cols, entries, field, symmetry, seed)
    d = tmp_path_factory.mktemp("gz")
    path = str(d / "m.mtx.gz")
    with gzip.open(path, "wb") as f:
        f.write(text.encode())
    h1, h2 = host.mm_from_text(text), host.mm_load(path)
    a, b = host.mm_entries(h1), host.mm_entries(h2)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))
    host.mm_free(h1)
    host.mm_free(h2)
