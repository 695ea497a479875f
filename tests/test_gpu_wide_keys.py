"""Dictionaries too large for 64-bit (predicate, subject, object) keys: the partitioner switches to 128-bit key words.
The whole path is first checked against the oracle on ordinary inputs with the wide words forced (DX_FORCE_WIDE_KEYS=1),
then on an input that really needs them: 40 000 predicates over 40 million terms (16 + 2 x 26 bits)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture()
def engine():
    from opus_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_forced_wide_keys_match_the_oracle(engine, lubm1, monkeypatch):
    from oracle import dap_oracle
    from parity_util import assert_same_as_oracle, engine_artifacts
    monkeypatch.setenv("DX_FORCE_WIDE_KEYS", "1")
    g = json.load(open(os.path.join(HERE, "golden", "appendix_c.json")))
    engine.partition_text(g["instances"].encode(), g["class_index"], g["n_partitions"])
    assert_same_as_oracle(engine_artifacts(engine), dap_oracle.run_dap(g["instances"], g["class_index"], g["n_partitions"]))
    st = engine.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
    assert st.n_triples == lubm1["n_triples"]
    assert_same_as_oracle(engine_artifacts(engine), lubm1["oracle"])
    # the same input with 64-bit words gives the same partitions row for row
    wide = [engine.partition_rows(q) for q in range(4)]
    monkeypatch.delenv("DX_FORCE_WIDE_KEYS")
    engine.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
    for q in range(4):
        for a, b in zip(wide[q], engine.partition_rows(q)):
            assert np.array_equal(a, b)


def test_forced_wide_keys_heterogeneous_schema(engine, monkeypatch):
    from opus_amd.workloads import hetero
    from oracle import dap_oracle
    from parity_util import assert_same_as_oracle, engine_artifacts
    monkeypatch.setenv("DX_FORCE_WIDE_KEYS", "1")
    text, ci, n = hetero.generate("swdf", 20000, seed=3)
    engine.partition_text(text.encode(), ci, n)
    assert_same_as_oracle(engine_artifacts(engine), dap_oracle.run_dap(text, ci, n))


def _lines(template: bytes, fields, n):
    """n copies of `template` with the zero-padded decimal fields (column, width, values) filled in."""
    buf = np.tile(np.frombuffer(template, dtype=np.uint8), (n, 1))
    for col, width, values in fields:
        v = values.astype(np.int64)
        for k in range(width):
            buf[:, col + width - 1 - k] = 48 + (v // 10 ** k) % 10
    return buf


@pytest.mark.slow
def test_forty_thousand_predicates_over_forty_million_terms(engine):
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 100e9:
        pytest.skip("needs tens of GB of HBM")
    NS, NP, NC, NQ = 20_000_000, 40_000, 16, 8
    s = np.arange(NS, dtype=np.int64)
    o = (s * 7 + 1) % NS
    p = s % NP
    t_type = b"<http://x/s00000000> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <http://x/C00> .\n"
    t_edge = b"<http://x/s00000000> <http://x/p00000> <http://x/s00000000> .\n"
    t_lit = b"<http://x/s00000000> <http://x/lit> \"v00000000\" .\n"
    a = _lines(t_type, [(11, 8, s), (t_type.index(b"C00") + 1, 2, s % NC)], NS)
    b = _lines(t_edge, [(11, 8, s), (t_edge.index(b"p00000") + 1, 5, p), (t_edge.rindex(b"s00000000") + 1, 8, o)], NS)
    c = _lines(t_lit, [(11, 8, s), (t_lit.index(b"v0") + 1, 8, s)], NS)
    text = np.concatenate([a, b, c], axis=1).reshape(-1)   # the three lines of a subject follow one another
    del a, b, c
    ci = "".join("<http://x/C%02d>\t%d\n" % (k, k % NQ) for k in range(NC))
    dev = torch.from_numpy(text).cuda()
    torch.cuda.synchronize()
    st = engine.partition_device(dev.data_ptr(), int(text.size), ci, NQ)
    assert st.n_triples == 3 * NS and st.n_unique == 3 * NS
    assert st.n_terms == 2 * NS + NP + NC + 2 and st.n_terms > 1 << 25       # subjects, literals, predicates, classes, rdf:type, lit
    assert st.n_predicates == NP + 2 and st.n_predicates > 1 << 14
    assert st.n_vertices == NS and st.n_edges == NS and st.n_node_index == NS
    # a subject's class, hence its partition, is s % 16 % 8 = s % 8; an edge lies in the partitions of both ends
    rows = [int(2 * np.count_nonzero(s % NQ == q) + np.count_nonzero((s % NQ == q) | (o % NQ == q))) for q in range(NQ)]
    assert [int(engine.lib.dx_part_rows(engine.ctx, q)) for q in range(NQ)] == rows and st.n_rows == sum(rows)
    # one vertical-partition table, decoded: predicate 123 (its subjects are the s = 123 mod 40000, all in partition 3;
    # their objects in partition 6)
    pid = int(engine.lookup(["<http://x/p00123>"])[0])
    assert sorted(t["partition"] for t in engine.tables() if t["predicate"] == pid) == [3, 6]
    tab = next(t for t in engine.tables() if t["partition"] == 3 and t["predicate"] == pid)
    ts, to = engine.table_rows(tab["index"], tab["rows"])
    got = sorted(zip(engine.decode(ts), engine.decode(to)))
    sel = (p == 123) & ((s % NQ == 3) | (o % NQ == 3))
    want = sorted(("<http://x/s%08d>" % x, "<http://x/s%08d>" % y) for x, y in zip(s[sel].tolist(), o[sel].tolist()))
    assert got == want and len(want) > 50
    # and the type table of partition 3 holds the subjects of class 3 and 11
    tid = int(engine.lookup(["<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"])[0])
    tab = next(t for t in engine.tables() if t["partition"] == 3 and t["predicate"] == tid)
    assert tab["rows"] == int(np.count_nonzero(s % NQ == 3))
    del dev
    torch.cuda.empty_cache()
