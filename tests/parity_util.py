"""Turns the engine's id-encoded results into the string-level structures the oracle produces."""
from collections import defaultdict

import numpy as np

from oracle import dap_oracle


def engine_measures(e, strs):
    """schema_cc / schema_node_freq / schema_node_count as the oracle's string-keyed dicts."""
    _, (c, f, k), (a, p, b, rows, dist) = e.measures()
    freq = {strs[ci]: int(v) for ci, v in zip(c.tolist(), f.tolist()) if v}
    count = {strs[ci]: int(v) for ci, v in zip(c.tolist(), k.tolist()) if v}
    cc = {(strs[x], strs[y], strs[z]): (int(r), int(d)) for x, y, z, r, d in zip(a.tolist(), p.tolist(), b.tolist(), rows.tolist(), dist.tolist())}
    assert len(cc) == len(a), "a (subType, pred, objType) group is reported twice"
    return {"schema_cc": cc, "schema_node_freq": freq, "schema_node_count": count}


def engine_artifacts(e, measures=True):
    strs = e.decode(np.arange(e.term_count(), dtype=np.int64))
    S = lambda a: [strs[i] for i in a]  # noqa: E731
    n = e.partition_count()
    parts, statistics = [], set()
    for q in range(n):
        s, o, p = e.partition_rows(q)
        rows = list(zip(S(s), S(o), S(p)))
        parts.append(rows)
        if rows:
            statistics.add(("0%d" % q if q < 10 else str(q)) + " " + str(len(rows)))
    vp = defaultdict(list)
    for t in e.tables():
        if not t["in_layout"]:
            continue
        s, o = e.table_rows(t["index"], t["rows"])
        key = (dap_oracle.partition_name(t["partition"]), "_3=" + dap_oracle.uri_to_storage(strs[t["predicate"]]))
        vp[key].extend(zip(S(s), S(o)))
    u, t = e.node_index()
    node_index = list(zip(S(u), S(t)))
    vp_index = {}
    for q in range(n):
        if not parts[q]:
            continue
        u, r, p = e.vp_index(q)
        d = defaultdict(set)
        for uu, rr, pp in zip(S(u), r, S(p)):
            d[uu].add(("s" if rr == 0 else "o") + dap_oracle.uri_to_storage(pp))
        vp_index[dap_oracle.partition_name(q)] = {k: frozenset(v) for k, v in d.items()}
    v, es, ed, ep = e.instance_graph()
    stat = defaultdict(int)
    for (name, _), rows in vp.items():
        stat[name] += len(rows)
    extra = engine_measures(e, strs) if measures else {}
    return {**extra, "strs": strs, "parts": parts, "statistics": statistics, "vp": dict(vp), "node_index": node_index,
            "vp_index": vp_index, "vertices": S(v), "edges": set(zip(es.tolist(), ed.tolist(), S(ep))),
            "partition_stat": dict(stat)}


def assert_same_as_oracle(got, ref):
    """Bit-exact comparison (sets / sorted lists of strings, integer counts)."""
    assert got["vertices"] == [u for (_, u, _, _) in ref["vertices"]], "vertex order (ids) differs"
    assert got["edges"] == ref["edges"], "edges differ"
    assert len(got["parts"]) == len(ref["parts"])
    for q, (g, r) in enumerate(zip(got["parts"], ref["parts"])):
        assert len(g) == len(set(g)), "duplicate rows in partition %d" % q
        assert set(g) == r, "partition %d differs: %d vs %d rows" % (q, len(g), len(r))
    assert got["statistics"] == ref["statistics"]
    assert len(got["node_index"]) == len(set(got["node_index"]))
    assert set(got["node_index"]) == ref["node_index"]
    assert set(got["vp"]) == set(ref["vp"]), "VP directory names differ"
    for k in ref["vp"]:
        assert sorted(got["vp"][k]) == sorted(ref["vp"][k]), "VP table %s differs" % (k,)
    assert got["vp_index"] == ref["vp_index"], "VP index differs"
    assert got["partition_stat"] == ref["partition_stat"]
    for k in ("schema_cc", "schema_node_freq", "schema_node_count"):
        if k in got and k in ref:
            assert got[k] == ref[k], "%s differs" % k
