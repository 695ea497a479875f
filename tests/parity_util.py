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


def assert_layout_equals_oracle(base, dataset, n, local_dir, ref):
    """The Parquet files of an on-disk layout (whoever wrote them) and the text files beside them against the
    oracle's artefacts: part files, vertical-partition directories, node index, VP index, partition_stat, statistics."""
    import os

    import pyarrow.parquet as pq
    from opus_amd.host import parquet_io as P
    root = P.cluster_dir(base, dataset, n)
    assert sorted(P.vp_table_keys(base, dataset, n)) == sorted("%s-%s" % k for k in ref["vp"])
    for (pname, sub), rows in ref["vp"].items():
        t = P.read_table(os.path.join(root, pname, sub))
        assert t.column_names == ["_1", "_2"]
        assert sorted(zip(t["_1"].to_pylist(), t["_2"].to_pylist())) == sorted(rows), (pname, sub)
        assert os.path.exists(os.path.join(root, pname, "_SUCCESS"))  # partitionBy leaves it at the root of the write
    for q, part in enumerate(ref["parts"]):
        files = [f for f in os.listdir(root) if f.startswith("part-%05d-" % q)]
        assert len(files) == (1 if part else 0)
        if part:
            t = pq.read_table(os.path.join(root, files[0]))
            assert t.column_names == ["_1", "_2", "_3"]
            assert set(zip(t["_1"].to_pylist(), t["_2"].to_pylist(), t["_3"].to_pylist())) == part
            assert t.num_rows == len(part)
    # nothing but what Spark would leave there: part files, partition directories, _SUCCESS
    assert all(f.startswith("part-") and f.endswith(".parquet") or f == "_SUCCESS" or os.path.isdir(os.path.join(root, f)) for f in os.listdir(root))
    t = P.read_table(os.path.join(base, dataset + "_data", "node_index_%d" % n))
    assert t.column_names == ["uri", "uriType"]
    assert set(zip(t["uri"].to_pylist(), t["uriType"].to_pylist())) == ref["node_index"] and t.num_rows == len(ref["node_index"])
    for pname, idx in ref["vp_index"].items():
        t = P.read_table(os.path.join(base, dataset + "_data", "classSubPart_index_%d__%s" % (n, pname)))
        assert t.column_names == ["uri", "parts"]
        got = {u: frozenset(p.split(",")) for u, p in zip(t["uri"].to_pylist(), t["parts"].to_pylist())}
        assert got == idx and t.num_rows == len(idx), pname
    lines = open(os.path.join(local_dir, dataset + "_local", "partition_stat.txt_%d_bal" % n)).read().split("\n")
    assert {ln.split(" ")[0]: int(ln.split(" ")[1]) for ln in lines if ln.startswith("00")} == ref["partition_stat"]
    part_lines = {ln.split(" ")[0]: int(ln.split(" ")[1]) for ln in lines if ln.startswith("part-")}
    assert part_lines == {f: len(ref["parts"][int(f[5:10])]) for f in os.listdir(root) if f.startswith("part-")}
    assert set(open(os.path.join(local_dir, dataset + "_local", "statistics.txt_%d" % n)).read().split("\n")) - {""} == ref["statistics"]
