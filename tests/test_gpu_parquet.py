"""Engine -> Parquet layout of the reference -> a fresh engine: files equal the oracle's artefacts, and the
queries answer the same from the imported tables."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_export_import_round_trip(tmp_path, lubm1):
    from opus_amd import lubm
    from opus_amd.engine import DX_QUERY_BINDINGS, Engine
    from opus_amd.host import parquet_io as P, translator as T
    from oracle.query_oracle import parse_translated_file, run_query
    ref = lubm1["oracle"]
    base = str(tmp_path) + "/"
    e = Engine(0)
    try:
        e.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
        stat = P.export_layout(e, base, "lubm", local_dir=str(tmp_path))
    finally:
        e.close()
    root = P.cluster_dir(base, "lubm", 4)
    # files against the oracle
    assert sorted(P.vp_table_keys(base, "lubm", 4)) == sorted("%s-%s" % k for k in ref["vp"])
    for (pname, sub), rows in ref["vp"].items():
        t = P.read_table(os.path.join(root, pname, sub))
        assert sorted(zip(t["_1"].to_pylist(), t["_2"].to_pylist())) == sorted(rows), (pname, sub)
    for q, part in enumerate(ref["parts"]):
        files = [f for f in os.listdir(root) if f.startswith("part-%05d-" % q)]
        assert len(files) == (1 if part else 0)
        if part:
            import pyarrow.parquet as pq
            t = pq.read_table(os.path.join(root, files[0]))
            assert set(zip(t["_1"].to_pylist(), t["_2"].to_pylist(), t["_3"].to_pylist())) == part
    t = P.read_table(os.path.join(base, "lubm_data", "node_index_4"))
    assert set(zip(t["uri"].to_pylist(), t["uriType"].to_pylist())) == ref["node_index"]
    for pname, idx in ref["vp_index"].items():
        t = P.read_table(os.path.join(base, "lubm_data", "classSubPart_index_4__" + pname))
        got = {u: frozenset(p.split(",")) for u, p in zip(t["uri"].to_pylist(), t["parts"].to_pylist())}
        assert got == idx, pname
    lines = open(os.path.join(str(tmp_path), "lubm_local", "partition_stat.txt_4_bal")).read().split("\n")
    assert {l.split(" ")[0]: int(l.split(" ")[1]) for l in lines if l.startswith("00")} == ref["partition_stat"]
    assert {k: v for k, v in stat.items() if k.startswith("00")} == ref["partition_stat"]
    assert set(open(os.path.join(str(tmp_path), "lubm_local", "statistics.txt_4")).read().split("\n")) - {""} == ref["statistics"]

    # a fresh engine that has only the Parquet files
    ni = {}
    for u, c in ref["node_index"]:
        ni.setdefault(u, []).append(c)
    cidx = T.load_class_index(lubm1["class_index"])
    e2 = Engine(0)
    try:
        for name in ("Q2.txt", "Q4.txt", "Q9.txt", "Q14.txt"):
            tr = T.translate_query(name, lubm.queries(0)[name], lambda u: ni.get(u, []), cidx, ref["partition_stat"])
            want = run_query(ref["vp"], tr)
            qm, parts, _ = parse_translated_file(tr)
            rows = P.import_layout(e2, base, "lubm", 4, keys=dict.fromkeys(parts))
            assert sum(rows[p] for p in parts) == want["load_data"]
            res = e2.query(next(iter(qm.values())), DX_QUERY_BINDINGS)
            got, cols = res.rows()
            assert res.count == want["count"] and cols == want["columns"] and got == want["rows"], name
        with pytest.raises(Exception):
            e2.load_table_ids("00000-_3=x", np.array([10 ** 9]), np.array([0]))  # id outside the dictionary
    finally:
        e2.close()
