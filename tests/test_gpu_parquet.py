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
    from parity_util import assert_layout_equals_oracle
    assert_layout_equals_oracle(base, "lubm", 4, str(tmp_path), ref)
    assert {k: v for k, v in stat.items() if k.startswith("00")} == ref["partition_stat"]

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
