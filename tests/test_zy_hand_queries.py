"""The query path on the GPU against answers worked out by hand (tests/golden/hand_derived_queries.json, made by
tests/golden/make_hand_derived_queries.py from the reference's rules, QueryProcessor.scala:184-232): no oracle in between."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_hand_derived_query_answers():
    from opus_amd.engine import DX_QUERY_BINDINGS, DX_QUERY_COUNT, Engine
    from oracle.query_oracle import parse_translated_file
    layout = json.load(open(os.path.join(HERE, "golden", "hand_derived_2.json")))
    cases = json.load(open(os.path.join(HERE, "golden", "hand_derived_queries.json")))["cases"]
    e = Engine(0)
    try:
        e.partition_text(layout["instances"].encode(), layout["class_index"], layout["n_partitions"])
        for c in cases:
            qm, parts, _ = parse_translated_file(c["text"])
            statements = list(qm.values())
            e.unload_all()
            assert sum(e.load_table(p) for p in parts) == c["load_data"], c["name"]   # Load_data, QueryProcessor.scala:222-232
            if len(statements) == 1:
                counted = e.query(statements[0], DX_QUERY_COUNT)
                bound = e.query(statements[0], DX_QUERY_BINDINGS)
            else:
                counted = e.query_joined(statements, DX_QUERY_COUNT)
                bound = e.query_joined(statements, DX_QUERY_BINDINGS)
            assert counted.count == c["count"], c["name"]
            rows, cols = bound.rows()
            assert bound.count == c["count"] and cols == c["columns"], c["name"]
            assert [list(r) for r in rows] == c["rows"], c["name"]
            counted.close()
            bound.close()
    finally:
        e.close()
