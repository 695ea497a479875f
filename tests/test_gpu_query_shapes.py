"""The DBpedia and SWDF query folders of the reference, as shapes (tests/golden/query_shapes.json), instantiated
over SWDF-shaped synthetic data and taken through the whole query path: translator (including the
variable-predicate UNION templates), table loading, engine, against SQLite running the same SQL."""
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_reference_query_shapes_on_heterogeneous_data():
    import shape_queries as SQ
    from opus_amd.engine import DX_QUERY_BINDINGS, DxError, Engine
    from opus_amd.host import translator as T
    from opus_amd.workloads import hetero
    from oracle import dap_oracle
    from oracle.query_oracle import parse_translated_file, run_query
    text, ci, k = hetero.generate("swdf", 20000, seed=0)
    ref = dap_oracle.run_dap(text, ci, k)
    index = SQ.TripleIndex(dap_oracle.parse_instances(text))
    ni = {}
    for u, c in ref["node_index"]:
        ni.setdefault(u, []).append(c)
    cidx = T.load_class_index(ci)
    sub_parts = lambda part, uri: sorted(ref["vp_index"].get(part, {}).get(uri, ()))  # noqa: E731
    shapes = json.load(open(os.path.join(HERE, "golden", "query_shapes.json")))
    rng = random.Random(11)
    e = Engine(0)
    answered, nonempty, with_var_pred, not_indexable, unsupported, missing = 0, 0, 0, 0, 0, 0
    try:
        e.partition_text(text.encode(), ci, k)
        for folder, lst in shapes.items():
            for sh in lst:
                for rep in range(3):
                    bind = SQ.instantiate(sh["patterns"], index, rng)
                    assert bind is not None, sh
                    q = SQ.sparql(sh["patterns"], bind)
                    try:
                        tr = T.translate_query("q", q, lambda u: ni.get(u, []), cidx, ref["partition_stat"], sub_parts)
                    except NotImplementedError:
                        unsupported += 1    # templates the reference itself cannot turn into SQL
                        continue
                    if tr is None:
                        not_indexable += 1  # "Query cannot be indexed" (QueryTranslator.scala:111-113)
                        continue
                    qm, parts, _ = parse_translated_file(tr)
                    e.unload_all()
                    try:
                        want = run_query(ref["vp"], tr)
                    except FileNotFoundError:   # a predicate of 150 characters or more has no VP directory (DAP.scala:434)
                        with pytest.raises(DxError) as ei:
                            for p in parts:
                                e.load_table(p)
                        assert ei.value.code == -2
                        missing += 1
                        continue
                    assert sum(e.load_table(p) for p in parts) == want["load_data"]
                    res = e.query(next(iter(qm.values())), DX_QUERY_BINDINGS)
                    rows, cols = res.rows()
                    assert res.count == want["count"], (q, tr)
                    assert cols == want["columns"] and rows == want["rows"], (q, tr)
                    # the same statement with every join that can bisect its sorted table doing so, and with every hash table
                    # built over the arriving table: other ways to the same rows
                    for switch in ("DX_JOIN_LOOKUP", "DX_JOIN_REVERSE"):
                        os.environ[switch] = "1"
                        try:
                            other = e.query(next(iter(qm.values())), DX_QUERY_BINDINGS)
                            assert other.count == want["count"] and other.rows()[0] == want["rows"], (switch, q, tr)
                            other.close()
                        finally:
                            del os.environ[switch]
                    answered += 1
                    nonempty += want["count"] > 0
                    with_var_pred += any(tp[1].startswith("?") for tp in sh["patterns"]) and want["count"] > 0
    finally:
        e.close()
    assert answered >= 60 and nonempty >= 40 and with_var_pred >= 4, (answered, nonempty, with_var_pred, not_indexable, unsupported, missing)
