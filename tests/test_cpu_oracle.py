"""CPU suite: the oracle against the golden vectors, the host-side tools, and the C-ABI surface."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


# the hand-derived known-answer examples: the only expected outputs there are for this path (the reference's own test
# asserts nothing, AppTest.scala:13-16)
@pytest.fixture(scope="module", params=["appendix_c.json", "hand_derived_2.json"])
def golden(request):
    return json.load(open(os.path.join(HERE, "golden", request.param)))


def test_python_oracle_matches_hand_derived_example(golden):
    from oracle import dap_oracle as O
    g = golden
    r = O.run_dap(g["instances"], g["class_index"], g["n_partitions"])
    assert [[i, u, sorted(t), [list(x) for x in l]] for i, u, t, l in r["vertices"]] == g["vertices"]
    assert r["edges"] == {tuple(e) for e in g["edges"]}
    assert [set(p) for p in r["parts"]] == [{tuple(t) for t in p} for p in g["parts"]]
    assert r["statistics"] == set(g["statistics"])
    assert r["node_index"] == {tuple(x) for x in g["node_index"]}
    assert {k[0] + "/" + k[1]: sorted(v) for k, v in r["vp"].items()} == {k: sorted(tuple(x) for x in v) for k, v in g["vp"].items()}
    assert {p: {u: set(s) for u, s in d.items()} for p, d in r["vp_index"].items()} == \
           {p: {u: set(s) for u, s in d.items()} for p, d in g["vp_index"].items()}
    assert r["partition_stat"] == g["partition_stat"]
    if "schema_cc" in g:
        assert r["schema_cc"] == {tuple(k): (rows, dist) for k, rows, dist in g["schema_cc"]}
        assert r["schema_node_freq"] == g["schema_node_freq"] and r["schema_node_count"] == g["schema_node_count"]


def test_cpp_restatement_matches_hand_derived_example(golden):
    from oracle import cpu_port
    g = golden
    r = cpu_port.run(np.frombuffer(g["instances"].encode(), dtype=np.uint8), g["class_index"], g["n_partitions"])
    assert r["n_triples"] == g.get("n_triples", 9) and r["n_vertices"] == len(g["vertices"]) and r["n_edges"] == len(g["edges"])
    assert r["part_rows"] == [len(p) for p in g["parts"]]
    # the restatement counts (partition, predicate) pairs: predicates that share a directory count apiece, and so does one too long for a directory
    assert r["n_node_index"] == len(g["node_index"]) and r["n_tables"] == len({(q, t[2]) for q, part in enumerate(g["parts"]) for t in part})
    want = [sum(cpu_port.row_digest(*t) for t in p) & ((1 << 64) - 1) for p in g["parts"]]
    assert r["part_digest"] == want
    if "schema_cc" in g:
        h = lambda t: cpu_port.term_hash(t.encode())  # noqa: E731
        assert r["measures"]["cc"] == {tuple(h(x) for x in k): (rows, dist) for k, rows, dist in g["schema_cc"]}
        assert r["measures"]["classes"] == {h(c): (g["schema_node_freq"][c], g["schema_node_count"][c]) for c in g["schema_node_freq"]}


def test_cpp_restatement_matches_python_oracle_on_lubm1(lubm1):
    from oracle import cpu_port
    ref = lubm1["oracle"]
    r = cpu_port.run(lubm1["bytes"], lubm1["class_index"], 4)
    assert r["n_triples"] == ref["n_triples"] and r["n_vertices"] == len(ref["vertices"]) and r["n_edges"] == len(ref["edges"])
    assert r["part_rows"] == [len(p) for p in ref["parts"]]
    assert r["n_node_index"] == len(ref["node_index"])
    assert r["part_digest"] == [sum(cpu_port.row_digest(*t) for t in p) & ((1 << 64) - 1) for p in ref["parts"]]


@pytest.mark.parametrize("line,expect", [
    ("<a> <p> <b> .", ("<a>", "<p>", "<b>")),
    ("<a>\t<p>   \"x y\"@en .", ("<a>", "<p>", "\"x y\"@en")),
    ("<a> <p> ", ("<a>", "<p>", "")),
    (" <a> <p> <b> .", ("", "<a>", "<p> <b>")),          # leading blank: java split keeps an empty first field
    ("<a> <p> <b>.", ("<a>", "<p>", "<b")),              # the last two characters go, whatever they are
])
def test_line_parsing_follows_java_split(line, expect):
    from oracle import dap_oracle as O
    assert O.parse_instances(line + "\n") == [expect]


@pytest.mark.parametrize("text", ["<a> <p>\n", "<a>\n", "<a> <p> <b> .\n\n<a> <p> <c> .\n"])
def test_short_lines_raise_like_the_reference(text):
    from oracle import cpu_port, dap_oracle as O
    with pytest.raises(ValueError):
        O.parse_instances(text)
    with pytest.raises(ValueError):
        cpu_port.run(np.frombuffer(text.encode(), dtype=np.uint8), "<A>\t0\n", 1)


def test_vertex_ids_follow_utf16_order():
    from oracle import dap_oracle as O
    t = "<\U00010400> <p> <a> .\n<￮> <p> <a> .\n"
    v, _ = O.create_instance_graph(O.parse_instances(t))
    assert [u for (_, u, _, _) in v] == ["<a>", "<\U00010400>", "<￮>"]  # surrogates sort below U+FFEE in UTF-16


def test_translator_reproduces_the_references_translated_q2():
    """The only translated file the reference carries (QueryOptimizer.scala:282), byte for byte."""
    from opus_amd import lubm
    from opus_amd.host import translator as T
    want = open(os.path.join(HERE, "golden", "translated_Q2_reference_example.txt")).read()
    O = "http://www.lehigh.edu/~zhp2/2004/0401/univ-bench.owl#"
    ci = {"<%sGraduateStudent>" % O: ["2"], "<%sUniversity>" % O: ["1"], "<%sDepartment>" % O: ["4"]}
    got = T.translate_query("Q2.txt", lubm.queries(0)["Q2.txt"], lambda u: [], ci, {"00001": 10, "00002": 10, "00004": 10})
    assert got == want


def test_translator_and_sqlite_oracle_on_lubm1(lubm1):
    from opus_amd import lubm
    from opus_amd.host import translator as T
    from oracle import query_oracle as Q
    ref = lubm1["oracle"]
    ni = {}
    for u, c in ref["node_index"]:
        ni.setdefault(u, []).append(c)
    ci = T.load_class_index(lubm1["class_index"])
    counts = {}
    for name, text in lubm.queries(0).items():
        if name == "Q2.txt":
            continue  # a 6-way join that takes SQLite several seconds; covered on the GPU side
        tr = T.translate_query(name, text, lambda u: ni.get(u, []), ci, ref["partition_stat"])
        assert tr is not None
        qm, parts, tp = Q.parse_translated_file(tr)
        assert list(qm) == ["X"] and tp == len(T.ParsedQuery(text).triple_patterns)
        counts[name] = Q.run_query(ref["vp"], tr)["count"]
    assert counts["Q1.txt"] == counts["Q10.txt"] > 0   # the two files are the same query
    assert counts["Q5.txt"] == 0                        # teacherOf never points at a department
    assert counts["Q6.txt"] > 0 and counts["Q14.txt"] > counts["Q6.txt"]


def test_generator_is_deterministic_and_ranges_concatenate():
    from opus_amd import lubm
    a, na = lubm.generate(0, 2, seed=5, degree_pool=10)
    b0, n0 = lubm.generate(0, 1, seed=5, degree_pool=10)
    b1, n1 = lubm.generate(1, 2, seed=5, degree_pool=10)
    assert na == n0 + n1 and bytes(a) == bytes(b0) + bytes(b1)
    c, _ = lubm.generate(0, 2, seed=6, degree_pool=10)
    assert bytes(c) != bytes(a)
    assert a[-1] == ord("\n") and bytes(a).count(b"\n") == na


def test_library_exports_every_declared_symbol():
    from opus_amd import engine
    header = open(os.path.join(ROOT, "include", "diaeresis_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(dx_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(engine.EXPORTS)
    lib = engine.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert ctypes.sizeof(engine.PartStats) == 8 * 9 + 4 * 4 + 8 + 8 * 12


def test_no_cpu_fallback():
    """Without a GPU the product path refuses to run instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from opus_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(0)


def test_product_code_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "opus_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.replace("the oracle", ""), os.path.join(dirpath, f)


def test_translator_variable_predicate_union():
    """QueryTranslator.mapSubPartitions + the Array[String] templates (QueryTranslator.scala:189-241, 500-533):
    a variable predicate with a constant subject becomes one UNION arm per 's' entry of the VP index."""
    from opus_amd.host import translator as T
    from oracle.query_oracle import parse_translated_file
    q = ("SELECT ?type ?prop ?val WHERE { <e> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> ?type . <e> ?prop ?val }")
    tr = T.translate_query("q", q, lambda u: {"<e>": ["<C>"]}.get(u, []), {"<C>": ["1", "2"]}, {"00001": 5, "00002": 3},
                           lambda part, uri: ["s_p1_", "o_p9_", "s_http___x_q_"])
    qm, parts, tp = parse_translated_file(tr)
    assert tp == 2
    assert sorted(parts) == ["00002-_3=_http___www_w3_org_1999_02_22_rdf_syntax_ns_type_", "00002-_3=_http___x_q_", "00002-_3=_p1_"]
    assert qm["X"] == (
        "SELECT tab0.type AS type,tab1.prop AS prop,tab1.val AS val FROM "
        "(SELECT o AS type FROM table00002__3_E__http___www_w3_org_1999_02_22_rdf_syntax_ns_type_ WHERE s == '<e>') AS tab0, "
        "( (SELECT o AS val, '<00002-_3=_p1_>' AS prop FROM table00002__3_E__p1_ WHERE s == '<e>')UNION"
        "(SELECT o AS val, '<00002-_3=_http___x_q_>' AS prop FROM table00002__3_E__http___x_q_ WHERE s == '<e>') ) AS tab1")
    with pytest.raises(NotImplementedError):  # without the VP index the reference cannot place the pattern either
        T.translate_query("q", q, lambda u: ["<C>"], {"<C>": ["1"]}, {"00001": 5})


def test_query_shapes_fixture_instantiates():
    import json
    import random
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shape_queries as SQ
    from opus_amd.workloads import hetero
    from oracle import dap_oracle
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "query_shapes.json")))
    assert sum(x["files"] for x in shapes["dbpedia"]) == 112 and sum(x["files"] for x in shapes["swdf/swdf(u)"]) == 279
    text, _, _ = hetero.generate("swdf", 5000, seed=2)
    index = SQ.TripleIndex(dap_oracle.parse_instances(text))
    rng = random.Random(0)
    done = sum(SQ.instantiate(sh["patterns"], index, rng) is not None for lst in shapes.values() for sh in lst)
    assert done >= 35


def test_instance_measures_hand_example():
    """computeCC / computeSchemaNodeFreq / computeSchemaNodeCount on a case worked out by hand from
    Preprocessor.scala:230-264, 312-339: a is typed [A, A, B] (one type stated twice), b is typed [B], c is
    untyped; the distinct edges are (a,p,b), (a,p,c), (c,q,b)."""
    from oracle import dap_oracle as D
    T = D.RDF_TYPE
    text = ("<a> %s <A> .\n<a> %s <A> .\n<a> %s <B> .\n<b> %s <B> .\n" % (T, T, T, T) +
            "<a> <p> <b> .\n<a> <p> <c> .\n<c> <q> <b> .\n<a> <p> <b> .\n")
    vertices, edges = D.create_instance_graph(D.parse_instances(text))
    assert D.compute_schema_node_count(vertices, edges) == {"<A>": 4, "<B>": 4}
    assert D.compute_schema_node_freq(vertices, edges) == {"<A>": 2, "<B>": 2}
    assert D.compute_cc(vertices, edges) == {("<A>", "<p>", "<B>"): (2, 1), ("<B>", "<p>", "<B>"): (1, 1)}


def test_instance_measures_cpp_restatement_agrees_with_python():
    """The two CPU restatements of computeCC / computeSchemaNodeFreq / computeSchemaNodeCount agree on a
    multi-typed, many-class input (strings compared through their FNV-1a hashes)."""
    import numpy as np
    from opus_amd.workloads import hetero
    from oracle import cpu_port, dap_oracle as D
    text, ci, k = hetero.generate("swdf", 6000, seed=3)
    ref = D.run_dap(text, ci, k)
    got = cpu_port.run(np.frombuffer(text.encode(), dtype=np.uint8), ci, k)["measures"]
    H = lambda s: cpu_port.term_hash(s.encode())  # noqa: E731
    classes = {H(c): (ref["schema_node_freq"].get(c, 0), ref["schema_node_count"].get(c, 0))
               for c in set(ref["schema_node_count"]) | set(ref["schema_node_freq"])}
    assert classes == got["classes"] and len(classes) > 10
    assert {(H(a), H(b), H(c)): v for (a, b, c), v in ref["schema_cc"].items()} == got["cc"]


def test_cpu_port_on_threads_equals_one_thread_and_answers_like_sqlite(lubm1):
    """The restatement on several threads (one shared dictionary) gives the statistics and digests of the
    single-threaded one, keeps the same vertical-partition tables as the string-level oracle, and its query leg
    (scans + hash joins, keyed on ids or on strings, one thread or several) counts what SQLite counts."""
    import numpy as np
    from opus_amd import lubm
    from opus_amd.host import translator as T
    from oracle import cpu_port, query_oracle
    arr, ci, ref = lubm1["bytes"], lubm1["class_index"], lubm1["oracle"]
    one = cpu_port.run(arr, ci, 4)
    for threads in (2, 5):
        many = cpu_port.run(arr, ci, 4, threads=threads)
        assert {k: v for k, v in many.items() if k not in ("seconds", "measures")} == \
               {k: v for k, v in one.items() if k not in ("seconds", "measures")}
    ds = cpu_port.Dataset(arr, ci, 4, threads=3)
    try:
        for (part, sub), rows in ref["vp"].items():
            assert ds.table_rows(part + "-" + sub) == len(rows)
        assert ds.table_rows("00000-_3=_no_such_table_") == -1
        ni = {}
        for u, c in ref["node_index"]:
            ni.setdefault(u, []).append(c)
        cidx = T.load_class_index(ci)
        for name, text in lubm.queries(0).items():
            tr = T.translate_query(name, text, lambda x: ni.get(x, []), cidx, ref["partition_stat"])
            qm, parts, _ = query_oracle.parse_translated_file(tr)
            want = query_oracle.run_query(ref["vp"], tr)["count"]
            sql = next(iter(qm.values()))
            for threads in (1, 3):
                for by_string in (False, True):
                    assert ds.count(sql, parts, threads=threads, string_keyed=by_string)[0] == want, (name, threads, by_string)
    finally:
        ds.close()


def test_sqlite_oracle_matches_hand_derived_query_answers():
    """The query path's checker against answers worked out by hand (tests/golden/make_hand_derived_queries.py) over
    the hand-derived layout: two- and three-pattern joins, constants on either column, bags, UNION as a set, tables of
    two first-level partitions, and files of two statements folded by their common column (or crossed when there is
    none), QueryProcessor.scala:184-208. Load_data is the sum over the `partitions` line."""
    from oracle.query_oracle import run_query
    layout = json.load(open(os.path.join(HERE, "golden", "hand_derived_2.json")))
    vp = {tuple(k.split("/", 1)): [tuple(r) for r in v] for k, v in layout["vp"].items()}
    cases = json.load(open(os.path.join(HERE, "golden", "hand_derived_queries.json")))["cases"]
    assert len(cases) == 9
    for c in cases:
        r = run_query(vp, c["text"])
        assert r["count"] == c["count"], c["name"]
        assert r["columns"] == c["columns"], c["name"]
        assert [list(x) for x in r["rows"]] == c["rows"], c["name"]
        assert r["load_data"] == c["load_data"], c["name"]
