"""Parity of the HIP query path with SQLite executing the same translated SQL text."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def loaded(lubm1):
    """Engine with LUBM-1 partitioned, plus translated queries produced from the engine's own
    node index / statistics by the host-side translator."""
    from opus_amd import lubm
    from opus_amd.engine import Engine
    from opus_amd.host import translator as T
    from parity_util import engine_artifacts
    e = Engine(0)
    e.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
    art = engine_artifacts(e)
    ni = {}
    for u, c in art["node_index"]:
        ni.setdefault(u, []).append(c)
    ci = T.load_class_index(lubm1["class_index"])
    translated = {}
    for name, text in lubm.queries(0).items():
        translated[name] = T.translate_query(name, text, lambda u: ni.get(u, []), ci, art["partition_stat"])
        assert translated[name] is not None, name
    yield e, translated
    e.close()


def _run_file(e, text, mode):
    from oracle.query_oracle import parse_translated_file
    qm, partitions, tp = parse_translated_file(text)
    e.unload_all()
    loaded = sum(e.load_table(p) for p in partitions)
    assert len(qm) == 1
    return e.query(next(iter(qm.values())), mode), loaded


@pytest.mark.parametrize("name", ["Q%d.txt" % i for i in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14)])
def test_lubm_query_counts_and_bindings(loaded, lubm1, name):
    from opus_amd.engine import DX_QUERY_BINDINGS, DX_QUERY_COUNT
    from oracle.query_oracle import run_query
    e, translated = loaded
    ref = run_query(lubm1["oracle"]["vp"], translated[name])
    res, load_rows = _run_file(e, translated[name], DX_QUERY_COUNT)
    assert res.count == ref["count"], name
    assert load_rows == ref["load_data"]
    res_b, _ = _run_file(e, translated[name], DX_QUERY_BINDINGS)
    assert res_b.count == ref["count"]
    rows, cols = res_b.rows()
    assert cols == ref["columns"]
    assert rows == ref["rows"], name  # sorted decoded bindings, bit-exact


@pytest.mark.parametrize("switches", [{"DX_JOIN_STRIPE_ROWS": "1"}, {"DX_JOIN_STRIPE_ROWS": "0"},
                                      {"DX_JOIN_STRIPE_ROWS": "1", "DX_JOIN_NO_LDS": "1"},
                                      {"DX_JOIN_REVERSE": "1", "DX_JOIN_NO_LDS": "1"},
                                      {"DX_JOIN_REVERSE": "1", "DX_JOIN_NO_LDS": "1", "DX_JOIN_STRIPE_ROWS": "1"},
                                      {"DX_JOIN_REVERSE": "0", "DX_JOIN_NO_LDS": "1"},
                                      {"DX_JOIN_LOOKUP": "1"}, {"DX_JOIN_LOOKUP": "1", "DX_JOIN_STRIPE_ROWS": "1"},
                                      {"DX_JOIN_LOOKUP": "0"}],
                         ids=lambda d: ",".join("%s=%s" % (k[3:], v) for k, v in sorted(d.items())))
def test_join_variants_change_nothing(loaded, lubm1, monkeypatch, switches):
    """The ways a join step can be carried out, each forced on every step it can apply to:
    * a large intermediate that a hash join reads is written in eight stripes with a row counter each (the blocks'
      claims then do not queue on one word); DX_JOIN_STRIPE_ROWS is the size from which (default 65536, more than
      LUBM-1 produces; 1: whenever possible, 0: never);
    * the hash table is built over the arriving table's rows and the intermediate streamed past it when the table
      is expected to be the smaller side (DX_JOIN_REVERSE 1: whenever possible, 0: never);
    * DX_JOIN_NO_LDS keeps every join's table in HBM (small intermediates otherwise get theirs in LDS);
    * no table at all when the arriving table is stored in ascending order of its subjects and the join's key holds
      that subject: every row of the intermediate bisects the table (DX_JOIN_LOOKUP 1: whenever possible, 0: never;
      by itself only for tables of 2^20 rows and more).
    Counts and bindings are the oracle's every time."""
    from opus_amd.engine import DX_QUERY_BINDINGS, DX_QUERY_COUNT
    from oracle.query_oracle import run_query
    e, translated = loaded
    for k, v in switches.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("DX_NO_PLAN_CACHE", "1")
    for name in ("Q2.txt", "Q9.txt", "Q8.txt", "Q4.txt", "Q7.txt", "Q12.txt", "Q1.txt"):
        ref = run_query(lubm1["oracle"]["vp"], translated[name])
        res, _ = _run_file(e, translated[name], DX_QUERY_COUNT)
        assert res.count == ref["count"], name
        res_b, _ = _run_file(e, translated[name], DX_QUERY_BINDINGS)
        rows, cols = res_b.rows()
        assert (res_b.count, cols) == (ref["count"], ref["columns"]), name
        assert rows == ref["rows"], name
        # the same statement again: the counters the stripes use are back at zero
        again, _ = _run_file(e, translated[name], DX_QUERY_COUNT)
        assert again.count == ref["count"], name


def test_striped_intermediates_at_a_size_that_stripes_by_itself(monkeypatch):
    """LUBM-16: the first intermediates of Q2 and Q9 hold more than 65536 rows, so they are striped without any
    switch; the bindings are those of the unstriped run (also with every table built over the arriving arm, which
    then streams a striped intermediate), and a statement whose stripes outgrow the buffer's first
    size (the estimate of a join's output is low) runs again and arrives at the same rows."""
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    from opus_amd import lubm
    from opus_amd.engine import DX_QUERY_BINDINGS, Engine
    arr, _ = lubm.generate(0, 16, seed=3, ontology=0, degree_pool=2)
    ci_text = lubm.class_index_text(4, 0)
    e = Engine(0)
    try:
        e.partition_text(arr, ci_text, 4)
        queries = bench.translated_queries(e, ci_text, 0)
        striped_somewhere = False
        for name in ("Q2.txt", "Q9.txt", "Q8.txt"):
            sql, tabs = bench.parse_translated(queries[name])
            got = {}
            for rows_from in ("0", "65536", "65536r"):
                monkeypatch.setenv("DX_JOIN_STRIPE_ROWS", rows_from.rstrip("r"))
                monkeypatch.setenv("DX_JOIN_REVERSE", "1" if rows_from.endswith("r") else "0")
                e.unload_all()
                for p in tabs:
                    e.load_table(p)
                r = e.query(sql, DX_QUERY_BINDINGS)
                ids = r.ids(sort=True)
                got[rows_from] = (r.count, r.columns, ids.tobytes())
                steps = [ln for ln in r.plan.split("\n") if "->" in ln]
                striped_somewhere |= rows_from != "0" and any(int(ln.rsplit("->", 1)[1].split()[0]) >= 65536 for ln in steps[:-1])
            assert got["0"] == got["65536"] == got["65536r"], name
            assert got["0"][0] > 0 and len(got["0"][2]) == 8 * got["0"][0] * len(got["0"][1]), name
        assert striped_somewhere
    finally:
        e.close()


def test_several_statements_in_one_file_are_joined_like_customJoin(loaded, lubm1):
    """A translated file may hold several statements (QueryProcessor.scala:73-88); executeFinalQuery runs each and folds the
    results with customJoin (:184-208): inner join on the output columns two results share, cross join when they share
    none. The translator never writes such a file, so these are made by hand over LUBM-1's tables: Q9's triangle as three
    one-pattern statements (joined on Y, then on X and Z), the same in another order, two statements without a common
    column, and one statement alone through the same entry point. The oracle executes every statement in SQLite and
    joins the materialised results the way the reference's code reads."""
    import re
    from opus_amd.engine import DX_QUERY_BINDINGS, DX_QUERY_COUNT
    from oracle.query_oracle import SqliteProcessor, clean_table_name, parse_translated_file
    e, translated = loaded
    qm, parts, _ = parse_translated_file(translated["Q9.txt"])
    by_pred = {}
    for p in parts:
        for k in ("advisor", "teacherof", "takescourse"):
            if k in p.lower():
                by_pred[k] = p
    assert len(by_pred) == 3
    t = {k: clean_table_name(p) for k, p in by_pred.items()}
    one = {
        "adv": "SELECT tab0.X AS X,tab0.Y AS Y FROM (SELECT s AS X, o AS Y FROM %s) AS tab0 " % t["advisor"],
        "tea": "SELECT tab0.Y AS Y,tab0.Z AS Z FROM (SELECT s AS Y, o AS Z FROM %s) AS tab0 " % t["teacherof"],
        "tak": "SELECT tab0.X AS X,tab0.Z AS Z FROM (SELECT s AS X, o AS Z FROM %s) AS tab0 " % t["takescourse"],
    }
    course = lubm1["oracle"]["vp"][tuple(by_pred["teacherof"].split("-", 1))][0][1]
    few = "SELECT tab0.W AS W FROM (SELECT s AS W FROM %s WHERE o == '%s') AS tab0 " % (t["teacherof"], course)
    cases = [["adv", "tea", "tak"], ["tak", "adv", "tea"], ["tea", "tak"], ["adv"]]
    files = [[(n, one[n]) for n in c] for c in cases] + [[("few", few), ("tea", one["tea"])], [("tea", one["tea"]), ("few", few), ("adv", one["adv"])]]
    proc = SqliteProcessor(lubm1["oracle"]["vp"])
    proc.preload(list(by_pred.values()))
    e.unload_all()
    for p in by_pred.values():
        e.load_table(p)
    seen_rows = 0
    for stmts in files:
        text = ">>>>> made by hand\npartitions %s\nTP %d\n%s" % (",".join(by_pred.values()), len(stmts), "\n".join("%s %s" % s for s in stmts))
        query_map, _, _ = parse_translated_file(text)
        assert list(query_map.values()) == [q for _, q in stmts]
        cols, rows = proc.execute(query_map)
        r = e.query_joined([q for _, q in stmts], DX_QUERY_COUNT)
        assert r.count == len(rows), [n for n, _ in stmts]
        rb = e.query_joined([q for _, q in stmts], DX_QUERY_BINDINGS)
        assert rb.columns == cols, [n for n, _ in stmts]      # Spark's order: shared columns, the left side's, the right side's
        got, got_cols = rb.rows()
        order = sorted(range(len(cols)), key=lambda i: cols[i])
        assert got_cols == [cols[i] for i in order]
        assert got == sorted(tuple(row[i] for i in order) for row in rows), [n for n, _ in stmts]
        seen_rows += len(rows)
    assert seen_rows > 1000
    # the triangle as three statements is Q9 without its type patterns: at least as many rows as Q9 itself
    from oracle.query_oracle import run_query
    assert e.query_joined([one["adv"], one["tea"], one["tak"]]).count >= run_query(lubm1["oracle"]["vp"], translated["Q9.txt"])["count"] > 0


def test_remembered_parse_and_join_order_change_nothing(loaded, lubm1, monkeypatch):
    """A statement that comes again skips the parser and the join-order search (its text and the statistics the
    search looked at are the key): same plan, same count as with DX_NO_PLAN_CACHE=1, which always parses and searches."""
    from opus_amd.engine import DX_QUERY_COUNT
    e, translated = loaded

    def plan_lines(r):
        return [ln for ln in r.plan.split("\n") if not ln.startswith("parse+plan")]
    for name in ("Q2.txt", "Q9.txt", "Q8.txt", "Q4.txt"):
        first, _ = _run_file(e, translated[name], DX_QUERY_COUNT)      # may or may not be remembered already
        again, _ = _run_file(e, translated[name], DX_QUERY_COUNT)      # remembered
        monkeypatch.setenv("DX_NO_PLAN_CACHE", "1")
        fresh, _ = _run_file(e, translated[name], DX_QUERY_COUNT)
        monkeypatch.delenv("DX_NO_PLAN_CACHE")
        assert first.count == again.count == fresh.count, name
        assert plan_lines(first) == plan_lines(again) == plan_lines(fresh), name
        assert len(plan_lines(fresh)) >= 3


@pytest.mark.parametrize("name", ["Q%d.txt" % i for i in (2, 4, 7, 8, 9, 12)])
def test_rows_seen_on_an_earlier_run_change_nothing(loaded, lubm1, monkeypatch, name):
    """A statement that has run before sizes its steps (LDS or HBM table, heads, stripes, buffers) by the rows each
    step produced then instead of the planner's estimates: same count, same bindings, with the feedback off
    (DX_NO_ROW_FEEDBACK=1), on the run that records the rows and on the runs that use them."""
    from opus_amd.engine import DX_QUERY_BINDINGS, DX_QUERY_COUNT
    from oracle.query_oracle import run_query
    e, translated = loaded
    ref = run_query(lubm1["oracle"]["vp"], translated[name])
    monkeypatch.setenv("DX_NO_ROW_FEEDBACK", "1")
    off, _ = _run_file(e, translated[name], DX_QUERY_BINDINGS)
    monkeypatch.delenv("DX_NO_ROW_FEEDBACK")
    assert off.count == ref["count"]
    assert off.rows()[0] == ref["rows"]
    for _ in range(3):
        c, _ = _run_file(e, translated[name], DX_QUERY_COUNT)
        assert c.count == ref["count"], name
        b, _ = _run_file(e, translated[name], DX_QUERY_BINDINGS)
        assert b.count == ref["count"]
        assert b.rows()[0] == ref["rows"], name


@pytest.fixture(scope="module")
def loaded_swat(lubm1_swat):
    """The same for the second namespace the reference ships queries for (queries/lubm10240)."""
    from opus_amd import lubm
    from opus_amd.engine import Engine
    from opus_amd.host import translator as T
    from parity_util import assert_same_as_oracle, engine_artifacts
    e = Engine(0)
    e.partition_text(lubm1_swat["bytes"], lubm1_swat["class_index"], 4)
    art = engine_artifacts(e)
    assert_same_as_oracle(art, lubm1_swat["oracle"])
    ni = {}
    for u, c in art["node_index"]:
        ni.setdefault(u, []).append(c)
    ci = T.load_class_index(lubm1_swat["class_index"])
    translated = {}
    for name, text in lubm.queries(1).items():
        translated[name] = T.translate_query(name, text, lambda u: ni.get(u, []), ci, art["partition_stat"])
        assert translated[name] is not None, name
    yield e, translated
    e.close()


@pytest.mark.parametrize("name", ["Q%d.txt" % i for i in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14)])
def test_lubm10240_folder_query_counts_and_bindings(loaded_swat, lubm1_swat, name):
    """The 13 queries of queries/lubm10240 (swat namespace) over data generated in that namespace."""
    from opus_amd.engine import DX_QUERY_BINDINGS, DX_QUERY_COUNT
    from oracle.query_oracle import run_query
    e, translated = loaded_swat
    ref = run_query(lubm1_swat["oracle"]["vp"], translated[name])
    res, load_rows = _run_file(e, translated[name], DX_QUERY_COUNT)
    assert res.count == ref["count"], name
    assert load_rows == ref["load_data"]
    res_b, _ = _run_file(e, translated[name], DX_QUERY_BINDINGS)
    rows, cols = res_b.rows()
    assert res_b.count == ref["count"] and cols == ref["columns"]
    assert rows == ref["rows"], name  # sorted decoded bindings, bit-exact


def test_lubm10240_folder_answers_are_not_all_empty(loaded_swat, lubm1_swat):
    from oracle.query_oracle import run_query
    _, translated = loaded_swat
    assert sum(run_query(lubm1_swat["oracle"]["vp"], translated[n])["count"] > 0 for n in translated) >= 10


def test_reference_translated_q2_example_parses_and_runs(loaded):
    """The one translated file the reference carries (QueryOptimizer.scala:282) names partitions
    00001/00002/00004 of an N >= 5 run; against this 4-partition data its tables do not exist."""
    from opus_amd.engine import DxError
    from oracle.query_oracle import parse_translated_file
    e, _ = loaded
    qm, partitions, tp = parse_translated_file(open(os.path.join(HERE, "golden", "translated_Q2_reference_example.txt")).read())
    assert tp == 6 and len(partitions) == 6
    e.unload_all()
    with pytest.raises(DxError) as ei:  # SQL parses; the first missing view is reported like Spark's "Table or view not found"
        e.query(qm["X"])
    assert ei.value.code == -2


def test_optimizer_style_rewrite_gives_same_count(loaded, lubm1):
    """QueryOptimizer reorders FROM and replaces the join chain by another spanning set of equalities."""
    from oracle.query_oracle import parse_translated_file, run_query
    e, translated = loaded
    text = translated["Q9.txt"]
    qm, partitions, _ = parse_translated_file(text)
    sql = qm["X"]
    import re
    head, where = sql.rsplit(" WHERE ", 1)
    sel, frm = head.split(" FROM ", 1)
    subs = re.split(r", (?=\()", frm)
    assert len(subs) == 6
    sql2 = sel + " FROM " + ", ".join(reversed(subs)) + " WHERE " + " AND ".join(reversed(where.strip().split(" AND ")))
    ref = run_query(lubm1["oracle"]["vp"], text)
    e.unload_all()
    for p in partitions:
        e.load_table(p)
    assert e.query(sql2).count == ref["count"]


def test_union_and_constant_projection(loaded, lubm1):
    """The variable-predicate templates (QueryTranslator.scala:476-533): UNION arms with a constant column."""
    from opus_amd.engine import DX_QUERY_BINDINGS
    from oracle.query_oracle import SqliteProcessor, clean_table_name
    e, translated = loaded
    O = "http://www.lehigh.edu/~zhp2/2004/0401/univ-bench.owl#"
    parts = ["00002-_3=" + s for s in ("_http___www_lehigh_edu__zhp2_2004_0401_univ_bench_owl_memberof_",
                                       "_http___www_lehigh_edu__zhp2_2004_0401_univ_bench_owl_takescourse_",
                                       "_http___www_w3_org_1999_02_22_rdf_syntax_ns_type_")]
    e.unload_all()
    for p in parts:
        e.load_table(p)
    arms = "UNION".join("(SELECT s AS X, '<%s>' AS P FROM %s WHERE o == '<http://www.Department0.University0.edu>')" % (p, clean_table_name(p))
                        for p in parts[:2])
    sql = ("SELECT tab0.X AS X,tab0.P AS P FROM ( %s ) AS tab0, (SELECT s AS X FROM %s WHERE o == '<%sGraduateStudent>') AS tab1 "
           "WHERE tab0.X=tab1.X " % (arms, clean_table_name(parts[2]), O))
    proc = SqliteProcessor(lubm1["oracle"]["vp"])
    proc.preload(parts)
    cols, rows = proc.execute({"X": sql})
    res = e.query(sql, DX_QUERY_BINDINGS)
    got, gcols = res.rows()
    order = sorted(range(len(cols)), key=lambda i: cols[i])
    assert gcols == [cols[i] for i in order]
    assert got == sorted(tuple(r[i] for i in order) for r in rows)
    assert res.count == len(rows) > 0


def test_union_is_a_set_also_across_different_arms(loaded, lubm1):
    """SQL's UNION (what QueryTranslator.scala:485-491, 507-526 write between the arms) returns every row once. The
    translator's own arms each project their own constant and both columns, so their rows never coincide; arms written
    by hand can: one column of two tables, the same constant in two arms, an arm that drops a column of its table, an
    arm with a filter beside one without. Counts and bindings against SQLite, alone and joined with another pattern."""
    from opus_amd.engine import DX_QUERY_BINDINGS
    from oracle.query_oracle import SqliteProcessor, clean_table_name
    e, translated = loaded
    O = "http://www.lehigh.edu/~zhp2/2004/0401/univ-bench.owl#"
    pre = "00002-_3=_http___www_lehigh_edu__zhp2_2004_0401_univ_bench_owl_"
    parts = [pre + "memberof_", pre + "takescourse_", pre + "advisor_", "00002-_3=_http___www_w3_org_1999_02_22_rdf_syntax_ns_type_"]
    mem, tak, adv, typ = (clean_table_name(p) for p in parts)
    dept = "<http://www.Department0.University0.edu>"
    unions = {
        "subjects of two tables": "( (SELECT s AS X FROM %s)UNION(SELECT s AS X FROM %s) )" % (mem, tak),
        "both columns, one arm turned round": "( (SELECT s AS X, o AS Y FROM %s)UNION(SELECT o AS X, s AS Y FROM %s)UNION(SELECT s AS X, o AS Y FROM %s) )" % (adv, adv, mem),
        "same constant twice, another once": "( (SELECT s AS X, '<p1>' AS P FROM %s)UNION(SELECT s AS X, '<p1>' AS P FROM %s)UNION(SELECT s AS X, '<p2>' AS P FROM %s) )" % (mem, tak, adv),
        "a filtered arm beside a plain one": "( (SELECT s AS X FROM %s WHERE o == '%s')UNION(SELECT s AS X FROM %s) )" % (mem, dept, adv),
        "one arm that drops a column": "( (SELECT o AS Y FROM %s)UNION(SELECT o AS Y FROM %s WHERE s == '<no such term>') )" % (tak, mem),
    }
    proc = SqliteProcessor(lubm1["oracle"]["vp"])
    proc.preload(parts)
    e.unload_all()
    for p in parts:
        e.load_table(p)
    shrunk = 0
    for what, u in unions.items():
        names = [c.split(" AS ")[1].strip() for c in u.split("FROM")[0].split("SELECT")[1].split(",")]
        alone = "SELECT %s FROM %s AS tab0 " % (",".join("tab0.%s AS %s" % (n, n) for n in names), u)
        joined = ("SELECT %s FROM %s AS tab0, (SELECT s AS %s FROM %s WHERE o == '<%sGraduateStudent>') AS tab1 WHERE tab0.%s=tab1.%s " %
                  (",".join("tab0.%s AS %s" % (n, n) for n in names), u, names[0], typ, O, names[0], names[0]))
        for sql in (alone, joined):
            cols, rows = proc.execute({"X": sql})
            for _ in range(2):                      # the second time the duplicate-free table of the arms is already there
                res = e.query(sql, DX_QUERY_BINDINGS)
                got, gcols = res.rows()
                order = sorted(range(len(cols)), key=lambda i: cols[i])
                assert gcols == [cols[i] for i in order], what
                assert got == sorted(tuple(r[i] for i in order) for r in rows), what
                assert res.count == e.query(sql).count == len(rows), what
        bag = sum(len(proc.execute({"X": "SELECT tab0.%s AS %s FROM %s AS tab0 " % (names[0], names[0], arm.strip().rstrip(")").rstrip() + ")")})[1])
                  for arm in u.strip()[1:-1].split("UNION"))
        shrunk += bag > len(proc.execute({"X": alone})[1])
    assert shrunk >= 4   # in nearly every case the arms' rows do coincide: concatenating them would count too many


def test_unknown_constant_and_empty_result(loaded):
    e, translated = loaded
    from oracle.query_oracle import parse_translated_file
    qm, partitions, _ = parse_translated_file(translated["Q1.txt"])
    e.unload_all()
    for p in partitions:
        e.load_table(p)
    sql = qm["X"].replace("GraduateCourse0>", "NoSuchCourse>")
    assert e.query(sql).count == 0


def test_sql_errors(loaded):
    from opus_amd.engine import DxError
    e, _ = loaded
    for bad, code in (("SELECT FROM", -74), ("SELECT tab0.X AS X FROM (SELECT p AS X FROM t) AS tab0", -95),
                      ("SELECT tab0.X AS X FROM (SELECT s AS X FROM nosuchtable) AS tab0", -2)):
        with pytest.raises(DxError) as ei:
            e.query(bad)
        assert ei.value.code == code, bad


def test_union_over_a_directory_shared_by_two_predicates():
    """<http://x/P-1> and <http://x/p_1> sanitise to one directory name, so their rows are one table, and a pair stated
    under both is in it twice. A plain sub-select keeps both rows (Spark reads both files); a UNION is a set union and
    returns the pair once -- also across a layout on disk, where the directory holds two files."""
    from opus_amd.engine import DX_QUERY_BINDINGS, Engine
    from oracle.query_oracle import SqliteProcessor
    from parity_util import engine_artifacts
    T = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"
    text = ("<a> %s <A> .\n<b> %s <A> .\n<c> %s <A> .\n" % (T, T, T) +
            "<a> <http://x/P-1> <b> .\n<a> <http://x/p_1> <b> .\n<a> <http://x/p_1> <c> .\n<b> <http://x/P-1> <c> .\n<a> <http://x/q> <c> .\n")
    e = Engine(0)
    try:
        e.partition_text(text.encode(), "<A>\t0\n", 1)
        art = engine_artifacts(e)
        shared, other = "00000-_3=_http___x_p_1_", "00000-_3=_http___x_q_"
        tab = lambda k: "table" + k.replace("-", "_").replace("=", "_E_")  # noqa: E731
        plain = "SELECT tab0.y AS y FROM (SELECT o AS y FROM %s WHERE s == '<a>') AS tab0" % tab(shared)
        union = ("SELECT tab0.p AS p,tab0.y AS y FROM ( (SELECT o AS y, '<%s>' AS p FROM %s WHERE s == '<a>')UNION"
                 "(SELECT o AS y, '<%s>' AS p FROM %s WHERE s == '<a>') ) AS tab0" % (shared, tab(shared), other, tab(other)))
        proc = SqliteProcessor(art["vp"])
        proc.preload([shared, other])
        want_plain = sorted(proc.execute({"X": plain})[1])
        want_union = sorted(proc.execute({"X": union})[1])
        assert want_plain == [("<b>",), ("<b>",), ("<c>",)] and len(want_union) == 3
        assert e.load_tables([shared, other]) == 5
        assert e.query(plain).count == 3
        r = e.query(union, DX_QUERY_BINDINGS)
        assert r.count == 3
        rows, cols = r.rows()
        assert cols == ["p", "y"] and rows == sorted(want_union)
        r.close()
    finally:
        e.close()


def test_table_statistics_of_small_and_large_tables():
    """The distinct subjects / objects the join-order search works with: counted in one launch for all small tables (a set
    in LDS per table), by marking terms for large ones; both against numpy over the tables' own rows. Tables whose
    predicates share a directory are concatenations and go through the set for both columns."""
    from opus_amd import lubm
    from opus_amd.engine import Engine
    from opus_amd.workloads import hetero
    from oracle import dap_oracle
    e = Engine(0)
    try:
        text, ci, k = hetero.generate("swdf", 30000, seed=4)
        e.partition_text(text.encode(), ci, k)
        strs = e.decode(np.arange(e.term_count(), dtype=np.int64))
        by_key = {}
        for t in e.tables():
            if t["in_layout"]:
                s, o = e.table_rows(t["index"], t["rows"])
                key = "%s-_3=%s" % (dap_oracle.partition_name(t["partition"]), dap_oracle.uri_to_storage(strs[t["predicate"]]))
                by_key.setdefault(key, []).append((s, o))
        assert any(len(v) > 1 for v in by_key.values())   # some directories are shared
        keys = sorted(by_key)
        e.load_tables(keys)
        for key in keys:
            s = np.concatenate([a for a, _ in by_key[key]])
            o = np.concatenate([b for _, b in by_key[key]])
            assert e.table_stats(key) == [len(s), len(np.unique(s)), len(np.unique(o))], key
        # tables above the small-table bound: LUBM-2's larger ones
        arr, _ = lubm.generate(0, 2, seed=2, ontology=0)
        e.partition_text(arr, lubm.class_index_text(4, 0), 4)
        strs = e.decode(np.arange(e.term_count(), dtype=np.int64))
        big = [t for t in e.tables() if t["in_layout"] and t["rows"] > 16384]
        assert big
        for t in big[:6]:
            s, o = e.table_rows(t["index"], t["rows"])
            key = "%s-_3=%s" % (dap_oracle.partition_name(t["partition"]), dap_oracle.uri_to_storage(strs[t["predicate"]]))
            e.load_table(key)
            assert e.table_stats(key) == [len(s), len(np.unique(s)), len(np.unique(o))], key
    finally:
        e.close()
