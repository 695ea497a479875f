"""The two CLIs end to end: dx-dap writes the reference's layout, dx-query answers from it (a separate
process with nothing resident), results file in the reference's format."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "opus_amd", "_lib")


def test_dap_then_query_round_trip(lubm1, tmp_path):
    from opus_amd import lubm
    from opus_amd.host import translator as T
    from oracle import dap_oracle, query_oracle
    ref = lubm1["oracle"]
    cwd = tmp_path
    (cwd / "instances.nt").write_bytes(bytes(lubm1["bytes"]))
    base = str(cwd / "store")
    env = dict(os.environ)

    # before the schema clustering has written a class index: the measures it needs are produced, then the tool stops
    p = subprocess.run([os.path.join(BIN, "dx-dap"), "LUBM", "4", base, str(cwd / "instances.nt"), "unused-schema-path"],
                       cwd=cwd, capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 1 and "class_index.txt_4_bal not found" in p.stderr, p.stdout + p.stderr
    freq = dict(ln.split("\t") for ln in (cwd / "LUBM_local" / "schema_node_freq.txt").read_text().splitlines())
    assert {k: int(v) for k, v in freq.items()} == ref["schema_node_freq"]
    count = dict(ln.split("\t") for ln in (cwd / "LUBM_local" / "schema_node_count.txt").read_text().splitlines())
    assert {k: int(v) for k, v in count.items()} == ref["schema_node_count"]
    cc = {}
    for line in open(os.path.join(base, "LUBM_data", "schema_cc", "part-00000")):
        x = line.rstrip("\n").split(",")  # Loader.loadCC (Loader.scala:318-322)
        cc[(x[0][2:], x[1], x[2][:-1])] = (int(x[3]), int(x[4][:-1]))
    assert cc == ref["schema_cc"]
    (cwd / "LUBM_local" / "class_index.txt_4_bal").write_text(lubm1["class_index"])

    # wrong argument count: message and exit code of System.exit(-1)
    p = subprocess.run([os.path.join(BIN, "dx-dap"), "LUBM"], cwd=cwd, capture_output=True, text=True)
    assert p.returncode == 255 and "-->Missing Arguments" in p.stdout

    p = subprocess.run([os.path.join(BIN, "dx-dap"), "LUBM", "4", base, str(cwd / "instances.nt"), "unused-schema-path"],
                       cwd=cwd, capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "DAP - partitioning time:" in p.stdout

    # The one command has left the reference's layout: Parquet files with the Scala tools' schemas where they look for
    # them (DAP.scala:364-367, 412, 433-437, 499), written by the library itself (no Python in between)
    from parity_util import assert_layout_equals_oracle
    assert_layout_equals_oracle(base + "/", "LUBM", 4, str(cwd), ref)
    clusters = os.path.join(base, "LUBM_data", "cluster", "clusters_4_bal")
    # VP directory names = the oracle's (partition, _3=<sanitised predicate>) keys
    got_dirs = {(d, s) for d in os.listdir(clusters) if os.path.isdir(os.path.join(clusters, d))
                for s in os.listdir(os.path.join(clusters, d)) if s.startswith("_3=")}
    assert got_dirs == set(ref["vp"])
    assert os.path.isdir(os.path.join(base, "LUBM_data", "node_index_4"))
    for q in range(4):
        assert os.path.isdir(os.path.join(base, "LUBM_data", "classSubPart_index_4__%05d" % q))
    stats = set((cwd / "LUBM_local" / "statistics.txt_4").read_text().split("\n")) - {""}
    assert stats == ref["statistics"]
    pstat = T.load_partition_stats((cwd / "LUBM_local" / "partition_stat.txt_4_bal").read_text())
    assert pstat == ref["partition_stat"]
    # statTables.txt, read the way QueryOptimizer.main does (QueryOptimizer.scala:270-276): "<table> <rows>" per line,
    # keyed by the view name the translated SQL uses; the values are the row counts of the vertical-partition tables
    stat_tables = {}
    for line in (cwd / "LUBM_local" / "statTables.txt").read_text().split("\n"):
        f = line.split(" ")
        if len(f) >= 2:
            stat_tables["table" + f[0].replace("-", "_").replace("=", "_E_").strip()] = float(f[1])
    assert stat_tables == {query_oracle.clean_table_name(part + "-" + sub): float(len(rows)) for (part, sub), rows in ref["vp"].items()}
    assert all(k.startswith("table0000") and "_3_E_" in k for k in stat_tables)

    # second run: output exists, stage is skipped (DAP.scala:86)
    p = subprocess.run([os.path.join(BIN, "dx-dap"), "LUBM", "4", base, str(cwd / "instances.nt"), "x"], cwd=cwd, capture_output=True, text=True)
    assert p.returncode == 0 and "nothing to do" in p.stdout

    # the translator's command line (scripts/run_translator.sh) over the layout dx-dap wrote: same arguments as
    # QueryTranslator.main, translated files under <queries>/translated_queries_4_bal
    ni = {}
    for u, c in ref["node_index"]:
        ni.setdefault(u, []).append(c)
    ci = T.load_class_index(lubm1["class_index"])
    (cwd / "queries").mkdir()
    for name, text in lubm.queries(0).items():
        (cwd / "queries" / name).write_text(text)
    tenv = dict(env, PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-m", "opus_amd.host.translator", "LUBM", "4", base], cwd=cwd, capture_output=True, text=True, env=tenv)
    assert p.returncode == 255 and "-->Missing Arguments" in p.stdout
    p = subprocess.run([sys.executable, "-m", "opus_amd.host.translator", "LUBM", "4", base, str(cwd / "queries")], cwd=cwd,
                       capture_output=True, text=True, env=tenv, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    qdir = cwd / "queries" / "translated_queries_4_bal"
    want = {}
    for name, text in lubm.queries(0).items():
        tr = T.translate_query(name, text, lambda u: ni.get(u, []), ci, pstat)
        assert (qdir / name).read_text() == tr, name
        if name == "Q2.txt":
            os.remove(qdir / name)  # SQLite needs seconds for this 6-way join; covered by test_gpu_queries
            continue
        r = query_oracle.run_query(ref["vp"], tr)
        want[name] = (r["count"], r["load_data"], r["tp"])
    # a file with several statements (QueryProcessor.scala:73-88, 184-208): Q9's three edge patterns as three statements whose
    # results share Y, then X and Z -- never written by the translator, made by hand here
    _, q9_parts, _ = query_oracle.parse_translated_file((qdir / "Q9.txt").read_text())
    edge = {k: next(p_ for p_ in q9_parts if k in p_.lower()) for k in ("advisor", "teacherof", "takescourse")}
    tbl = {k: query_oracle.clean_table_name(v) for k, v in edge.items()}
    multi = (">>>>> Q9 as three statements\npartitions %s\nTP 3\n" % ",".join(edge.values()) +
             "a SELECT tab0.X AS X,tab0.Y AS Y FROM (SELECT s AS X, o AS Y FROM %s) AS tab0 \n" % tbl["advisor"] +
             "b SELECT tab0.Y AS Y,tab0.Z AS Z FROM (SELECT s AS Y, o AS Z FROM %s) AS tab0 \n" % tbl["teacherof"] +
             "c SELECT tab0.X AS X,tab0.Z AS Z FROM (SELECT s AS X, o AS Z FROM %s) AS tab0 " % tbl["takescourse"])
    (qdir / "Q90_three_statements.txt").write_text(multi)
    r = query_oracle.run_query(ref["vp"], multi)
    assert r["count"] > 0
    want["Q90_three_statements.txt"] = (r["count"], r["load_data"], r["tp"])

    p = subprocess.run([os.path.join(BIN, "dx-query"), "LUBM", "4", base, str(qdir)], cwd=cwd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    seen = {}
    for tp in {w[2] for w in want.values()}:
        lines = (cwd / "results" / "LUBM" / "4_bal" / ("results_%d_bal.txt" % tp)).read_text().split("\n")
        rec = {}
        for line in lines:
            if ": " in line:
                k, v = line.split(": ", 1)
                rec[k] = v
                if k == "partitions":
                    seen[rec["Query"]] = (int(rec["Result_count"]), int(rec["Load_data"]), tp)
                    assert set(rec) == {"Query", "Time", "Time_us", "Time_load", "Load_data", "Result_count", "partitions"}
                    rec = {}
    assert seen == want

    # QueryProcessorCold: one query, bindings as CSV under <base><dataset>/result_cold (QueryProcessorCold.scala:28-94)
    import csv
    p = subprocess.run([os.path.join(BIN, "dx-query-cold"), "4"], cwd=cwd, capture_output=True, text=True)
    assert p.returncode == 255 and p.stdout.startswith("Arguments: number of partitions, query file")
    for name in ("Q4.txt", "Q9.txt", "Q14.txt", "Q90_three_statements.txt"):
        tr = (qdir / name).read_text()
        qm, partitions, tp = query_oracle.parse_translated_file(tr)
        proc = query_oracle.SqliteProcessor(ref["vp"])
        proc.preload(partitions)
        cols, rows = proc.execute(qm)  # columns in the order of the SELECT list, like Spark's
        p = subprocess.run([os.path.join(BIN, "dx-query-cold"), "4", str(qdir / name), "LUBM", "1", base], cwd=cwd, capture_output=True,
                           text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        cold = os.path.join(base, "LUBM", "result_cold")
        files = sorted(f for f in os.listdir(cold) if f.endswith(".csv"))
        assert len(files) == 1 and os.path.exists(os.path.join(cold, "_SUCCESS"))  # the previous query's directory was replaced
        with open(os.path.join(cold, files[0]), newline="") as f:
            got = [tuple(r) for r in csv.reader(f, delimiter=",", quotechar='"', escapechar="\\", doublequote=False)]
        assert sorted(got) == sorted(tuple(r) for r in rows), name
        rec = dict(ln.split(": ", 1) for ln in (cwd / "results" / "LUBM" / "cold_4_bal" / ("results_%d.txt" % tp)).read_text().splitlines()[-5:])
        assert rec["Query"] == name and int(rec["Result_count"]) == len(rows) and rec["partitions"] == ",".join(partitions)


def test_layout_written_by_dx_dap_serves_a_fresh_engine_through_its_parquet_files(lubm1, tmp_path):
    """Only dx-dap runs; the translator's command line and parquet_io.import_layout then work from the files alone, as
    the Scala tools would (spark.read.load of the same directories), and the queries answer as the oracle says."""
    from opus_amd import lubm
    from opus_amd.engine import DX_QUERY_BINDINGS, Engine
    from opus_amd.host import parquet_io as P
    from oracle import query_oracle
    ref = lubm1["oracle"]
    cwd = tmp_path
    (cwd / "instances.nt").write_bytes(bytes(lubm1["bytes"]))
    (cwd / "LUBM_local").mkdir()
    (cwd / "LUBM_local" / "class_index.txt_4_bal").write_text(lubm1["class_index"])
    base = str(cwd / "store")
    p = subprocess.run([os.path.join(BIN, "dx-dap"), "LUBM", "4", base, str(cwd / "instances.nt"), "unused"], cwd=cwd,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    (cwd / "queries").mkdir()
    for name, text in lubm.queries(0).items():
        (cwd / "queries" / name).write_text(text)
    tenv = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-m", "opus_amd.host.translator", "LUBM", "4", base, str(cwd / "queries")], cwd=cwd,
                       capture_output=True, text=True, env=tenv, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    e = Engine(0)
    try:
        for name in ("Q1.txt", "Q4.txt", "Q7.txt", "Q9.txt", "Q14.txt"):
            tr = (cwd / "queries" / "translated_queries_4_bal" / name).read_text()
            want = query_oracle.run_query(ref["vp"], tr)
            qm, parts, _ = query_oracle.parse_translated_file(tr)
            rows = P.import_layout(e, base + "/", "LUBM", 4, keys=dict.fromkeys(parts))
            assert sum(rows[k] for k in parts) == want["load_data"]
            res = e.query(next(iter(qm.values())), DX_QUERY_BINDINGS)
            got, cols = res.rows()
            assert res.count == want["count"] and cols == want["columns"] and got == want["rows"], name
    finally:
        e.close()
