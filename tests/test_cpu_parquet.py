"""The Parquet side of the reference's layout: file names, schemas, codec, and a write/read round trip of
the oracle's artefacts (no GPU: the writers take plain id arrays and term strings)."""
import glob
import json
import os
import re

import numpy as np
import pytest
import pyarrow.parquet as pq

HERE = os.path.dirname(os.path.abspath(__file__))


def _ids(strings, index):
    return np.array([index[s] for s in strings], dtype=np.int64)


def test_layout_round_trip(tmp_path):
    from opus_amd.host import parquet_io as P
    from oracle import dap_oracle
    g = json.load(open(os.path.join(HERE, "golden", "appendix_c.json")))
    ref = dap_oracle.run_dap(g["instances"], g["class_index"], g["n_partitions"])
    n = g["n_partitions"]
    all_terms = sorted({x for part in ref["parts"] for row in part for x in row})
    index = {t: i for i, t in enumerate(all_terms)}
    terms = P.Terms.from_strings(all_terms)
    base = str(tmp_path) + "/"
    for q, part in enumerate(ref["parts"]):
        rows = sorted(part)
        if rows:
            P.write_partition_file(base, "ds", n, q, terms, _ids([r[0] for r in rows], index), _ids([r[1] for r in rows], index),
                                   _ids([r[2] for r in rows], index))
    by_pred = {}
    for q, part in enumerate(ref["parts"]):
        for s, o, p in sorted(part):
            by_pred.setdefault((q, p), []).append((s, o))
    for (q, p), rows in by_pred.items():
        P.write_vp_table(base, "ds", n, q, p, terms, _ids([r[0] for r in rows], index), _ids([r[1] for r in rows], index))
    ni = sorted(ref["node_index"])
    P.write_node_index(base, "ds", n, terms, _ids([r[0] for r in ni], index), _ids([r[1] for r in ni], index))

    root = P.cluster_dir(base, "ds", n)
    part_files = sorted(f for f in os.listdir(root) if f.startswith("part-"))
    assert all(re.fullmatch(r"part-\d{5}-[0-9a-f-]{36}-c000\.snappy\.parquet", f) for f in part_files)
    assert os.path.exists(os.path.join(root, "_SUCCESS"))
    for f in part_files:
        q = int(f[5:10])
        md = pq.ParquetFile(os.path.join(root, f))
        assert md.schema_arrow.names == ["_1", "_2", "_3"]
        assert all(str(t) == "string" for t in md.schema_arrow.types)
        col = md.metadata.row_group(0).column(0)
        assert col.compression == "SNAPPY" and col.physical_type == "BYTE_ARRAY"
        assert not (md.metadata.metadata or {}).get(b"ARROW:schema")
        t = md.read()
        assert set(zip(t["_1"].to_pylist(), t["_2"].to_pylist(), t["_3"].to_pylist())) == ref["parts"][q]
    # VP directories: Hive-style names, rows equal to the oracle's tables
    assert sorted(P.vp_table_keys(base, "ds", n)) == sorted("%s-%s" % k for k in ref["vp"])
    for (pname, sub), rows in ref["vp"].items():
        t = P.read_table(os.path.join(root, pname, sub))
        assert t.schema.names == ["_1", "_2"]
        assert sorted(zip(t["_1"].to_pylist(), t["_2"].to_pylist())) == sorted(rows)
        assert os.path.exists(os.path.join(root, pname, sub, "_SUCCESS"))
    t = P.read_table(os.path.join(base, "ds_data", "node_index_%d" % n))
    assert t.schema.names == ["uri", "uriType"]
    assert set(zip(t["uri"].to_pylist(), t["uriType"].to_pylist())) == ref["node_index"]


def test_vp_index_file(tmp_path):
    from opus_amd.host import parquet_io as P
    terms = P.Terms.from_strings(["<a>", "<b>", "<http://x/P-1>", "<http://x/q>"])
    P.write_vp_index(str(tmp_path) + "/", "ds", 4, 2, terms, np.array([0, 0, 1, 0]), np.array([0, 1, 1, 0]),
                     ["<http://x/P-1>", "<http://x/q>", "<http://x/q>", "<http://x/P-1>"])
    d = os.path.join(str(tmp_path), "ds_data", "classSubPart_index_4__00002")
    t = P.read_table(d)
    assert t.schema.names == ["uri", "parts"]
    got = dict(zip(t["uri"].to_pylist(), t["parts"].to_pylist()))
    assert got == {"<a>": "o_http___x_q_,s_http___x_p_1_", "<b>": "o_http___x_q_"}
    assert len(glob.glob(os.path.join(d, "part-00000-*-c000.snappy.parquet"))) == 1


def test_translator_command_line_over_parquet_layout(tmp_path, monkeypatch, capsys):
    """run_translator.sh's arguments (QueryTranslator.scala:25-41) over a layout in the Scala tools' Parquet format:
    node index and per-partition VP index read from Parquet, class index and partition statistics from
    <dataset>_local in the working directory; one translated file per query that can be indexed."""
    from opus_amd.host import parquet_io as P
    from opus_amd.host import translator as T
    T_ = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"
    all_terms = ["<a>", "<b>", "<A>", "<B>", "<http://x/p>", "<http://x/q>"]
    terms = P.Terms.from_strings(all_terms)
    base = str(tmp_path / "store") + "/"
    P.write_node_index(base, "ds", 2, terms, np.array([0, 1]), np.array([2, 3]))       # a: A, b: B
    for q in (0, 1):
        P.write_vp_index(base, "ds", 2, q, terms, np.array([0, 0, 1]), np.array([0, 0, 1]), ["<http://x/p>", "<http://x/q>", "<http://x/p>"])
    local = tmp_path / "ds_local"
    local.mkdir()
    (local / "class_index.txt_2_bal").write_text("<A>\t0,1\n<B>\t1\n")
    (local / "partition_stat.txt_2_bal").write_text("00000 10\n00001 7\npart-00000-x.parquet 10\n")
    qdir = tmp_path / "queries"
    qdir.mkdir()
    (qdir / "bound.txt").write_text("SELECT ?x WHERE { ?x <http://x/p> <b> . ?x %s <A> }" % T_)
    (qdir / "varpred.txt").write_text("SELECT ?p ?y WHERE { <a> ?p ?y }")
    (qdir / "nowhere.txt").write_text("SELECT ?x WHERE { ?x <http://x/p> <zzz> }")
    monkeypatch.chdir(tmp_path)
    assert T.main(["ds", "2", base[:-1]]) == 255
    assert "-->Missing Arguments" in capsys.readouterr().out
    assert T.main(["ds", "2", base[:-1], str(qdir)]) == 0
    assert "Query cannot be indexed:nowhere.txt" in capsys.readouterr().out
    out = qdir / "translated_queries_2_bal"
    assert sorted(os.listdir(out)) == ["bound.txt", "varpred.txt"]
    ni = {"<a>": ["<A>"], "<b>": ["<B>"]}
    ci = T.load_class_index((local / "class_index.txt_2_bal").read_text())
    st = T.load_partition_stats((local / "partition_stat.txt_2_bal").read_text())
    sub = lambda part, uri: {"<a>": ["s_http___x_p_", "s_http___x_q_"], "<b>": ["o_http___x_p_"]}.get(uri, [])  # noqa: E731
    for name in ("bound.txt", "varpred.txt"):
        want = T.translate_query(name, (qdir / name).read_text(), lambda u: ni.get(u, []), ci, st, sub)
        assert (out / name).read_text() == want
    assert "UNION" in (out / "varpred.txt").read_text() and "partitions 00001-_3=_http___x_p_,00001-_3=_http___x_q_" in (out / "varpred.txt").read_text()
    # a second run does not translate its own output
    assert T.main(["ds", "2", base, str(qdir)]) == 0
    assert sorted(os.listdir(out)) == ["bound.txt", "varpred.txt"]


def test_native_parquet_writer_is_read_back_by_pyarrow(tmp_path):
    """The library's own writer (opus_amd/csrc/dx_parquet.cpp, host code), driven without a GPU by a small harness
    built with AddressSanitizer: three million rows over several row groups, two dictionary-encoded id columns and a
    plain string column, read back value by value."""
    import subprocess

    import pyarrow.parquet as pq
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pqh")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-I" + os.path.join(root, "opus_amd", "csrc"), "-I" + os.path.join(root, "include"), "-o", exe,
           os.path.join(root, "tests", "native", "parquet_writer_harness.cpp"), os.path.join(root, "opus_amd", "csrc", "dx_parquet.cpp")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if p.returncode != 0:
        pytest.skip("host build of the writer harness failed: " + p.stderr[-400:])
    T = 50000

    def term(k):
        return '"h\u00e9llo \U0001F642"@fr' if k == 1 else "<t%d>" % k + "x" * (k % 97)
    for rows in (0, 1, 9, 3000000):
        out = str(tmp_path / ("t%d.parquet" % rows))
        r = subprocess.run([exe, out, str(rows), str(T)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        t = pq.read_table(out)
        assert t.column_names == ["_1", "_2", "parts"] and t.num_rows == rows
        md = pq.ParquetFile(out).metadata
        assert md.num_row_groups >= (rows + (1 << 20) - 1) // (1 << 20) and md.num_rows == rows
        a, b, c = t["_1"].to_pylist(), t["_2"].to_pylist(), t["parts"].to_pylist()
        for i in list(range(0, rows, 9973)) + [x for x in (1, rows - 1, (1 << 20) - 1, 1 << 20) if 0 <= x < rows]:
            assert a[i] == term(i % T) and b[i] == term((7 * i + 3) % T) and c[i] == "row%d" % i, i
