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
    # "plan": the file laid out by plan_parquet_file from page sizes, the bodies dropped in (what the GPU encoder's host side does)
    # "plan-snappy": the same with every page a snappy stream that carries its body as one literal; "plan-packed": the pages
    # of terms packed by the rules the device follows (restated in the harness), read here by pyarrow's snappy
    for rows, codec in ((0, ""), (1, ""), (9, "snappy"), (3000000, ""), (3000000, "snappy"), (0, "snappy"), (0, "plan"), (1, "plan"), (9, "plan"),
                        (2500000, "plan"), (0, "plan-snappy"), (1, "plan-snappy"), (2500000, "plan-snappy"), (1, "plan-packed"), (9, "plan-packed"),
                        (2500000, "plan-packed")):
        out = str(tmp_path / ("t%d%s.parquet" % (rows, codec)))
        r = subprocess.run([exe, out, str(rows), str(T)] + ([codec] if codec.startswith("plan") else []), capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, DX_PARQUET_CODEC=codec if codec == "snappy" else "uncompressed"))
        assert r.returncode == 0, r.stderr
        t = pq.read_table(out)
        assert t.column_names == ["_1", "_2", "parts"] and t.num_rows == rows
        md = pq.ParquetFile(out).metadata
        if rows:
            assert md.row_group(0).column(0).compression == ("SNAPPY" if codec in ("snappy", "plan-snappy", "plan-packed") else "UNCOMPRESSED")
        if rows == 2500000 and codec == "plan-packed":  # "row<i>" after "row<i-1>", "<t<k>>xxx" after "<t<k-1>>xx": mostly copies
            assert os.path.getsize(out) < 0.7 * os.path.getsize(str(tmp_path / "t2500000plan-snappy.parquet"))
        if rows == 3000000 and codec == "snappy":  # the terms of the test compress well; the bit-packed indices hardly
            assert os.path.getsize(out) < 0.7 * os.path.getsize(str(tmp_path / "t3000000.parquet"))
        assert md.num_row_groups >= (rows + (1 << 20) - 1) // (1 << 20) and md.num_rows == rows
        a, b, c = t["_1"].to_pylist(), t["_2"].to_pylist(), t["parts"].to_pylist()
        for i in list(range(0, rows, 9973)) + [x for x in (1, rows - 1, (1 << 20) - 1, 1 << 20) if 0 <= x < rows]:
            assert a[i] == term(i % T) and b[i] == term((7 * i + 3) % T) and c[i] == "row%d" % i, i


def _read_with_harness(exe, path, cols, tmp_path):
    import subprocess
    out = str(tmp_path / "dump.txt")
    r = subprocess.run([exe, path, out] + cols, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return None, r.stderr
    data = open(out, "rb").read()
    per_col = [[] for _ in cols]
    at = 0
    summary = None
    while at < len(data):
        if data.startswith(b"chunks ", at):
            summary = data[at:].decode().split()
            break
        t1 = data.index(b"\t", at)
        t2 = data.index(b"\t", t1 + 1)
        col, n = int(data[at:t1]), int(data[t1 + 1:t2])
        per_col[col].append(data[t2 + 1:t2 + 1 + n])
        at = t2 + 1 + n + 1
    return per_col, dict(zip(summary[0::2], map(int, summary[1::2])))


def test_native_parquet_reader_reads_what_spark_style_writers_write(tmp_path):
    """The library's own reader (opus_amd/csrc/dx_parquet_read.cpp, host code), driven without a GPU by a small
    harness built with AddressSanitizer, against files written by pyarrow's parquet-cpp in the shapes a layout can
    come in: snappy and uncompressed, dictionary pages and plain pages (and the fall-back from one to the other inside
    a chunk), data pages of both versions, nullable and required columns, several row groups, small pages, an empty
    file; and against the library's own writer. Unsupported things are refused, not misread."""
    import subprocess

    import pyarrow as pa
    import pyarrow.parquet as pq
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pqr")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-I" + os.path.join(root, "opus_amd", "csrc"), "-I" + os.path.join(root, "include"), "-o", exe,
           os.path.join(root, "tests", "native", "parquet_reader_harness.cpp"), os.path.join(root, "opus_amd", "csrc", "dx_parquet_read.cpp")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if p.returncode != 0:
        pytest.skip("host build of the reader harness failed: " + p.stderr[-400:])
    rng = np.random.default_rng(5)
    n = 200000
    terms = ['"h\u00e9llo \U0001F642"@fr', "", "<a>"] + ["<http://example.org/t%d>" % k + "y" * (k % 61) for k in range(30000)]
    s = [terms[i] for i in rng.integers(0, len(terms), n)]
    o = [terms[i] for i in rng.integers(0, 40, n)]                      # few distinct values: narrow indices, long runs
    u = ["<unique/%d>" % i + "z" * (i % 300) for i in range(n)]          # all distinct: the writer gives the dictionary up
    table = pa.table({"_1": s, "_2": o, "extra": u})
    required = pa.schema([pa.field(name, pa.string(), nullable=False) for name in ("_1", "_2", "extra")])
    cases = {
        "snappy_v1": dict(compression="snappy", data_page_version="1.0", row_group_size=70000),
        "plain_v1": dict(compression="none", use_dictionary=False, data_page_version="1.0"),
        "snappy_v2": dict(compression="snappy", data_page_version="2.0", data_page_size=4096),
        "none_v2_smalldict": dict(compression="none", data_page_version="2.0", dictionary_pagesize_limit=20000, row_group_size=50000),
    }
    want = [[x.encode() for x in col] for col in (s, o, u)]
    for name, kw in cases.items():
        for label, tab in (("nullable", table), ("required", table.cast(required))):
            path = str(tmp_path / (name + "_" + label + ".parquet"))
            pq.write_table(tab, path, **kw)
            got, info = _read_with_harness(exe, path, ["_1", "_2", "extra"], tmp_path)
            assert got is not None, (name, label, info)
            assert got == want, (name, label)
            assert info["indexed"] + info["plain"] == 3 * n
            if name == "plain_v1":
                assert info["dictionaries"] == 0 and info["indexed"] == 0
            else:
                assert info["dictionaries"] >= 3 and info["indexed"] >= n and info["plain"] >= n  # `extra` gives its dictionary up
    # column order and a subset of the columns
    got, _ = _read_with_harness(exe, str(tmp_path / "snappy_v1_nullable.parquet"), ["_2", "_1"], tmp_path)
    assert got == [want[1], want[0]]
    # no rows
    pq.write_table(table.slice(0, 0), str(tmp_path / "empty.parquet"), compression="snappy")
    got, info = _read_with_harness(exe, str(tmp_path / "empty.parquet"), ["_1", "_2"], tmp_path)
    assert got == [[], []]
    # the library's own writer
    wexe = str(tmp_path / "pqw")
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "opus_amd", "csrc"),
           "-I" + os.path.join(root, "include"), "-o", wexe, os.path.join(root, "tests", "native", "parquet_writer_harness.cpp"),
           os.path.join(root, "opus_amd", "csrc", "dx_parquet.cpp")]
    assert subprocess.run(cmd, capture_output=True, text=True, timeout=300).returncode == 0
    for codec in ("", "snappy"):
        own = str(tmp_path / ("own%s.parquet" % codec))
        assert subprocess.run([wexe, own, "1200000", "5000"], capture_output=True, timeout=300, env=dict(os.environ, DX_PARQUET_CODEC=codec or "uncompressed")).returncode == 0
        got, info = _read_with_harness(exe, own, ["_1", "_2", "parts"], tmp_path)
        t = pq.read_table(own)
        assert got == [[x.encode() for x in t[c].to_pylist()] for c in ("_1", "_2", "parts")]
    # refused: a null, a codec that is not read, a column that is not a string, a column that is not there, not a Parquet file
    pq.write_table(pa.table({"_1": ["a", None], "_2": ["b", "c"]}), str(tmp_path / "null.parquet"))
    got, err = _read_with_harness(exe, str(tmp_path / "null.parquet"), ["_1", "_2"], tmp_path)
    assert got is None and "null" in err
    pq.write_table(table.slice(0, 10), str(tmp_path / "gzip.parquet"), compression="gzip")
    got, err = _read_with_harness(exe, str(tmp_path / "gzip.parquet"), ["_1"], tmp_path)
    assert got is None and "codec" in err
    pq.write_table(pa.table({"_1": [1, 2]}), str(tmp_path / "ints.parquet"))
    got, err = _read_with_harness(exe, str(tmp_path / "ints.parquet"), ["_1"], tmp_path)
    assert got is None and "not a string column" in err
    got, err = _read_with_harness(exe, str(tmp_path / "ints.parquet"), ["nope"], tmp_path)
    assert got is None and "no column" in err
    open(str(tmp_path / "junk.parquet"), "wb").write(b"PAR1 this is not a parquet file PAR1")
    got, err = _read_with_harness(exe, str(tmp_path / "junk.parquet"), ["_1"], tmp_path)
    assert got is None
    # damaged files: any outcome but a memory error (the harness runs under AddressSanitizer and UBSan)
    small = str(tmp_path / "small.parquet")
    pq.write_table(table.slice(0, 3000), small, compression="snappy", data_page_size=2048, row_group_size=1000)
    good = bytearray(open(small, "rb").read())
    footer_len = int.from_bytes(good[-8:-4], "little")
    for trial in range(300):
        b = bytearray(good)
        lo = len(b) - 8 - footer_len if trial % 2 else 4          # every other trial damages the footer
        for _ in range(1 + trial % 3):
            b[int(rng.integers(lo, len(b) - 8))] = int(rng.integers(0, 256))
        open(str(tmp_path / "fuzz.parquet"), "wb").write(bytes(b))
        r = subprocess.run([exe, str(tmp_path / "fuzz.parquet"), str(tmp_path / "fuzz.txt"), "_1", "_2", "extra"], capture_output=True, text=True, timeout=120)
        assert r.returncode in (0, 1) and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (trial, r.stderr[-400:])
    # truncated files never read outside the mapping
    blob = open(str(tmp_path / "snappy_v1_nullable.parquet"), "rb").read()
    for cut in (len(blob) // 3, len(blob) - 9, len(blob) - 100):
        open(str(tmp_path / "cut.parquet"), "wb").write(blob[:cut] + blob[-8:])
        got, err = _read_with_harness(exe, str(tmp_path / "cut.parquet"), ["_1", "_2"], tmp_path)
        assert got is None and "AddressSanitizer" not in err, err[-300:]
