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


def test_native_import_of_a_layout_written_the_way_spark_writes_it(tmp_path, lubm1, monkeypatch):
    """The library reads a layout's Parquet files itself (dx_import_layout; dx_open_dataset falls back to it when the
    layout has no id columns beside it): its own files, and the same tables rewritten the way Spark leaves them --
    snappy, nullable columns, several part files per directory, no dx_<N> directory. Counts and bindings equal the
    oracle's; tables come and go by key without being read again."""
    import shutil

    import pyarrow.parquet as pq
    from opus_amd import lubm
    from opus_amd.engine import DX_QUERY_BINDINGS, Engine
    from opus_amd.host import parquet_io as P, translator as T
    from oracle.query_oracle import parse_translated_file, run_query
    ref = lubm1["oracle"]
    base = str(tmp_path / "own") + "/"
    os.makedirs(base)
    e = Engine(0)
    try:
        e.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
        monkeypatch.setenv("DX_PARQUET_CODEC", "uncompressed")
        e.write_layout(base, "lubm", str(tmp_path / "own"))
        # the library's writer with Spark's default codec (its own default too): same layout, the files say so by name, and they are smaller
        monkeypatch.delenv("DX_PARQUET_CODEC")
        os.makedirs(str(tmp_path / "own_snappy"))
        e.write_layout(str(tmp_path / "own_snappy") + "/", "lubm", str(tmp_path / "own_snappy"))
    finally:
        e.close()
    from parity_util import assert_layout_equals_oracle
    assert_layout_equals_oracle(str(tmp_path / "own_snappy") + "/", "lubm", 4, str(tmp_path / "own_snappy"), ref)

    def parquet_bytes(root):
        return sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(os.path.join(root, "lubm_data", "cluster")) for f in fs if f.endswith(".parquet"))
    names = [f for _, _, fs in os.walk(str(tmp_path / "own_snappy" / "lubm_data" / "cluster")) for f in fs if f.endswith(".parquet")]
    assert names and all(f.endswith(".snappy.parquet") for f in names)
    assert parquet_bytes(str(tmp_path / "own_snappy")) < 0.8 * parquet_bytes(str(tmp_path / "own"))
    # the same layout as Spark would have left it
    spark = str(tmp_path / "spark") + "/"
    src_root, dst_root = P.cluster_dir(base, "lubm", 4), P.cluster_dir(spark, "lubm", 4)
    n_rows = 0
    for key in P.vp_table_keys(base, "lubm", 4):
        part, sub = key.split("-", 1)
        t = P.read_table(os.path.join(src_root, part, sub))
        n_rows += t.num_rows
        if t.num_rows > 1:  # and in no particular order: the library's own tables are sorted by subject, Spark's are not
            t = t.take(np.random.default_rng(len(key) + t.num_rows).permutation(t.num_rows))
        d = os.path.join(dst_root, part, sub)
        os.makedirs(d)
        half = t.num_rows // 2
        pieces = [t.slice(0, half), t.slice(half)] if t.num_rows > 3 else [t]
        for i, piece in enumerate(pieces):
            pq.write_table(piece, os.path.join(d, "part-%05d-4e0c-c000.snappy.parquet" % i), compression="snappy", data_page_version="1.0")
        open(os.path.join(d, "_SUCCESS"), "w").close()
        open(os.path.join(d, ".part-00000-4e0c-c000.snappy.parquet.crc"), "wb").write(b"crc")
    assert not os.path.exists(os.path.join(spark, "lubm_data", "dx_4"))

    ni = {}
    for u, c in ref["node_index"]:
        ni.setdefault(u, []).append(c)
    cidx = T.load_class_index(lubm1["class_index"])
    for how in ("own files, dx_import_layout", "own snappy files, dx_import_layout", "spark files, dx_open_dataset"):
        e2 = Engine(0)
        # every join that may bisect its table does (DX_JOIN_LOOKUP=1): the library's own files come back in ascending
        # order of the subjects and may, the shuffled ones must be found out and joined by hashing
        if "snappy" in how or how.startswith("spark"):
            monkeypatch.setenv("DX_JOIN_LOOKUP", "1")
        else:
            monkeypatch.delenv("DX_JOIN_LOOKUP", raising=False)
        try:
            if how.startswith("own"):
                tables, rows = e2.import_layout(str(tmp_path / "own_snappy") + "/" if "snappy" in how else base, "lubm", 4)
                assert (tables, rows) == (len(ref["vp"]), n_rows)
            else:
                e2.open_dataset(spark, "lubm", 4)
            for name in ("Q2.txt", "Q4.txt", "Q7.txt", "Q9.txt", "Q14.txt"):
                tr = T.translate_query(name, lubm.queries(0)[name], lambda u: ni.get(u, []), cidx, ref["partition_stat"])
                want = run_query(ref["vp"], tr)
                qm, parts, _ = parse_translated_file(tr)
                for again in range(2):  # unloadTables, then the same tables by key once more: nothing is read again
                    e2.unload_all()
                    assert sum(e2.load_table(p) for p in parts) == want["load_data"], (how, name)
                    res = e2.query(next(iter(qm.values())), DX_QUERY_BINDINGS)
                    got, cols = res.rows()
                    assert res.count == want["count"] and cols == want["columns"] and got == want["rows"], (how, name)
                    res.close()
            with pytest.raises(Exception):
                e2.load_table("00000-_3=no_such_predicate")
        finally:
            e2.close()
    monkeypatch.delenv("DX_JOIN_LOOKUP", raising=False)
    # a layout that is not there
    e3 = Engine(0)
    try:
        with pytest.raises(Exception):
            e3.import_layout(str(tmp_path / "nowhere") + "/", "lubm", 4)
    finally:
        e3.close()
    shutil.rmtree(str(tmp_path / "spark"))
