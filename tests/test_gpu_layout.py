"""dx_part_write_layout with its pages encoded on the GPU: the files it leaves are read back with pyarrow (and by the
library's own reader elsewhere) and compared with the oracle's artefacts -- part files (DAP.scala:364-367), VP directories
(:433-437), node index (:405-412), VP index (:496-499), the text files beside them."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["uncompressed", "snappy"], autouse=True)
def codec(request, monkeypatch):
    """Every test of this file runs with plain pages and with the pages as snappy streams (the codec Spark writes by
    default, and this writer too; DX_PARQUET_CODEC=uncompressed switches it off; the streams are made on the GPU as well)."""
    monkeypatch.setenv("DX_PARQUET_CODEC", request.param)
    return request.param


def _layout_of(text, ci, n, tmp, name="ds"):
    from opus_amd.engine import Engine
    from oracle import dap_oracle
    base = str(tmp) + "/"
    e = Engine(0)
    try:
        e.partition_text(text.encode() if isinstance(text, str) else text, ci, n)
        e.write_layout(base, name, str(tmp))
        times = e.layout_times()
    finally:
        e.close()
    ref = dap_oracle.run_dap(text if isinstance(text, str) else text.tobytes().decode(), ci, n)
    return base, ref, times


def test_edge_case_graphs(tmp_path):
    """Every hand-made graph of the partitioner's edge cases as files: empty partitions, predicates that share a
    directory, predicates of 150 units and more (in the part files, in no directory), non-BMP characters, an input
    that places nothing at all."""
    import json
    from parity_util import assert_layout_equals_oracle
    from test_gpu_partitioner import EDGE_CASES
    cases = dict(EDGE_CASES)
    for name in ("appendix_c.json", "hand_derived_2.json"):   # the hand-derived examples as files too
        g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)))
        cases["golden " + name] = (g["instances"], g["class_index"], g["n_partitions"])
    # pages whose records do what the packing rules branch on: terms of 64 KiB and more (nothing can reach back over them),
    # neighbours of equal and of different length with long and with short common beginnings, equal rows of predicates
    T = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"
    long_a, long_b = "x" * 70000, "x" * 69990 + "y" * 10
    lines = ["<http://e/s%03d> %s <A> ." % (i, T) for i in range(40)]
    lines += ["<http://e/s%03d> <http://e/knows> <http://e/s%03d> ." % (i, (i * 7 + 1) % 40) for i in range(40)]
    lines += ['<http://e/s000> <http://e/note> "%s" .' % long_a, '<http://e/s001> <http://e/note> "%s" .' % long_b,
              '<http://e/s002> <http://e/note> "%s" .' % ("p" * 200), '<http://e/s003> <http://e/note> "%s" .' % ("p" * 199 + "q"),
              '<http://e/s004> <http://e/note> "%s" .' % ("p" * 130), '<http://e/s005> <http://e/note> "" .', '<http://e/s006> <http://e/note> "ab" .',
              '<http://e/s007> <http://e/note> "abc" .']
    cases["records the packing rules branch on"] = ("\n".join(lines) + "\n", "<A>\t0,1\n", 2)
    for i, name in enumerate(sorted(cases)):
        text, ci, n = cases[name]
        d = tmp_path / ("c%d" % i)
        d.mkdir()
        base, ref, _ = _layout_of(text, ci, n, d)
        assert_layout_equals_oracle(base, "ds", n, str(d), ref)


@pytest.mark.parametrize("batch_values,threads", [(None, None), (20000, 3), (1, 1)])
def test_lubm1_in_one_batch_and_in_many(tmp_path, lubm1, monkeypatch, batch_values, threads, codec):
    import pyarrow.parquet as pq
    from parity_util import assert_layout_equals_oracle
    if batch_values:
        monkeypatch.setenv("DX_LAYOUT_BATCH_VALUES", str(batch_values))
        monkeypatch.setenv("DX_WRITE_THREADS", str(threads))
    base, _, times = _layout_of(lubm1["bytes"], lubm1["class_index"], 4, tmp_path, "lubm")
    assert_layout_equals_oracle(base, "lubm", 4, str(tmp_path), lubm1["oracle"])
    on_disk = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(base) for f in fs if f.endswith((".parquet", ".dxcol", ".bin")))
    assert times["bytes"] == on_disk and times["threads"] == (threads or times["threads"])
    assert times["total_ms"] >= times["encode_ms"] > 0
    # the files say which codec they carry, in their names as Spark's do and in their metadata; packed pages are smaller
    parquet = [os.path.join(r, f) for r, _, fs in os.walk(base) for f in fs if f.endswith(".parquet")]
    assert all(f.endswith(".snappy.parquet") == (codec == "snappy") for f in parquet)
    biggest = max(parquet, key=os.path.getsize)
    md = pq.ParquetFile(biggest).metadata.row_group(0).column(0)
    assert md.compression == ("SNAPPY" if codec == "snappy" else "UNCOMPRESSED")
    if codec == "snappy":
        assert md.total_compressed_size < 0.6 * md.total_uncompressed_size
    # the id columns and the dictionary beside the Parquet files serve a fresh engine
    from opus_amd.engine import Engine
    e = Engine(0)
    try:
        e.open_dataset(base, "lubm", 4)
        key = sorted("%s-%s" % k for k in lubm1["oracle"]["vp"])[0]
        assert e.load_table(key) == len(lubm1["oracle"]["vp"][tuple(key.split("-", 1))])
    finally:
        e.close()


def test_swdf_shaped_many_small_tables(tmp_path):
    """120 predicates over 4 partitions: hundreds of small files in one pass of the encoder."""
    from opus_amd.workloads import hetero
    from parity_util import assert_layout_equals_oracle
    text, ci, n = hetero.generate("swdf", 20000, seed=3, n_partitions=4)
    base, ref, _ = _layout_of(text, ci, n, tmp_path)
    assert len(ref["vp"]) > 150
    assert_layout_equals_oracle(base, "ds", n, str(tmp_path), ref)


@pytest.mark.slow
def test_row_groups_of_a_large_partition(tmp_path):
    """LUBM-24 into four partitions: part files and the largest tables hold more than 2^20 rows, so they have several row
    groups, each with its own dictionary. Compared with the engine's own id-level accessors, decoded."""
    import pyarrow.parquet as pq
    from opus_amd import lubm
    from opus_amd.engine import Engine
    from opus_amd.host import parquet_io as P
    from oracle import dap_oracle
    arr, _ = lubm.generate(0, 24, seed=1, ontology=0)
    ci = lubm.class_index_text(4, 0)
    base = str(tmp_path) + "/"
    e = Engine(0)
    try:
        e.partition_text(arr, ci, 4)
        e.write_layout(base, "big", str(tmp_path))
        strs = np.array(e.decode(np.arange(e.term_count(), dtype=np.int64)), dtype=object)
        root = P.cluster_dir(base, "big", 4)
        seen_groups = 0
        for q in range(4):
            s, o, p = e.partition_rows(q)
            f = [x for x in os.listdir(root) if x.startswith("part-%05d-" % q)]
            assert len(f) == 1
            pf = pq.ParquetFile(os.path.join(root, f[0]))
            seen_groups = max(seen_groups, pf.metadata.num_row_groups)
            t = pf.read()
            assert t.num_rows == len(s)
            assert t["_1"].to_pylist() == strs[s].tolist() and t["_2"].to_pylist() == strs[o].tolist() and t["_3"].to_pylist() == strs[p].tolist()
        assert seen_groups >= 2
        for tinfo in e.tables():
            if not tinfo["in_layout"] or tinfo["rows"] < (1 << 19):
                continue
            s, o = e.table_rows(tinfo["index"], tinfo["rows"])
            d = os.path.join(root, dap_oracle.partition_name(tinfo["partition"]), "_3=" + dap_oracle.uri_to_storage(strs[tinfo["predicate"]]))
            t = P.read_table(d)
            assert t["_1"].to_pylist() == strs[s].tolist() and t["_2"].to_pylist() == strs[o].tolist()
        # VP index: rows of every partition against the id-level accessor
        for q in range(4):
            u, r, p = e.vp_index(q)
            want = {}
            for uu, rr, pp in zip(strs[u].tolist(), r.tolist(), strs[p].tolist()):
                want.setdefault(uu, set()).add(("s" if rr == 0 else "o") + dap_oracle.uri_to_storage(pp))
            t = P.read_table(os.path.join(base, "big_data", "classSubPart_index_4__%05d" % q))
            got = dict(zip(t["uri"].to_pylist(), t["parts"].to_pylist()))
            assert len(got) == t.num_rows == len(want)
            assert all(set(v.split(",")) == want[k] and v.split(",") == sorted(v.split(",")) for k, v in got.items())
    finally:
        e.close()


def test_a_failed_write_leaves_no_layout_that_looks_finished(tmp_path, lubm1):
    """A path the call cannot create: the error comes back, no _SUCCESS marker is left, and a second attempt is not
    refused with "already exists"."""
    from opus_amd.engine import DxError, Engine
    base = str(tmp_path) + "/"
    e = Engine(0)
    try:
        e.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
        os.makedirs(os.path.join(base, "lubm_data"))
        open(os.path.join(base, "lubm_data", "node_index_4"), "w").write("in the way")  # a file where a directory must go
        with pytest.raises(DxError):
            e.write_layout(base, "lubm", str(tmp_path))
        assert not any(f == "_SUCCESS" for _, _, fs in os.walk(base) for f in fs)
        assert not os.path.exists(os.path.join(base, "lubm_data", "cluster", "clusters_4_bal"))
        os.remove(os.path.join(base, "lubm_data", "node_index_4"))
        e.write_layout(base, "lubm", str(tmp_path))
    finally:
        e.close()
    from parity_util import assert_layout_equals_oracle
    assert_layout_equals_oracle(base, "lubm", 4, str(tmp_path), lubm1["oracle"])
