"""The reference's on-disk layout in Parquet, written and read on the host with pyarrow.

What the Scala tools write and read (paths from Constatnts.scala:9-40):
  <base><dataset>_data/cluster/clusters_<N>_bal/part-0000i-<uuid>-c000.snappy.parquet   (_1=s, _2=o, _3=p)   DAP.scala:364-367
  <base><dataset>_data/cluster/clusters_<N>_bal/0000i/_3=<dir>/part-*.snappy.parquet     (_1=s, _2=o)         DAP.scala:433-437
  <base><dataset>_data/node_index_<N>/part-*.snappy.parquet                              (uri, uriType)       DAP.scala:405-412
  <base><dataset>_data/classSubPart_index_<N>__0000i/part-*.snappy.parquet               (uri, parts)         DAP.scala:496-499
  <local><dataset>_local/statistics.txt_<N>, partition_stat.txt_<N>_bal                  text                 DAP.scala:373-381, 101-114
every directory Spark writes carries an empty _SUCCESS file. All columns are strings (raw N-Triples terms).

The writers take plain arrays (term ids plus the term strings), so they serve the engine's results as
well as any other producer; `export_layout` / `import_layout` connect them to an Engine. Parquet encoding
and decoding stay on the host: the GPU sees term ids only.
"""
from __future__ import annotations

import glob
import os
import re
import uuid
from collections import defaultdict

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.parquet as pq


def uri_to_storage(uri: str) -> str:
    """DAP.uriToStorage (DAP.scala:512-514)."""
    return re.sub("[^A-Za-z0-9]", "_", uri).lower()


def partition_name(q: int) -> str:
    """"part-0000i-..." -> "0000i" (DAP.scala:427-428)."""
    return "%05d" % q


def cluster_dir(base: str, dataset: str, n: int) -> str:
    return os.path.join(base, dataset + "_data", "cluster", "clusters_%d_bal" % n)


class Terms:
    """All term strings of a dictionary as one Arrow array; columns are cut out of it by id."""

    def __init__(self, data: np.ndarray, offsets: np.ndarray):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        self.array = pa.LargeStringArray.from_buffers(len(offsets) - 1, pa.py_buffer(offsets), pa.py_buffer(data))

    @classmethod
    def from_strings(cls, strings):
        t = cls.__new__(cls)
        t.array = pa.array(list(strings), type=pa.large_string())
        return t

    def column(self, ids: np.ndarray) -> pa.Array:
        """Strings of `ids`, dictionary-encoded over the ids that occur (Parquet then stores each once)."""
        ids = np.asarray(ids, dtype=np.int64)
        uniq, inv = np.unique(ids, return_inverse=True)
        return pa.DictionaryArray.from_arrays(pa.array(inv.astype(np.int32)), self.array.take(pa.array(uniq)).cast(pa.string()))


def _write(dirpath: str, columns: dict, stem: str = "part-00000"):
    os.makedirs(dirpath, exist_ok=True)
    name = "%s-%s-c000.snappy.parquet" % (stem, uuid.uuid4())
    table = pa.table(columns)
    # plain string columns for the reader: no Arrow schema blob, dictionary pages + snappy like Spark's writer
    pq.write_table(table, os.path.join(dirpath, name), compression="snappy", use_dictionary=True, store_schema=False)
    open(os.path.join(dirpath, "_SUCCESS"), "w").close()
    return name


def write_partition_file(base, dataset, n, q, terms: Terms, s, o, p):
    """One first-level partition as the part file partitionClusters writes (DAP.scala:364-367)."""
    return _write(cluster_dir(base, dataset, n), {"_1": terms.column(s), "_2": terms.column(o), "_3": terms.column(p)},
                  stem="part-%05d" % q)


def write_vp_table(base, dataset, n, q, predicate: str, terms: Terms, s, o):
    """One vertical-partition table under <partition>/_3=<sanitised predicate>/ (DAP.scala:433-437). Predicates
    whose names sanitise alike land in one directory as several files, as Spark's partitionBy does."""
    d = os.path.join(cluster_dir(base, dataset, n), partition_name(q), "_3=" + uri_to_storage(predicate))
    name = _write(d, {"_1": terms.column(s), "_2": terms.column(o)})
    open(os.path.join(cluster_dir(base, dataset, n), partition_name(q), "_SUCCESS"), "w").close()
    return name


def write_node_index(base, dataset, n, terms: Terms, uri, uri_type):
    return _write(os.path.join(base, dataset + "_data", "node_index_%d" % n), {"uri": terms.column(uri), "uriType": terms.column(uri_type)})


def write_vp_index(base, dataset, n, q, terms: Terms, uri, role, predicate_strings):
    """classSubPart_index: one row per uri with the comma-joined set of 's'/'o' + sanitised predicate
    (DAP.scala:479-499; the set's order is unspecified there, sorted here)."""
    groups = defaultdict(set)
    for u, r, ps in zip(np.asarray(uri).tolist(), np.asarray(role).tolist(), predicate_strings):
        groups[u].add(("s" if r == 0 else "o") + uri_to_storage(ps))
    us = np.fromiter(groups.keys(), dtype=np.int64, count=len(groups))
    parts = pa.array([",".join(sorted(groups[u])) for u in us.tolist()], type=pa.string())
    d = os.path.join(base, dataset + "_data", "classSubPart_index_%d__%s" % (n, partition_name(q)))
    return _write(d, {"uri": terms.column(us), "parts": parts})


def export_layout(engine, base: str, dataset: str, local_dir: str | None = None):
    """Everything the partitioner holds in HBM, as the files DAP.main leaves behind (DAP.scala:85-94)."""
    n = engine.partition_count()
    data, offs = engine.decode_raw(np.arange(engine.term_count(), dtype=np.int64))
    terms = Terms(data, offs)
    strs = None
    statistics, stat = [], {}
    for q in range(n):
        s, o, p = engine.partition_rows(q)
        if len(s) == 0:
            continue
        name = write_partition_file(base, dataset, n, q, terms, s, o, p)
        statistics.append("%s %d" % ("0%d" % q if q < 10 else str(q), len(s)))
        stat[name] = len(s)
    pred_cache = {}
    for t in engine.tables():
        if not t["in_layout"]:
            continue
        s, o = engine.table_rows(t["index"], t["rows"])
        if t["predicate"] not in pred_cache:
            pred_cache[t["predicate"]] = engine.decode([t["predicate"]])[0]
        write_vp_table(base, dataset, n, t["partition"], pred_cache[t["predicate"]], terms, s, o)
        stat[partition_name(t["partition"])] = stat.get(partition_name(t["partition"]), 0) + t["rows"]
    u, ty = engine.node_index()
    write_node_index(base, dataset, n, terms, u, ty)
    for q in range(n):
        u, r, p = engine.vp_index(q)
        if len(u) == 0 and engine.partition_rows(q)[0].size == 0:
            continue
        if strs is None:
            strs = {}
        for pid in np.unique(p).tolist():
            if pid not in strs:
                strs[pid] = engine.decode([pid])[0]
        write_vp_index(base, dataset, n, q, terms, u, r, [strs[x] for x in p.tolist()])
    local = os.path.join(local_dir if local_dir is not None else os.getcwd(), dataset + "_local")
    os.makedirs(local, exist_ok=True)
    with open(os.path.join(local, "statistics.txt_%d" % n), "w") as f:
        f.write("".join(line + "\n" for line in statistics))
    with open(os.path.join(local, "partition_stat.txt_%d_bal" % n), "w") as f:
        f.write("".join("%s %d\n" % kv for kv in sorted(stat.items())))
    return stat


def read_table(dirpath: str) -> pa.Table:
    """All part files of one Spark output directory."""
    files = sorted(glob.glob(os.path.join(dirpath, "part-*.parquet")))
    if not files:
        raise FileNotFoundError("no part files in " + dirpath)
    return pa.concat_tables([pq.read_table(f) for f in files])


def vp_table_keys(base: str, dataset: str, n: int) -> list:
    """The "PPPPP-_3=<dir>" keys of every VP table of a layout."""
    root = cluster_dir(base, dataset, n)
    out = []
    for part in sorted(os.listdir(root)):
        if re.fullmatch(r"\d{5}", part) and os.path.isdir(os.path.join(root, part)):
            out += ["%s-%s" % (part, sub) for sub in sorted(os.listdir(os.path.join(root, part))) if sub.startswith("_3=")]
    return out


def import_layout(engine, base: str, dataset: str, n: int, keys=None) -> dict:
    """Loads VP tables written by the Scala tools (or by export_layout) into the engine: decodes the Parquet
    strings on the host, builds one dictionary over the terms of the requested tables, installs it and
    registers every table by ids (QueryProcessor.loadDataset, QueryProcessor.scala:308-339). Returns
    {key: rows}."""
    keys = list(keys) if keys is not None else vp_table_keys(base, dataset, n)
    root = cluster_dir(base, dataset, n)
    tables = {}
    for key in keys:
        part, sub = key.split("-", 1)
        t = read_table(os.path.join(root, part, sub))
        tables[key] = (pc.cast(t.column("_1"), pa.large_string()), pc.cast(t.column("_2"), pa.large_string()))
    chunks = [c for so in tables.values() for col in so for c in col.chunks]
    dictionary = pc.unique(pa.chunked_array(chunks, type=pa.large_string())) if chunks else pa.array([], type=pa.large_string())
    bufs = dictionary.buffers()
    offsets = np.frombuffer(bufs[1], dtype=np.int64, count=len(dictionary) + 1 + dictionary.offset)[dictionary.offset:]
    data = np.frombuffer(bufs[2], dtype=np.uint8) if bufs[2] is not None else np.zeros(0, dtype=np.uint8)
    base_off = int(offsets[0]) if len(offsets) else 0
    engine.dictionary_load(data[base_off:int(offsets[-1])] if len(offsets) else data, (offsets - base_off).astype(np.uint64))
    rows = {}
    for key, (s, o) in tables.items():
        sid = pc.index_in(s, value_set=dictionary).to_numpy(zero_copy_only=False).astype(np.int64)
        oid = pc.index_in(o, value_set=dictionary).to_numpy(zero_copy_only=False).astype(np.int64)
        engine.load_table_ids(key, sid, oid)
        rows[key] = len(sid)
    return rows
