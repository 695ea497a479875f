"""ctypes front end of oracle/cpu_port.cpp (TEST INFRASTRUCTURE ONLY: checker and CPU baseline)."""
import ctypes
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(os.path.join(_HERE, "_build", "libcpu_port.so"))
        lib.cpu_port_run.restype = ctypes.c_void_p
        lib.cpu_port_run.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        lib.cpu_port_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.cpu_port_free.argtypes = [ctypes.c_void_p]
        lib.cpu_port_measure_sizes.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.cpu_port_measures.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.cpu_port_rows_digest.restype = ctypes.c_uint64
        lib.cpu_port_rows_digest.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_uint64]
        lib.cpu_port_run_mt.restype = ctypes.c_void_p
        lib.cpu_port_run_mt.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_double)]
        lib.cpu_port_table_rows.restype = ctypes.c_int64
        lib.cpu_port_table_rows.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        lib.cpu_port_query.restype = ctypes.c_int64
        lib.cpu_port_query.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        _LIB = lib
    return _LIB


FNV_OFF, FNV_PRIME, M64 = 1469598103934665603, 1099511628211, (1 << 64) - 1


def _fnv(b: bytes) -> int:
    h = FNV_OFF
    for c in b:
        h = ((h ^ c) * FNV_PRIME) & M64
    return h


def _mix(h: int) -> int:
    h ^= h >> 33
    h = (h * 0xFF51AFD7ED558CCD) & M64
    h ^= h >> 33
    h = (h * 0xC4CEB9FE1A85EC53) & M64
    return h ^ (h >> 33)


def row_digest(s: str, o: str, p: str) -> int:
    """The per-row term of cpu_port.cpp's order-independent partition digest."""
    e = lambda x: x.encode("utf-8", "surrogateescape")  # noqa: E731
    return _mix(_fnv(e(s)) ^ _mix((_fnv(e(o)) + 0x9E3779B97F4A7C15) & M64) ^ _mix((_fnv(e(p)) + 0x632BE59BD9B4E019) & M64))


def run(text_u8: np.ndarray, class_index_text: str, n_partitions: int, threads: int = 1) -> dict:
    """threads > 1: the same restatement on that many threads with one shared dictionary (same statistics and digests)."""
    text_u8 = np.ascontiguousarray(text_u8, dtype=np.uint8)
    secs = ctypes.c_double(0)
    if threads > 1:
        h = _lib().cpu_port_run_mt(text_u8.ctypes.data, text_u8.size, class_index_text.encode(), n_partitions, threads, 0, ctypes.byref(secs))
    else:
        h = _lib().cpu_port_run(text_u8.ctypes.data, text_u8.size, class_index_text.encode(), n_partitions, ctypes.byref(secs))
    out = np.zeros(6 + 2 * n_partitions, dtype=np.uint64)
    _lib().cpu_port_stats(h, out.ctypes.data)
    sizes = np.zeros(2, dtype=np.uint64)
    _lib().cpu_port_measure_sizes(h, sizes.ctypes.data)
    cls, cc = np.zeros((int(sizes[0]), 3), dtype=np.uint64), np.zeros((int(sizes[1]), 5), dtype=np.uint64)
    _lib().cpu_port_measures(h, cls.ctypes.data, cc.ctypes.data)
    _lib().cpu_port_free(h)
    if out[0]:
        raise ValueError("malformed N-Triples input")
    return {"measures": measures_by_hash(cls, cc), "seconds": secs.value, "n_triples": int(out[1]), "n_vertices": int(out[2]), "n_edges": int(out[3]),
            "n_node_index": int(out[4]), "n_tables": int(out[5]),
            "part_rows": [int(x) for x in out[6:6 + n_partitions]],
            "part_digest": [int(x) for x in out[6 + n_partitions:6 + 2 * n_partitions]]}


def measures_by_hash(cls, cc):
    """{"classes": {hash(class): (node freq, node count)}, "cc": {(hash(subType), hash(pred), hash(objType)): (rows, dist)}}"""
    return {"classes": {int(r[0]): (int(r[1]), int(r[2])) for r in cls},
            "cc": {(int(r[0]), int(r[1]), int(r[2])): (int(r[3]), int(r[4])) for r in cc}}


def term_hash(b: bytes) -> int:
    """FNV-1a of a term's bytes: the key of measures_by_hash."""
    return _fnv(b)


def rows_digest(pool: np.ndarray, offs: np.ndarray, s: np.ndarray, o: np.ndarray, p: np.ndarray) -> int:
    """Digest of id-encoded rows over a string pool (term i = pool[offs[i]:offs[i+1]]), comparable with part_digest."""
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    s, o, p = (np.ascontiguousarray(a, dtype=np.int64) for a in (s, o, p))
    return int(_lib().cpu_port_rows_digest(pool.ctypes.data, offs.ctypes.data, offs.size - 1, s.ctypes.data, o.ctypes.data,
                                           p.ctypes.data, s.size))


_SUB = re.compile(r"\(SELECT (.*?) FROM (\S+?)(?: WHERE (.*?))?\) AS (tab\d+)")
_FILTER = re.compile(r"([so]) == '(.*?)'(?= AND [so] == '|$)")


def parse_statement(sql: str, partitions):
    """The sub-selects of a translated statement without UNION (QueryTranslator.scala:363-392, 417-455) as
    (table key, s constant, o constant, s variable, o variable); equal variable names join."""
    key_of = {"table" + p.replace("-", "_").replace("=", "_E_").strip(): p for p in partitions}
    subs, var_no = [], {}
    body = sql.split(" FROM ", 1)[1]
    for proj, table, where, _alias in _SUB.findall(body):
        s_var = o_var = -1
        for item in proj.split(","):
            col, name = item.strip().split(" AS ")
            no = var_no.setdefault(name.strip(), len(var_no))
            if col.strip() == "s":
                s_var = no
            elif col.strip() == "o":
                o_var = no
            else:
                raise ValueError("projected constant: not a plain basic graph pattern")
        consts = dict(_FILTER.findall(where)) if where else {}
        subs.append((key_of[table], consts.get("s"), consts.get("o"), s_var, o_var))
    if "UNION" in sql or not subs:
        raise ValueError("statement outside the subset the CPU query leg runs")
    return subs


class Dataset:
    """A partitioning kept for queries: the vertical-partition tables of cpu_port.cpp over the given text."""

    def __init__(self, text_u8: np.ndarray, class_index_text: str, n_partitions: int, threads: int = 1):
        self._text = np.ascontiguousarray(text_u8, dtype=np.uint8)  # the tables' strings point into it
        secs = ctypes.c_double(0)
        self._h = _lib().cpu_port_run_mt(self._text.ctypes.data, self._text.size, class_index_text.encode(), n_partitions, threads, 1,
                                         ctypes.byref(secs))
        self.seconds = secs.value
        out = np.zeros(6 + 2 * n_partitions, dtype=np.uint64)
        _lib().cpu_port_stats(self._h, out.ctypes.data)
        if out[0]:
            self.close()
            raise ValueError("malformed N-Triples input")
        self.n_triples = int(out[1])

    def table_rows(self, key: str) -> int:
        return int(_lib().cpu_port_table_rows(self._h, key.encode("utf-8", "surrogateescape")))

    def count(self, sql: str, partitions, threads: int = 1, string_keyed: bool = False):
        """(count, seconds) of the statement: scans with constant filters, hash joins on the shared variables, count."""
        subs = parse_statement(sql, partitions)
        n = len(subs)
        enc = lambda x: None if x is None else x.encode("utf-8", "surrogateescape")  # noqa: E731
        tables = (ctypes.c_char_p * n)(*[enc(s[0]) for s in subs])
        sc = (ctypes.c_char_p * n)(*[enc(s[1]) for s in subs])
        oc = (ctypes.c_char_p * n)(*[enc(s[2]) for s in subs])
        sv = (ctypes.c_int * n)(*[s[3] for s in subs])
        ov = (ctypes.c_int * n)(*[s[4] for s in subs])
        secs = ctypes.c_double(0)
        c = _lib().cpu_port_query(self._h, n, tables, sc, oc, sv, ov, threads, 1 if string_keyed else 0, ctypes.byref(secs))
        if c < 0:
            raise KeyError("statement names a table the partitioning does not have")
        return int(c), secs.value

    def close(self):
        if self._h:
            _lib().cpu_port_free(self._h)
            self._h = None

    def __del__(self):
        self.close()
