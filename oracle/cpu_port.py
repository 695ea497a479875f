"""ctypes front end of oracle/cpu_port.cpp (TEST INFRASTRUCTURE ONLY: checker and CPU baseline)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(os.path.join(_HERE, "_ref", "libcpu_port.so"))
        lib.cpu_port_run.restype = ctypes.c_void_p
        lib.cpu_port_run.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        lib.cpu_port_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.cpu_port_free.argtypes = [ctypes.c_void_p]
        lib.cpu_port_measure_sizes.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.cpu_port_measures.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.cpu_port_rows_digest.restype = ctypes.c_uint64
        lib.cpu_port_rows_digest.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_uint64]
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


def run(text_u8: np.ndarray, class_index_text: str, n_partitions: int) -> dict:
    text_u8 = np.ascontiguousarray(text_u8, dtype=np.uint8)
    secs = ctypes.c_double(0)
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
