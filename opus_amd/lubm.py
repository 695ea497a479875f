"""ctypes front end of the LUBM-shaped N-Triples generator (tools/lubmgen.cpp)."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

WORKLOAD_DIR = os.path.join(_HERE, "workloads", "lubm")
ONTOLOGY_TAG = {0: "zhp2", 1: "swat"}
QUERY_DIR = {0: "queries_lubm100_1300_2300", 1: "queries_lubm10240"}


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_lib", "liblubmgen.so")
        if not os.path.exists(path):
            raise RuntimeError("liblubmgen.so is not built: run `make -C opus_amd/csrc` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.lubm_generate.restype = ctypes.c_void_p
        lib.lubm_generate.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32,
                                      ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        lib.lubm_free.argtypes = [ctypes.c_void_p]
        _LIB = lib
    return _LIB


def generate(univ_begin: int, univ_end: int, seed: int = 0, ontology: int = 0, degree_pool: int = 1000):
    """Universities [univ_begin, univ_end) as one uint8 numpy array of N-Triples
    text plus the triple count."""
    nb, nt = ctypes.c_uint64(0), ctypes.c_uint64(0)
    p = _lib().lubm_generate(univ_begin, univ_end, seed, ontology, degree_pool, ctypes.byref(nb), ctypes.byref(nt))
    if not p:
        raise RuntimeError("lubm_generate failed")
    try:
        arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(nb.value,)).copy()
    finally:
        _lib().lubm_free(p)
    return arr, nt.value


def class_index_text(n_partitions: int, ontology: int = 0) -> str:
    with open(os.path.join(WORKLOAD_DIR, "class_index_%s.txt_%d_bal" % (ONTOLOGY_TAG[ontology], n_partitions))) as f:
        return f.read()


def queries(ontology: int = 0) -> dict:
    """{file name: SPARQL text} of the reference's LUBM query folder for that namespace."""
    d = os.path.join(WORKLOAD_DIR, QUERY_DIR[ontology])
    return {n: open(os.path.join(d, n)).read() for n in sorted(os.listdir(d))}
