"""ctypes binding of libdiaeresis_hip.so (the C ABI declared in include/diaeresis_hip.h).

There is no CPU fallback: if the library is missing or no GPU is visible, every
entry point raises. PyTorch is used by callers only to hold device memory; the
library itself takes raw device pointers.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libdiaeresis_hip.so")

DX_QUERY_COUNT = 0
DX_QUERY_BINDINGS = 1

ERRORS = {-2: "DX_E_NOENT", -5: "DX_E_IO", -12: "DX_E_NOMEM", -22: "DX_E_INVAL", -34: "DX_E_RANGE",
          -74: "DX_E_MALFORMED", -95: "DX_E_UNSUPPORTED", -1000: "DX_E_HIP"}

# every symbol include/diaeresis_hip.h declares
EXPORTS = [
    "dx_init", "dx_free", "dx_last_error", "dx_partition_device", "dx_partition_file", "dx_part_stage_name",
    "dx_term_count", "dx_terms_bytes", "dx_terms_decode", "dx_terms_lookup", "dx_part_count", "dx_part_rows",
    "dx_part_fetch", "dx_table_count", "dx_table_info_at", "dx_table_fetch", "dx_node_index_rows",
    "dx_node_index_fetch", "dx_vp_index_rows", "dx_vp_index_fetch", "dx_instance_graph_sizes",
    "dx_instance_graph_fetch", "dx_part_write_layout", "dx_layout_times", "dx_open_dataset", "dx_load_table", "dx_unload_all",
    "dx_query", "dx_query_joined", "dx_result_count", "dx_result_columns", "dx_result_column_name", "dx_result_fetch_ids",
    "dx_result_virtual_term", "dx_result_plan", "dx_result_device_ms", "dx_result_free",
    "dx_dictionary_load", "dx_load_table_ids", "dx_set_comm", "dx_declare_remote_table", "dx_table_stats", "dx_set_table_stats", "dx_result_local_rows",
    "dx_measures_compute", "dx_measures_classes", "dx_measures_cc", "dx_measures_write", "dx_load_tables",
    "dx_query_count_noalloc", "dx_allocator_stats", "dx_rccl_unique_id", "dx_comm_rccl_init", "dx_comm_rccl_init_file", "dx_comm_rccl_init_job", "dx_comm_selftest",
    "dx_load_tables_sharded", "dx_import_layout",
]


class DxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "DX_E_?"), code, message))
        self.code = code


class PartStats(ctypes.Structure):
    _fields_ = [("n_bytes", c_uint64), ("n_triples", c_uint64), ("n_unique", c_uint64), ("n_terms", c_uint64),
                ("n_vertices", c_uint64), ("n_edges", c_uint64), ("n_rows", c_uint64), ("n_node_index", c_uint64),
                ("n_vp_index", c_uint64), ("n_predicates", c_uint32), ("n_tables", c_uint32),
                ("hash_retries", c_uint32), ("n_dict_entries", c_uint32), ("ms_total", c_double), ("ms_stage", c_double * 12)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "ms_stage"}
        lib = load_library()
        d["ms_stage"] = {lib.dx_part_stage_name(i).decode(): self.ms_stage[i] for i in range(12)
                         if lib.dx_part_stage_name(i)}
        return d


class MeasureSizes(ctypes.Structure):
    _fields_ = [("n_classes", c_uint64), ("n_cc", c_uint64), ("n_type_entries", c_uint64), ("n_cc_rows", c_uint64),
                ("device_ms", c_double)]


class TableInfo(ctypes.Structure):
    _fields_ = [("partition", c_int32), ("in_layout", c_int32), ("predicate", c_int64), ("rows", c_int64)]


_LIB = None


def load_library():
    """Loads the shared library and declares the signatures. Raises if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: build it with `make -C opus_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    # PyTorch wheels bundle their own libamdhip64.so.7. A process must hold exactly one HIP runtime:
    # importing torch first makes the dynamic linker resolve this library's libamdhip64.so.7
    # dependency to the copy torch has already loaded (same SONAME) instead of /opt/rocm's.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    i64p, u64p, i32p = POINTER(c_int64), POINTER(c_uint64), POINTER(c_int32)
    sig = {
        "dx_init": (c_void_p, [c_int]),
        "dx_free": (None, [c_void_p]),
        "dx_last_error": (c_char_p, [c_void_p]),
        "dx_partition_device": (c_int, [c_void_p, c_void_p, c_uint64, c_char_p, c_int, POINTER(PartStats)]),
        "dx_partition_file": (c_int, [c_void_p, c_char_p, c_char_p, c_int, POINTER(PartStats)]),
        "dx_part_stage_name": (c_char_p, [c_int]),
        "dx_term_count": (c_int64, [c_void_p]),
        "dx_terms_bytes": (c_int, [c_void_p, i64p, c_uint64, u64p]),
        "dx_terms_decode": (c_int, [c_void_p, i64p, c_uint64, c_void_p, c_uint64, u64p]),
        "dx_terms_lookup": (c_int, [c_void_p, c_char_p, u64p, c_uint64, i64p]),
        "dx_part_count": (c_int, [c_void_p]),
        "dx_part_rows": (c_int64, [c_void_p, c_int]),
        "dx_part_fetch": (c_int, [c_void_p, c_int, i64p, i64p, i64p]),
        "dx_table_count": (c_int, [c_void_p]),
        "dx_table_info_at": (c_int, [c_void_p, c_int, POINTER(TableInfo)]),
        "dx_table_fetch": (c_int, [c_void_p, c_int, i64p, i64p]),
        "dx_node_index_rows": (c_int64, [c_void_p]),
        "dx_node_index_fetch": (c_int, [c_void_p, i64p, i64p]),
        "dx_vp_index_rows": (c_int64, [c_void_p, c_int]),
        "dx_vp_index_fetch": (c_int, [c_void_p, c_int, i64p, i32p, i64p]),
        "dx_instance_graph_sizes": (c_int, [c_void_p, i64p, i64p]),
        "dx_instance_graph_fetch": (c_int, [c_void_p, i64p, i64p, i64p, i64p]),
        "dx_measures_compute": (c_int, [c_void_p, ctypes.POINTER(MeasureSizes)]),
        "dx_measures_classes": (c_int, [c_void_p, i64p, i64p, i64p]),
        "dx_measures_cc": (c_int, [c_void_p, i64p, i64p, i64p, i64p, i64p]),
        "dx_measures_write": (c_int, [c_void_p, c_char_p, c_char_p, c_char_p]),
        "dx_part_write_layout": (c_int, [c_void_p, c_char_p, c_char_p, c_char_p]),
        "dx_layout_times": (c_int, [c_void_p, POINTER(c_double), c_int]),
        "dx_open_dataset": (c_int, [c_void_p, c_char_p, c_char_p, c_int]),
        "dx_import_layout": (c_int, [c_void_p, c_char_p, c_char_p, c_int, u64p, u64p]),
        "dx_load_table": (c_int, [c_void_p, c_char_p, u64p]),
        "dx_load_tables": (c_int, [c_void_p, POINTER(c_char_p), c_int, u64p]),
        "dx_unload_all": (c_int, [c_void_p]),
        "dx_query": (c_int, [c_void_p, c_char_p, c_int, POINTER(c_void_p)]),
        "dx_query_joined": (c_int, [c_void_p, POINTER(c_char_p), c_int, c_int, POINTER(c_void_p)]),
        "dx_query_count_noalloc": (c_int, [c_void_p, c_char_p, u64p]),
        "dx_allocator_stats": (c_int, [c_void_p, u64p, u64p]),
        "dx_result_count": (c_uint64, [c_void_p]),
        "dx_result_columns": (c_int, [c_void_p]),
        "dx_result_column_name": (c_char_p, [c_void_p, c_int]),
        "dx_result_fetch_ids": (c_int, [c_void_p, i64p, c_int]),
        "dx_result_virtual_term": (c_char_p, [c_void_p, c_int64]),
        "dx_result_plan": (c_char_p, [c_void_p]),
        "dx_result_device_ms": (c_double, [c_void_p]),
        "dx_result_free": (None, [c_void_p]),
        "dx_dictionary_load": (c_int, [c_void_p, c_void_p, u64p, c_uint64]),
        "dx_load_table_ids": (c_int, [c_void_p, c_char_p, i64p, i64p, c_uint64]),
        "dx_set_comm": (c_int, [c_void_p, c_void_p]),
        "dx_rccl_unique_id": (c_int, [c_void_p]),
        "dx_comm_rccl_init": (c_int, [c_void_p, c_void_p, c_int, c_int]),
        "dx_comm_rccl_init_file": (c_int, [c_void_p, c_char_p, c_int, c_int, c_int]),
        "dx_comm_rccl_init_job": (c_int, [c_void_p, c_char_p, c_uint64, c_int, c_int, c_int]),
        "dx_comm_selftest": (c_int, [c_void_p]),
        "dx_load_tables_sharded": (c_int, [c_void_p, POINTER(c_char_p), c_int, u64p]),
        "dx_declare_remote_table": (c_int, [c_void_p, c_char_p]),
        "dx_table_stats": (c_int, [c_void_p, c_char_p, u64p]),
        "dx_set_table_stats": (c_int, [c_void_p, c_char_p, u64p]),
        "dx_result_local_rows": (c_uint64, [c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def _i64(n):
    return np.empty(int(n), dtype=np.int64)


def _p(a, t=c_int64):
    return a.ctypes.data_as(POINTER(t))


class QueryResult:
    # count is read at once (it is what the reference's timed `result.count` returns); the other attributes are
    # fetched from the handle when first asked for, or when the handle is closed, whichever comes first
    _LAZY = ("local_rows", "columns", "device_ms", "plan")

    def __init__(self, engine, handle):
        self._e, self._h = engine, handle
        self.count = int(engine.lib.dx_result_count(handle))     # over all ranks

    def __getattr__(self, name):
        if name not in QueryResult._LAZY:
            raise AttributeError(name)
        lib, handle = self._e.lib, self._h
        if not handle:
            raise AttributeError(name + " was not read before the result was closed")
        if name == "local_rows":                                 # held by this rank (== count on one GPU)
            v = int(lib.dx_result_local_rows(handle))
        elif name == "columns":
            v = [lib.dx_result_column_name(handle, i).decode() for i in range(lib.dx_result_columns(handle))]
        elif name == "device_ms":
            v = float(lib.dx_result_device_ms(handle))
        else:
            v = lib.dx_result_plan(handle).decode()
        self.__dict__[name] = v
        return v

    def ids(self, sort=False):
        """(count, n_columns) int64 array of term ids (needs DX_QUERY_BINDINGS)."""
        a = _i64(self.local_rows * max(1, len(self.columns)))
        self._e._check(self._e.lib.dx_result_fetch_ids(self._h, _p(a), 1 if sort else 0))
        return a.reshape(len(self.columns), self.local_rows).T.copy() if self.columns else a.reshape(0, 0)

    def rows(self):
        """Decoded bindings (this rank's share when the query ran on several GPUs): sorted list of tuples
        of raw N-Triples terms, columns in name order."""
        ids = self.ids()
        order = sorted(range(len(self.columns)), key=lambda i: self.columns[i])
        flat = ids[:, order].reshape(-1)
        real = flat[flat >= 0]
        uniq, inv = np.unique(real, return_inverse=True)
        strs = self._e.decode(uniq)
        out_flat, k = [], 0
        for v in flat:
            if v >= 0:
                out_flat.append(strs[inv[k]])
                k += 1
            else:
                out_flat.append(self._e.lib.dx_result_virtual_term(self._h, int(v)).decode())
        nc = len(order)
        return sorted(tuple(out_flat[i * nc:(i + 1) * nc]) for i in range(self.local_rows)), [self.columns[i] for i in order]

    def close(self):
        if self._h:
            for name in QueryResult._LAZY:   # what the handle knows stays readable after it is gone
                getattr(self, name)
            self._e.lib.dx_result_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Engine:
    """One dx_ctx: one GPU, one host thread."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.ctx = self.lib.dx_init(device)
        if not self.ctx:
            raise RuntimeError("dx_init(%d) failed: no such HIP device (this engine has no CPU path)" % device)
        self._results = weakref.WeakSet()  # a dx_result holds memory of its context: freed before the context is

    def close(self):
        if self.ctx:
            for r in list(self._results):
                r.close()
            self.lib.dx_free(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise DxError(rc, self.lib.dx_last_error(self.ctx).decode(errors="replace"))

    # ---- partitioner -------------------------------------------------------------
    def partition_device(self, device_ptr: int, n_bytes: int, class_index_text: str, n_partitions: int) -> PartStats:
        st = PartStats()
        self._check(self.lib.dx_partition_device(self.ctx, c_void_p(device_ptr), n_bytes, class_index_text.encode(),
                                                 n_partitions, byref(st)))
        return st

    def partition_text(self, text, class_index_text: str, n_partitions: int) -> PartStats:
        """Convenience for tests: copies host bytes / numpy uint8 to the GPU through torch, then partitions."""
        import torch
        arr = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.asarray(text, dtype=np.uint8)
        if arr.size == 0:
            return self.partition_device(0, 0, class_index_text, n_partitions)
        t = torch.from_numpy(arr.copy()).to("cuda:0")
        torch.cuda.synchronize()
        st = self.partition_device(t.data_ptr(), arr.size, class_index_text, n_partitions)
        del t
        return st

    def partition_file(self, instances_path: str, class_index_path: str, n_partitions: int) -> PartStats:
        st = PartStats()
        self._check(self.lib.dx_partition_file(self.ctx, instances_path.encode(), class_index_path.encode(), n_partitions, byref(st)))
        return st

    def term_count(self) -> int:
        return int(self.lib.dx_term_count(self.ctx))

    def decode_raw(self, ids):
        """(bytes as a uint8 array, n+1 offsets): term k of `ids` is bytes[offs[k]:offs[k+1]]."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        n = ids.size
        total = c_uint64(0)
        if n:
            self._check(self.lib.dx_terms_bytes(self.ctx, _p(ids), n, byref(total)))
        buf = np.empty(max(1, total.value), dtype=np.uint8)
        offs = np.zeros(n + 1, dtype=np.uint64)
        if n:
            self._check(self.lib.dx_terms_decode(self.ctx, _p(ids), n, buf.ctypes.data_as(c_void_p), total.value, _p(offs, c_uint64)))
        return buf[:total.value], offs

    def decode(self, ids) -> list:
        buf, offs = self.decode_raw(ids)
        raw = buf.tobytes()
        return [raw[int(offs[i]):int(offs[i + 1])].decode("utf-8", "surrogateescape") for i in range(len(offs) - 1)]

    def lookup(self, terms) -> np.ndarray:
        enc = [t.encode("utf-8", "surrogateescape") for t in terms]
        offs = np.zeros(len(enc) + 1, dtype=np.uint64)
        for i, b in enumerate(enc):
            offs[i + 1] = offs[i] + len(b)
        out = _i64(len(enc))
        if enc:
            self._check(self.lib.dx_terms_lookup(self.ctx, b"".join(enc), _p(offs, c_uint64), len(enc), _p(out)))
        return out

    def partition_count(self) -> int:
        return int(self.lib.dx_part_count(self.ctx))

    def partition_rows(self, q: int):
        n = int(self.lib.dx_part_rows(self.ctx, q))
        s, o, p = _i64(n), _i64(n), _i64(n)
        if n:
            self._check(self.lib.dx_part_fetch(self.ctx, q, _p(s), _p(o), _p(p)))
        return s, o, p

    def tables(self) -> list:
        out = []
        for i in range(int(self.lib.dx_table_count(self.ctx))):
            ti = TableInfo()
            self._check(self.lib.dx_table_info_at(self.ctx, i, byref(ti)))
            out.append({"index": i, "partition": ti.partition, "in_layout": bool(ti.in_layout), "predicate": ti.predicate, "rows": ti.rows})
        return out

    def table_rows(self, i: int, rows: int):
        s, o = _i64(rows), _i64(rows)
        if rows:
            self._check(self.lib.dx_table_fetch(self.ctx, i, _p(s), _p(o)))
        return s, o

    def node_index(self):
        n = int(self.lib.dx_node_index_rows(self.ctx))
        u, t = _i64(n), _i64(n)
        if n:
            self._check(self.lib.dx_node_index_fetch(self.ctx, _p(u), _p(t)))
        return u, t

    def vp_index(self, q: int):
        n = int(self.lib.dx_vp_index_rows(self.ctx, q))
        u, p = _i64(n), _i64(n)
        r = np.empty(n, dtype=np.int32)
        if n:
            self._check(self.lib.dx_vp_index_fetch(self.ctx, q, _p(u), _p(r, c_int32), _p(p)))
        return u, r, p

    def instance_graph(self):
        nv, ne = c_int64(0), c_int64(0)
        self._check(self.lib.dx_instance_graph_sizes(self.ctx, byref(nv), byref(ne)))
        v, es, ed, ep = _i64(nv.value), _i64(ne.value), _i64(ne.value), _i64(ne.value)
        self._check(self.lib.dx_instance_graph_fetch(self.ctx, _p(v), _p(es), _p(ed), _p(ep)))
        return v, es, ed, ep

    def measures(self):
        """Instance-scale inputs of the schema clustering (Preprocessor.scala:230-264, 312-339) from the last
        partition run: (sizes, classes, cc) with classes = (class term, node freq, node count) arrays and
        cc = (subType, pred, objType, rows, distinct object instances) arrays, all as term ids."""
        sz = MeasureSizes()
        self._check(self.lib.dx_measures_compute(self.ctx, byref(sz)))
        c, f, k = _i64(sz.n_classes), _i64(sz.n_classes), _i64(sz.n_classes)
        self._check(self.lib.dx_measures_classes(self.ctx, _p(c), _p(f), _p(k)))
        cc = tuple(_i64(sz.n_cc) for _ in range(5))
        self._check(self.lib.dx_measures_cc(self.ctx, *[_p(a) for a in cc]))
        sizes = {k_: getattr(sz, k_) for k_, _ in MeasureSizes._fields_}
        return sizes, (c, f, k), cc

    def write_measures(self, base: str, dataset: str, local_dir: str = ""):
        self._check(self.lib.dx_measures_write(self.ctx, base.encode(), dataset.encode(), local_dir.encode()))

    def write_layout(self, base: str, dataset: str, local_dir: str = ""):
        self._check(self.lib.dx_part_write_layout(self.ctx, base.encode(), dataset.encode(), local_dir.encode()))

    def layout_times(self) -> dict:
        """Where the time of the last write_layout went (ms), the bytes it wrote and the threads that wrote them."""
        v = (c_double * 8)()
        self._check(self.lib.dx_layout_times(self.ctx, v, 8))
        keys = ("total_ms", "prepare_ms", "encode_ms", "d2h_ms", "write_wall_ms", "write_busy_ms", "bytes", "threads")
        return {k: (int(x) if k in ("bytes", "threads") else float(x)) for k, x in zip(keys, v)}

    # ---- query processor ---------------------------------------------------------------
    def open_dataset(self, base: str, dataset: str, n_partitions: int):
        self._check(self.lib.dx_open_dataset(self.ctx, base.encode(), dataset.encode(), n_partitions))

    def import_layout(self, base: str, dataset: str, n_partitions: int):
        """Every vertical-partition table of a layout from its Parquet files (the library's own reader): one dictionary
        over their strings, the tables resident by key. Returns (tables, rows)."""
        nt, nr = c_uint64(0), c_uint64(0)
        self._check(self.lib.dx_import_layout(self.ctx, base.encode(), dataset.encode(), n_partitions, byref(nt), byref(nr)))
        return nt.value, nr.value

    def load_table(self, key: str) -> int:
        rows = c_uint64(0)
        self._check(self.lib.dx_load_table(self.ctx, key.encode(), byref(rows)))
        return rows.value

    def load_tables(self, keys) -> int:
        """preloadTables: all tables of a `partitions` line at once; returns the rows loaded."""
        keys = [k.encode() for k in keys]
        arr = (c_char_p * len(keys))(*keys)
        rows = c_uint64(0)
        self._check(self.lib.dx_load_tables(self.ctx, arr, len(keys), byref(rows)))
        return rows.value

    def dictionary_load(self, data: np.ndarray, offsets: np.ndarray):
        """Installs a dictionary: term i = data[offsets[i]:offsets[i+1]] (uint8 bytes, uint64 offsets)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self.lib.dx_dictionary_load(self.ctx, data.ctypes.data_as(c_void_p), _p(offsets, c_uint64), offsets.size - 1))

    def load_table_ids(self, key: str, s, o):
        s = np.ascontiguousarray(s, dtype=np.int64)
        o = np.ascontiguousarray(o, dtype=np.int64)
        self._check(self.lib.dx_load_table_ids(self.ctx, key.encode(), _p(s), _p(o), s.size))

    def unload_all(self):
        self._check(self.lib.dx_unload_all(self.ctx))

    # ---- several GPUs ---------------------------------------------------------------------
    def set_comm(self, comm):
        """comm: an object with a `struct` attribute holding a dx_comm (opus_amd.host.comm.TorchComm), or None."""
        self._comm = comm  # keeps the callbacks alive
        self._check(self.lib.dx_set_comm(self.ctx, ctypes.byref(comm.struct) if comm is not None else None))

    @staticmethod
    def rccl_unique_id() -> bytes:
        """128 bytes that rank 0 hands to the other ranks before they all call comm_rccl_init."""
        buf = ctypes.create_string_buffer(128)
        rc = load_library().dx_rccl_unique_id(buf)
        if rc != 0:
            raise DxError(rc, "dx_rccl_unique_id failed (is librccl.so.1 loadable?)")
        return buf.raw

    def comm_rccl_init(self, unique_id: bytes, rank: int, world: int):
        """Joins the group of `world` ranks over RCCL (collective: returns when all ranks have called it)."""
        if len(unique_id) != 128:
            raise ValueError("an RCCL unique id has 128 bytes")
        self._comm = None
        self._check(self.lib.dx_comm_rccl_init(self.ctx, ctypes.create_string_buffer(unique_id, 128), rank, world))

    def comm_rccl_init_file(self, path: str, rank: int, world: int, timeout_ms: int = 60000, job: int = 0):
        """job: a number the ranks of this launch share and other launches do not (it goes into the file's name)."""
        self._comm = None
        self._check(self.lib.dx_comm_rccl_init_job(self.ctx, path.encode(), job, rank, world, timeout_ms))

    def comm_selftest(self):
        self._check(self.lib.dx_comm_selftest(self.ctx))

    def load_tables_sharded(self, keys) -> int:
        """Collective preloadTables: partition q's tables on rank q % world, statistics shared; rows over all ranks."""
        keys = [k.encode() for k in keys]
        arr = (c_char_p * len(keys))(*keys)
        rows = c_uint64(0)
        self._check(self.lib.dx_load_tables_sharded(self.ctx, arr, len(keys), byref(rows)))
        return rows.value

    def declare_remote_table(self, key: str):
        self._check(self.lib.dx_declare_remote_table(self.ctx, key.encode()))

    def table_stats(self, key: str):
        out = (c_uint64 * 3)()
        self._check(self.lib.dx_table_stats(self.ctx, key.encode(), out))
        return [int(out[0]), int(out[1]), int(out[2])]

    def set_table_stats(self, key: str, stats):
        arr = (c_uint64 * 3)(int(stats[0]), int(stats[1]), int(stats[2]))
        self._check(self.lib.dx_set_table_stats(self.ctx, key.encode(), arr))

    def allocator_stats(self):
        """(buffers the context's device allocator has handed out so far, bytes of those still in use)"""
        n, b = c_uint64(0), c_uint64(0)
        self._check(self.lib.dx_allocator_stats(self.ctx, byref(n), byref(b)))
        return n.value, b.value

    def query_count_noalloc(self, sql: str) -> int:
        """The count of query(sql), computed without any device allocation (fixed scratch array; one GPU)."""
        n = c_uint64(0)
        self._check(self.lib.dx_query_count_noalloc(self.ctx, sql.encode("utf-8", "surrogateescape"), byref(n)))
        return n.value

    def query_joined(self, statements, mode: int = DX_QUERY_COUNT) -> QueryResult:
        """The statements of one translated file executed and their results joined left to right on the columns they
        share (QueryProcessor.executeFinalQuery with a query map of several statements)."""
        arr = (c_char_p * len(statements))(*[q.encode("utf-8", "surrogateescape") for q in statements])
        h = c_void_p(None)
        self._check(self.lib.dx_query_joined(self.ctx, arr, len(statements), mode, byref(h)))
        r = QueryResult(self, h)
        self._results.add(r)
        return r

    def query(self, sql: str, mode: int = DX_QUERY_COUNT) -> QueryResult:
        h = c_void_p(None)
        self._check(self.lib.dx_query(self.ctx, sql.encode("utf-8", "surrogateescape"), mode, byref(h)))
        r = QueryResult(self, h)
        self._results.add(r)
        return r
