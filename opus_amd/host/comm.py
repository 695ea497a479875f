"""The three collectives the library asks the host for when a query runs on several GPUs, over
torch.distributed: RCCL ("nccl") on device buffers in production, gloo through host copies in tests.

The library is handed plain C function pointers (struct dx_comm in include/diaeresis_hip.h); this module
keeps the ctypes trampolines alive for as long as the communicator object lives.
"""
from __future__ import annotations

import ctypes
import sys
import traceback
from ctypes import CFUNCTYPE, POINTER, c_int, c_uint64, c_void_p

import numpy as np

COUNTS_FN = CFUNCTYPE(c_int, c_void_p, POINTER(c_uint64), POINTER(c_uint64), c_int)
EXCHANGE_FN = CFUNCTYPE(c_int, c_void_p, c_void_p, POINTER(c_uint64), c_void_p, POINTER(c_uint64), c_int)
ALLREDUCE_FN = CFUNCTYPE(c_int, c_void_p, POINTER(c_uint64), c_int)
ALLGATHER_FN = CFUNCTYPE(c_int, c_void_p, c_void_p, c_uint64, c_void_p, POINTER(c_uint64), c_int)
ALLREDUCE_DEV_FN = CFUNCTYPE(c_int, c_void_p, c_void_p, c_uint64, c_int)


class DxComm(ctypes.Structure):
    _fields_ = [("rank", c_int), ("world", c_int), ("user", c_void_p), ("exchange_host", COUNTS_FN),
                ("exchange", EXCHANGE_FN), ("allreduce_sum", ALLREDUCE_FN), ("allgather", ALLGATHER_FN),
                ("allreduce_sum_dev", ALLREDUCE_DEV_FN), ("stream_ordered", c_int)]


class _DevicePointer:
    """Lets torch wrap memory the library owns without copying it."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_tensor(ptr: int, n: int, elem_bytes: int):
    """The library's buffer as a torch tensor over the same memory. Without a GPU (the host-logic tests hand
    in host memory) the pointer is wrapped as a CPU tensor instead."""
    import torch
    if elem_bytes == 8:
        dtype, typestr, npdt = torch.int64, "<i8", np.int64
    elif elem_bytes == 4:
        dtype, typestr, npdt = torch.int32, "<i4", np.int32
    else:
        dtype, typestr, npdt, n = torch.uint8, "|u1", np.uint8, n * elem_bytes
    on_gpu = torch.cuda.is_available()
    if n == 0:
        return torch.empty(0, dtype=dtype, device="cuda" if on_gpu else "cpu")
    if not on_gpu:
        raw = (ctypes.c_char * (n * np.dtype(npdt).itemsize)).from_address(ptr)
        return torch.from_numpy(np.frombuffer(raw, dtype=npdt))
    return torch.as_tensor(_DevicePointer(ptr, n, typestr), device="cuda")


def _sync():
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class TorchComm:
    """dx_comm over the default torch.distributed process group (or `group`)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.on_device = dist.get_backend(group) == "nccl"
        self._fns = (COUNTS_FN(self._counts), EXCHANGE_FN(self._exchange), ALLREDUCE_FN(self._allreduce),
                     ALLGATHER_FN(self._allgather), ALLREDUCE_DEV_FN(self._allreduce_dev))
        # stream_ordered = 0: torch works on its own streams, so every callback returns when its result is complete
        self.struct = DxComm(self.rank, self.world, None, *self._fns, 0)

    # every callback returns 0, or 1 after printing the exception: the library turns that into DX_E_IO
    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception:  # noqa: BLE001
            traceback.print_exc(file=sys.stderr)
            return 1

    def _counts(self, user, send, recv, n):
        """exchange_host: `world` groups of n words each way; group p of recv = what peer p sent at position `rank`."""
        def run():
            import torch
            words = n * self.world
            mine = torch.from_numpy(np.ctypeslib.as_array(send, shape=(words,)).astype(np.int64))
            if self.on_device:
                mine = mine.cuda()
            everyone = torch.empty(self.world * words, dtype=torch.int64, device=mine.device)
            self.dist.all_gather_into_tensor(everyone, mine, group=self.group)
            got = everyone.view(self.world, self.world, n)[:, self.rank, :].contiguous().cpu().numpy()
            np.ctypeslib.as_array(recv, shape=(words,))[:] = got.reshape(-1).astype(np.uint64)
        return self._guard(run)

    def _exchange(self, user, send_dev, send_counts, recv_dev, recv_counts, elem_bytes):
        def run():
            import torch
            scale = 1 if elem_bytes in (4, 8) else elem_bytes
            sc = [int(send_counts[p]) * scale for p in range(self.world)]
            rc = [int(recv_counts[p]) * scale for p in range(self.world)]
            send_t = device_tensor(send_dev or 0, sum(sc) // scale, elem_bytes)
            recv_t = device_tensor(recv_dev or 0, sum(rc) // scale, elem_bytes)
            if self.on_device:
                self.dist.all_to_all_single(recv_t, send_t, rc, sc, group=self.group)
                _sync()
                return
            # gloo has no all-to-all: host copies and pairwise sends (tests only)
            hs = send_t.cpu()
            hr = torch.empty(sum(rc), dtype=hs.dtype)
            reqs, so, ro = [], 0, 0
            for p in range(self.world):
                if p == self.rank:
                    hr[ro:ro + rc[p]] = hs[so:so + sc[p]]
                else:
                    if sc[p]:
                        reqs.append(self.dist.isend(hs[so:so + sc[p]].contiguous(), p, group=self.group))
                    if rc[p]:
                        reqs.append(self.dist.irecv(hr[ro:ro + rc[p]], p, group=self.group))
                so += sc[p]
                ro += rc[p]
            for r in reqs:
                r.wait()
            if hr.numel():
                recv_t.copy_(hr)
            _sync()
        return self._guard(run)

    def _allreduce(self, user, values, n):
        def run():
            import torch
            view = np.ctypeslib.as_array(values, shape=(n,))
            t = torch.from_numpy(view.astype(np.int64))
            if self.on_device:
                t = t.cuda()
            self.dist.all_reduce(t, group=self.group)
            view[:] = t.cpu().numpy().astype(np.uint64)
        return self._guard(run)


    def _allgather(self, user, send_dev, send_count, recv_dev, recv_counts, elem_bytes):
        def run():
            import torch
            rc = [int(recv_counts[p]) for p in range(self.world)]
            recv_t = device_tensor(recv_dev or 0, sum(rc), elem_bytes)
            send_t = device_tensor(send_dev or 0, int(send_count), elem_bytes)
            scale = recv_t.numel() // max(1, sum(rc)) if sum(rc) else 1   # odd element sizes travel as bytes
            at = 0
            for p in range(self.world):  # uneven sizes: one broadcast per rank, straight into its place
                n = rc[p] * scale
                if n:
                    piece = recv_t[at:at + n]
                    if p == self.rank:
                        piece.copy_(send_t)
                    if self.on_device:
                        self.dist.broadcast(piece, src=self._global_rank(p), group=self.group)
                    else:
                        host = piece.cpu()
                        self.dist.broadcast(host, src=self._global_rank(p), group=self.group)
                        if p != self.rank:
                            piece.copy_(host)
                at += n
            _sync()
        return self._guard(run)

    def _allreduce_dev(self, user, dev, n, elem_bytes):
        def run():
            import torch
            if elem_bytes not in (1, 8):
                raise ValueError("allreduce_sum_dev: elements of 1 or 8 bytes")
            t = device_tensor(dev or 0, int(n), elem_bytes)
            if not t.numel():
                return
            if self.on_device:
                self.dist.all_reduce(t, group=self.group)
            else:
                host = t.cpu()
                self.dist.all_reduce(host, group=self.group)
                t.copy_(host)
            _sync()
        return self._guard(run)

    def _global_rank(self, p):
        return p if self.group is None else self.dist.get_global_rank(self.group, p)


def load_tables_sharded(engine, comm: TorchComm, keys, owner) -> int:
    """Each key ("PPPPP-_3=<pred>") is loaded on the rank `owner(partition number)` returns and declared as
    remote elsewhere; the owners' row / distinct counts are then shared, so that every rank plans alike.
    Returns the rows loaded over all ranks."""
    import torch
    keys = list(dict.fromkeys(keys))
    stats = np.zeros((len(keys), 3), dtype=np.int64)
    for i, key in enumerate(keys):
        if owner(int(key.split("-", 1)[0])) == comm.rank:
            engine.load_table(key)
            stats[i] = engine.table_stats(key)
        else:
            engine.declare_remote_table(key)
    t = torch.from_numpy(stats)
    if comm.on_device:
        t = t.cuda()
    comm.dist.all_reduce(t, group=comm.group)  # one owner per table: the sum is the owner's numbers
    stats = t.cpu().numpy()
    for i, key in enumerate(keys):
        engine.set_table_stats(key, stats[i])
    return int(stats[:, 0].sum())
