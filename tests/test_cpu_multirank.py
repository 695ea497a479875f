"""The N>1 host path on CPU: two gloo ranks shard the input, work on their blocks, combine measurements."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from opus_amd import lubm
        from opus_amd.host import shard
        from oracle import cpu_port
        dev = torch.device("cpu")
        b, e = shard.university_block(rank, world, 1)
        arr, nt = lubm.generate(b, e, seed=3, degree_pool=5)
        ci = lubm.class_index_text(4, 0)
        res = cpu_port.run(arr, ci, 4)  # the rank's local work, done here by the CPU checker
        shard.barrier(dev)
        elapsed, part, total = shard.combine_timings(1.0 + rank, 0.5 + 0.25 * rank, nt, dev)
        counts = shard.combine_counts({"rows": sum(res["part_rows"]), "triples": res["n_triples"]}, dev)
        out.put((rank, b, e, nt, bytes(arr[:64]), elapsed, part, total, counts))
    finally:
        dist.destroy_process_group()


def test_two_ranks_shard_and_combine():
    from opus_amd import lubm
    from opus_amd.host import shard
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole, n_whole = lubm.generate(0, 2, seed=3, degree_pool=5)
    (r0, b0, e0, n0, head0, el0, pt0, tot0, c0), (r1, b1, e1, n1, head1, el1, pt1, tot1, c1) = got
    assert (b0, e0, b1, e1) == (0, 1, 1, 2)                       # blocks tile the input
    assert n0 + n1 == n_whole and head0 == bytes(whole[:64])      # and concatenate to the unsharded text
    assert el0 == el1 == 2.0 and pt0 == pt1 == 0.75               # MAX over ranks
    assert tot0 == tot1 == n_whole                                # SUM over ranks
    assert c0 == c1 and c0["triples"] == n_whole
    assert [shard.partition_owner(q, 2) for q in range(5)] == [0, 1, 0, 1, 0]
    with pytest.raises(ValueError):
        shard.university_block(2, 2, 1)


def _comm_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes
        from opus_amd.host.comm import TorchComm
        c = TorchComm()
        st = c.struct
        # the library calls these through the C function pointers: do the same
        send = (ctypes.c_uint64 * world)(*[100 * rank + p for p in range(world)])
        recv = (ctypes.c_uint64 * world)()
        assert st.exchange_host(None, send, recv, 1) == 0
        # three words per peer: group p of recv is what peer p sent at position `rank`
        send3 = (ctypes.c_uint64 * (3 * world))(*[1000 * rank + 10 * p + k for p in range(world) for k in range(3)])
        recv3 = (ctypes.c_uint64 * (3 * world))()
        assert st.exchange_host(None, send3, recv3, 3) == 0
        assert list(recv3) == [1000 * p + 10 * rank + k for p in range(world) for k in range(3)]
        vals = (ctypes.c_uint64 * 2)(rank + 1, 10)
        assert st.allreduce_sum(None, vals, 2) == 0
        assert st.stream_ordered == 0
        out.put((rank, list(recv), list(vals), st.rank, st.world))
    finally:
        dist.destroy_process_group()


def test_comm_callbacks_over_gloo():
    """The count exchange and the sum the library asks the host for, through their C function pointers."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == (0, [0, 100], [3, 20], 0, 2)     # recv[p] = what p addressed to me
    assert got[1] == (1, [1, 101], [3, 20], 1, 2)


def _collectives_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes

        import numpy as np
        from opus_amd.host.comm import TorchComm
        st = TorchComm().struct
        u64 = lambda xs: (ctypes.c_uint64 * len(xs))(*xs)  # noqa: E731
        ptr = lambda a: ctypes.c_void_p(a.ctypes.data)     # noqa: E731
        # all-to-all with counts: rank r sends r+1 elements to peer 0 and 2 elements to peer 1
        sc = [rank + 1, 2]
        send = np.arange(sum(sc), dtype=np.int64) + 100 * rank
        rc_c = u64([0] * world)
        assert st.exchange_host(None, u64(sc), rc_c, 1) == 0
        rc = list(rc_c)
        recv = np.full(sum(rc), -1, dtype=np.int64)
        assert st.exchange(None, ptr(send), u64(sc), ptr(recv), u64(rc), 8) == 0
        # all-gather with uneven sizes, bytes
        mine = np.full(3 + rank, 7 + rank, dtype=np.uint8)
        counts = [3, 4]
        gathered = np.zeros(sum(counts), dtype=np.uint8)
        assert st.allgather(None, ptr(mine), mine.size, ptr(gathered), u64(counts), 1) == 0
        # sums of byte flags and of 8-byte words, in place
        flags = np.array([1, 0, rank, 1 - rank], dtype=np.uint8)
        words = np.array([10 + rank, 1 << 40], dtype=np.int64)
        assert st.allreduce_sum_dev(None, ptr(flags), flags.size, 1) == 0
        assert st.allreduce_sum_dev(None, ptr(words), words.size, 8) == 0
        out.put((rank, rc, recv.tolist(), gathered.tolist(), flags.tolist(), words.tolist()))
    finally:
        dist.destroy_process_group()


def test_comm_data_collectives_over_gloo():
    """All-to-all with counts, uneven all-gather and in-place sums, through the C function pointers, on host buffers."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_collectives_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, rc0, recv0, g0, f0, w0), (_, rc1, recv1, g1, f1, w1) = got
    assert rc0 == [1, 2] and recv0 == [0, 100, 101]            # rank 0's first element, rank 1's first two
    assert rc1 == [2, 2] and recv1 == [1, 2, 102, 103]         # what each addressed to rank 1
    assert g0 == g1 == [7, 7, 7, 8, 8, 8, 8]
    assert f0 == f1 == [2, 0, 1, 1]
    assert w0 == w1 == [21, 1 << 41]
