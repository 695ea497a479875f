"""Worker of test_gpu_comm_rccl.py: one rank, backend "nccl" (RCCL). Drives every callback of the
communicator through the same C function pointers the library calls, on buffers in device memory."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(out_path):
    import numpy as np
    import torch
    import torch.distributed as dist
    from opus_amd.host.comm import TorchComm

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    comm = TorchComm()
    c = comm.struct
    res = {"on_device": comm.on_device, "world": comm.world}
    u64 = ctypes.c_uint64

    send = (u64 * 1)(7)
    recv = (u64 * 1)(0)
    res["counts_rc"] = c.exchange_host(None, send, recv, 1)
    res["counts"] = int(recv[0])

    for name, dtype, eb in (("i64", torch.int64, 8), ("i32", torch.int32, 4), ("u8x12", torch.uint8, 12)):
        n = 1000
        src = torch.arange(n * (12 if eb == 12 else 1), device="cuda").to(dtype)
        dst = torch.zeros_like(src)
        cnt = (u64 * 1)(n)
        rc = c.exchange(None, src.data_ptr(), cnt, dst.data_ptr(), cnt, eb)
        res["exchange_" + name] = [rc, bool(torch.equal(src, dst))]
    zero = (u64 * 1)(0)
    res["exchange_empty_rc"] = c.exchange(None, None, zero, None, zero, 8)

    vals = (u64 * 2)(5, 11)
    res["allreduce_rc"] = c.allreduce_sum(None, vals, 2)
    res["allreduce"] = [int(vals[0]), int(vals[1])]

    src = torch.arange(513, device="cuda", dtype=torch.int64)
    dst = torch.zeros(513, device="cuda", dtype=torch.int64)
    cnt = (u64 * 1)(513)
    rc = c.allgather(None, src.data_ptr(), 513, dst.data_ptr(), cnt, 8)
    res["allgather"] = [rc, bool(torch.equal(src, dst))]

    for name, dtype, eb in (("u8", torch.uint8, 1), ("i64", torch.int64, 8)):
        t = (torch.arange(300, device="cuda") % 3).to(dtype)
        want = t.clone()
        rc = c.allreduce_sum_dev(None, t.data_ptr(), 300, eb)
        res["allreduce_dev_" + name] = [rc, bool(torch.equal(t, want))]

    # the same collectives through the engine's self-test, over the torch communicator ...
    from opus_amd.engine import Engine
    e = Engine(0)
    e.set_comm(comm)
    e.comm_selftest()
    res["selftest_torch"] = True
    # ... and over the communicator the library makes itself on librccl (a group of one rank: RCCL admits one rank
    # per GPU), formed once from a unique id handed over in memory and once through the rendezvous file
    e.comm_rccl_init(Engine.rccl_unique_id(), 0, 1)
    e.comm_selftest()
    res["selftest_native"] = True
    e.comm_rccl_init_file(out_path + ".rccl_id", 0, 1)
    e.comm_selftest()
    res["selftest_native_file"] = True
    # The meeting point of a group of two with only one rank present: nothing may wait for ever.
    from opus_amd.engine import DxError
    import time
    # rank 1 finds only the file of another launch (another job number): no id within its time limit
    stale = out_path + ".meet"
    open(stale + ".7", "wb").write(b"\x01" * 128)
    t0 = time.time()
    try:
        e.comm_rccl_init_file(stale, 1, 2, timeout_ms=300, job=8)
        res["waiter_without_rank0"] = 0
    except DxError as ex:
        res["waiter_without_rank0"] = ex.code
    res["waiter_seconds"] = time.time() - t0
    # rank 0 replaces a file left at the plain name, waits for a peer that never comes, gives up after its time limit and
    # takes its file away again
    open(stale, "wb").write(b"\x02" * 128)
    t0 = time.time()
    try:
        e.comm_rccl_init_file(stale, 0, 2, timeout_ms=1500)
        res["rank0_without_peer"] = 0
    except DxError as ex:
        res["rank0_without_peer"] = ex.code
    res["rank0_seconds"] = time.time() - t0
    res["file_left"] = os.path.exists(stale)
    e.comm_rccl_init(Engine.rccl_unique_id(), 0, 1)   # the context still forms groups
    e.comm_selftest()
    # with a group of one rank the engine works as on a single GPU
    st = e.partition_text(b"<a> <p> <b> .\n<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <C> .\n", "<C>\t0", 1)
    res["partition_rows"] = int(st.n_rows)
    e.set_comm(None)   # leaves the group
    e.close()

    dist.barrier()
    dist.destroy_process_group()
    with open(out_path, "w") as f:
        json.dump(res, f)
    # the thread that still waits for the peer that never came cannot be called back: leave without the interpreter's
    # and the libraries' exit handlers, which might wait for it
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main(sys.argv[1])
