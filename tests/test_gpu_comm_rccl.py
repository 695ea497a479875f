"""The communicator's RCCL side on the one GPU of the test box: a process group of one rank with backend
"nccl", every callback called through the C function pointers the library uses, on device buffers. (Two
ranks cannot share a GPU under RCCL; the two-rank tests use gloo.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_callbacks_over_rccl_single_rank(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "rccl.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_comm_worker.py"), str(out)], env=env, capture_output=True, text=True,
                       timeout=200)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    r = json.load(open(out))
    assert r["on_device"] and r["world"] == 1
    assert r["counts_rc"] == 0 and r["counts"] == 7
    for k in ("exchange_i64", "exchange_i32", "exchange_u8x12", "allgather", "allreduce_dev_u8", "allreduce_dev_i64"):
        assert r[k] == [0, True], k
    assert r["exchange_empty_rc"] == 0
    assert r["allreduce_rc"] == 0 and r["allreduce"] == [5, 11]
    assert r["selftest_torch"] and r["selftest_native"] and r["selftest_native_file"]
    assert r["partition_rows"] == 2
    assert r["waiter_without_rank0"] == -5 and r["waiter_seconds"] < 5
    assert r["rank0_without_peer"] == -5 and r["rank0_seconds"] < 20 and not r["file_left"]
