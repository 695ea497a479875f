"""Queries answered by two ranks together (tables split by first-level partition, joins through the
exchange) against the same engine answering alone. Both ranks share the one GPU of the test box and talk
over gloo; on a multi-GPU node the same code runs over RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_answer_like_one(tmp_path):
    out = tmp_path / "sharded.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "sharded_query_worker.py"), str(out)]
    env = dict(os.environ, DX_TEST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = json.load(open(out))
    failures = res.pop("__failures__")
    assert len(failures) == 2
    for f in failures:   # one rank's planner did not find a table / a table's files are missing: both ranks return that error
        assert f["missing_table"] == -2 and f["missing_file"] == -2, failures
        assert f["after"] == f["want_after"]
        assert f["owner_lost_table"] == -2 and f["owner_after"] == f["owner_want_after"], failures
        assert f["other_placement_shortcut"] == -22 and f["other_placement_exchanges"] == f["owner_want_after"], failures
        assert f["no_allgather"] == -22, failures
    assert len(res) >= 42
    # statements whose tables all lie in one rank's partitions: answered there, one sum, no rows moved
    alone_runs = [n for n, r in res.items() if n.endswith("#owner") and "alone: all tables in its partitions" in r["plan"]]
    assert len(alone_runs) >= 5
    for n in alone_runs:
        assert sorted(res[n]["share_sizes"])[:-1] == [0] * (len(res[n]["share_sizes"]) - 1), n
        assert "0 host exchanges, 0 data, 1 sum" in res[n]["plan"]
    assert any(n.endswith("#owner") and "alone" not in r["plan"] for n, r in res.items())  # and some that span both ranks
    spread = 0
    for name, r in res.items():
        a = r["alone"]
        assert r["count"] == r["count_bindings"] == a["count"] == len(a["rows"]), name
        assert r["columns"] == a["columns"], name
        assert [tuple(x) for x in r["rows"]] == [tuple(x) for x in a["rows"]], name  # merged shares == single-GPU bindings
        assert r["loaded"] == a["loaded"], name
        assert sum(r["share_sizes"]) == r["count"]
        spread += all(s > 0 for s in r["share_sizes"])
    assert spread >= 5  # the shuffles really spread the larger results over both ranks
    assert "remote" in res["Q2.txt"]["plan"] and "local" in res["Q2.txt"]["plan"]
    # small sides travel whole (one data collective per join), redistribution moves both sides
    assert ", 5 data," in res["Q2.txt"]["plan"] and "6 host exchanges" not in res["Q2.txt"]["plan"]
    assert res["Q2.txt#hashed"]["plan"] != res["Q2.txt"]["plan"]


# (ranks, first-level partitions, 128-bit key words): two ranks as before; four ranks over eight partitions (two each);
# eight ranks over four partitions, where half the ranks own no partition at all and still take part in every collective
@pytest.mark.parametrize("ranks,parts,wide_keys", [(2, 4, "0"), (2, 4, "1"), (4, 8, "0"), (8, 4, "0"), (8, 8, "1")])
def test_ranks_partition_like_one(tmp_path, ranks, parts, wide_keys):
    """Each rank tokenises its piece of the text; the ranks agree on one dictionary, exchange vertex facts and
    route the triples to the owners of their partitions. Together they must hold exactly what one GPU
    produces from the whole text, and answer the queries alike."""
    out = tmp_path / "sharded_partition.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks, "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "sharded_partition_worker.py"), str(out)]
    # wide_keys: the 128-bit key words of a very large dictionary, forced on this small one (16 bytes per routed triple)
    env = dict(os.environ, DX_TEST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", DX_FORCE_WIDE_KEYS=wide_keys, DX_TEST_PARTS=str(parts),
               DX_TEST_UNIVERSITIES="2" if ranks == 2 else "3")
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = json.load(open(out))
    shares, alone = res["shares"], res["alone"]
    world = len(shares)
    assert world == ranks and len(alone["parts"]) == parts
    assert sum(s["n_triples_in"] for s in shares) == res["n_triples"]
    for s in shares:
        assert s["n_terms"] == alone["n_terms"]                       # one dictionary
        assert s["terms_head"] == shares[0]["terms_head"]             # with the same ids on every rank
        assert (s["n_unique"], s["n_edges"], s["n_vertices"]) == (shares[0]["n_unique"], shares[0]["n_edges"], shares[0]["n_vertices"])
    assert shares[0]["n_vertices"] == alone["n_vertices"]
    assert len(alone["measures"]["cc"]) > 20
    for s in shares:   # schema_cc, node frequency and node count over all ranks' triples, the same on every rank
        assert s["measures"] == alone["measures"]
    # distinct triples and edges of the whole input, placed or not, counted once wherever their copies were read
    assert shares[0]["n_edges"] == alone["n_edges"] and shares[0]["n_unique"] == alone["n_unique"]
    assert alone["n_unique"] == res["n_distinct_lines"] < res["n_triples"]   # the worker states one triple twice, in two ranks' parts
    for q, want in enumerate(alone["parts"]):
        for s in shares:
            if s["rank"] == q % world:
                assert s["parts"][q] == want, "partition %d" % q
            else:
                assert s["parts"][q] == []                           # nobody else holds it
    merged_ni = sorted(tuple(x) for s in shares for x in s["node_index"])
    assert merged_ni == sorted(tuple(x) for x in alone["node_index"])   # disjoint shares of the node index
    merged_vp = sorted((k, v) for s in shares for k, v in s["vp"])
    assert merged_vp == sorted((k, v) for k, v in alone["vp"])
    for pname, want in alone["vp_index"].items():
        owner = int(pname) % world
        assert shares[owner]["vp_index"][pname] == want, pname
        for s in shares:
            if s["rank"] != owner:
                assert s["vp_index"].get(pname, []) == [], "rank %d holds VP-index rows of partition %s" % (s["rank"], pname)
    assert sum(s["n_vp_index"] for s in shares) == alone["n_vp_index"]
    assert res["malformed"] == [-74] * world   # one rank's text is malformed: every rank returns DX_E_MALFORMED
    assert len(res["queries"]) == 13
    for name, r in res["queries"].items():
        assert r["count"] == r["want_count"] and r["columns"] == r["want_columns"] and r["rows"] == r["want_rows"], name
