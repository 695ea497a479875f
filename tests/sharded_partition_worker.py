"""One rank of the multi-GPU partitioner test (launched by test_gpu_sharded.py through torch.distributed.run).

The ranks split one N-Triples text at a line end near the middle (inside a university, so that types of one
rank's vertices are declared in the other's part), partition it together, and then answer queries together
over the partitions each of them ended up with. Rank 0 also partitions the whole text alone; both results
are written as JSON.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_path):
    import numpy as np
    import torch.distributed as dist
    dist.init_process_group(os.environ.get("DX_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    from opus_amd import lubm
    from opus_amd.engine import DX_QUERY_BINDINGS, Engine
    from opus_amd.host import shard, translator as T
    from opus_amd.host.comm import TorchComm, load_tables_sharded
    from oracle.query_oracle import parse_translated_file
    from parity_util import engine_artifacts

    n_parts = int(os.environ.get("DX_TEST_PARTS", "4"))
    arr, n_triples = lubm.generate(0, int(os.environ.get("DX_TEST_UNIVERSITIES", "2")), seed=1, ontology=0, degree_pool=2)
    # triples that no partition takes (no typed end; a class the index does not know), one of them in two ranks' parts:
    # they count among the distinct triples and edges of the input wherever they were read
    loose = b"<http://x/u1> <http://x/rel> <http://x/u2> .\n<http://x/u3> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <http://x/NoSuchClass> .\n"
    arr = np.concatenate([np.frombuffer(loose, dtype=np.uint8), arr, np.frombuffer(loose[:loose.index(b"\n") + 1], dtype=np.uint8)])
    n_triples += 3
    ci = lubm.class_index_text(n_parts, 0)
    cuts = [0]
    for r in range(1, world):
        at = arr.size * r // world
        while arr[at - 1] != 10:
            at += 1
        cuts.append(at)
    cuts.append(arr.size)
    mine = arr[cuts[rank]:cuts[rank + 1]]

    comm = TorchComm()
    e = Engine(0)
    e.set_comm(comm)
    st = e.partition_text(mine, ci, n_parts)
    art = engine_artifacts(e, measures=False)
    from parity_util import engine_measures
    meas = engine_measures(e, art["strs"])  # collective: every rank gathers all ranks' triples

    def jsonable(m):
        return {"cc": sorted([list(k), list(v)] for k, v in m["schema_cc"].items()), "freq": sorted(m["schema_node_freq"].items()),
                "count": sorted(m["schema_node_count"].items())}
    share = {"rank": rank, "measures": jsonable(meas), "n_terms": int(st.n_terms), "n_unique": int(st.n_unique), "n_edges": int(st.n_edges),
             "n_vertices": int(st.n_vertices), "n_triples_in": int(st.n_triples), "n_vp_index": int(st.n_vp_index),
             "parts": [sorted(p) for p in art["parts"]], "node_index": sorted(art["node_index"]),
             "vp": sorted(("%s/%s" % k, sorted(v)) for k, v in art["vp"].items()),
             "vp_index": {k: sorted((u, sorted(v)) for u, v in d.items()) for k, d in art["vp_index"].items()},
             "terms_head": art["strs"][:50]}
    shares = [None] * world
    dist.all_gather_object(shares, share)

    # the whole text on one GPU, by rank 0 alone
    box = [None]
    alone = None
    if rank == 0:
        e1 = Engine(0)
        st1 = e1.partition_text(arr, ci, n_parts)
        a1 = engine_artifacts(e1)
        alone = {"measures": jsonable(a1), "n_terms": int(st1.n_terms), "n_unique": int(st1.n_unique), "n_edges": int(st1.n_edges), "n_vertices": int(st1.n_vertices), "n_vp_index": int(st1.n_vp_index),
                 "parts": [sorted(p) for p in a1["parts"]], "node_index": sorted(a1["node_index"]),
                 "vp": sorted(("%s/%s" % k, sorted(v)) for k, v in a1["vp"].items()),
                 "vp_index": {k: sorted((u, sorted(v)) for u, v in d.items()) for k, d in a1["vp_index"].items()}}
        ni = {}
        for u, c in a1["node_index"]:
            ni.setdefault(u, []).append(c)
        cidx = T.load_class_index(ci)
        statements = {}
        for name, text in sorted(lubm.queries(0).items()):
            qm, parts, _ = parse_translated_file(T.translate_query(name, text, lambda u: ni.get(u, []), cidx, a1["partition_stat"]))
            sql = next(iter(qm.values()))
            e1.unload_all()
            for p in dict.fromkeys(parts):
                e1.load_table(p)
            r1 = e1.query(sql, DX_QUERY_BINDINGS)
            rows1, cols1 = r1.rows()
            statements[name] = (sql, parts, r1.count, [list(r) for r in rows1], cols1)
        box[0] = statements
        e1.close()
    dist.broadcast_object_list(box, src=0)
    statements = box[0]

    queries = {}
    for name, (sql, parts, want_count, want_rows, want_cols) in statements.items():
        e.unload_all()
        load_tables_sharded(e, comm, parts, lambda q: shard.partition_owner(q, world))
        rb = e.query(sql, DX_QUERY_BINDINGS)
        rows, cols = rb.rows()
        got = [None] * world
        dist.all_gather_object(got, rows)
        queries[name] = {"count": rb.count, "want_count": want_count, "columns": cols, "want_columns": want_cols,
                         "rows": sorted([list(r) for sh in got for r in sh]), "want_rows": want_rows}
    # a malformed line in one rank's part of the text: every rank returns DX_E_MALFORMED, nobody waits for ever
    from opus_amd.engine import DxError
    bad = np.concatenate([mine, np.frombuffer(b"<only-two> <fields>\n", dtype=np.uint8)]) if rank == 1 else mine
    try:
        e.partition_text(bad, ci, n_parts)
        code = 0
    except DxError as ex:
        code = ex.code
    codes = [None] * world
    dist.all_gather_object(codes, code)
    # and the group partitions again afterwards
    st2 = e.partition_text(mine, ci, n_parts)
    assert int(st2.n_terms) == int(st.n_terms)
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"shares": shares, "alone": alone, "queries": queries, "n_triples": int(n_triples), "malformed": codes,
                       "n_distinct_lines": len(set(arr.tobytes().split(b"\n")) - {b""})}, f)
    e.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
