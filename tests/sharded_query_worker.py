"""One rank of the multi-GPU query test (launched by test_gpu_sharded.py through torch.distributed.run).

Every rank partitions the same LUBM-1 text (ids are deterministic, so the dictionaries agree), keeps only
the tables of the first-level partitions it owns, and the ranks answer every query together. Rank 0 also
answers alone over all tables and writes both results as JSON.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_path):
    import torch
    import torch.distributed as dist
    dist.init_process_group(os.environ.get("DX_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    from opus_amd import lubm
    from opus_amd.engine import DX_QUERY_BINDINGS, DX_QUERY_COUNT, Engine
    from opus_amd.host import shard, translator as T
    from opus_amd.host.comm import TorchComm, load_tables_sharded
    from oracle.query_oracle import clean_table_name, parse_translated_file
    from parity_util import engine_artifacts

    arr, _ = lubm.generate(0, 1, seed=0, ontology=0, degree_pool=3)
    ci = lubm.class_index_text(4, 0)
    e = Engine(0)
    e.partition_text(arr, ci, 4)
    art = engine_artifacts(e)
    ni = {}
    for u, c in art["node_index"]:
        ni.setdefault(u, []).append(c)
    cidx = T.load_class_index(ci)
    statements = {}
    for name, text in sorted(lubm.queries(0).items()):
        qm, parts, _ = parse_translated_file(T.translate_query(name, text, lambda u: ni.get(u, []), cidx, art["partition_stat"]))
        statements[name] = (next(iter(qm.values())), parts)
    # a join on two columns (the same predicate as two first-level partitions hold it) and a cross product
    O = "_http___www_lehigh_edu__zhp2_2004_0401_univ_bench_owl_"
    adv = [k for k in ("%05d-_3=%sadvisor_" % (q, O) for q in range(4)) if (k.split("-")[0], k.split("-", 1)[1]) in art["vp"]]
    if len(adv) >= 2:
        statements["two_column_key"] = (
            "SELECT tab0.X AS X,tab0.Y AS Y FROM (SELECT s AS X, o AS Y FROM %s) AS tab0, (SELECT s AS X, o AS Y FROM %s) AS tab1 "
            "WHERE tab0.X=tab1.X AND tab0.Y=tab1.Y " % (clean_table_name(adv[0]), clean_table_name(adv[1])), adv[:2])
    q1_sql, q1_parts = statements["Q1.txt"]
    statements["cross_product"] = (
        "SELECT tab0.X AS X,tab1.Y AS Y FROM (SELECT s AS X FROM %s WHERE o == '<http://www.Department0.University0.edu/GraduateCourse0>') AS tab0, "
        "(SELECT s AS Y FROM %s WHERE o == '<http://www.Department0.University0.edu/GraduateCourse1>') AS tab1 " %
        tuple(clean_table_name(p) for p in [p for p in q1_parts if "takescourse" in p] * 2), q1_parts)

    comm = TorchComm()
    owner = lambda q: shard.partition_owner(q, world)  # noqa: E731
    results = {}
    # every statement twice: with the default rule (at this size every join sends its small side to all ranks) and with
    # every join redistributing both sides by the join key
    # and a third time as the library runs them by default: a statement whose tables all lie in one rank's partitions is
    # answered by that rank alone (the first two runs switch that off, so that every statement takes the exchanges)
    runs = [(name, st, None, "1") for name, st in statements.items()] + [(name + "#hashed", st, "0", "1") for name, st in statements.items()] + \
           [(name + "#owner", st, None, None) for name, st in statements.items()]
    for name, (sql, parts), small_side, no_owner in runs:
        for var, val in (("DX_SMALL_SIDE_ROWS", small_side), ("DX_NO_SINGLE_OWNER", no_owner)):
            if val is None:
                os.environ.pop(var, None)
            else:
                os.environ[var] = val
        # together
        e.set_comm(comm)
        e.unload_all()
        # the library's own collective preloadTables on every other statement, the host-side variant on the rest
        loaded = e.load_tables_sharded(parts) if len(results) % 2 == 0 else load_tables_sharded(e, comm, parts, owner)
        rc = e.query(sql, DX_QUERY_COUNT)
        rb = e.query(sql, DX_QUERY_BINDINGS)
        rows, cols = rb.rows()
        shares = [None] * world
        dist.all_gather_object(shares, rows)
        entry = {"count": rc.count, "count_bindings": rb.count, "local_rows": rb.local_rows, "columns": cols, "loaded": loaded,
                 "rows": sorted(tuple(r) for sh in shares for r in sh), "share_sizes": [len(sh) for sh in shares],
                 "plan": rb.plan}
        # alone
        if rank == 0:
            e.set_comm(None)
            e.unload_all()
            alone_loaded = sum(e.load_table(p) for p in dict.fromkeys(parts))
            ra = e.query(sql, DX_QUERY_BINDINGS)
            arows, acols = ra.rows()
            entry["alone"] = {"count": ra.count, "columns": acols, "rows": [tuple(r) for r in arows], "loaded": alone_loaded}
        results[name] = entry
        dist.barrier()
    # ---- a rank that fails must not strand its peer: both return the same error
    from opus_amd.engine import DxError
    failures = {}
    os.environ.pop("DX_SMALL_SIDE_ROWS", None)
    os.environ["DX_NO_SINGLE_OWNER"] = "1"   # the failing statement takes the exchanges; the single-owner case follows below
    e.set_comm(comm)
    e.comm_selftest()
    sql, parts = statements["Q2.txt"]
    e.unload_all()
    e.load_tables_sharded(parts)
    if rank == 1:   # this rank forgets one table of the statement: its planner fails, the other's does not
        e.unload_all()
        for p in parts[1:]:
            (e.load_table if owner(int(p.split("-")[0])) == rank else e.declare_remote_table)(p)
    try:
        e.query(sql, DX_QUERY_COUNT)
        failures["missing_table"] = 0
    except DxError as ex:
        failures["missing_table"] = ex.code
    # a table whose files do not exist on its owner: the collective load fails on every rank
    try:
        e.unload_all()
        e.load_tables_sharded(["00001-_3=_no_such_predicate_", "00000-_3=_no_such_predicate_"])
        failures["missing_file"] = 0
    except DxError as ex:
        failures["missing_file"] = ex.code
    # and the group still works afterwards
    e.unload_all()
    e.load_tables_sharded(parts)
    failures["after"] = e.query(sql, DX_QUERY_COUNT).count
    failures["want_after"] = results["Q2.txt"]["count"]
    # the single-owner case: the owner has lost the statement's tables, the other rank has nothing to do with them
    os.environ.pop("DX_NO_SINGLE_OWNER", None)
    own_name = next(n for n, r in results.items() if n.endswith("#owner") and "alone" in r["plan"] and "Q" in n)
    sql1, parts1 = statements[own_name.split("#")[0]]
    e.unload_all()
    e.load_tables_sharded(parts1)
    if owner(int(parts1[0].split("-")[0])) == rank:
        e.unload_all()
    try:
        e.query(sql1, DX_QUERY_COUNT)
        failures["owner_lost_table"] = 0
    except DxError as ex:
        failures["owner_lost_table"] = ex.code
    e.unload_all()
    e.load_tables_sharded(parts1)
    failures["owner_after"] = e.query(sql1, DX_QUERY_COUNT).count
    failures["owner_want_after"] = results[own_name]["count"]
    # tables placed by another rule than partition % world (the public calls allow it): the rank the names point at holds
    # none of them, so the one-rank shortcut must refuse on every rank instead of counting nothing; through the
    # exchanges the same placement answers correctly
    other = lambda q: (owner(q) + 1) % world  # noqa: E731
    e.unload_all()
    load_tables_sharded(e, comm, parts1, other)
    try:
        e.query(sql1, DX_QUERY_COUNT)
        failures["other_placement_shortcut"] = 0
    except DxError as ex:
        failures["other_placement_shortcut"] = ex.code
    os.environ["DX_NO_SINGLE_OWNER"] = "1"
    failures["other_placement_exchanges"] = e.query(sql1, DX_QUERY_COUNT).count
    os.environ.pop("DX_NO_SINGLE_OWNER", None)
    # a communicator of several ranks without allgather is refused when it is set, not when a join first needs it
    import ctypes
    from opus_amd.host.comm import ALLGATHER_FN, DxComm
    st = comm.struct
    bare = DxComm(st.rank, st.world, None, st.exchange_host, st.exchange, st.allreduce_sum, ALLGATHER_FN(), st.allreduce_sum_dev, 0)
    failures["no_allgather"] = e.lib.dx_set_comm(e.ctx, ctypes.byref(bare))
    e.set_comm(comm)
    all_f = [None] * world
    dist.all_gather_object(all_f, failures)
    if rank == 0:
        results["__failures__"] = all_f
        with open(out_path, "w") as f:
            json.dump(results, f)
    e.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
