#!/usr/bin/env python3
"""Headline benchmark: DAP post-stage partitioning throughput + LUBM Q1-Q14 query latency on MI355X.

One step = one pass of both hot paths over one synthetic LUBM batch per GPU:
  1. partition the N-Triples text (already resident in HBM) into first-level partitions, vertical
     partition tables and indexes (dx_partition_device);
  2. run the 13 LUBM queries of the reference's query folder (there is no Q13) the way
     QueryProcessor.executeNonTypeQuery does: tables resident, timed count (dx_query).

Prints ONE JSON line (rank 0). `value` is partitioning throughput in triples/s over all ranks;
the query geo-mean in ms is carried next to it. With --gpus N > 1 (launched by torchrun, one rank
per GPU) the ranks work on ONE data set together: every rank tokenises its block of universities, the
ranks agree on one dictionary, route the triples to the owners of their first-level partitions
(partition q lives on rank q % N) and answer the 13 queries together, with the join inputs exchanged by
join key -- all over the communicator the library makes on RCCL (dx_comm_rccl_init).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def translated_queries(engine, class_index_text, ontology, gather=None):
    """Host-side work that stays outside the timed path: the reference's translator turns SPARQL into
    SQL from the node index, class index and partition statistics (QueryTranslator.scala:25-119)."""
    from opus_amd import lubm
    from opus_amd.host import translator as T
    strs_cache = {}
    u, t = engine.node_index()
    consts = set()
    parsed = {}
    for name, text in lubm.queries(ontology).items():
        pq = T.ParsedQuery(text)
        parsed[name] = text
        consts.update(pq.mappings.values())
    ids = engine.lookup(sorted(consts))
    want = {int(i): c for i, c in zip(ids, sorted(consts)) if i >= 0}
    sel = np.isin(u, np.fromiter(want.keys(), dtype=np.int64, count=len(want))) if want else np.zeros(len(u), bool)
    ni = {}
    type_strs = engine.decode(np.unique(t[sel])) if sel.any() else []
    tmap = dict(zip(np.unique(t[sel]).tolist(), type_strs))
    for uu, tt in zip(u[sel].tolist(), t[sel].tolist()):
        ni.setdefault(want[uu], []).append(tmap[tt])
    stat = {}
    for tb in engine.tables():
        if tb["in_layout"]:
            k = "%05d" % tb["partition"]
            stat[k] = stat.get(k, 0) + tb["rows"]
    if gather is not None:  # several ranks hold shares of the node index and of the partitions
        merged_ni, merged_stat = {}, {}
        for ni_r, stat_r in gather((ni, stat)):
            for k, v in ni_r.items():
                merged_ni.setdefault(k, [])
                merged_ni[k] += [x for x in v if x not in merged_ni[k]]
            for k, v in stat_r.items():
                merged_stat[k] = merged_stat.get(k, 0) + v
        ni, stat = merged_ni, merged_stat
    ci = T.load_class_index(class_index_text)
    out = {}
    for name, text in parsed.items():
        tr = T.translate_query(name, text, lambda x: ni.get(x, []), ci, stat)
        if tr is None:
            raise RuntimeError("query %s cannot be indexed" % name)
        out[name] = tr
    del strs_cache
    return out


def parse_translated(text):
    """QueryProcessor.main (QueryProcessor.scala:68-88)."""
    partitions, sql = [], None
    for line in text.split("\n"):
        if line.startswith("TP") or line.startswith(">>>>>"):
            continue
        if line.startswith("partitions"):
            partitions = line.split(" ")[1].split(",")
        elif line:
            sql = " ".join(line.split(" ")[1:])
    return sql, partitions


def cpu_baseline(arr, n_triples, class_index, n_partitions, plans, gpu_counts):
    """The CPU restatement (oracle/cpu_port.cpp) timed on this host's cores, rank 0 only, on the benchmark's own
    data set: the partitioner on one thread and on all the cores this process may use (one shared dictionary), then
    the 13 translated statements over the restatement's own vertical-partition tables as scans + hash joins + count,
    keyed on term ids and keyed on the terms' strings (what the reference's Spark plan compares), on one thread and
    on all. Protocol as for the GPU: tables resident, one untimed run, then the timed count. The counts must equal
    the GPU's."""
    from oracle import cpu_port
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    one = cpu_port.run(arr, class_index, n_partitions)
    ds = cpu_port.Dataset(arr, class_index, n_partitions, threads=threads)   # the same run on all cores; keeps its tables
    legs = {}
    for label, t, by_string in (("ids_1_thread", 1, False), ("ids_all_threads", threads, False),
                                ("strings_1_thread", 1, True), ("strings_all_threads", threads, True)):
        times = {}
        for name, (sql, parts) in plans.items():
            c, _ = ds.count(sql, parts, threads=t, string_keyed=by_string)          # untimed run
            if c != gpu_counts[name]:
                raise RuntimeError("CPU restatement counts %d rows for %s, the GPU %d" % (c, name, gpu_counts[name]))
            c, secs = ds.count(sql, parts, threads=t, string_keyed=by_string)
            times[name] = secs * 1e3
        legs[label] = {"geomean_ms": math.exp(sum(math.log(max(v, 1e-6)) for v in times.values()) / len(times)),
                       "query_ms": times}
    all_cores = n_triples / ds.seconds
    ds.close()
    import shutil
    spark = "present" if (shutil.which("java") and shutil.which("spark-submit")) else \
        "absent on this host (no java / spark-submit): the reference itself cannot be timed beside"
    return {"spark_local": spark, "value": n_triples / one["seconds"], "unit": "triples/s", "cores": 1, "kind": "port",
            "sample": "oracle/cpu_port.cpp on the benchmark's own data (%d triples): partitioner on one thread in %.1f s and on %d "
                      "threads in %.1f s; the 13 statements as scans + hash joins over its tables, counts equal to the GPU's"
                      % (n_triples, one["seconds"], threads, n_triples / all_cores),
            "all_cores": {"value": all_cores, "unit": "triples/s", "cores": threads,
                          "sample": "the same restatement on %d threads with one shared dictionary" % threads},
            "query_geomean_ms": legs["strings_1_thread"]["geomean_ms"],
            "query_legs": legs}


def file_leg(eng, arr, text, n_bytes, class_index, n_partitions):
    """The drop-in calls as dx-dap makes them, outside the timed steps and never part of `value`: the N-Triples file
    through the pinned staging buffers into HBM and through the partitioner (dx_partition_file), then the reference's
    on-disk layout written by the library (dx_part_write_layout: Parquet + text files)."""
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="dx_file_leg_")
    try:
        path = os.path.join(d, "instances.nt")
        (arr if arr is not None else text.cpu().numpy()).tofile(path)
        cip = os.path.join(d, "class_index.txt")
        open(cip, "w").write(class_index)
        eng.partition_file(path, cip, n_partitions)   # first call: page cache warm, staging buffers allocated
        t0 = time.perf_counter()
        st = eng.partition_file(path, cip, n_partitions)
        t_file = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.write_layout(os.path.join(d, "store"), "bench", d)
        t_layout = time.perf_counter() - t0
        lt = eng.layout_times()
        size = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(os.path.join(d, "store")) for f in fs)
        return {"partition_file_ms": t_file * 1e3, "of_which_device_ms": st.ms_total,
                "file_GBps": n_bytes / t_file / 1e9,
                "write_layout_ms": t_layout * 1e3, "layout_bytes": size,
                "write_layout_legs_ms": {"directories_and_text_files": lt["prepare_ms"], "device_encode": lt["encode_ms"],
                                         "device_to_host": lt["d2h_ms"], "file_write_first_to_last": lt["write_wall_ms"],
                                         "file_write_summed_over_threads": lt["write_busy_ms"], "threads": lt["threads"]},
                "file_write_GBps": lt["bytes"] / max(lt["write_wall_ms"], 1e-6) / 1e6,
                "end_to_end_triples_per_s": st.n_triples / (t_file + t_layout),
                "note": "file in the page cache -> pinned staging (read and upload overlapped) -> HBM -> partitions; then the "
                        "reference's layout on local disk: Parquet pages (dictionary pages, snappy streams like Spark's) encoded on the GPU, the "
                        "files' bytes copied to pinned host memory piece by piece while host threads write the pieces that have "
                        "landed (device_to_host and file_write overlap); end_to_end = triples / (partition_file + write_layout)"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def join_group(eng, args, rank, world, device):
    """Puts this rank's context into the group of all ranks. Default: the communicator the library makes itself on
    librccl (rank 0's unique id travels over torch.distributed, which the launcher has set up anyway). Every rank
    first proves that it can load RCCL, so that no rank waits in ncclCommInitRank for one that never comes; if any
    cannot, or with --comm torch / a gloo rehearsal, all ranks use torch.distributed through callbacks instead."""
    import torch
    import torch.distributed as dist
    from opus_amd.engine import DxError, Engine
    from opus_amd.host.comm import TorchComm
    red = device if args.backend == "nccl" else torch.device("cpu")
    kind = "torch.distributed callbacks (%s)" % args.backend
    if args.comm == "rccl" and args.backend == "nccl" and not args.single_device:
        try:
            uid = Engine.rccl_unique_id()
            ok = 1
        except (DxError, OSError) as ex:
            print("rank %d: cannot load RCCL natively (%s)" % (rank, ex), file=sys.stderr)
            uid, ok = None, 0
        flag = torch.tensor([ok], dtype=torch.int64, device=red)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:
            box = [uid if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            os.environ.setdefault("DX_RCCL_INIT_TIMEOUT_MS", "120000")  # a bench does not wait ten minutes for a peer
            try:
                eng.comm_rccl_init(box[0], rank, world)
                eng.comm_selftest()       # every collective with known patterns, before anything relies on the group
                ok = 1
            except DxError as ex:
                print("rank %d: the library's RCCL group failed (%s)" % (rank, ex), file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int64, device=red)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 1:
                return None, "library-native RCCL (ncclSend/ncclRecv groups on the context's stream)"
            kind += " after the library's own RCCL group failed its self-test on some rank"
    comm = TorchComm()
    eng.set_comm(comm)
    eng.comm_selftest()
    return comm, kind


def crosscheck_against_one_gpu(args, rank, world, local_rank, device, ontology, ci):
    """The sharded paths against one GPU on a data set small enough for rank 0 to hold alone (LUBM-8): all ranks
    partition it together and answer the 13 queries together, rank 0 then does the same alone on a second context,
    and the counts must agree. Returns (universities, counts) on rank 0."""
    import torch
    import torch.distributed as dist
    from opus_amd import lubm
    from opus_amd.engine import DX_QUERY_COUNT, Engine
    n_univ = max(8, world)
    e = Engine(local_rank)
    try:
        join_group(e, args, rank, world, device)
        arr, _ = lubm.generate(rank * n_univ // world, (rank + 1) * n_univ // world, seed=0, ontology=ontology, degree_pool=3)
        t = torch.from_numpy(arr).to(device)
        torch.cuda.synchronize()
        e.partition_device(t.data_ptr(), int(arr.size), ci, args.partitions)
        def gather(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
        queries = translated_queries(e, ci, ontology, gather=gather)
        names = sorted(queries, key=lambda n: int(n[1:-4]))
        together = {}
        for n in names:
            sql, parts = parse_translated(queries[n])
            e.unload_all()
            e.load_tables_sharded(parts)
            r = e.query(sql, DX_QUERY_COUNT)
            together[n] = int(r.count)
            r.close()
    finally:
        e.close()
    if rank != 0:
        return None
    whole, _ = lubm.generate(0, n_univ, seed=0, ontology=ontology, degree_pool=3)
    e1 = Engine(local_rank)
    try:
        t = torch.from_numpy(whole).to(device)
        torch.cuda.synchronize()
        e1.partition_device(t.data_ptr(), int(whole.size), ci, args.partitions)
        alone = {}
        for n in names:
            sql, parts = parse_translated(queries[n])
            e1.unload_all()
            e1.load_tables(list(dict.fromkeys(parts)))
            r = e1.query(sql, DX_QUERY_COUNT)
            alone[n] = int(r.count)
            r.close()
    finally:
        e1.close()
    if alone != together:
        raise RuntimeError("sharded query counts differ from one GPU's on LUBM-%d: %r vs %r" % (n_univ, together, alone))
    if not any(alone.values()):
        raise RuntimeError("cross-check data set answers every query with 0 rows")
    return n_univ, alone


def sharded_query_leg(args, eng0, plans, qnames, want_counts, rank, world, local_rank, barrier):
    """Rank 0 writes its partitions as the reference lays them out; every rank opens that layout (one
    dictionary), loads the tables of the first-level partitions it owns, and all ranks answer the 13 queries
    together. Returns rank 0's timings; the counts must equal the ones rank 0 got alone."""
    import shutil
    import tempfile

    import torch.distributed as dist
    from opus_amd.engine import DX_QUERY_COUNT, Engine
    from opus_amd.host import shard
    from opus_amd.host.comm import TorchComm, load_tables_sharded
    box = [tempfile.mkdtemp(prefix="dx_sharded_") if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    base = box[0]
    try:
        if rank == 0:
            eng0.write_layout(base, "bench", os.path.join(base, "bench_local"))
        barrier()
        names = [None]
        if rank == 0:
            names[0] = (plans, want_counts)
        dist.broadcast_object_list(names, src=0)   # every rank runs rank 0's statements against rank 0's data
        plans0, want0 = names[0]
        e = Engine(local_rank)
        e.open_dataset(base, "bench", args.partitions)
        comm = TorchComm()
        e.set_comm(comm)
        times, got = {n: [] for n in qnames}, {}
        for it in range(args.warmup + args.steps):
            for n in qnames:
                sql, parts = plans0[n]
                e.unload_all()
                load_tables_sharded(e, comm, parts, lambda q: shard.partition_owner(q, world))
                barrier()
                t0 = time.perf_counter()
                r = e.query(sql, DX_QUERY_COUNT)
                dt = (time.perf_counter() - t0) * 1e3
                got[n] = r.count
                r.close()
                if it >= args.warmup:
                    times[n].append(dt)
        e.close()
        bad = {n: (got[n], want0[n]) for n in qnames if got[n] != want0[n]}
        if bad:
            raise RuntimeError("sharded query counts differ from the single-GPU counts: %r" % bad)
        med = {n: float(np.median(times[n])) for n in qnames}
        geo = math.exp(sum(math.log(max(v, 1e-9)) for v in med.values()) / len(med))
        out = {"geomean_ms": geo, "query_ms": med, "dataset": "rank 0's block, tables owned by partition %% %d" % world,
               "counts_equal_single_gpu": True}
    except BaseException:
        # a rank that fails must leave at once: waiting for the others here would keep them waiting for it
        if rank == 0:
            shutil.rmtree(base, ignore_errors=True)
        raise
    barrier()
    if rank == 0:
        shutil.rmtree(base, ignore_errors=True)
    return out


def hetero_leg(args):
    """`--workload swdf|dbpedia`: the heterogeneous-schema configurations of BASELINE.json (configs[3], configs[4]) on one
    GPU: a synthetic data set of that shape (many small vertical partitions / thousands of predicates with skew), the
    partitioner timed as in the default run, then the query shapes of the reference's SWDF and DBpedia folders
    (tests/golden/query_shapes.json: no constants of the real data sets, only the patterns) instantiated over the data
    and run warm, with the cost of loading the many small tables a statement names reported per table."""
    import random

    import torch
    from opus_amd.engine import DX_QUERY_COUNT, DxError, Engine
    from opus_amd.host import translator as T
    from opus_amd.workloads import hetero
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shape_queries as SQ
    from oracle import dap_oracle  # only parses the sample lines and names directories (the checker's helpers, nothing timed)

    shape = args.workload
    n_triples = args.triples or 12_000_000
    text, ci, k, sample = hetero.generate_large(shape, n_triples, seed=0, n_partitions=args.partitions)
    n_bytes = int(text.size)
    dev = torch.from_numpy(text).cuda()
    del text
    torch.cuda.synchronize()
    eng = Engine(0)
    eng.partition_device(dev.data_ptr(), n_bytes, ci, k)
    for _ in range(args.warmup):
        eng.partition_device(dev.data_ptr(), n_bytes, ci, k)
    part_s, stage_ms = [], []
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        st = eng.partition_device(dev.data_ptr(), n_bytes, ci, k)
        part_s.append(time.perf_counter() - t0)
        stage_ms.append([st.ms_stage[i] for i in range(12)])
    elapsed = time.perf_counter() - t_begin
    stage_names = [eng.lib.dx_part_stage_name(i).decode() for i in range(12)]
    stage_avg = {stage_names[i]: float(np.mean([x[i] for x in stage_ms])) for i in range(12) if stage_names[i]}
    tables = [t for t in eng.tables() if t["in_layout"]]
    rows_by_part = {}
    for t in tables:
        rows_by_part["%05d" % t["partition"]] = rows_by_part.get("%05d" % t["partition"], 0) + t["rows"]
    per_part = {}
    for t in tables:
        per_part[t["partition"]] = per_part.get(t["partition"], 0) + 1

    # ---- the query shapes, instantiated over the first lines of the text (what matches there matches in the whole)
    index = SQ.TripleIndex(dap_oracle.parse_instances(sample))
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "query_shapes.json")))
    rng = random.Random(5)
    sparqls = []
    for folder, lst in shapes.items():
        for i, sh in enumerate(lst):
            bind = SQ.instantiate(sh["patterns"], index, rng)
            if bind is not None:
                sparqls.append(("%s/%d" % (folder, i), sh["patterns"], SQ.sparql(sh["patterns"], bind), bind))
    consts = sorted({v for _, _, _, b in sparqls for key, v in b.items() if key[0] == "C"})
    ids = eng.lookup(consts)
    id_of = {c: int(i) for c, i in zip(consts, ids) if i >= 0}
    u, t = eng.node_index()
    sel = np.isin(u, np.fromiter(id_of.values(), dtype=np.int64, count=len(id_of))) if id_of else np.zeros(len(u), bool)
    tvals = np.unique(t[sel])
    tmap = dict(zip(tvals.tolist(), eng.decode(tvals))) if tvals.size else {}
    name_of = {i: c for c, i in id_of.items()}
    ni = {}
    for uu, tt in zip(u[sel].tolist(), t[sel].tolist()):
        ni.setdefault(name_of[uu], []).append(tmap[tt])
    vpi_cache = {}

    def sub_parts(part, uri):  # the VP index rows of one uri in one partition: "s<dir>" / "o<dir>"
        q = int(part)
        if q not in vpi_cache:
            vpi_cache[q] = eng.vp_index(q)
        vu, vr, vp = vpi_cache[q]
        i = id_of.get(uri, -1)
        m = vu == i
        if not m.any():
            return []
        preds = eng.decode(vp[m])
        return sorted({("s" if r == 0 else "o") + dap_oracle.uri_to_storage(pp) for r, pp in zip(vr[m].tolist(), preds)})

    cidx = T.load_class_index(ci)
    times, dev_ms, load_us, counts, tables_named = {}, {}, [], {}, []
    skipped = {"not_indexable": 0, "unsupported_template": 0, "no_directory": 0}
    first = True
    for name, patterns, q, _ in sparqls:
        try:
            tr = T.translate_query(name, q, lambda x: ni.get(x, []), cidx, rows_by_part, sub_parts)
        except NotImplementedError:
            skipped["unsupported_template"] += 1
            continue
        if tr is None:
            skipped["not_indexable"] += 1
            continue
        sql, parts = parse_translated(tr)
        eng.unload_all()
        try:
            t0 = time.perf_counter()
            eng.load_tables(parts)
            t_load = time.perf_counter() - t0
        except DxError:
            skipped["no_directory"] += 1
            continue
        load_us.append((len(parts), t_load * 1e6))
        try:
            if first:
                eng.query(sql, DX_QUERY_COUNT).close()
                first = False
            eng.query(sql, DX_QUERY_COUNT).close()   # a statement's first run sizes its buffers (and may run twice for that)
        except DxError as ex:
            if ex.code != -95:
                raise
            skipped["beyond_an_engine_limit"] = skipped.get("beyond_an_engine_limit", 0) + 1
            continue
        t0 = time.perf_counter()
        r = eng.query(sql, DX_QUERY_COUNT)
        times[name] = (time.perf_counter() - t0) * 1e3
        dev_ms[name] = r.device_ms
        counts[name] = int(r.count)
        tables_named.append(len(parts))
        r.close()
    geo = lambda d: math.exp(sum(math.log(max(v, 1e-9)) for v in d.values()) / max(1, len(d)))  # noqa: E731
    n_rec, n_ent = st.n_triples, int(st.n_dict_entries)
    tok_bytes = n_bytes + 13 * n_rec + 16 * n_ent
    out = {
        "metric": "LUBM Q1-Q14 geo-mean query ms + DAP partitioning triples/sec",
        "value": st.n_triples * args.steps / sum(part_s), "unit": "triples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": "%s-shaped synthetic data (%d triples, %.2f GB of N-Triples text, %d predicates, %d classes), %d first-level "
                               "partitions; query shapes of the reference's swdf / dbpedia folders instantiated over it"
                               % (shape.upper(), st.n_triples, n_bytes / 1e9, int(st.n_predicates), hetero.SHAPES[shape].n_class, k),
                   "parallelism": "single GPU"},
        "value_basis": "triples / partitioner wall time (a step = 1 partitioner call); the queries run after the timed steps",
        "partition_ms": float(np.median(part_s)) * 1e3, "partition_stage_ms": stage_avg,
        "terms": int(st.n_terms), "replication_factor": st.n_rows / max(1, st.n_triples),
        "vp_tables": len(tables), "vp_tables_per_partition": {"min": min(per_part.values()), "max": max(per_part.values())},
        "median_rows_per_table": float(np.median([t["rows"] for t in tables])),
        "queries_run": len(times), "queries_with_answers": sum(1 for c in counts.values() if c > 0), "queries_skipped": skipped,
        "query_geomean_ms": geo(times), "query_geomean_device_ms": geo(dev_ms),
        "tables_named_per_query": {"median": float(np.median(tables_named)), "max": int(max(tables_named))} if tables_named else None,
        "load_tables": {"calls": len(load_us), "tables": int(sum(n for n, _ in load_us)),
                        "us_per_table": float(sum(t for _, t in load_us) / max(1, sum(n for n, _ in load_us))),
                        "worst_call": {"tables": int(max(load_us, key=lambda x: x[1])[0]), "us": float(max(load_us, key=lambda x: x[1])[1])} if load_us else None},
        "roofline": {"bound": "hbm", "kernel": "k_tokenise", "ms": stage_avg["tokenise_hash"], "algorithmic_bytes": tok_bytes,
                     "achieved": tok_bytes / (stage_avg["tokenise_hash"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": tok_bytes / (stage_avg["tokenise_hash"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None},
    }
    if not args.no_cpu_baseline:
        from oracle import cpu_port
        n_sample = min(n_bytes, 300 << 20)   # a bounded piece of the same text: about 2.5 M triples, 10-20 s on one core
        host = dev[:n_sample].cpu().numpy()
        cut = int(np.flatnonzero(host == 10)[-1]) + 1
        one = cpu_port.run(host[:cut], ci, k)
        out["cpu_baseline"] = {"value": one["n_triples"] / one["seconds"], "unit": "triples/s", "cores": 1, "kind": "port",
                               "sample": "oracle/cpu_port.cpp, partitioner only, on the first %d triples of the same text in %.1f s" % (one["n_triples"], one["seconds"])}
    print(json.dumps(out))
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--universities", type=int, default=0,
                    help="universities in total (default: 100 = LUBM-100 on one GPU; 1300 * N / 8 on N > 1 GPUs, so that 8 GPUs "
                         "hold LUBM-1300)")
    ap.add_argument("--partitions", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-file-leg", dest="file_leg", action="store_false",
                    help="skip the extra leg of the single-GPU run that times the drop-in calls on files: dx_partition_file "
                         "(file -> partitions) and dx_part_write_layout (partitions -> the reference's layout on disk)")
    ap.add_argument("--sharded-queries", action="store_true",
                    help="extra leg for --gpus N > 1: rank 0's partitions are split over the ranks by first-level "
                         "partition and the 13 queries run on all GPUs together (hash exchange before every join)")
    ap.add_argument("--replicas", action="store_true",
                    help="--gpus N > 1: independent replicas instead (every rank partitions and queries its own block, no exchange)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="communicator of the sharded run: the library's own over librccl, or torch.distributed through callbacks")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo only to rehearse on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--workload", default="lubm", choices=["lubm", "swdf", "dbpedia"],
                    help="lubm: the headline run. swdf / dbpedia: heterogeneous-schema data of that shape on one GPU (see hetero_leg)")
    ap.add_argument("--triples", type=int, default=0, help="--workload swdf|dbpedia: triples to generate (default 12 million)")
    args = ap.parse_args()
    if args.workload != "lubm":
        if args.gpus != 1:
            raise SystemExit("--workload %s runs on one GPU" % args.workload)
        return hetero_leg(args)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from opus_amd import lubm
    from opus_amd.engine import DX_QUERY_COUNT, Engine
    from opus_amd.host import shard
    device = torch.device("cuda", local_rank)

    # ---- synthetic input, resident in HBM before anything is timed
    ontology = 0
    total_univ = args.universities or (100 if world == 1 else max(world, round(1300 * world / 8)))
    u0, u1 = rank * total_univ // world, (rank + 1) * total_univ // world
    arr, n_triples = lubm.generate(u0, u1, seed=0, ontology=ontology, degree_pool=1000)
    text = torch.from_numpy(arr).to("cuda:%d" % local_rank)
    n_bytes = int(arr.size)
    if world > 1 or args.no_cpu_baseline:
        del arr   # the CPU baseline beside the single-GPU run works on the same bytes
    ci = lubm.class_index_text(args.partitions, ontology)
    torch.cuda.synchronize()
    eng = Engine(local_rank)
    together = world > 1 and not args.replicas
    comm, comm_kind = None, "none"
    crosscheck = None
    sharding_gave_up = None
    if together:
        # The sharded paths are used only if every rank got into the group and the cross-check against one GPU held;
        # otherwise all ranks say so and run independent replicas (a line is still printed, and it says which it was).
        ok, why = 1, ""
        try:
            comm, comm_kind = join_group(eng, args, rank, world, device)
            crosscheck = crosscheck_against_one_gpu(args, rank, world, local_rank, device, ontology, ci)
            if os.environ.get("DX_BENCH_REHEARSE_FAILURE_ON_RANK") == str(rank):  # how the fallback below is rehearsed
                raise RuntimeError("rehearsal of a failed cross-check")
        except Exception as ex:  # noqa: BLE001  (whatever it was, the other ranks have to hear of it)
            ok, why = 0, "rank %d: %s: %s" % (rank, type(ex).__name__, ex)
            print(why, file=sys.stderr)
        red = device if args.backend == "nccl" else torch.device("cpu")
        flag = torch.tensor([ok], dtype=torch.int64, device=red)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            reasons = [None] * world
            dist.all_gather_object(reasons, why)
            sharding_gave_up = "; ".join(r for r in reasons if r) or "some rank failed"
            eng.close()
            eng = Engine(local_rank)
            together, comm, comm_kind, crosscheck = False, None, "none", None

    def gather_objects(obj):
        box = [None] * world
        dist.all_gather_object(box, obj)
        return box

    def partition():
        t0 = time.perf_counter()
        st = eng.partition_device(text.data_ptr(), n_bytes, ci, args.partitions)
        return time.perf_counter() - t0, st

    _, st = partition()  # sizes the allocator's cache and gives the translator its inputs
    queries = translated_queries(eng, ci, ontology, gather=gather_objects if together else None)
    qnames = sorted(queries, key=lambda n: int(n[1:-4]))
    plans = {n: parse_translated(queries[n]) for n in qnames}

    load_rows = {}                                 # rows of the tables a statement reads (the reference's Load_data)

    def run_queries(first):
        times, counts, dev = {}, {}, {}
        for n in qnames:
            sql, parts = plans[n]
            eng.unload_all()                       # unloadTables of the previous query
            if together:                           # tables stay with the rank that owns their partition (q % world)
                eng.load_tables_sharded(parts)
                shard.barrier(device)
            else:
                rows = 0
                for p in parts:                    # preloadTables: resident in HBM, reported as Time_load by the reference
                    rows += eng.load_table(p)      # (one call per table, as the reference loads them: the timed count that
                                                   # follows a single batched load_tables call measures ~5 us longer)
                load_rows[n] = rows
            if first and n == qnames[0]:
                eng.query(sql, DX_QUERY_COUNT).close()   # the reference's one untimed run of the first query
            t0 = time.perf_counter()
            r = eng.query(sql, DX_QUERY_COUNT)
            times[n] = (time.perf_counter() - t0) * 1e3
            counts[n] = r.count
            dev[n] = r.device_ms
            r.close()
        return times, counts, dev

    def barrier():
        shard.barrier(device)

    for w in range(args.warmup):
        partition()
        run_queries(first=(w == 0))

    part_s, q_ms, q_dev, stage_ms = [], [], [], []
    barrier()
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        dt, st = partition()
        part_s.append(dt)
        stage_ms.append([st.ms_stage[i] for i in range(12)])
        times, counts, dev = run_queries(first=False)
        q_ms.append(times)
        q_dev.append(dev)
    barrier()
    elapsed = time.perf_counter() - t_begin

    red_device = device if args.backend == "nccl" else torch.device("cpu")
    elapsed, part_total, total_triples = shard.combine_timings(elapsed, sum(part_s), n_triples, red_device)
    local_counts = {n: int(counts[n]) for n in qnames}
    counts = local_counts if together else shard.combine_counts(local_counts, red_device)  # together: already over all ranks
    # the clustering's instance-scale inputs (computed once per dataset by the reference, so outside the timed step)
    measures = eng.measures()[0]  # with --global-partition a collective call: over all ranks' triples
    # (with --global-partition the timed queries already run over all ranks, and no rank holds all partitions to write out)
    sharded = sharded_query_leg(args, eng, plans, qnames, local_counts, rank, world, local_rank, barrier) \
        if (args.sharded_queries and world > 1 and not together) else None

    if rank == 0:
        med = lambda xs: float(np.median(xs))  # noqa: E731
        q_med = {n: med([q[n] for q in q_ms]) for n in qnames}
        q_dev_med = {n: med([q[n] for q in q_dev]) for n in qnames}
        geo = math.exp(sum(math.log(max(v, 1e-9)) for v in q_med.values()) / len(q_med))
        geo_dev = math.exp(sum(math.log(max(v, 1e-9)) for v in q_dev_med.values()) / len(q_dev_med))
        stage_names = [eng.lib.dx_part_stage_name(i).decode() for i in range(12)]
        stage_avg = {stage_names[i]: float(np.mean([s[i] for s in stage_ms])) for i in range(12) if stage_names[i]}
        # Dominant kernel of the step = the longest of the kernels the library times on its own stream with
        # a hipEvent pair each (tokenise, table clear, slot claims). Algorithmic bytes per launch: what the
        # kernel has to read and write once.
        n_rec, n_occ = st.n_triples, 3 * st.n_triples
        n_ent = int(st.n_dict_entries)                          # occurrences left after the folding inside the tiles
        kernels = {
            # one pass over the text: the text itself read (and, when the line scan ran as a pass of its own, its 2-byte
            # record-start mask per 16 bytes); per record the three occurrences' entry numbers (12 B) and the kind (1 B)
            # written; per dictionary entry its span and its hash (16 B) written
            "k_tokenise (text -> record starts, term spans, hashes, terms of a tile folded onto their first occurrence there)":
                (stage_avg["tokenise_hash"], n_bytes + (n_bytes // 8 if stage_avg["line_scan"] > 0 else 0) + 13 * n_rec + 16 * n_ent),
            "k_ht_clear (16-byte slots)":
                (stage_avg["dict_clear_kernel"], 16 * (1 << max(10, math.ceil(math.log2(max(n_rec // 2, 1)))))),
            # per entry: span and hash read (16 B), its 16-byte slot read, the slot's number written (4 B)
            "k_entry_claim (dictionary slots of the tiles' entries)":
                (stage_avg["dict_claim_kernel"], 36 * n_ent),
        }
        dom = max(kernels, key=lambda k: kernels[k][0])
        dom_ms, dom_bytes = kernels[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        # Only these three kernels have a timer of their own; every other kernel is part of a stage. No stage's time
        # outside them may exceed the dominant kernel's, or one of its kernels might be the longer one (checked from all
        # stage timers of the same steps; the dictionary stage holds the table clear and the claims).
        residual = dict(stage_avg)
        residual["dictionary"] = stage_avg["dictionary"] - stage_avg["dict_clear_kernel"] - stage_avg["dict_claim_kernel"]
        for name in ("tokenise_hash", "dict_clear_kernel", "dict_claim_kernel"):
            residual.pop(name, None)
        longer = {k: v for k, v in residual.items() if v > dom_ms}
        # HBM-side bytes per launch of that kernel from the PMC passes kept under profiles/ (same command,
        # separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs; see profiles/summarise.py); null when the
        # workload is not the profiled one
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
        if os.path.exists(pmc_path) and total_univ == 100 and world == 1 and args.partitions == 8:
            pmc = json.load(open(pmc_path))
            names = [w for w in dom.replace("+", " ").split() if w.startswith("k_")]
            # a templated kernel is listed with its arguments (k_tokenise<true>): the instantiation this run used is the
            # one the profile of the same command holds
            found = [[k for k in pmc if k == nm or k.startswith(nm + "<")] for nm in names]
            if names and all(len(f) == 1 for f in found):
                traffic = sum(pmc[f[0]]["bytes"] for f in found)
        # algorithmic bytes of the whole partitioner: the text once + R * (24 + 16) B per triple (R = replication:
        # 24 B per row of a first-level partition, 16 B per row of a vertical-partition table)
        R = st.n_rows / max(1, st.n_triples)
        hp1_alg_bytes = n_bytes + st.n_rows * 40
        out = {
            "metric": "LUBM Q1-Q14 geo-mean query ms + DAP partitioning triples/sec",
            "value": total_triples * args.steps / part_total,
            "unit": "triples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": "LUBM-%d (%d triples in total; rank 0's share: %d triples, %.2f GB of N-Triples text), %d first-level "
                                   "partitions, 13 LUBM queries (Q1-Q12, Q14)" % (total_univ, total_triples, n_triples, n_bytes / 1e9, args.partitions),
                       "parallelism": ("one data set over %d GPUs: text split by university, one dictionary over all ranks, first-level "
                                       "partition q owned by rank q %% %d, triples routed to the owners, join inputs exchanged by key or the "
                                       "small side sent to all; communicator: %s" % (world, world, comm_kind)
                                       if together else "independent university blocks per GPU, no cross-GPU exchange") if world > 1 else "single GPU"},
            # the metric has two halves: `value` is input triples / wall time of the partitioner calls of the timed
            # steps (slowest rank); the 13 queries of the same steps are reported as query_geomean_ms; ms_per_step covers both
            "value_basis": "triples / partitioner wall time inside the timed steps; a step = 1 partitioner call + 13 timed queries (with their table loads)",
            "partition_ms": med(part_s) * 1e3,
            "partition_stage_ms": stage_avg,
            "partition_algorithmic_GBps": hp1_alg_bytes / med(part_s) / 1e9,
            "replication_factor": R,
            "dictionary_entries": n_ent,
            "query_geomean_ms": geo,
            "query_geomean_device_ms": geo_dev,
            "query_ms": q_med,
            "query_counts": {n: int(counts[n]) for n in qnames},
            "roofline": {"bound": "hbm", "kernel": dom, "ms": dom_ms, "algorithmic_bytes": dom_bytes,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "stages_that_could_hold_a_longer_kernel": longer or None},
            "kernel_GBps": {k: v[1] / (v[0] * 1e-3) / 1e9 for k, v in kernels.items()},
        }
        if load_rows:
            # Query path against the HBM roofline: algorithmic bytes of a counting statement = 16 B per row of every
            # table it names (each (s, o) pair read once; nothing is written for a count), over the statement's device time
            per = {n: {"bytes": 16 * load_rows[n], "device_ms": q_dev_med[n],
                       "GBps": 16 * load_rows[n] / max(q_dev_med[n], 1e-9) / 1e6,
                       "frac": 16 * load_rows[n] / max(q_dev_med[n], 1e-9) / 1e6 / HBM_PEAK_GBS} for n in qnames}
            out["roofline_hp2"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "geomean_frac": math.exp(sum(math.log(max(v["frac"], 1e-12)) for v in per.values()) / len(per)),
                                   "total_frac": sum(v["bytes"] for v in per.values()) / max(sum(v["device_ms"] for v in per.values()), 1e-9) / 1e6 / HBM_PEAK_GBS,
                                   "per_query": per,
                                   "note": "bytes = 16 per row of every table the statement names (the survey's figure), device time by the "
                                           "device's clock from the first kernel's begin to the last kernel's end. A statement that filters on "
                                           "one column reads only that column of the rows that fail, and one that bisects a sorted table "
                                           "reads a few lines of it: both move fewer bytes than the figure, so frac is no measure of HBM "
                                           "traffic and can pass 1 for a single-column count. At LUBM-100 the tables hold 0.03-2.4 M rows "
                                           "(1-40 MB a statement): bound by the latency of a chain of short kernels, not by bandwidth"}
        if measures is not None:
            out["instance_measures"] = {"device_ms": measures["device_ms"], "classes": measures["n_classes"],
                                        "cc_groups": measures["n_cc"], "cc_rows": measures["n_cc_rows"]}
        if sharded is not None:
            out["sharded_queries"] = sharded
        if sharding_gave_up is not None:
            out["sharded_paths_not_used"] = sharding_gave_up
        if crosscheck is not None:
            out["counts_equal_single_gpu"] = {"verified_on": "LUBM-%d partitioned and queried by all ranks together vs by rank 0 alone" % crosscheck[0],
                                              "counts": crosscheck[1]}
        if args.file_leg and world == 1:
            out["file_leg"] = file_leg(eng, arr if not args.no_cpu_baseline else None, text, n_bytes, ci, args.partitions)
            # what the reference's own timer covers (DAP.scala:96-97: from reading the input to the layout on disk), beside
            # `value`, which is HBM-resident text -> HBM-resident tables
            out["end_to_end_triples_per_s"] = out["file_leg"].get("end_to_end_triples_per_s")
        if not args.no_cpu_baseline and world == 1:  # timed beside the single-GPU run only: the other ranks would wait for it
            out["cpu_baseline"] = cpu_baseline(arr, n_triples, ci, args.partitions, plans, {n: int(counts[n]) for n in qnames})
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
