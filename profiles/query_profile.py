"""Per-statement plans and device times at a given LUBM size: python profiles/query_profile.py <universities> [query names...]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # the repository
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from opus_amd import lubm
from opus_amd.engine import DX_QUERY_COUNT, Engine

U = int(sys.argv[1])
only = sys.argv[2:]
dev = torch.device("cuda:0")
parts = []
for b in range(0, U, 50):
    arr, nt = lubm.generate(b, min(U, b + 50), seed=0, ontology=0)
    parts.append(torch.from_numpy(arr).to(dev))
text = torch.cat(parts) if len(parts) > 1 else parts[0]
del parts
ci = lubm.class_index_text(8, 0)
e = Engine(0)
t0 = time.perf_counter()
st = e.partition_device(text.data_ptr(), text.numel(), ci, 8)
print("partition", st.n_triples, "triples", (time.perf_counter() - t0) * 1e3, "ms (first call)", flush=True)
st = e.partition_device(text.data_ptr(), text.numel(), ci, 8)
print("partition again", st.ms_total, "ms device", flush=True)
queries = bench.translated_queries(e, ci, 0)
names = sorted(queries, key=lambda n: int(n[1:-4]))
out = {}
for n in names:
    if only and n not in only and n[:-4] not in only:
        continue
    sql, tabs = bench.parse_translated(queries[n])
    e.unload_all()
    rows = sum(e.load_table(p) for p in tabs)
    best, wall = 1e9, 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        r = e.query(sql, DX_QUERY_COUNT)
        w = (time.perf_counter() - t0) * 1e3
        d = r.device_ms
        plan = r.plan
        cnt = r.count
        r.close()
        best, wall = min(best, d), min(wall, w)
    print("%s count=%d rows_loaded=%d device=%.4f ms wall=%.4f ms GB/s=%.0f" % (n, cnt, rows, best, wall, rows * 16 / best / 1e6), flush=True)
    print(plan, flush=True)
    out[n] = {"device_ms": best, "wall_ms": wall, "count": cnt, "bytes": rows * 16}
geo = float(np.exp(np.mean([np.log(v["device_ms"]) for v in out.values()])))
print("geomean device", geo, "wall", float(np.exp(np.mean([np.log(v["wall_ms"]) for v in out.values()]))))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "qprof_%d.json" % U), "w"))
e.close()
