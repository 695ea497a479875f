#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of one round into the files kept under profiles/.

  python profiles/summarise.py r01 gpurun_out/prof_r1 gpurun_out/pmc_r1_fetch gpurun_out/pmc_r1_write

inputs : <stats dir>/<tag>_kernel_stats.csv            rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...
         <fetch dir>/<tag>_counter_collection.csv      rocprofv3 --pmc FETCH_SIZE  (its own run)
         <write dir>/<tag>_counter_collection.csv      rocprofv3 --pmc WRITE_SIZE  (its own run)
outputs: profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_traffic.json, profiles/<tag>_summary.md

FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads
(128-byte requests tallied as 64), so reads are doubled; WRITE_SIZE is taken as it is.
"""
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("dx::", "").replace("void ", "")
    return re.split(r"\(", name, maxsplit=1)[0]


def pmc(path):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in acc.items()}


def main(tag, stats_dir, fetch_dir, write_dir):
    src = os.path.join(stats_dir, tag + "_kernel_stats.csv")
    shutil.copyfile(src, os.path.join(HERE, tag + "_kernel_stats.csv"))
    rows = list(csv.DictReader(open(src)))
    fetch = pmc(os.path.join(fetch_dir, tag + "_counter_collection.csv"))
    write = pmc(os.path.join(write_dir, tag + "_counter_collection.csv"))
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        fk = fetch.get(k, (0, 0.0))[1]
        wk = write.get(k, (0, 0.0))[1]
        traffic[k] = {"launches_sampled": fetch.get(k, write.get(k))[0], "fetch_size_kib": fk, "write_size_kib": wk,
                      "read_bytes": fk * 1024 * 2, "write_bytes": wk * 1024, "bytes": fk * 1024 * 2 + wk * 1024}
    json.dump(traffic, open(os.path.join(HERE, tag + "_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    with open(os.path.join(HERE, tag + "_summary.md"), "w") as f:
        f.write("# %s: rocprofv3 summaries (MI355X, LUBM-100, `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-file-leg`, see `collect.sh`)\n\n" % tag)
        f.write("Kernel trace (`rocprofv3 --kernel-trace --stats`; 7 partition calls = 1 sizing + 1 warm-up + 5 timed; "
                "full CSV in `%s_kernel_stats.csv`). HBM traffic per launch from separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` "
                "runs of the same command with `--steps 2` (KiB as the counters report them; `read GB` doubles FETCH_SIZE, see "
                "`summarise.py`).\n\n" % tag)
        f.write("| kernel | calls | total ms | avg us | % | FETCH_SIZE KiB | read GB | WRITE_SIZE KiB | write GB |\n|---|---|---|---|---|---|---|---|---|\n")
        for r in rows[:32]:
            k = short(r["Name"])
            t = traffic.get(k)
            f.write("| %s | %s | %.3f | %.2f | %s | %s | %s | %s | %s |\n" % (
                k, r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"],
                "%.0f" % t["fetch_size_kib"] if t else "", "%.3f" % (t["read_bytes"] / 1e9) if t else "",
                "%.0f" % t["write_size_kib"] if t else "", "%.3f" % (t["write_bytes"] / 1e9) if t else ""))


if __name__ == "__main__":
    main(*sys.argv[1:5])
