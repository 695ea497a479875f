#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of one round into the files kept under profiles/.

  python profiles/summarise.py r03 gpurun_out/prof_r03 gpurun_out/pmc_r03_fetch gpurun_out/pmc_r03_write [gpurun_out/pmc_r03_sqa gpurun_out/pmc_r03_sqb]

inputs : <stats dir>/<tag>_kernel_stats.csv            rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...
         <fetch dir>/<tag>_counter_collection.csv      rocprofv3 --pmc FETCH_SIZE  (its own run)
         <write dir>/<tag>_counter_collection.csv      rocprofv3 --pmc WRITE_SIZE  (its own run)
         <sq dirs>/<tag>_counter_collection.csv        rocprofv3 --pmc SQ_...      (optional; a run each, see collect.sh)
outputs: profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_traffic.json, profiles/<tag>_summary.md, profiles/<tag>_pmc_sq.json

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


def pmc_all(path):
    """kernel -> counter -> mean value per launch"""
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: {c: v / n for c, (n, v) in d.items()} for k, d in acc.items()}


SQ_KERNELS = ["k_tokenise<true>", "k_entry_claim", "k_entry_first", "k_entry_assign", "k_encode_tiles", "k_split_scatter<unsigned long>",
              "k_vpi_flags", "k_rs_scatter<unsigned long, false, 8>"]


def main(tag, stats_dir, fetch_dir, write_dir, *sq_dirs):
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
        if sq_dirs:
            sq = defaultdict(dict)
            for d in sq_dirs:
                for k, c in pmc_all(os.path.join(d, tag + "_counter_collection.csv")).items():
                    sq[k].update(c)
            json.dump({k: sq[k] for k in sorted(sq)}, open(os.path.join(HERE, tag + "_pmc_sq.json"), "w"), indent=1, sort_keys=True)
            avg_us = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}
            f.write("\nWave and instruction counters per launch (two more `--pmc` runs; all values in `%s_pmc_sq.json`). "
                    "`VALU issue` = SQ_INSTS_VALU x 4 (a wave's vector instruction occupies its SIMD for four cycles) over the SIMD cycles "
                    "the kernel had (1024 SIMDs x duration x 2.4 GHz); `waves/SIMD` = SQ_WAVE_CYCLES x 4 (the counter ticks once per four cycles a wave is resident) over the same SIMD "
                    "cycles, i.e. averaged over the kernel's life including its ramp-up and tail; `LDS wait`, `any wait` = SQ_WAIT_INST_LDS, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES.\n\n" % tag)
            f.write("| kernel | avg us | SQ_WAVES | SQ_INSTS_VALU (M) | SQ_INSTS_LDS (M) | SQ_INSTS_SALU (M) | VMEM rd/wr (M) | VALU issue | waves/SIMD | LDS wait | any wait | bank-conflict cycles / LDS active |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
            for k in SQ_KERNELS:
                c = sq.get(k)
                if not c or k not in avg_us:
                    continue
                simd_cycles = 1024 * avg_us[k] * 1e-6 * 2.4e9
                g = lambda n: c.get(n, float("nan"))  # noqa: E731
                f.write("| %s | %.1f | %.0f | %.1f | %.1f | %.1f | %.2f / %.2f | %.2f | %.1f | %.3f | %.3f | %.2f |\n" % (
                    k, avg_us[k], g("SQ_WAVES"), g("SQ_INSTS_VALU") / 1e6, g("SQ_INSTS_LDS") / 1e6, g("SQ_INSTS_SALU") / 1e6,
                    g("SQ_INSTS_VMEM_RD") / 1e6, g("SQ_INSTS_VMEM_WR") / 1e6, g("SQ_INSTS_VALU") * 4 / simd_cycles,
                    g("SQ_WAVE_CYCLES") * 4 / simd_cycles, g("SQ_WAIT_INST_LDS") / max(g("SQ_WAVE_CYCLES"), 1),
                    g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_ACTIVE_INST_LDS"), 1)))


if __name__ == "__main__":
    main(*sys.argv[1:7])
