#!/usr/bin/env python
"""Condense rocprofv3 output (gpurun_out/<dir>) into the small files that are
committed under profiles/:

    python tools/prof_summary.py gpurun_out/prof3 profiles/r02_v1 ["<workload>"]

writes  <prefix>_kernel_stats.csv  (copy of rocprofv3 --kernel-trace --stats)
        <prefix>_pmc.json          per (kernel, grid size) mean FETCH_SIZE /
                                   WRITE_SIZE (KB, one --pmc pass each) and the
                                   derived HBM traffic = (2*FETCH + WRITE) *
                                   1024 B -- the gfx950 correction of
                                   MI355X_MICROARCH.md, section HBM (FETCH_SIZE
                                   counts 128-B requests as 64 B for wide
                                   coalesced reads); "workload" = the
                                   config.workload string of the bench line the
                                   passes were taken on (bench.py only uses a
                                   summary whose workload matches its own)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import sources_digest  # noqa: E402  (the kernels the passes were taken on)


def mean_counter(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main(src, prefix, workload=None):
    stats = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], prefix + "_kernel_stats.csv")
    f = glob.glob(os.path.join(src, "**", "fetch*counter_collection.csv"), recursive=True)
    w = glob.glob(os.path.join(src, "**", "write*counter_collection.csv"), recursive=True)
    fetch = mean_counter(f[0]) if f else {}
    write = mean_counter(w[0]) if w else {}
    out = collections.defaultdict(list)
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k), write.get(k)
        out[k[0]].append({
            "grid": k[1], "launches": (fk or wk)[1],
            "FETCH_SIZE_KB": fk and fk[0], "WRITE_SIZE_KB": wk and wk[0],
            "hbm_bytes_corrected": None if fk is None or wk is None
            else int((2 * fk[0] + wk[0]) * 1024)})
    if out:
        with open(prefix + "_pmc.json", "w") as fh:
            json.dump({"note": "mean per launch, by kernel and grid size (work "
                               "items); traffic = (2*FETCH_SIZE + WRITE_SIZE) * "
                               "1024 (gfx950 correction)",
                       "workload": workload, "sources": sources_digest(),
                       "kernels": out}, fh, indent=1)
    print("wrote", prefix + "_*")


if __name__ == "__main__":
    main(*sys.argv[1:4])
