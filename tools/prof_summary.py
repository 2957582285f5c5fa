#!/usr/bin/env python
"""Condense rocprofv3 output (gpurun_out/<dir>) into the small files that are
committed under profiles/:

    python tools/prof_summary.py gpurun_out/prof3 profiles/r01_v2

writes  <prefix>_kernel_stats.csv  (copy of rocprofv3 --kernel-trace --stats)
        <prefix>_pmc.json          per-kernel mean FETCH_SIZE / WRITE_SIZE (KB,
                                   one --pmc pass each) and the derived HBM
                                   traffic = (2*FETCH + WRITE) * 1024 B -- the
                                   gfx950 correction of MI355X_MICROARCH.md,
                                   section HBM (FETCH_SIZE counts 128-B
                                   requests as 64 B for wide coalesced reads)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def mean_counter(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main(src, prefix):
    stats = glob.glob(os.path.join(src, "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], prefix + "_kernel_stats.csv")
    out = {}
    f = glob.glob(os.path.join(src, "fetch*counter_collection.csv"))
    w = glob.glob(os.path.join(src, "write*counter_collection.csv"))
    fetch = mean_counter(f[0]) if f else {}
    write = mean_counter(w[0]) if w else {}
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k), write.get(k)
        out[k] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                  "hbm_bytes_corrected": None if fk is None or wk is None
                  else int((2 * fk + wk) * 1024)}
    if out:
        with open(prefix + "_pmc.json", "w") as fh:
            json.dump({"note": "mean per launch; traffic = (2*FETCH_SIZE + "
                               "WRITE_SIZE) * 1024 (gfx950 correction)",
                       "kernels": out}, fh, indent=1)
    print("wrote", prefix + "_*")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
