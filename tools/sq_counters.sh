#!/bin/bash
# SQ occupancy / stall counters of the step's kernels (one rocprofv3 --pmc pass):
#   bash tools/sq_counters.sh <out-prefix> [bench args]
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles, MI355X_MICROARCH.md)
set -e
PREFIX=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/sq_$PREFIX
rm -rf $OUT; mkdir -p $OUT $R/profiles
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc -o sq -- python $R/bench.py --steps 6 --warmup 2 --no-cpu --no-wallclock "$@" > /dev/null 2> $OUT/err.log || true
rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/lds -o lds -- python $R/bench.py --steps 6 --warmup 2 --no-cpu --no-wallclock "$@" > /dev/null 2>> $OUT/err.log || true
cd $R
python - $OUT profiles/${PREFIX}_sq_counters.txt <<'P'
import csv, glob, sys, collections, os
out, dst = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"][:64], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (k, g), c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc < 1e5: continue
    rows.append((wc, k, g, m))
rows.sort(reverse=True)
with open(dst, "w") as fh:
    fh.write("SQ counters per launch (mean), rocprofv3 --pmc, bench default workload; shares of SQ_WAVE_CYCLES:\n"
             "wait = SQ_WAIT_ANY (parked on s_waitcnt / barrier), stall = SQ_WAIT_INST_ANY (issue stall),\n"
             "active = SQ_ACTIVE_INST_ANY, valu = SQ_ACTIVE_INST_VALU; insts = SQ_INSTS_VALU per wave\n\n")
    for wc, k, g, m in rows[:40]:
        sh = lambda n: 100.0 * m.get(n, 0) / wc
        fh.write("%-64s grid %9s  wait %5.1f%% stall %5.1f%% active %5.1f%% valu %5.1f%%  VALU insts/wave %8.0f  waves %8.0f\n" % (
            k, g, sh("SQ_WAIT_ANY"), sh("SQ_WAIT_INST_ANY"), sh("SQ_ACTIVE_INST_ANY"), sh("SQ_ACTIVE_INST_VALU"),
            m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_WAVES", 1), 1), m.get("SQ_WAVES", 0)))
    # second pass: the LDS array (summed over the chip's 256 CUs; GRBM_GUI_ACTIVE over its 8 XCDs)
    g = glob.glob(os.path.join(out, "lds", "**", "*counter_collection.csv"), recursive=True)
    if g:
        agg2 = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(g[0])):
            agg2[(r["Kernel_Name"][:64], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        fh.write("\nLDS array, second pass: busy = SQ_LDS_IDX_ACTIVE / 256 CUs as a share of the kernel's cycles "
                 "(GRBM_GUI_ACTIVE / 8 XCDs),\nconflict = SQ_LDS_BANK_CONFLICT share of the busy cycles, LDS instructions per launch\n\n")
        rows2 = []
        for (k, gs), c in agg2.items():
            m = {n: sum(v) / len(v) for n, v in c.items()}
            cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
            if cyc < 2e4: continue
            rows2.append((m.get("SQ_LDS_IDX_ACTIVE", 0), k, gs, m, cyc))
        for idx, k, gs, m, cyc in sorted(rows2, reverse=True)[:16]:
            fh.write("%-64s grid %9s  LDS busy %5.1f%%  conflict %5.1f%%  LDS insts %10.0f  kernel cycles %9.0f\n" % (
                k, gs, 100.0 * idx / 256.0 / cyc, 100.0 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(idx, 1),
                m.get("SQ_INSTS_LDS", 0), cyc))
print(open(dst).read())
P
mkdir -p gpurun_out/profiles_out && cp profiles/${PREFIX}_sq_counters.txt gpurun_out/profiles_out/
rm -rf $OUT/pmc $OUT/lds
