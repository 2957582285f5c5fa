#!/bin/bash
# The drop-in CLI on the bench workload under rocprofv3 (kernel + memory-copy
# trace): what the GPU does between the parse and the printed tables.
#   bash tools/cli_trace.sh <out-prefix> [--config 3s]
set -e
PREFIX=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/cli_$PREFIX
D=/tmp/taoamd_cli_trace
rm -rf $OUT $D; mkdir -p $OUT $D
cd $R
python - "$D" "$@" <<'P'
import sys, os
sys.path.insert(0, os.getcwd())
import bench
from tao_amodal_amd.synth import synth
d = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
a = bench.parse()
gt, dt = synth(seed=a.seed, V=a.videos, F=a.frames, C=a.cats, dets_per_frame=a.dets)
gt.write_json(os.path.join(d, "gt.json")); dt.write_json(os.path.join(d, "pred.json"))
P
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/eval_on_tao_amodal.py --track_result $D/pred.json --annotation $D/gt.json --output_log $D/eval.log"
TAOAMD_TIMING=1 $CMD > $OUT/plain.out 2> $OUT/plain.err || true
TAOAMD_TIMING=1 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace --stats -d $OUT/tr -o tr -- $CMD > $OUT/traced.out 2> $OUT/traced.err || true
cd $R
python - $OUT <<'P'
import csv, glob, os, sys, collections
out = sys.argv[1]
k = glob.glob(os.path.join(out, "tr", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(k)))
m = glob.glob(os.path.join(out, "tr", "**", "*memory_copy_trace.csv"), recursive=True)
cop = list(csv.DictReader(open(m[0]))) if m else []
t0 = min([int(r["Start_Timestamp"]) for r in rows] + [int(r["Start_Timestamp"]) for r in cop])
t1 = max([int(r["End_Timestamp"]) for r in rows] + [int(r["End_Timestamp"]) for r in cop])
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    a = agg[r["Kernel_Name"][:70]]; a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
with open(os.path.join(out, "summary.txt"), "w") as f:
    f.write("GPU activity span %.1f ms; kernels %.1f ms busy in %d launches\n" % (
        (t1 - t0) / 1e6, sum(a[0] for a in agg.values()) / 1e6, len(rows)))
    byd = collections.defaultdict(lambda: [0, 0, 0])
    for r in cop:
        b = byd[r.get("Direction", "?")]; b[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); b[1] += 1
        b[2] += int(r.get("Size") or r.get("Bytes") or 0)
    for d, b in byd.items():
        f.write("copies %-22s %8.1f ms  %5d calls  %9.1f MB\n" % (d, b[0] / 1e6, b[1], b[2] / 1e6))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
        f.write("%9.2f ms %5d  %s\n" % (a[0] / 1e6, a[1], name))
    # a coarse timeline: busy ms per 50 ms window (kernels / copies)
    W = 50e6
    nb = int((t1 - t0) / W) + 1
    kb, cb = [0] * nb, [0] * nb
    for r in rows:
        kb[int((int(r["Start_Timestamp"]) - t0) / W)] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for r in cop:
        cb[int((int(r["Start_Timestamp"]) - t0) / W)] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    f.write("\nwindow(ms)  kernel-busy(ms)  copy-busy(ms)\n")
    for i in range(nb):
        f.write("%6d %8.1f %8.1f\n" % (i * 50, kb[i] / 1e6, cb[i] / 1e6))
print(open(os.path.join(out, "summary.txt")).read())
P
grep "taoamd timing" $OUT/plain.err $OUT/traced.err || true
find $OUT/tr -name "*.csv" -size +8M -delete
rm -rf $D
