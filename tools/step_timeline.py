#!/usr/bin/env python
"""One step of the bench out of a rocprofv3 --kernel-trace:

    python tools/step_timeline.py gpurun_out/prof_<prefix>/ktr profiles/<prefix>_step_timeline.txt

Start / end / duration (us, relative to the step's first kernel) of every
kernel of ONE timed step, with the queue it ran on: which chain is the step,
what runs beside what.  The step is cut at the launches of the anchor kernel
(the image level's first kernel of a pass); a step in the middle of the run is
taken."""
import csv
import glob
import os
import sys

ANCHORS = ("ss_split", "seg_tile_kernel")       # first image-level kernel of a pass


def main(src, out):
    path = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"]
    anchor = next(a for a in ANCHORS if any(name(r).startswith(a) or ("void " + a) in name(r)
                                            for r in rows))
    # image-level launches of the anchor: the larger grid of the two evaluators'
    marks = [i for i, r in enumerate(rows) if anchor in name(r)]
    grid = lambda r: int(r.get("Grid_Size") or r["Grid_Size_X"])  # noqa: E731
    big = max(grid(rows[i]) for i in marks)
    marks = [i for i in marks if grid(rows[i]) == big]
    # (both evaluators may launch the anchor with the same grid: the image level's
    # is the one on the busiest stream -- the caller's, which carries its chain)
    sid = lambda r: r.get("Stream_Id") or r["Queue_Id"]  # noqa: E731
    busy = {}
    for r in rows:
        busy[sid(r)] = busy.get(sid(r), 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    main = max(busy, key=busy.get)
    if any(sid(rows[i]) == main for i in marks):
        marks = [i for i in marks if sid(rows[i]) == main]
    k = len(marks) // 2
    t0 = int(rows[marks[k]]["Start_Timestamp"])
    t1 = int(rows[marks[k + 1]]["Start_Timestamp"])
    queues = {}
    lines = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < t0 - 60000 or s >= t1 - 60000:
            continue
        q = queues.setdefault(r.get("Stream_Id") or r["Queue_Id"], len(queues))
        lines.append("%8.1f %8.1f %7.1f  stream %d  %s" % ((s - t0) / 1e3, (e - t0) / 1e3,
                                                        (e - s) / 1e3, q, name(r)[:90]))
    with open(out, "w") as f:
        f.write("One timed step under rocprofv3 --kernel-trace (step %d of %d): start / end / "
                "duration in us\nrelative to the image level's first kernel, stream (in order of appearance), "
                "kernel.  Next step starts at %.1f.\n(from %s)\n\n"
                % (k, len(marks), (t1 - t0) / 1e3, path))
        f.write("\n".join(lines) + "\n")
    print("wrote", out, len(lines), "kernels, step", (t1 - t0) / 1e3, "us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
