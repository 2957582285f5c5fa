#!/usr/bin/env python
"""End-to-end wall-clock of the drop-in CLI on Config-2-sized JSON files
(BASELINE.json metric, second half: "wall-clock for ... Track-mAP").

    python tools/wallclock_config2.py [--videos 200 --frames 300 --dets 50 --cats 1203]

Writes the synthetic pair as JSON under /tmp (excluded from the timing), then
runs tools/eval_on_tao_amodal.py on it with TAOAMD_TIMING=1 and prints one JSON
line: total seconds and the parse / flatten / upload / kernels / download /
summarize split."""
import argparse
import io
import contextlib
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=200)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--dets", type=int, default=50)
    ap.add_argument("--cats", type=int, default=1203)
    ap.add_argument("--dir", default="/tmp/taoamd_wallclock")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    gt_p, pr_p = os.path.join(a.dir, "gt.json"), os.path.join(a.dir, "pred.json")
    from tao_amodal_amd.synth import synth
    gt, dt = synth(V=a.videos, F=a.frames, C=a.cats, dets_per_frame=a.dets)
    with open(gt_p, "w") as f:
        json.dump(gt.to_json(), f, separators=(",", ":"))
    with open(pr_p, "w") as f:
        json.dump(dt.to_json(), f, separators=(",", ":"))
    sizes = {"gt_MB": round(os.path.getsize(gt_p) / 1e6, 1),
             "pred_MB": round(os.path.getsize(pr_p) / 1e6, 1)}
    os.environ["TAOAMD_TIMING"] = "1"
    spec = importlib.util.spec_from_file_location(
        "cli", os.path.join(ROOT, "tools", "eval_on_tao_amodal.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    import torch
    torch.zeros(1, device="cuda")          # context creation is not the CLI's
    from tao_amodal_amd.evaluation._core import TIMING
    out = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(out):
        cli.main(["--track_result", pr_p, "--annotation", gt_p, "--output_log",
                  os.path.join(a.dir, "eval.log")])
    total = time.perf_counter() - t0
    print(json.dumps({"workload": vars(a), "files": sizes,
                      "total_s": round(total, 2),
                      "split_s": {k: round(v, 3) for k, v in TIMING.items()},
                      "first_lines": out.getvalue().splitlines()[:2]}))


if __name__ == "__main__":
    main()
