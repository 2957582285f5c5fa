#!/usr/bin/env python
""""Reference, 1 core": the ACTUAL reference evaluator timed on one host core
 -- development container only, it
imports /root/reference through tests/golden/refenv.py; nothing of it travels.

    python tools/time_reference.py [--out profiles/reference_1core.json]

Workloads: golden fixture F1, and a down-scaled Config 2 (20 videos x 300
frames x 50 dets, 100 categories: the full Config 2 through the reference is
infeasible, it materialises 60 000 x 1203 cells).  Box pairs are counted where
the reference computes them: every m x n handed to ``mask_utils.iou`` (image
level) and every ``bb_intersect_union`` call (track level), so the Mpair/s
figure is the same unit bench.py reports.  Beside it: the C port
(oracle/tao_oracle.c) on the same inputs, the number bench.py's cpu_baseline
gives on the GPU box."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def time_reference(gt_path, pred_path):
    import refenv
    ref_lvis, ref_tao = refenv.import_reference()
    import tao_amodal.evaluation.lvis_amodal.eval as le_mod
    import tao_amodal.evaluation.tao_amodal.eval as te_mod
    counts = {"lvis": 0, "tao": 0}
    real_iou = le_mod.mask_utils.iou

    def counting_iou(dt, gt, crowd):
        counts["lvis"] += len(dt) * len(gt)
        return real_iou(dt, gt, crowd)
    real_biu = te_mod.bb_intersect_union

    def counting_biu(d, g):
        counts["tao"] += 1
        return real_biu(d, g)

    class Proxy:
        def __getattr__(self, k):
            return counting_iou if k == "iou" else getattr(real_mask_utils, k)
    real_mask_utils = le_mod.mask_utils
    le_mod.mask_utils = Proxy()
    te_mod.bb_intersect_union = counting_biu
    try:
        t0 = time.perf_counter()
        le = ref_lvis.LVISEval(gt_path, pred_path, "bbox")
        le.run()
        t_lvis = time.perf_counter() - t0
        preds = json.load(open(pred_path))
        import make_golden
        make_golden.reference_make_track_ids_unique()(preds)
        t0 = time.perf_counter()
        te = ref_tao.TaoEval(ref_tao.Tao(gt_path), preds)
        te.run()
        t_tao = time.perf_counter() - t0
    finally:
        le_mod.mask_utils = real_mask_utils
        te_mod.bb_intersect_union = real_biu
    return {"lvis_s": round(t_lvis, 2), "tao_s": round(t_tao, 2),
            "lvis_pairs": counts["lvis"], "tao_pairs": counts["tao"],
            "Mpair_per_s": round((counts["lvis"] + counts["tao"]) / (t_lvis + t_tao) / 1e6, 5),
            "AP": [float(le.results["AP"]), float(te.results["AP"])]}


def time_port(gt, dt):
    import orclib
    from tao_amodal_amd import flatten
    fl = flatten.flatten_lvis(gt, dt)
    dt.track_id, _ = flatten.make_track_ids_unique(dt)
    ft = flatten.flatten_tao(gt, dt)
    orclib.set_threads(1)
    t0 = time.perf_counter()
    orclib.run_flat(fl, detail=False)
    ot = orclib.run_flat(ft, detail=False)
    t = time.perf_counter() - t0
    pairs = fl.n_pairs + ot["pairs"]
    return {"s": round(t, 4), "pairs": int(pairs), "Mpair_per_s": round(pairs / t / 1e6, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "reference_1core.json"))
    a = ap.parse_args()
    import logging
    logging.disable(logging.WARNING)
    from tao_amodal_amd.columns import DTColumns, GTColumns
    from tao_amodal_amd.synth import synth
    import goldenio
    out = {"what": "the reference evaluator (tools/eval_on_tao_amodal.py's two "
                   "evaluators, class API) on ONE core of the development "
                   "container, pairs counted at mask_utils.iou / "
                   "bb_intersect_union; 'port' = oracle/tao_oracle.c, 1 thread, "
                   "same inputs",
           "host": {"cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
                    "python": sys.version.split()[0]},
           "workloads": {}}
    work = "/tmp/taoamd_time_reference"
    os.makedirs(work, exist_ok=True)
    for name, make in (
            ("F1 (5 videos x 20 frames x 12 dets, 6 categories)",
             lambda: tuple(c.from_json(j) for c, j in zip(
                 (GTColumns, DTColumns), goldenio.load_inputs("f1")))),
            ("down-scaled Config 2 (20 videos x 300 frames x 50 dets, 100 categories)",
             lambda: synth(seed=20240807, V=20, F=300, C=100, dets_per_frame=50))):
        gt, dt = make()
        gp, pp = os.path.join(work, "gt.json"), os.path.join(work, "pred.json")
        gt.write_json(gp)
        dt.write_json(pp)
        ref = time_reference(gp, pp)
        port = time_port(gt, dt)
        assert port["pairs"] == ref["lvis_pairs"] + ref["tao_pairs"], (port, ref)
        out["workloads"][name] = {"boxes": len(dt), "reference": ref, "port": port,
                                  "port_over_reference": round(
                                      port["Mpair_per_s"] / ref["Mpair_per_s"], 1)}
        print(name, out["workloads"][name], flush=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
