"""Golden vectors of the reference TaoEval for the modes the CLI does not use
(SURVEY.md 8(f) rank 2): iou_3d_type in {avg_iou, imagenetvid} and
use_cats = 0.  Same conventions as make_golden.py; writes
tests/golden/<name>/tao_modes.json.gz (per-cell IoUs) and tao_modes.npz
(precision / recall, categories that are not all -1); and of the reference
LVISEval with use_cats = 0 (evaluate + accumulate; its summarize() raises
IndexError on the frequency groups): lvis_nocats.json.gz (per-image IoUs,
per-range dt_matches / dt_ignore / gt_ignore) and lvis_nocats.npz."""
import gzip
import json
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402
from make_golden import reference_make_track_ids_unique  # noqa: E402

MODES = {"avg_iou": dict(iou_3d_type="avg_iou", use_cats=1),
         "imagenetvid": dict(iou_3d_type="imagenetvid", use_cats=1),
         "nocats": dict(iou_3d_type="3d_iou", use_cats=0)}


def run(name):
    _, ref_tao = refenv.import_reference()
    out = os.path.join(HERE, name)
    gt_path, pred_path = os.path.join(out, "gt.json"), os.path.join(out, "pred.json")
    cells, arrays = {}, {}
    for mode, cfg in MODES.items():
        preds = json.load(open(pred_path))
        reference_make_track_ids_unique()(preds)
        lg = logging.getLogger("golden.modes")
        lg.propagate = False
        te = ref_tao.TaoEval(ref_tao.Tao(gt_path), preds, logger=lg,
                             iou_3d_type=cfg["iou_3d_type"])
        te.params.use_cats = cfg["use_cats"]
        te.run()
        cells[mode] = [{"key": [int(k[0]), int(k[1])], "ious": np.asarray(v).tolist()}
                       for k, v in te.ious.items() if len(v) > 0 or True
                       if not (isinstance(v, list) and len(v) == 0)]
        p, r = te.eval["precision"], te.eval["recall"]
        k = np.flatnonzero((p.reshape(p.shape[0], p.shape[1], p.shape[2], -1) > -1)
                           .any(axis=(0, 1, 3)))
        arrays[mode + "_k"] = k
        arrays[mode + "_precision"] = p[:, :, k]
        arrays[mode + "_recall"] = r[:, k]
        arrays[mode + "_shape"] = np.array(p.shape)
        arrays[mode + "_results"] = np.array([float(v) for v in te.results.values()])
    np.savez_compressed(os.path.join(out, "tao_modes.npz"), **arrays)
    with gzip.GzipFile(os.path.join(out, "tao_modes.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(cells, separators=(",", ":")).encode())
    print(name, {m: float(arrays[m + "_results"][0]) for m in MODES})


def run_lvis_nocats(name):
    ref_lvis, _ = refenv.import_reference()
    out = os.path.join(HERE, name)
    le = ref_lvis.LVISEval(os.path.join(out, "gt.json"),
                           os.path.join(out, "pred.json"), "bbox")
    le.params.use_cats = 0
    le.evaluate()
    le.accumulate()
    try:
        le.summarize()
        summarize_error = ""
    except Exception as e:          # noqa: BLE001 -- the behaviour is the datum
        summarize_error = type(e).__name__
    cells = [{"key": int(k[0]), "ious": np.asarray(v).tolist()}
             for k, v in le.ious.items()
             if not (isinstance(v, list) and len(v) == 0)]
    per_rng = []
    for e in le.eval_imgs:
        if e is None:
            continue
        per_rng.append({"image_id": int(e["image_id"]),
                        "rng": [float(x) for x in e["visibility_rng"]],
                        "dt_ids": [int(x) for x in e["dt_ids"]],
                        "gt_ids": [int(x) for x in e["gt_ids"]],
                        "dt_matches": np.asarray(e["dt_matches"]).astype(int).tolist(),
                        "dt_ignore": np.asarray(e["dt_ignore"]).astype(int).tolist(),
                        "gt_ignore": np.asarray(e["gt_ignore"]).astype(int).tolist()})
    np.savez_compressed(os.path.join(out, "lvis_nocats.npz"),
                        precision=le.eval["precision"], recall=le.eval["recall"])
    with gzip.GzipFile(os.path.join(out, "lvis_nocats.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps({"cells": cells, "eval_imgs": per_rng,
                            "summarize_error": summarize_error},
                           separators=(",", ":")).encode())
    p = le.eval["precision"]
    print(name, "lvis nocats", p.shape, float(np.mean(p[p > -1])), summarize_error)


if __name__ == "__main__":
    for n in sys.argv[1:] or ["f1", "f2", "f4"]:
        run(n)
        run_lvis_nocats(n)
