"""Fixture F6 and its golden vectors: the reference LVISEval with
iou_type="segm" (L/eval.py:54-58,70-73,179-191,
L/lvis.py:171-193, L/results.py:42-62).

Inputs (written to tests/golden/f6/):
  gt.json        images carry height / width; every annotation carries a
                 "segmentation" in one of the four forms ann_to_rle accepts:
                 one polygon, several polygons, uncompressed RLE, compressed
                 RLE (the last two made with the reference's own mask_utils
                 from a polygon -- they are inputs, not expectations)
  pred.json      detections with "bbox"; every third one also brings its own
                 polygon, the others get the box polygon of L/results.py:48-49
  pred_rle.json  the same detections as compressed RLE only (no "bbox"): area
                 and bbox then come from the mask (L/results.py:54-60)

Golden (lvis_segm.json.gz, lvis_segm.npz), for both prediction files: the
compressed RLE the reference builds for every ground truth / detection it
evaluates, mask areas and boxes, the per-cell IoUs and per-range decisions,
precision, recall, results.

Run in the development container only (needs /root/reference)."""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import refenv  # noqa: E402
from tao_amodal_amd.synth import synth  # noqa: E402

H, W = 90, 120          # small frames: the masks stay a few dozen runs long
SCALE = W / 1280.0


def _poly_around(rng, box, n):
    """A star-shaped polygon around a box, float vertices, partly outside the
    frame when the (amodal) box is."""
    x, y, w, h = box
    cx, cy = x + w / 2.0, y + h / 2.0
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = rng.uniform(0.85, 1.0, n)
    px = cx + rad * (w / 2.0) * np.cos(ang)
    py = cy + rad * (h / 2.0) * np.sin(ang)
    return [round(float(v), 2) for xy in zip(px, py) for v in xy]


def build_inputs(mask_utils):
    gt, dt = synth(seed=66, V=3, F=6, C=6, dets_per_frame=9,
                   gt_tracks_per_video=5, n_present=3, n_neg=1)
    gtj, predj = gt.to_json(), dt.to_json()
    rng = np.random.default_rng(6)
    for im in gtj["images"]:
        im["height"], im["width"] = H, W
    for k, a in enumerate(gtj["annotations"]):
        a["bbox"] = [round(v * SCALE, 1) for v in a["bbox"]]
        a["area"] = a["bbox"][2] * a["bbox"][3]
        x, y, w, h = a["bbox"]
        form = k % 4
        if form == 0:
            seg = [[x, y, x, y + h, x + w, y + h, x + w, y]]
        elif form == 1:
            seg = [_poly_around(rng, [x, y, w * 0.6, h], 6),
                   _poly_around(rng, [x + w * 0.5, y, w * 0.5, h * 0.7], 5)]
        else:
            poly = [_poly_around(rng, a["bbox"], 11)]
            rle = mask_utils.merge(mask_utils.frPyObjects(poly, H, W))
            if form == 2:       # uncompressed: run lengths, column-major
                m = mask_utils.decode(rle).reshape(-1, order="F")
                edges = np.flatnonzero(np.diff(m)) + 1
                counts = np.diff(np.r_[0, edges, m.size]).tolist()
                if m[0] == 1:
                    counts = [0] + counts
                seg = {"size": [H, W], "counts": counts}
            else:
                seg = {"size": [H, W], "counts": rle["counts"].decode("ascii")}
        a["segmentation"] = seg
    for k, p in enumerate(predj):
        p["bbox"] = [round(v * SCALE, 1) for v in p["bbox"]]
        if k % 3 == 0:
            p["segmentation"] = [_poly_around(rng, p["bbox"], 12)]
    pred_rle = []
    for p in predj:
        x, y, w, h = p["bbox"]
        seg = p.get("segmentation", [[x, y, x, y + h, x + w, y + h, x + w, y]])
        rle = mask_utils.merge(mask_utils.frPyObjects(seg, H, W))
        q = {k: v for k, v in p.items() if k not in ("bbox", "segmentation")}
        q["segmentation"] = {"size": [H, W],
                             "counts": rle["counts"].decode("ascii")}
        pred_rle.append(q)
    return gtj, predj, pred_rle


def _counts_str(rle):
    c = rle["counts"]
    return c.decode("ascii") if isinstance(c, bytes) else c


def run_reference(ref_lvis, gt_path, pred_path):
    le = ref_lvis.LVISEval(gt_path, pred_path, "segm")
    le.run()
    P = le.params
    n_img, n_rng = len(P.img_ids), len(P.visibility_rng)
    cells = []
    for (im, c), v in le.ious.items():
        if isinstance(v, list) and len(v) == 0:
            continue
        cells.append({"key": [int(im), int(c)], "ious": np.asarray(v).tolist()})
    evals = []
    for c in range(len(P.cat_ids)):
        for a in range(n_rng):
            for i in range(n_img):
                e = le.eval_imgs[(c * n_rng + a) * n_img + i]
                if e is None:
                    continue
                evals.append({
                    "key": [int(e["image_id"]), int(e["category_id"]), a],
                    "dt_ids": [int(x) for x in e["dt_ids"]],
                    "gt_ids": [int(x) for x in e["gt_ids"]],
                    "dt_matches": np.asarray(e["dt_matches"]).astype(int).tolist(),
                    "dt_ignore": np.asarray(e["dt_ignore"]).astype(int).tolist(),
                    "gt_ignore": np.asarray(e["gt_ignore"]).astype(int).tolist()})
    # the masks the evaluator actually compared (after _to_mask)
    gt_rle = {int(a["id"]): _counts_str(a["segmentation"])
              for lst in le._gts.values() for a in lst}
    dt_rle = {int(a["id"]): _counts_str(a["segmentation"])
              for lst in le._dts.values() for a in lst}
    dt_area = {int(a["id"]): float(a["area"]) for a in le.lvis_dt.anns.values()}
    dt_bbox = {int(a["id"]): [float(v) for v in a["bbox"]]
               for a in le.lvis_dt.anns.values()}
    return {"cells": cells, "evals": evals, "gt_rle": gt_rle, "dt_rle": dt_rle,
            "dt_area": dt_area, "dt_bbox": dt_bbox,
            "results": [[k, float(v)] for k, v in le.results.items()]}, \
        le.eval["precision"], le.eval["recall"]


def main():
    ref_lvis, _ = refenv.import_reference()
    import pycocotools.mask as mask_utils
    out = os.path.join(HERE, "f6")
    os.makedirs(out, exist_ok=True)
    gtj, predj, pred_rle = build_inputs(mask_utils)
    paths = {}
    for name, obj in (("gt", gtj), ("pred", predj), ("pred_rle", pred_rle)):
        paths[name] = os.path.join(out, name + ".json")
        with open(paths[name], "w") as f:
            json.dump(obj, f, separators=(",", ":"))
    golden, arrays = {}, {}
    for which in ("pred", "pred_rle"):
        g, p, r = run_reference(ref_lvis, paths["gt"], paths[which])
        golden[which] = g
        arrays[which + "_precision"], arrays[which + "_recall"] = p, r
        print(which, p.shape, "AP", dict(map(tuple, g["results"]))["AP"],
              len(g["cells"]), "cells")
    np.savez_compressed(os.path.join(out, "lvis_segm.npz"), **arrays)
    with gzip.GzipFile(os.path.join(out, "lvis_segm.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(golden, separators=(",", ":")).encode())


if __name__ == "__main__":
    main()
