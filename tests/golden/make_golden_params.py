"""Golden vectors of the reference evaluators run with EDITED ``params``
(reference lvis_amodal/eval.py:51-52,59-105, tao_amodal/eval.py:178-233): a
subset of the images / videos and a subset of the categories in another order
than the sorted one.  Writes tests/golden/<name>/params_subset.npz: the
subsets, precision / recall, the result values and the printed lines' keys.

The subsets are a function of the fixture alone (below), so the test applies
the same ones.  Development container only (needs /root/reference)."""
import json
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402
from make_golden import reference_make_track_ids_unique  # noqa: E402


def subsets(gt):
    """(image ids, video ids, category ids) the restricted runs evaluate."""
    vids = sorted(v["id"] for v in gt["videos"])
    keep_v = vids[1::2] if len(vids) > 2 else vids[:1]
    imgs = sorted(i["id"] for i in gt["images"] if i["video_id"] in set(keep_v))
    imgs = imgs[: max(3, (2 * len(imgs)) // 3)]
    cats = sorted(c["id"] for c in gt["categories"])
    keep_c = list(reversed(cats[::2]))            # every other one, descending
    return imgs, keep_v, keep_c


def run(name):
    ref_lvis, ref_tao = refenv.import_reference()
    out = os.path.join(HERE, name)
    gt_path, pred_path = os.path.join(out, "gt.json"), os.path.join(out, "pred.json")
    gt = json.load(open(gt_path))
    imgs, vids, cats = subsets(gt)
    le = ref_lvis.LVISEval(gt_path, pred_path, "bbox")
    le.params.img_ids = list(imgs)
    le.params.cat_ids = list(cats)
    le.run()
    preds = json.load(open(pred_path))
    reference_make_track_ids_unique()(preds)
    lg = logging.getLogger("golden.params")
    lg.propagate = False
    te = ref_tao.TaoEval(ref_tao.Tao(gt_path), preds, logger=lg)
    te.params.vid_ids = list(vids)
    te.params.cat_ids = list(cats)
    te.run()
    np.savez_compressed(
        os.path.join(out, "params_subset.npz"),
        img_ids=np.array(imgs), vid_ids=np.array(vids), cat_ids=np.array(cats),
        lvis_precision=le.eval["precision"], lvis_recall=le.eval["recall"],
        lvis_results=np.array([float(v) for v in le.results.values()]),
        lvis_freq_groups=np.array([len(g) for g in le.freq_groups]),
        tao_precision=te.eval["precision"], tao_recall=te.eval["recall"],
        tao_results=np.array([float(v) for v in te.results.values()]))
    print(name, "imgs", len(imgs), "vids", vids, "cats", cats,
          "LVIS AP", le.results["AP"], "TAO AP", te.results["AP"])


if __name__ == "__main__":
    for n in sys.argv[1:] or ["f1", "f5"]:
        run(n)
