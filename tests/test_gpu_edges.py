"""-m gpu: degenerate inputs through the class API against the Python oracle
(empty cell tables, no surviving detection, nothing to evaluate, one box)."""
import copy

import numpy as np
import pytest

from goldenio import load_inputs
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _run_both(gt, pred):
    from tao_amodal_amd.evaluation.lvis_amodal import LVIS, LVISEval, LVISResults
    from tao_amodal_amd.evaluation.tao_amodal import Tao, TaoEval, TaoResults
    lg = LVIS(copy.deepcopy(gt))
    le = LVISEval(lg, LVISResults(lg, copy.deepcopy(pred)), "bbox")
    le.run()
    want = pyoracle.lvis_eval(gt, pred)
    assert np.array_equal(le.eval["precision"], want["precision"])
    assert np.array_equal(le.eval["recall"], want["recall"])
    assert [float(v) for v in le.results.values()] == \
        [float(v) for v in want["results"].values()]
    assert le.result_lines() == want["printed"]
    p2 = copy.deepcopy(pred)
    pyoracle.make_track_ids_unique(p2)
    tg = Tao(copy.deepcopy(gt))
    te = TaoEval(tg, TaoResults(tg, copy.deepcopy(p2)))
    te.run()
    wt = pyoracle.tao_eval(gt, p2, frame_order="timeline")
    assert np.array_equal(te.eval["precision"], wt["precision"])
    assert np.array_equal(te.eval["recall"], wt["recall"])
    assert te.result_lines() == wt["printed"]
    return le, te


def test_single_detection_single_ground_truth():
    gt, pred = load_inputs("f1")
    a = gt["annotations"][0]
    keep_img = a["image_id"]
    gt["annotations"] = [a]
    gt["tracks"] = [t for t in gt["tracks"] if t["id"] == a["track_id"]]
    pred = [{"image_id": keep_img, "category_id": a["category_id"],
             "bbox": list(a["bbox"]), "score": 0.5, "track_id": 1,
             "video_id": [i for i in gt["images"] if i["id"] == keep_img][0]["video_id"]}]
    le, te = _run_both(gt, pred)
    # one TP, no FP: tp / (fp + tp + eps) = 1 / (1 + 2**-52), as in the reference
    assert le.results["AP"] == 1 / (1 + np.spacing(1)) == te.results["AP"]


def test_no_detection_survives_the_federated_filter_lvis():
    """Every prediction sits in a category that is neither present nor
    negative in its image: all cells have ground truth only."""
    gt, pred = load_inputs("f1")
    cats = {c["id"] for c in gt["categories"]}
    by_img = {}
    for a in gt["annotations"]:
        by_img.setdefault(a["image_id"], set()).add(a["category_id"])
    imgs = {i["id"]: i for i in gt["images"]}
    out = []
    for p in pred:
        free = cats - by_img.get(p["image_id"], set()) - \
            set(imgs[p["image_id"]]["neg_category_ids"])
        if free:
            q = dict(p)
            q["category_id"] = sorted(free)[0]
            out.append(q)
    from tao_amodal_amd.evaluation.lvis_amodal import LVIS, LVISEval, LVISResults
    lg = LVIS(copy.deepcopy(gt))
    le = LVISEval(lg, LVISResults(lg, copy.deepcopy(out)), "bbox")
    le.run()
    want = pyoracle.lvis_eval(gt, out)
    assert np.array_equal(le.eval["precision"], want["precision"])
    assert np.array_equal(le.eval["recall"], want["recall"])
    assert le.flat.dt_score.size == 0


def test_everything_ignored_gives_minus_one():
    gt, pred = load_inputs("f1")
    for a in gt["annotations"]:
        a["ignore"] = 1
    for t in gt["tracks"]:
        t["ignore"] = 1
    le, te = _run_both(gt, pred)
    assert le.results["AP"] == -1 and te.results["AP"] == -1
    assert (le.eval["precision"] == -1).all()


def test_tao_raises_like_the_reference_without_usable_predictions():
    from tao_amodal_amd.evaluation.tao_amodal import Tao, TaoResults
    gt, pred = load_inputs("f1")
    for p in pred:
        p["bbox"][2] = 0          # zero area: dropped by the strict filter
    with pytest.raises(ValueError, match="no predicted annotations"):
        TaoResults(Tao(gt), pred)


def test_reference_doctest_boxes_through_the_kernels():
    """The known answers of the reference's own doctests for
    bb_intersect_union (tao_amodal/eval.py:21-30, non-crowd cases):
    (i, u) = (400, 400), (100, 400), (25, 100), (400, 900) -- as IoUs of
    one-frame tracks (track level) and of single boxes (image level)."""
    from tao_amodal_amd import engine, flatten
    from tao_amodal_amd.columns import DTColumns, GTColumns
    pairs = [([0, 0, 20, 20], [0, 0, 20, 20], 400 / 400), ([0, 0, 20, 20], [0, 0, 10, 10], 100 / 400),
             ([10, 20, 10, 10], [10, 20, 5, 5], 25 / 100), ([0, 0, 20, 20], [0, 0, 30, 30], 400 / 900)]
    cats = [{"id": c + 1, "name": "c%d" % c, "frequency": "f"} for c in range(len(pairs))]
    gt = {"info": {}, "categories": cats,
          "videos": [{"id": 1, "name": "v", "neg_category_ids": [], "not_exhaustive_category_ids": []}],
          "images": [{"id": 1, "video_id": 1, "frame_index": 0, "neg_category_ids": [],
                      "not_exhaustive_category_ids": []}],
          "tracks": [{"id": c + 1, "category_id": c + 1, "video_id": 1} for c in range(len(pairs))],
          "annotations": [{"id": c + 1, "image_id": 1, "track_id": c + 1, "category_id": c + 1,
                           "bbox": g, "area": g[2] * g[3], "visibility": 1.0, "out_of_frame": False}
                          for c, (d, g, _) in enumerate(pairs)]}
    preds = [{"image_id": 1, "category_id": c + 1, "bbox": d, "score": 0.9, "track_id": c + 1,
              "video_id": 1} for c, (d, g, _) in enumerate(pairs)]
    G, D = GTColumns.from_json(gt), DTColumns.from_json(preds)
    want = np.array([w for _, _, w in pairs])
    got = engine.evaluate_flat(flatten.flatten_lvis(G, D), detail=True)
    assert np.array_equal(got["iou"], want)
    D.track_id, _ = flatten.make_track_ids_unique(D)
    got = engine.evaluate_flat(flatten.flatten_tao(G, D), detail=True)
    assert np.array_equal(got["iou"], want)
