"""Host side of the evaluator classes (no GPU): summaries, printed lines, the
CLI's table, argument/type errors -- against the reference's golden text."""
import importlib.util
import os

import numpy as np
import pytest

from goldenio import FIXTURES, load_eval, load_inputs, load_json_gz, path
from tao_amodal_amd import flatten
from tao_amodal_amd.columns import DTColumns, GTColumns
from tao_amodal_amd.evaluation.lvis_amodal import LVIS, LVISEval, LVISResults
from tao_amodal_amd.evaluation.tao_amodal import Tao, TaoEval, TaoResults

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cli():
    spec = importlib.util.spec_from_file_location(
        "cli", os.path.join(ROOT, "tools", "eval_on_tao_amodal.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _results_list(res):
    return [[k if isinstance(k, str) else list(k), float(v)] for k, v in res.items()]


@pytest.mark.parametrize("name", FIXTURES)
def test_lvis_summarize_and_lines_from_golden_tensors(name):
    want = load_json_gz(name, "lvis.json.gz")
    ev = LVISEval(path(name, "gt.json"), path(name, "pred.json"), "bbox")
    assert ev.params.img_ids == want["img_ids"] and ev.params.cat_ids == want["cat_ids"]
    ev.flat = flatten.flatten_lvis(ev.lvis_gt.columns, ev.lvis_dt.columns_dt)
    ev.freq_groups = ev._prepare_freq_group()
    assert ev.freq_groups == want["freq_groups"]
    p, r = load_eval(name)["lvis"]
    ev.eval = {"precision": p, "recall": r}
    ev.summarize()
    assert _results_list(ev.results) == want["results"]
    assert ev.result_lines() == want["printed"]


@pytest.mark.parametrize("name", FIXTURES)
def test_tao_summarize_and_lines_from_golden_tensors(name):
    want = load_json_gz(name, "tao.json.gz")
    gtj, predj = load_inputs(name)
    dt = DTColumns.from_json(predj)
    dt.track_id, _ = flatten.make_track_ids_unique(dt)
    gt = Tao(gtj)
    ev = TaoEval(gt, TaoResults(gt, dt))
    assert ev.params.vid_ids == want["vid_ids"] and ev.params.cat_ids == want["cat_ids"]
    p, r = load_eval(name)["tao"]
    ev.eval = {"precision": p, "recall": r}
    ev.summarize()
    assert _results_list(ev.results) == want["results"]
    assert ev.result_lines() == want["printed"]


def test_small_table_matches_reference_log():
    cli = _cli()
    want = load_json_gz("f2", "lvis.json.gz")
    res = dict((k, v) for k, v in want["results"])
    table = cli.create_small_table({m: res[m] * 100 for m in cli.LVIS_METRICS})
    log = open(path("f2", "cli_log.txt")).read().splitlines()
    assert table.splitlines() == log[2:5]


def test_constructor_errors_mirror_the_reference():
    gt_path, pred_path = path("f1", "gt.json"), path("f1", "pred.json")
    with pytest.raises(ValueError):
        LVISEval(gt_path, pred_path, "keypoints")
    with pytest.raises(TypeError):
        LVISEval(123, pred_path, "bbox")
    with pytest.raises(TypeError):
        LVISEval(gt_path, 123, "bbox")
    with pytest.raises(ValueError):
        TaoEval(gt_path, pred_path, iou_type="keypoints")
    with pytest.raises(TypeError):
        TaoEval(123, pred_path)
    with pytest.raises(RuntimeError):
        LVISEval(gt_path, pred_path, "bbox").summarize()
    with pytest.raises(IndexError):
        LVISResults(LVIS(gt_path), [])
    # a track id used in two videos is rejected (reference results.py:111-119)
    _, predj = load_inputs("f5")
    with pytest.raises(AssertionError):
        TaoResults(Tao(path("f5", "gt.json")), predj)
    # prediction on an image the ground truth does not know
    _, predj = load_inputs("f1")
    predj[0]["image_id"] = 10 ** 9
    with pytest.raises(AssertionError):
        LVISResults(LVIS(gt_path), predj)
    # Tao(dict) requires the six top-level keys
    gtj, _ = load_inputs("f1")
    del gtj["info"]
    with pytest.raises(AssertionError):
        Tao(gtj)


def test_product_path_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ev = LVISEval(path("f1", "gt.json"), path("f1", "pred.json"), "bbox")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ev.run()


def test_range_tables_of_any_length_are_cut_into_the_kernels_slots():
    """The reference loops over whatever params.visibility_rng / area_rng /
    time_rng hold (L/eval.py:140-145, T/eval.py:271-276); the kernels evaluate
    5 visibility ranges + the out-of-frame one (the LAST of the caller's), 4
    areas + the occlusion one (the LAST area range) x 4 durations per pass:
    EvalConstants cuts the caller's tables into such blocks and says, per
    kernel slot, which of the caller's ranges it holds (goldens from the
    reference: tests/test_gpu_constants.py, cases ranges3 / ranges8)."""
    import numpy as np
    from tao_amodal_amd.evaluation._core import EvalConstants
    from tao_amodal_amd.evaluation.lvis_amodal.eval import Params as LP
    from tao_amodal_amd.evaluation.tao_amodal.eval import Params as TP
    P = LP("bbox")
    P.visibility_rng = P.visibility_rng[:2]              # one range + out-of-frame
    c = EvalConstants(P, LP("bbox"), "lvis")
    assert not c.default and not c.single and c.n_rng == 2
    (tab, slots), = c.rng_blocks
    assert tab["visibility_rng"].shape == (5, 2) and slots == [(0, 0), (5, 1)]
    assert (tab["visibility_rng"] == [0, 1.0]).all()
    P.visibility_rng = [[0, 0.1 * k] for k in range(1, 13)] + [[0, 1.0]]   # 12 + oof
    c = EvalConstants(P, LP("bbox"), "lvis")
    assert c.n_rng == 13 and len(c.rng_blocks) == 3
    seen = sorted(i for _, sl in c.rng_blocks for _, i in sl)
    assert seen == list(range(13))                        # every range once
    assert c.rng_blocks[0][1][-1] == (5, 12)              # the last one in the oof slot
    assert np.array_equal(c.rng_blocks[2][0]["visibility_rng"],
                          [[0, 0.1 * 11], [0, 0.1 * 12]] + [[0, 0.1 * 12]] * 3)   # padded
    P.visibility_rng = [[0, 1.0]]                         # the out-of-frame range alone
    c = EvalConstants(P, LP("bbox"), "lvis")
    assert c.n_rng == 1 and c.rng_blocks[0][1] == [(5, 0)]
    for bad in ([], [[0, 1.0, 2.0]], "all"):
        P.visibility_rng = bad
        with pytest.raises(NotImplementedError, match="params.visibility_rng"):
            EvalConstants(P, LP("bbox"), "lvis")
    # track level: areas x durations
    Q = TP("bbox")
    c = EvalConstants(Q, TP("bbox"), "tao")
    assert c.default and c.single and c.n_rng == 20
    assert c.rng_blocks[0][1] == [(k, k) for k in range(20)]
    Q.area_rng = Q.area_rng[:1]                           # the occlusion range alone
    Q.time_rng = [[0, 5]]
    c = EvalConstants(Q, TP("bbox"), "tao")
    assert c.n_rng == 1 and c.rng_blocks[0][1] == [(16, 0)] and not c.single
    Q.area_rng = [[0, 10.0 * k] for k in range(1, 7)] + [[0, 1e10]]   # 6 + occlusion
    Q.time_rng = [[0, k] for k in range(1, 6)]                        # 5 durations
    c = EvalConstants(Q, TP("bbox"), "tao")
    assert c.n_rng == 35 and len(c.rng_blocks) == 2 * 2
    seen = sorted(i for _, sl in c.rng_blocks for _, i in sl)
    assert seen == list(range(35))
    # the occlusion range (area index 6) sits in area slot 4 of the first area block
    occ = [(k, i) for _, sl in c.rng_blocks[:2] for k, i in sl if i // 5 == 6]
    assert sorted(i for _, i in occ) == [30, 31, 32, 33, 34] and all(k // 4 == 4 for k, _ in occ)
    assert all(np.array_equal(t["area_rng"][4], [0, 1e10]) for t, _ in c.rng_blocks)
    Q.time_rng = []
    with pytest.raises(NotImplementedError, match="params.time_rng"):
        EvalConstants(Q, TP("bbox"), "tao")


def test_edited_thresholds_are_cut_into_the_kernels_blocks():
    """_core.EvalConstants: the caller's thresholds, any number and order, as
    ascending blocks of the kernels' sizes (10 IoU / 101 recall thresholds), a
    short block padded with copies of its last value; idx = where a block's
    values go in the caller's arrays."""
    import numpy as np
    from tao_amodal_amd.evaluation._core import EvalConstants
    from tao_amodal_amd.evaluation.lvis_amodal.eval import Params
    P = Params("bbox")
    c = EvalConstants(P, Params("bbox"), "lvis")
    assert c.default and c.single
    P.iou_thrs = np.array([0.75, 0.3, 0.9, 0.5])
    P.rec_thrs = np.linspace(0, 1, 201)
    c = EvalConstants(P, Params("bbox"), "lvis")
    assert not c.default and not c.single and (c.T, c.R) == (4, 201)
    (idx, val), = c.thr_blocks
    assert idx.tolist() == [1, 3, 0, 2]
    assert val.tolist() == [0.3, 0.5, 0.75, 0.9] + [0.9] * 6
    assert [len(i) for i, _ in c.rec_blocks] == [101, 100]
    assert all(len(v) == 101 and np.all(np.diff(v) >= 0) for _, v in c.rec_blocks)
    assert np.array_equal(np.sort(np.concatenate([i for i, _ in c.rec_blocks])), np.arange(201))
    assert c.rec_sorted
    P.rec_thrs = np.array([0.9, 0.1])
    assert not EvalConstants(P, Params("bbox"), "lvis").rec_sorted
    # range values: five visibility ranges travel, the sixth has no bounds
    P = Params("bbox")
    P.visibility_rng = [[0, 1.0], [0, 0.3], [0.3, 0.6], [0.6, 1.0], [0.05, 0.9], [7, 7]]
    c = EvalConstants(P, Params("bbox"), "lvis")
    assert not c.default and c.single and c.ranges["visibility_rng"].shape == (5, 2)
    P.iou_thrs = np.array([])
    with pytest.raises(NotImplementedError):
        EvalConstants(P, Params("bbox"), "lvis")


def test_restrict_to_params_is_the_references_prepare_filter():
    """_core.restrict_to_params on columns == the reference's get_ann_ids
    filter (images of the subset, annotations on them, predictions on them);
    category subsets become a selection of result columns."""
    from tao_amodal_amd.columns import GTColumns
    from tao_amodal_amd.evaluation._core import restrict_to_params
    gtj, predj = load_inputs("f1")
    gt, dt = GTColumns.from_json(gtj), DTColumns.from_json(predj)
    imgs = sorted(i["id"] for i in gtj["images"])[3:17]
    cats = sorted(c["id"] for c in gtj["categories"])
    g2, d2, pos = restrict_to_params(gt, dt, "image", imgs, cats[::-1][:2], True)
    assert sorted(g2.img_id.tolist()) == imgs
    assert g2.ann_id.tolist() == [a["id"] for a in gtj["annotations"] if a["image_id"] in imgs]
    assert d2.image_id.tolist() == [p["image_id"] for p in predj if p["image_id"] in imgs]
    assert pos.tolist() == [len(cats) - 1, len(cats) - 2]
    vids = sorted(v["id"] for v in gtj["videos"])[1:3]
    g3, d3, pos = restrict_to_params(gt, dt, "video", vids, cats, True)
    in_v = {i["id"] for i in gtj["images"] if i["video_id"] in vids}
    assert pos is None and sorted(g3.vid_id.tolist()) == vids
    assert sorted(g3.img_id.tolist()) == sorted(in_v)
    assert g3.trk_id.tolist() == [t["id"] for t in gtj["tracks"] if t["video_id"] in vids]
    assert d3.image_id.tolist() == [p["image_id"] for p in predj if p["image_id"] in in_v]
    g4, d4, pos = restrict_to_params(gt, dt, "image", sorted(i["id"] for i in gtj["images"]),
                                     cats, True)
    assert g4 is gt and d4 is dt and pos is None
    with pytest.raises(KeyError):
        restrict_to_params(gt, dt, "video", [10 ** 9], cats, True)
    with pytest.raises(KeyError):
        restrict_to_params(gt, dt, "image", imgs, [10 ** 9], True)
    with pytest.raises(NotImplementedError):
        restrict_to_params(gt, dt, "image", imgs, cats[:1], False)


@pytest.mark.parametrize("name", ["f1", "f2", "f4"])
def test_list_inputs_are_rewritten_in_place_like_the_reference(name):
    """A list of prediction dicts is the caller's object and the reference
    rewrites it (L/results.py:39-65: segmentation, area, id of the boxes that
    survive the per-image cut; T/results.py:47-98: merged categories on every
    dict, then the same, then averaged scores where a track's kept boxes
    differ).  Goldens: the lists after the reference's constructors ran
    (tests/golden/make_golden_mutation.py; F2 holds > 300 boxes in an image,
    merged categories and tracks with non-uniform scores)."""
    import gzip
    import json
    want = json.load(gzip.open(path(name, "mutated.json.gz"), "rt"))
    _, predj = load_inputs(name)
    LVISResults(LVIS(path(name, "gt.json")), predj)
    assert json.loads(json.dumps(predj)) == want["lvis"]
    _, predj = load_inputs(name)
    # (what the CLI does first: tools/eval_on_tao_amodal.py:44-66 on the list)
    new_ids, _ = flatten.make_track_ids_unique(DTColumns.from_json(predj))
    for p, t in zip(predj, new_ids.tolist()):
        p["track_id"] = t
    TaoResults(Tao(path(name, "gt.json")), predj)
    assert json.loads(json.dumps(predj, default=float)) == want["tao"]


def test_tao_counts_the_annotations_with_negative_coordinates(caplog):
    """tao_amodal/tao.py:143-158: one warning with the number of annotations whose
    box has x < 0, y < 0, w <= 0 or h <= 0; none when there is no such box."""
    import copy
    import logging
    gtj, _ = load_inputs("f2")
    gtj = copy.deepcopy(gtj)
    for a in gtj["annotations"]:
        a["bbox"] = [0.0, 0.0, 2.5, 1.0]
    with caplog.at_level(logging.INFO, logger="tao.tao"):
        Tao(copy.deepcopy(gtj))
    assert not [r for r in caplog.records if "negative values" in r.getMessage()]
    bad = copy.deepcopy(gtj)
    boxes = ([-1.0, 2.0, 3.0, 4.0], [1.0, -0.5, 3.0, 4.0], [1.0, 2.0, 0.0, 4.0],
             [1.0, 2.0, 3.0, -4.0], [0.0, 0.0, 1.0, 1.0])
    for a, b in zip(bad["annotations"], boxes):
        a["bbox"] = list(b)
    caplog.clear()
    with caplog.at_level(logging.INFO, logger="tao.tao"):
        Tao(bad)
    said = [r.getMessage() for r in caplog.records]
    at = said.index("4 annotations had negative values in coordinates!")
    assert said[at - 1] == "Creating index." and said[at + 1] == "Index created."
