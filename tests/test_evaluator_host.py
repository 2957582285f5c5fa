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


@pytest.mark.parametrize("edit", [
    ("img_ids", lambda v: v[:1]), ("cat_ids", lambda v: v[:1]),
    ("iou_thrs", lambda v: v[:3]), ("rec_thrs", lambda v: v[::2]),
    ("visibility_rng", lambda v: v[:2])])
def test_lvis_params_edits_the_kernels_cannot_honour_raise(edit):
    """The reference lets a caller restrict / change params before evaluate();
    this path evaluates the whole ground truth at the compiled-in thresholds,
    so such edits must raise instead of silently giving full-set numbers."""
    ev = LVISEval(path("f1", "gt.json"), path("f1", "pred.json"), "bbox")
    name, fn = edit
    setattr(ev.params, name, fn(getattr(ev.params, name)))
    with pytest.raises(NotImplementedError, match="params." + name):
        ev.evaluate()


@pytest.mark.parametrize("edit", [
    ("vid_ids", lambda v: v[:1]), ("cat_ids", lambda v: v[1:]),
    ("iou_thrs", lambda v: v + 0.01), ("area_rng", lambda v: v[:1]),
    ("time_rng", lambda v: [[0, 5]])])
def test_tao_params_edits_the_kernels_cannot_honour_raise(edit):
    gtj, predj = load_inputs("f1")
    dt = DTColumns.from_json(predj)
    dt.track_id, _ = flatten.make_track_ids_unique(dt)
    gt = Tao(gtj)
    ev = TaoEval(gt, TaoResults(gt, dt))
    name, fn = edit
    setattr(ev.params, name, fn(getattr(ev.params, name)))
    with pytest.raises(NotImplementedError, match="params." + name):
        ev.evaluate()
