"""-m gpu: the drop-in CLI and the class API end to end against the text and
numbers the reference printed for the same files."""
import importlib.util
import io
import os
import contextlib

import numpy as np
import pytest

from goldenio import FIXTURES, INTEGER_FIXTURES, load_eval, load_json_gz, path
from test_oracle_golden import _check_cells

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cli():
    spec = importlib.util.spec_from_file_location(
        "cli", os.path.join(ROOT, "tools", "eval_on_tao_amodal.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", FIXTURES)
def test_cli_text_is_identical_to_the_reference(name, tmp_path):
    log = tmp_path / "out" / "eval.log"
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        _cli().main(["--track_result", path(name, "pred.json"), "--annotation",
                     path(name, "gt.json"), "--output_log", str(log)])
    assert buf.getvalue() == open(path(name, "cli_stdout.txt")).read()
    want = open(path(name, "cli_log.txt")).read()
    got = log.read_text().replace(os.path.join(path(name, "")), "<DIR>/")
    assert got == want


@pytest.mark.parametrize("name", ["f1", "f2"])
def test_class_api_state_matches_reference(name):
    from tao_amodal_amd.evaluation.lvis_amodal import LVISEval
    want = load_json_gz(name, "lvis.json.gz")
    ev = LVISEval(path(name, "gt.json"), path(name, "pred.json"), "bbox")
    ev.run()
    p, r = load_eval(name)["lvis"]
    assert np.array_equal(ev.eval["precision"], p)
    assert np.array_equal(ev.eval["recall"], r)
    assert ev.eval["counts"] == list(p.shape)
    # eval_imgs / ious views (reference order: category, range, image)
    n_img, n_rng = len(ev.params.img_ids), 6
    cells = {}
    for c, cat in enumerate(ev.params.cat_ids):
        for i, img in enumerate(ev.params.img_ids):
            es = [ev.eval_imgs[(c * n_rng + a) * n_img + i] for a in range(n_rng)]
            if es[0] is None:
                assert all(e is None for e in es)
                assert len(ev.ious[img, cat]) == 0
                continue
            cells[img, cat] = {"ious": ev.ious[img, cat], "ranges": es}
    _check_cells(cells, want["cells"])
    ptr = {tuple(p_["idx"]): p_ for p_ in want["dt_pointers"]}
    for (k, a), w in ptr.items():
        g = ev.eval["dt_pointers"][k][a]
        assert list(g["dt_ids"]) == w["dt_ids"]
        assert np.array_equal(g["tps"].astype(int), np.asarray(w["tps"]).reshape(g["tps"].shape))
        assert np.array_equal(g["fps"].astype(int), np.asarray(w["fps"]).reshape(g["fps"].shape))
    assert ev.eval["dt_pointers"][0][0] == {} or (0, 0) in ptr


@pytest.mark.parametrize("name", ["f1", "f2"])
def test_tao_class_api_state_matches_reference(name):
    import json
    from tao_amodal_amd import flatten
    from tao_amodal_amd.columns import DTColumns
    from tao_amodal_amd.evaluation.tao_amodal import Tao, TaoEval, TaoResults
    want = load_json_gz(name, "tao.json.gz")
    dt = DTColumns.from_json(path(name, "pred.json"))
    dt.track_id, _ = flatten.make_track_ids_unique(dt)
    gt = Tao(path(name, "gt.json"))
    ev = TaoEval(gt, TaoResults(gt, dt))
    ev.run()
    p, r = load_eval(name)["tao"]
    assert np.array_equal(ev.eval["precision"], p)
    assert np.array_equal(ev.eval["recall"], r)
    P = ev.params
    cells = {}
    for v, vid in enumerate(P.vid_ids):
        for c, cat in enumerate(P.cat_ids):
            es = [ev.eval_vids[v, c, a, t] for a in range(5) for t in range(4)]
            if es[0] is None:
                continue
            cells[vid, cat] = {"ious": ev.ious[vid, cat], "ranges": es}
    _check_cells(cells, want["cells"], exact_iou=name in INTEGER_FIXTURES)
    for p_ in want["dt_pointers"]:
        k, a, t = p_["idx"]
        g = ev.eval["dt_pointers"][k][a][t]
        assert list(g["dt_ids"]) == p_["dt_ids"]
        assert np.array_equal(g["tps"].astype(int), np.asarray(p_["tps"]).reshape(g["tps"].shape))


def _subset_golden(name):
    import json
    z = np.load(path(name, "params_subset.npz"))
    gt = json.load(open(path(name, "gt.json")))
    return z, gt


@pytest.mark.parametrize("name", ["f1", "f5"])
def test_edited_params_subsets_match_the_reference(name):
    """params.img_ids / vid_ids / cat_ids edited before run() (reference
    lvis_amodal/eval.py:59-105, tao_amodal/eval.py:178-233): golden vectors
    from the reference run with the same subsets (tests/golden/
    make_golden_params.py) -- a subset of the images / videos, every other
    category in descending order."""
    from tao_amodal_amd import flatten
    from tao_amodal_amd.columns import DTColumns
    from tao_amodal_amd.evaluation.lvis_amodal import LVISEval
    from tao_amodal_amd.evaluation.tao_amodal import Tao, TaoEval, TaoResults
    z, _ = _subset_golden(name)
    ev = LVISEval(path(name, "gt.json"), path(name, "pred.json"), "bbox")
    ev.params.img_ids = z["img_ids"].tolist()
    ev.params.cat_ids = z["cat_ids"].tolist()
    ev.run()
    assert np.array_equal(ev.eval["precision"], z["lvis_precision"])
    assert np.array_equal(ev.eval["recall"], z["lvis_recall"])
    assert [float(v) for v in ev.results.values()] == z["lvis_results"].tolist()
    assert [len(g) for g in ev.freq_groups] == z["lvis_freq_groups"].tolist()
    # the views follow params.cat_ids' order
    c0 = int(z["cat_ids"][0])
    assert all(e is None or e["category_id"] == c0
               for e in ev.eval_imgs[: len(ev.params.img_ids)])
    dt = DTColumns.from_json(path(name, "pred.json"))
    dt.track_id, _ = flatten.make_track_ids_unique(dt)
    gt = Tao(path(name, "gt.json"))
    te = TaoEval(gt, TaoResults(gt, dt))
    te.params.vid_ids = z["vid_ids"].tolist()
    te.params.cat_ids = z["cat_ids"].tolist()
    te.run()
    assert np.array_equal(te.eval["precision"], z["tao_precision"])
    assert np.array_equal(te.eval["recall"], z["tao_recall"])
    assert [float(v) for v in te.results.values()] == z["tao_results"].tolist()


def test_edited_params_the_path_cannot_honour_still_raise():
    from tao_amodal_amd.evaluation.lvis_amodal import LVISEval
    ev = LVISEval(path("f1", "gt.json"), path("f1", "pred.json"), "bbox")
    # (thresholds and range tables of any values, order and length are
    # honoured: tests/test_gpu_constants.py; a table that is no list of
    # [lo, hi] pairs is not)
    ev.params.visibility_rng = [[0, 1.0, 2.0], [0, 0.5, 1.0]]
    with pytest.raises(NotImplementedError):
        ev.evaluate()
    # with another number of ranges the per-cell views are not kept
    ev = LVISEval(path("f1", "gt.json"), path("f1", "pred.json"), "bbox")
    ev.params.visibility_rng = [[0, 1.0], [0, 0.5], [0, 1.0]]
    ev.evaluate()
    ev.accumulate()
    assert ev.eval["precision"].shape[-1] == 3
    with pytest.raises(NotImplementedError):
        ev.eval_imgs[0]
    ev = LVISEval(path("f1", "gt.json"), path("f1", "pred.json"), "bbox")
    ev.params.img_ids = [10 ** 9]
    with pytest.raises(KeyError):
        ev.evaluate()
    ev = LVISEval(path("f1", "gt.json"), path("f1", "pred.json"), "bbox")
    ev.params.max_dets = 100            # a label only (eval.py:464-545)
    ev.run()
    assert "AR@100" in ev.results


def test_cli_image_level_failure_shows_nothing_of_the_track_level(tmp_path):
    """The track level runs beside the image level (worker thread); when the
    image level fails -- here: a prediction on an image the annotation file
    does not have, reference lvis_amodal/results.py:62-65 -- its exception
    leaves the CLI and nothing the track level logged reaches the log file,
    as in the reference, which never got that far."""
    import json
    preds = json.load(open(path("f1", "pred.json")))
    preds[3]["image_id"] = 10 ** 9
    bad = tmp_path / "pred.json"
    bad.write_text(json.dumps(preds))
    log = tmp_path / "out" / "eval.log"
    with pytest.raises(AssertionError, match="Results do not correspond"):
        with contextlib.redirect_stdout(io.StringIO()):
            _cli().main(["--track_result", str(bad), "--annotation",
                         path("f1", "gt.json"), "--output_log", str(log)])
    text = log.read_text()
    assert "Evaluating" in text
    assert "Loading gt" not in text and "Running per video evaluation" not in text


def test_cli_serial_switch_gives_the_same_text(tmp_path, monkeypatch):
    monkeypatch.setenv("TAOAMD_CLI_SERIAL", "1")
    log = tmp_path / "out" / "eval.log"
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        _cli().main(["--track_result", path("f5", "pred.json"), "--annotation",
                     path("f5", "gt.json"), "--output_log", str(log)])
    assert buf.getvalue() == open(path("f5", "cli_stdout.txt")).read()
    got = log.read_text().replace(os.path.join(path("f5", "")), "<DIR>/")
    assert got == open(path("f5", "cli_log.txt")).read()


@pytest.mark.parametrize("name", FIXTURES)
def test_cli_text_with_the_prediction_file_read_on_the_device(name, tmp_path, monkeypatch):
    """The reference's text again with the device-side reader taking files of
    any size (csrc/json_ingest.hip; by default from 32 MB on): columns that
    stay on the device through both levels' table builds, or the host reader
    where the device reader steps aside."""
    monkeypatch.setenv("TAOAMD_DEVICE_INGEST_MIN_BYTES", "0")
    log = tmp_path / "out" / "eval.log"
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        _cli().main(["--track_result", path(name, "pred.json"), "--annotation",
                     path(name, "gt.json"), "--output_log", str(log)])
    assert buf.getvalue() == open(path(name, "cli_stdout.txt")).read()
    want = open(path(name, "cli_log.txt")).read()
    got = log.read_text().replace(os.path.join(path(name, "")), "<DIR>/")
    assert got == want


def test_cli_as_a_fresh_process_reads_on_the_device(tmp_path):
    """A fresh process: the reader waits for the early HIP start, loads the
    kernel library without torch and returns host arrays -- same text."""
    import subprocess
    import sys
    name = FIXTURES[0]
    env = dict(os.environ, TAOAMD_DEVICE_INGEST_MIN_BYTES="0", TAOAMD_INGEST_TIMING="1")
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, "tools", "eval_on_tao_amodal.py"),
         "--track_result", path(name, "pred.json"), "--annotation", path(name, "gt.json"),
         "--output_log", str(tmp_path / "eval.log")], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == open(path(name, "cli_stdout.txt")).read()
    assert "taoamd ingest (device)" in r.stderr


def test_cli_leaves_nothing_for_the_cycle_collector(tmp_path, monkeypatch):
    """A call's tables, columns and evaluators are released by reference
    counting on the CLI's helper thread (no reference cycle ties them to the
    interpreter's collector: round 6 found 5 GB of device tensors a call
    waiting for it): with the collector switched off, the device memory in use
    is back where it was once the helper is done."""
    import gc
    import threading
    import torch
    monkeypatch.setenv("TAOAMD_DEVICE_INGEST_MIN_BYTES", "1")
    cli = _cli()

    def run():
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(["--track_result", path("f5", "pred.json"), "--annotation",
                      path("f5", "gt.json"), "--output_log", str(tmp_path / "eval.log")])
        for th in threading.enumerate():
            if th is not threading.current_thread() and \
                    not th.name.startswith("ThreadPoolExecutor"):
                th.join(timeout=10.0)
    run()                       # (caches of the process: libraries, allocator)
    gc.collect()
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    gc.disable()
    try:
        run()
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() <= before
    finally:
        gc.enable()
