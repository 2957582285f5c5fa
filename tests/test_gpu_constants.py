"""-m gpu: ``params.iou_thrs`` / ``rec_thrs`` / range values edited before
run(): the reference reads them when it runs (lvis_amodal/
eval.py:143,205,234,319-322,407; tao_amodal/eval.py:272-275,385,473-477,562),
so they are public state of the class API.  Goldens: the reference run with
the same edits (tests/golden/make_golden_constants.py)."""
import os
import sys

import numpy as np
import pytest

from goldenio import GOLDEN, path

sys.path.insert(0, GOLDEN)
from constants_cases import cases, edit  # noqa: E402

pytestmark = pytest.mark.gpu


def _golden(name):
    return np.load(path(name, "constants.npz"))


@pytest.mark.parametrize("case", list(cases()))
@pytest.mark.parametrize("name", ["f1", "f4"])
def test_edited_constants_match_the_reference(name, case):
    from tao_amodal_amd import flatten
    from tao_amodal_amd.columns import DTColumns
    from tao_amodal_amd.evaluation.lvis_amodal import LVISEval
    from tao_amodal_amd.evaluation.tao_amodal import Tao, TaoEval, TaoResults
    z = _golden(name)

    def run_all(e, side):
        """run(); a summarize() that raises in the reference (fewer ranges than
        its labels name: IndexError) must raise the same here, after the same
        partial results."""
        e.evaluate()
        e.accumulate()
        err = str(z[case + "_%s_summarize_error" % side]) \
            if case + "_%s_summarize_error" % side in z else ""
        if err:
            with pytest.raises(Exception) as info:
                e.summarize()
            assert type(info.value).__name__ == err
        else:
            e.summarize()
    ev = LVISEval(path(name, "gt.json"), path(name, "pred.json"), "bbox")
    edit(ev.params, cases()[case], "lvis")
    run_all(ev, "lvis")
    assert ev.eval["precision"].shape == z[case + "_lvis_precision"].shape
    assert np.array_equal(ev.eval["precision"], z[case + "_lvis_precision"])
    assert np.array_equal(ev.eval["recall"], z[case + "_lvis_recall"])
    assert [float(v) for v in ev.results.values()] == z[case + "_lvis_results"].tolist()
    assert ev.eval["counts"][:2] == [len(ev.params.iou_thrs), len(ev.params.rec_thrs)]
    dt = DTColumns.from_json(path(name, "pred.json"))
    dt.track_id, _ = flatten.make_track_ids_unique(dt)
    gt = Tao(path(name, "gt.json"))
    te = TaoEval(gt, TaoResults(gt, dt))
    edit(te.params, cases()[case], "tao")
    run_all(te, "tao")
    want_p, want_r = z[case + "_tao_precision"], z[case + "_tao_recall"]
    assert te.eval["precision"].shape == want_p.shape
    assert np.array_equal(te.eval["precision"], want_p)
    assert np.array_equal(te.eval["recall"], want_r)
    assert [float(v) for v in te.results.values()] == z[case + "_tao_results"].tolist()


def test_views_follow_the_edited_thresholds():
    """dt_matches / dt_ignore of the per-cell views and the tps / fps of
    eval['dt_pointers'] have one row per threshold of the caller, in the
    caller's order: equal to the default run's rows at the same values."""
    from tao_amodal_amd.evaluation.lvis_amodal import LVISEval
    base = LVISEval(path("f1", "gt.json"), path("f1", "pred.json"), "bbox")
    base.run()
    ev = LVISEval(path("f1", "gt.json"), path("f1", "pred.json"), "bbox")
    pick = [5, 0, 9]                                   # 0.75, 0.5, 0.95
    ev.params.iou_thrs = base.params.iou_thrs[pick]
    ev.run()
    assert np.array_equal(ev.eval["precision"], base.eval["precision"][pick])
    n = 0
    for i in range(len(ev.eval_imgs)):
        a, b = ev.eval_imgs[i], base.eval_imgs[i]
        assert (a is None) == (b is None)
        if a is None:
            continue
        n += 1
        assert np.array_equal(a["dt_matches"], b["dt_matches"][pick])
        assert np.array_equal(a["gt_matches"], b["gt_matches"][pick])
        assert np.array_equal(a["dt_ignore"], b["dt_ignore"][pick])
    assert n > 0
    for k in range(len(ev.params.cat_ids)):
        for r in range(6):
            a, b = ev.eval["dt_pointers"][k][r], base.eval["dt_pointers"][k][r]
            assert (a == {}) == (b == {})
            if a:
                assert np.array_equal(a["tps"], b["tps"][pick])
                assert np.array_equal(a["fps"], b["fps"][pick])


def test_the_constants_do_not_leak_into_the_next_run():
    from tao_amodal_amd.evaluation.lvis_amodal import LVISEval
    ev = LVISEval(path("f1", "gt.json"), path("f1", "pred.json"), "bbox")
    ev.params.iou_thrs = np.array([0.2, 0.4])
    ev.run()
    base = LVISEval(path("f1", "gt.json"), path("f1", "pred.json"), "bbox")
    base.run()
    z = np.load(path("f1", "eval.npz"))
    k = z["lvis_valid_k"]
    assert np.array_equal(base.eval["precision"][:, :, k], z["lvis_precision"])
