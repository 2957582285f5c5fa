"""-m gpu: the device-side cell-table build (flatten_dev / csrc/flatten.hip)
against flatten.py, the numpy statement of the same stage (itself pinned to
the reference's golden vectors in test_flat_oracle_golden.py): every table
equal, and the evaluation on the device-built tables equal to the oracle."""
import numpy as np
import pytest

import orclib
from goldenio import FIXTURES, load_inputs
from tao_amodal_amd import flatten as fl
from tao_amodal_amd.columns import DTColumns, GTColumns
from tao_amodal_amd.synth import synth

pytestmark = pytest.mark.gpu

LVIS_FIELDS = ("img_ids", "cat_ids", "cat_freq", "cell_unit", "cell_cat",
               "cell_dt_off", "cell_gt_off", "dt_box", "dt_row", "gt_row",
               "dt_score", "dt_flags", "dt_id", "dt_cat", "dt_cell", "gt_box",
               "gt_vis", "gt_flags", "gt_id", "gt_cat", "gt_cell")


def _same(a, b, fields):
    assert a.n_cells == b.n_cells and a.n_pairs == b.n_pairs
    for k in fields:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape, k
        assert np.array_equal(x.astype(y.dtype), y), k


def _lvis_both(gt, dt, max_dets=300):
    from tao_amodal_amd import engine, flatten_dev
    want = fl.flatten_lvis(gt, dt, max_dets)
    got = flatten_dev.flatten_lvis_device(gt, dt, "cuda:0", max_dets)
    _same(got, want, LVIS_FIELDS)
    # the evaluation consumes the device tables directly
    res = engine.evaluate_flat(got, "cuda:0")
    ref = orclib.run_flat(want, detail=False)
    assert np.array_equal(res["matched"], ref["matched"])
    assert np.array_equal(res["precision"], ref["precision"])
    assert np.array_equal(res["recall"], ref["recall"])


@pytest.mark.parametrize("name", FIXTURES)
def test_lvis_device_tables_equal_the_numpy_tables_on_the_fixtures(name):
    gtj, predj = load_inputs(name)
    _lvis_both(GTColumns.from_json(gtj), DTColumns.from_json(predj))


@pytest.mark.parametrize("kw,max_dets", [
    (dict(seed=1, V=6, F=30, C=40, dets_per_frame=25), 300),
    (dict(seed=2, V=3, F=8, C=1203, dets_per_frame=60), 300),
    (dict(seed=3, V=10, F=5, C=7, dets_per_frame=340, n_present=4), 300),   # top-300 cut
    (dict(seed=4, V=4, F=6, C=9, dets_per_frame=50, n_present=4, shuffle_image_ids=True), 20),
    (dict(seed=5, V=40, F=100, C=300, dets_per_frame=50), 300)])
def test_lvis_device_tables_equal_the_numpy_tables_on_synthetic_sets(kw, max_dets):
    gt, dt = synth(**kw)
    _lvis_both(gt, dt, max_dets)


def test_lvis_device_build_rejects_results_of_unknown_images():
    from tao_amodal_amd import flatten_dev
    gt, dt = synth(seed=6, V=2, F=4, C=8, dets_per_frame=5, n_present=4)
    dt.image_id = dt.image_id.copy()
    dt.image_id[3] = 10 ** 6
    with pytest.raises(AssertionError, match="do not correspond"):
        flatten_dev.flatten_lvis_device(gt, dt, "cuda:0")


TAO_FIELDS = ("vid_ids", "cat_ids", "cell_unit", "cell_cat", "cell_dt_off",
              "cell_gt_off", "cell_iou_off", "cell_span", "dt_score", "dt_area",
              "dt_len", "dt_flags", "dt_id", "dt_cat", "dt_cell", "dt_frame_off",
              "dt_frame_pos", "dt_frame_box", "gt_area", "gt_len", "gt_nhp",
              "gt_flags", "gt_id", "gt_cat", "gt_cell", "gt_frame_off",
              "gt_frame_pos", "gt_frame_box")


def _tao_both(gt, dt, max_dets=300):
    from tao_amodal_amd import engine, flatten_dev
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    want = fl.flatten_tao(gt, dt, max_dets)
    got = flatten_dev.flatten_tao_device(gt, dt, "cuda:0", max_dets)
    _same(got, want, TAO_FIELDS)
    assert got.required_average == want.required_average
    assert got.track_scores == want.track_scores
    res = engine.evaluate_flat(got, "cuda:0")
    ref = orclib.run_flat(want, detail=False)
    assert np.array_equal(res["iou"], ref["iou"])
    assert np.array_equal(res["matched"], ref["matched"])
    assert np.array_equal(res["precision"], ref["precision"])
    assert np.array_equal(res["recall"], ref["recall"])


@pytest.mark.parametrize("name", FIXTURES)
def test_tao_device_tables_equal_the_numpy_tables_on_the_fixtures(name):
    gtj, predj = load_inputs(name)
    _tao_both(GTColumns.from_json(gtj), DTColumns.from_json(predj))


@pytest.mark.parametrize("kw,max_dets", [
    (dict(seed=1, V=6, F=30, C=40, dets_per_frame=25), 300),
    (dict(seed=2, V=3, F=8, C=1203, dets_per_frame=60), 300),
    (dict(seed=3, V=10, F=5, C=7, dets_per_frame=340, n_present=4), 300),   # top-300 cut
    (dict(seed=4, V=4, F=6, C=9, dets_per_frame=50, n_present=4, shuffle_image_ids=True), 20),
    (dict(seed=7, V=5, F=12, C=20, dets_per_frame=30, n_present=4, collide_track_ids=True,
          n_merged=3), 300),
    (dict(seed=5, V=40, F=100, C=300, dets_per_frame=50), 300)])
def test_tao_device_tables_equal_the_numpy_tables_on_synthetic_sets(kw, max_dets):
    gt, dt = synth(**kw)
    _tao_both(gt, dt, max_dets)


def test_tao_device_tables_with_holes_and_wide_track_ids():
    from test_gpu_parity import _drop_frames
    gt, dt = synth(seed=8, V=4, F=40, C=12, dets_per_frame=30, n_present=4)
    gt, dt = _drop_frames(gt, dt, 8)
    dt.track_id = dt.track_id + (1 << 40)          # ids beyond 31 bits: two radix sorts
    _tao_both(gt, dt)


def test_tao_device_build_hands_rejected_inputs_to_the_numpy_path():
    from tao_amodal_amd import flatten_dev
    gt, dt = synth(seed=9, V=3, F=6, C=10, dets_per_frame=10, n_present=4)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    bad = dt.take(np.arange(len(dt)))
    bad.video_id = bad.video_id.copy()
    bad.video_id[0] = bad.video_id[0] % 3 + 1 if bad.video_id[0] != 2 else 3
    with pytest.raises(flatten_dev.Rejected):
        flatten_dev.flatten_tao_device(gt, bad, "cuda:0")
    with pytest.raises(AssertionError, match="more than one video"):
        flatten_dev.flatten_tao(gt, bad, device="cuda:0")
    bad = dt.take(np.arange(len(dt)))
    bad.category_id = bad.category_id.copy()
    t = bad.track_id[0]
    sel = np.flatnonzero(bad.track_id == t)
    if len(sel) > 1:
        bad.category_id[sel[-1]] = bad.category_id[sel[-1]] % 10 + 1
        with pytest.raises(AssertionError, match="multiple categories"):
            flatten_dev.flatten_tao(gt, bad, device="cuda:0")
