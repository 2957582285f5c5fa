"""Pin the Python oracle to the golden vectors produced by the reference."""
import numpy as np
import pytest

from goldenio import (ADVERSARIAL_FIXTURES, DECIMAL_SCALE_FIXTURES, FIXTURES,
                      INTEGER_FIXTURES, load_eval, load_inputs, load_json_gz)
from oracle import pyoracle


def _check_cells(got_cells, want_cells, exact_iou=True):
    want = {tuple(c["key"]): c for c in want_cells}
    assert set(got_cells) == set(want)
    for key, g in got_cells.items():
        w = want[key]
        wi = np.asarray(w["ious"], dtype=float)
        gi = np.asarray(g["ious"], dtype=float)
        if gi.size == 0 and wi.size == 0:
            gi = wi = np.zeros(0)
        assert gi.shape == wi.shape, key
        if exact_iou:
            assert np.array_equal(gi, wi), key
        else:
            assert np.allclose(gi, wi, rtol=0, atol=1e-12), key
        assert len(g["ranges"]) == len(w["ranges"])
        for a, (gr, wr) in enumerate(zip(g["ranges"], w["ranges"])):
            for f in ("dt_ids", "gt_ids", "dt_scores"):
                assert list(gr[f]) == list(wr[f]), (key, a, f)
            for f in ("dt_matches", "gt_matches", "dt_ignore", "gt_ignore"):
                assert np.array_equal(np.asarray(gr[f], dtype=float),
                                      np.asarray(wr[f], dtype=float)), (key, a, f)


def _check_pointers(got, want):
    want = {tuple(p["idx"]): p for p in want}
    assert set(got) == set(want)
    for k, g in got.items():
        w = want[k]
        assert list(g["dt_ids"]) == list(w["dt_ids"]), k
        assert np.array_equal(g["tps"].astype(int), np.asarray(w["tps"]).reshape(g["tps"].shape))
        assert np.array_equal(g["fps"].astype(int), np.asarray(w["fps"]).reshape(g["fps"].shape))


def _check_results(got, want):
    gk = [list(k) if isinstance(k, tuple) else k for k in got]
    assert gk == [k for k, _ in want]
    for (k, gv), (_, wv) in zip(got.items(), want):
        assert float(gv) == wv, k


@pytest.mark.parametrize("name", FIXTURES + ADVERSARIAL_FIXTURES)
def test_lvis_oracle_matches_reference(name):
    gt, pred = load_inputs(name)
    want = load_json_gz(name, "lvis.json.gz")
    got = pyoracle.lvis_eval(gt, pred)
    assert got["img_ids"] == want["img_ids"] and got["cat_ids"] == want["cat_ids"]
    _check_cells(got["cells"], want["cells"])
    p, r = load_eval(name)["lvis"]
    assert np.array_equal(got["precision"], p)
    assert np.array_equal(got["recall"], r)
    _check_pointers(got["pointers"], want["dt_pointers"])
    _check_results(got["results"], want["results"])
    assert got["printed"] == want["printed"]
    assert got["freq_groups"] == want["freq_groups"]


@pytest.mark.parametrize("name", FIXTURES + ADVERSARIAL_FIXTURES)
def test_tao_oracle_matches_reference(name):
    gt, pred = load_inputs(name)
    want = load_json_gz(name, "tao.json.gz")
    n = pyoracle.make_track_ids_unique(pred)
    assert n == want["n_track_ids_changed"]
    assert [p["track_id"] for p in pred] == want["unique_track_ids"]
    got = pyoracle.tao_eval(gt, pred, frame_order="set")
    assert got["vid_ids"] == want["vid_ids"] and got["cat_ids"] == want["cat_ids"]
    assert {str(k): v for k, v in got["track_scores"].items()} == want["track_scores"]
    _check_cells(got["cells"], want["cells"])
    p, r = load_eval(name)["tao"]
    assert np.array_equal(got["precision"], p)
    assert np.array_equal(got["recall"], r)
    _check_pointers(got["pointers"], want["dt_pointers"])
    _check_results(got["results"], want["results"])
    assert got["printed"] == want["printed"]


@pytest.mark.parametrize("name", DECIMAL_SCALE_FIXTURES)
def test_tao_oracle_matches_reference_on_long_decimal_tracks(name):
    """F8: tracks of hundreds of frames with decimal boxes -- the oracle's
    set-order sums equal the reference's bit for bit (the image level of this
    fixture is held by the flatten + C oracle test)."""
    gt, pred = load_inputs(name)
    want = load_json_gz(name, "tao.json.gz")
    pyoracle.make_track_ids_unique(pred)
    got = pyoracle.tao_eval(gt, pred, frame_order="set")
    _check_cells(got["cells"], want["cells"])
    p, r = load_eval(name)["tao"]
    assert np.array_equal(got["precision"], p)
    assert np.array_equal(got["recall"], r)
    _check_results(got["results"], want["results"])
    assert got["printed"] == want["printed"]


@pytest.mark.parametrize("name", INTEGER_FIXTURES)
def test_timeline_frame_order_is_exact_on_integer_boxes(name):
    gt, pred = load_inputs(name)
    want = load_json_gz(name, "tao.json.gz")
    pyoracle.make_track_ids_unique(pred)
    got = pyoracle.tao_eval(gt, pred, frame_order="timeline")
    _check_cells(got["cells"], want["cells"])
    p, r = load_eval(name)["tao"]
    assert np.array_equal(got["precision"], p)
    assert np.array_equal(got["recall"], r)


def test_timeline_frame_order_on_decimal_boxes_is_within_ulps():
    """F4 has arbitrary decimal coordinates: the canonical (timeline) frame
    order may differ from the reference's set order in the last bits of the
    3D IoU, but no match decision flips on this fixture."""
    gt, pred = load_inputs("f4")
    want = load_json_gz("f4", "tao.json.gz")
    pyoracle.make_track_ids_unique(pred)
    got = pyoracle.tao_eval(gt, pred, frame_order="timeline")
    _check_cells(got["cells"], want["cells"], exact_iou=False)
    p, r = load_eval("f4")["tao"]
    assert np.array_equal(got["precision"], p)


def test_known_answers_from_reference_doctests():
    """The non-crowd doctest values of reference T/eval.py:21-30."""
    assert pyoracle.bb_intersect_union([0, 0, 20, 20], [0, 0, 20, 20]) == (400, 400)
    assert pyoracle.bb_intersect_union([0, 0, 20, 20], [0, 0, 10, 10]) == (100, 400)
    assert pyoracle.bb_intersect_union([10, 20, 10, 10], [10, 20, 5, 5]) == (25, 100)
    assert pyoracle.bb_intersect_union([0, 0, 20, 20], [0, 0, 30, 30]) == (400, 900)
    assert pyoracle.bb_iou([0, 0, 20, 20], [0, 0, 10, 10]) == 0.25


from goldenio import MODE_FIXTURES, MODES, load_modes


@pytest.mark.parametrize("name", MODE_FIXTURES)
@pytest.mark.parametrize("mode", list(MODES))
def test_tao_oracle_other_modes_match_reference(name, mode):
    """avg_iou / imagenetvid / use_cats=0 (not on the CLI path)."""
    gt, pred = load_inputs(name)
    pyoracle.make_track_ids_unique(pred)
    got = pyoracle.tao_eval(gt, pred, frame_order="set", **MODES[mode])
    cells, p, r, res = load_modes(name)[mode]
    for key, cell in got["cells"].items():
        w = cells[key]
        g = np.asarray(cell["ious"], dtype=float)
        if g.size == 0 and w.size == 0:
            continue
        assert np.array_equal(g, w.reshape(g.shape)), (key, mode)
    assert np.array_equal(got["precision"], p)
    assert np.array_equal(got["recall"], r)
    assert [float(v) for v in got["results"].values()] == res.tolist()


from goldenio import load_lvis_nocats


@pytest.mark.parametrize("name", MODE_FIXTURES)
def test_lvis_oracle_without_categories_matches_reference(name):
    """params.use_cats = 0 of LVISEval: per-image
    IoUs, every range's matches and ignore flags, precision, recall."""
    gt, pred = load_inputs(name)
    got = pyoracle.lvis_eval(gt, pred, use_cats=False)
    cells, eval_imgs, p, r, err = load_lvis_nocats(name)
    assert err == "IndexError" and got["results"] is None
    # (the reference stores [] where an image has only GT or only detections)
    assert set(k[0] for k, c in got["cells"].items()
               if np.asarray(c["ious"]).size) == set(cells)
    for (im, c), cell in got["cells"].items():
        assert c == -1
        g = np.asarray(cell["ious"], dtype=float)
        if g.size:
            assert np.array_equal(g, cells[im].reshape(g.shape)), im
    # eval_imgs is range-major, image-minor (L/eval.py:138-143), None dropped;
    # the first and the out-of-frame range share the bounds [0, 1]
    A = len(pyoracle.VIS_RNG)
    per = len(eval_imgs) // A
    want = {(e["image_id"], i // per): e for i, e in enumerate(eval_imgs)}
    n = 0

    def same(mine, ref):
        mine = np.asarray(mine).astype(int)
        return np.array_equal(mine, np.asarray(ref).reshape(mine.shape))

    for (im, c), cell in got["cells"].items():
        for a, e in enumerate(cell["ranges"]):
            w = want[im, a]
            assert w["rng"] == [float(x) for x in pyoracle.VIS_RNG[a]]
            assert [int(x) for x in e["dt_ids"]] == w["dt_ids"]
            assert [int(x) for x in e["gt_ids"]] == w["gt_ids"]
            assert same(e["dt_matches"], w["dt_matches"]), (im, a)
            assert same(e["dt_ignore"], w["dt_ignore"]), (im, a)
            assert same(e["gt_ignore"], w["gt_ignore"]), (im, a)
            n += 1
    assert n == len(eval_imgs)
    assert np.array_equal(got["precision"], p)
    assert np.array_equal(got["recall"], r)


def test_timeline_frame_order_flips_matches_on_the_adversarial_fixture():
    """F7: every 3D IoU sits on a threshold in real arithmetic.  Adding the
    frames in timeline order instead of the reference's set order moves some of
    them across it -- the reason the HIP path guards these pairs
    (engine.apply_iou_guard, tests/test_gpu_guard.py)."""
    gt, pred = load_inputs("f7")
    want = {tuple(c["key"]): c for c in load_json_gz("f7", "tao.json.gz")["cells"]}
    pyoracle.make_track_ids_unique(pred)
    got = pyoracle.tao_eval(gt, pred, frame_order="timeline")
    flips = 0
    for key, g in got["cells"].items():
        for gr, wr in zip(g["ranges"], want[key]["ranges"]):
            flips += int((np.asarray(gr["dt_matches"], dtype=float)
                          != np.asarray(wr["dt_matches"], dtype=float)).sum())
    assert flips > 0
