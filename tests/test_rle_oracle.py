"""oracle/rle.py + pyoracle.lvis_eval(iou_type="segm") (test infrastructure)
pinned to (i) the reference's own maskApi.c compiled where it lies
(oracle/_ref) and (ii) the golden vectors of fixture F6 -- the reference
LVISEval run with iou_type="segm" (tests/golden/make_golden_segm.py)."""
import gzip
import json
import os

import numpy as np
import pytest

import orclib
from goldenio import path
from oracle import pyoracle, rle

needs_ref = pytest.mark.skipif(not os.path.exists(orclib.REF_SO),
                               reason="oracle/_ref not built")


def _rand_poly(rng, h, w, k=None, spread=0.5):
    k = k or int(rng.integers(3, 9))
    cx, cy = rng.uniform(0, w), rng.uniform(0, h)
    return np.c_[cx + rng.uniform(-w * spread, w * spread, k),
                 cy + rng.uniform(-h * spread, h * spread, k)].ravel()


@needs_ref
def test_polygon_rasterisation_and_text_form_against_reference_c():
    rng = np.random.default_rng(1)
    for it in range(300):
        h, w = int(rng.integers(5, 60)), int(rng.integers(5, 80))
        xy = _rand_poly(rng, h, w, spread=0.7)
        if it % 3 == 0:
            xy = np.round(xy)
        if it % 7 == 0:
            xy[2:4] = xy[0:2]                   # repeated vertex
        a = rle.fr_poly(xy.tolist(), h, w)
        assert a == orclib.ref_rle_fr_poly(xy, h, w), it
        s = rle.to_string(a)
        assert s == orclib.ref_rle_to_string(a)
        assert rle.fr_string(s, h, w) == a == orclib.ref_rle_fr_string(s, h, w)
        assert rle.to_bbox(a) == orclib.ref_rle_to_bbox(a)
        assert rle.area(a) == orclib.ref_rle_area(a)


@needs_ref
def test_merge_and_iou_against_reference_c():
    rng = np.random.default_rng(2)
    for it in range(150):
        h, w = int(rng.integers(8, 50)), int(rng.integers(8, 60))
        mk = lambda hh=h: rle.fr_poly(_rand_poly(rng, hh, w).tolist(), hh, w)
        ms = [mk() for _ in range(int(rng.integers(1, 5)))]
        for inter in (False, True):
            assert rle.merge(ms, inter) == orclib.ref_rle_merge(ms, inter), it
        ds, gs = [mk() for _ in range(4)], [mk() for _ in range(3)]
        if it % 10 == 0:
            gs[0] = mk(h + 1)                               # another frame size
        if it % 9 == 0:
            ds[1] = {"h": h, "w": w, "counts": [h * w]}     # empty mask
        if it % 8 == 0:
            ds[2] = {"h": h, "w": w, "counts": [0, h * w]}  # full mask
        assert np.array_equal(rle.iou_matrix(ds, gs), orclib.ref_rle_iou(ds, gs)), it


def test_known_answers():
    full = {"h": 4, "w": 5, "counts": [0, 20]}
    half = {"h": 4, "w": 5, "counts": [0, 8, 12]}          # the two left columns
    assert rle.iou_pair(full, half) == 8 / 20
    assert rle.iou_pair(half, half) == 1.0
    right = {"h": 4, "w": 5, "counts": [12, 8]}
    assert rle.iou_pair(half, right) == 0.0
    assert rle.to_bbox(right) == [3.0, 0.0, 2.0, 4.0]
    assert rle.merge([half, right])["counts"] == [0, 8, 4, 8]
    assert rle.merge([half, full], intersect=True)["counts"] == [0, 8, 12]
    # a 3 x 2 box drawn as a polygon on a 6 x 8 frame
    box = rle.fr_bbox([2, 1, 3, 2], 6, 8)
    assert rle.area(box) == 6 and rle.to_bbox(box) == [2.0, 1.0, 3.0, 2.0]


def load_segm():
    with gzip.open(path("f6", "lvis_segm.json.gz")) as f:
        g = json.load(f)
    z = np.load(path("f6", "lvis_segm.npz"))
    return g, z


@pytest.mark.parametrize("which", ["pred", "pred_rle"])
def test_lvis_oracle_segm_matches_reference(which):
    golden, z = load_segm()
    w = golden[which]
    gt = json.load(open(path("f6", "gt.json")))
    pred = json.load(open(path("f6", which + ".json")))
    got = pyoracle.lvis_eval(gt, pred, iou_type="segm")
    # every mask the evaluator compared, as compressed text
    assert {str(k): v for k, v in got["gt_rle"].items()} == w["gt_rle"]
    assert {str(k): v for k, v in got["dt_rle"].items()} == w["dt_rle"]
    assert {str(k): v for k, v in got["dt_area"].items()} == w["dt_area"]
    assert {str(k): v for k, v in got["dt_bbox"].items()} == w["dt_bbox"]
    want_cells = {tuple(c["key"]): np.asarray(c["ious"], dtype=float) for c in w["cells"]}
    have = {k: c for k, c in got["cells"].items() if np.asarray(c["ious"]).size}
    assert set(have) == set(want_cells)
    for k, c in have.items():
        assert np.array_equal(np.asarray(c["ious"]), want_cells[k].reshape(np.asarray(c["ious"]).shape)), k
    evals = {tuple(e["key"]): e for e in w["evals"]}
    n = 0
    for (im, c), cell in got["cells"].items():
        for a, e in enumerate(cell["ranges"]):
            r = evals[im, c, a]
            assert [int(x) for x in e["dt_ids"]] == r["dt_ids"]
            assert [int(x) for x in e["gt_ids"]] == r["gt_ids"]
            for key in ("dt_matches", "dt_ignore", "gt_ignore"):
                mine = np.asarray(e[key]).astype(int)
                assert np.array_equal(mine, np.asarray(r[key]).reshape(mine.shape)), (im, c, a, key)
            n += 1
    assert n == len(evals)
    assert np.array_equal(got["precision"], z[which + "_precision"])
    assert np.array_equal(got["recall"], z[which + "_recall"])
    assert [[k, float(v)] for k, v in got["results"].items()] == w["results"]
