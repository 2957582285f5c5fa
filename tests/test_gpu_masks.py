"""iou_type="segm" on the GPU: the run-length IoU kernel against the oracle
(oracle/rle.py, pinned to the reference C), and LVISEval(iou_type="segm")
through the class API against the golden run of the reference (fixture F6)."""
import gzip
import json

import numpy as np
import pytest

from goldenio import path
from oracle import rle
from tao_amodal_amd.masks import MaskBatch

pytestmark = pytest.mark.gpu


def _rand_poly(rng, h, w, spread=0.45):
    k = int(rng.integers(3, 10))
    cx, cy = rng.uniform(0, w), rng.uniform(0, h)
    return np.c_[cx + rng.uniform(-w * spread, w * spread, k),
                 cy + rng.uniform(-h * spread, h * spread, k)].ravel().tolist()


def _device_iou(cells_d, cells_g, dt, gt):
    """cells_* = per-cell counts; dt / gt = MaskArrays.  Returns the flat IoUs."""
    import torch
    from tao_amodal_amd import _lib
    lib = _lib.load()
    dev = "cuda"
    d_off = np.r_[0, np.cumsum(cells_d)].astype(np.int32)
    g_off = np.r_[0, np.cumsum(cells_g)].astype(np.int32)
    i_off = np.r_[0, np.cumsum(np.asarray(cells_d, np.int64) * cells_g)].astype(np.int64)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    head = [t(d_off), t(g_off), t(i_off)]
    sides = []
    for m in (dt, gt):
        sides.append([t(m.off), t(m.counts.view(np.int32) if len(m.counts) else np.zeros(1, np.int32)),
                      t(m.hw if len(m) else np.zeros((1, 2), np.int32)),
                      t(m.bbox if len(m) else np.zeros((1, 4)))])
    out = torch.full((max(int(i_off[-1]), 1),), -7.0, dtype=torch.float64, device=dev)
    nb = lib.taoamd_rle_iou_workspace(len(dt), int(dt.off[-1]), len(gt), int(gt.off[-1]))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = lib.taoamd_rle_iou(
        len(cells_d), *[x.data_ptr() for x in head],
        len(dt), int(dt.off[-1]), *[x.data_ptr() for x in sides[0]],
        len(gt), int(gt.off[-1]), *[x.data_ptr() for x in sides[1]],
        out.data_ptr(), ws.data_ptr(), nb, None)
    assert st == 0
    torch.cuda.synchronize()
    return out.cpu().numpy()[:int(i_off[-1])], i_off


def test_rle_iou_kernel_equals_the_oracle_on_random_cells():
    rng = np.random.default_rng(9)
    db, gb = MaskBatch(), MaskBatch()
    cells_d, cells_g, frames = [], [], []
    for c in range(60):
        h, w = int(rng.integers(6, 90)), int(rng.integers(6, 120))
        D, G = int(rng.integers(0, 12)), int(rng.integers(0, 7))
        for _ in range(D):
            db.add([_rand_poly(rng, h, w)], h, w)
        for k in range(G):
            kind = (c + k) % 9
            if kind == 0:
                gb.add({"size": [h, w], "counts": [h * w]}, h, w)         # empty
            elif kind == 1:
                gb.add({"size": [h, w], "counts": [0, h * w]}, h, w)      # full
            elif kind == 2:
                gb.add([_rand_poly(rng, h + 1, w)], h + 1, w)            # other frame
            else:
                gb.add([_rand_poly(rng, h, w), _rand_poly(rng, h, w, 0.2)], h, w)
        cells_d.append(D)
        cells_g.append(G)
    dt, gt = db.arrays(), gb.arrays()
    got, off = _device_iou(cells_d, cells_g, dt, gt)
    d0 = g0 = 0
    n_pos = n_neg = 0
    for c, (D, G) in enumerate(zip(cells_d, cells_g)):
        if D and G:
            want = rle.iou_matrix([dt.mask(d0 + i) for i in range(D)],
                                  [gt.mask(g0 + j) for j in range(G)])
            assert np.array_equal(got[off[c]:off[c + 1]].reshape(D, G), want), c
            n_pos += int((want > 0).sum())
            n_neg += int((want == -1).sum())
        d0 += D
        g0 += G
    assert n_pos > 100 and n_neg > 3


def test_rle_iou_kernel_long_run_lists():
    """A frame of 720 x 1280 with ragged masks: thousands of runs per mask."""
    rng = np.random.default_rng(10)
    h, w = 720, 1280
    db, gb = MaskBatch(), MaskBatch()
    for _ in range(6):
        db.add([_rand_poly(rng, h, w, 0.3) for _ in range(3)], h, w)
    for _ in range(4):
        gb.add([_rand_poly(rng, h, w, 0.3) for _ in range(3)], h, w)
    dt, gt = db.arrays(), gb.arrays()
    assert dt.off[-1] > 3000
    got, _ = _device_iou([6], [4], dt, gt)
    want = rle.iou_matrix([dt.mask(i) for i in range(6)], [gt.mask(j) for j in range(4)])
    assert np.array_equal(got.reshape(6, 4), want)


def test_rle_iou_kernel_cells_beyond_the_lds_table_and_empty_runs():
    """40 ground truths of ~300 runs each do not fit the 6144-run LDS table
    (the kernel then reads the prefix sums through the caches); run lists with
    empty runs inside (legal in an uncompressed RLE) count the same pixels."""
    rng = np.random.default_rng(11)
    h, w = 300, 400
    db, gb = MaskBatch(), MaskBatch()
    for _ in range(5):
        db.add([_rand_poly(rng, h, w, 0.4) for _ in range(2)], h, w)
    for _ in range(40):
        gb.add([_rand_poly(rng, h, w, 0.4) for _ in range(2)], h, w)
    # the same masks again with empty runs spliced in
    base = gb.arrays()
    for j in range(3):
        c = base.mask(j)["counts"]
        spliced = []
        for k, v in enumerate(c):
            spliced += [v, 0, 0] if k % 5 == 2 else [v]
        gb.add({"size": [h, w], "counts": spliced}, h, w)
    dt, gt = db.arrays(), gb.arrays()
    assert gt.off[-1] > 6144
    got, _ = _device_iou([5], [43], dt, gt)
    want = rle.iou_matrix([dt.mask(i) for i in range(5)], [gt.mask(j) for j in range(43)])
    assert np.array_equal(got.reshape(5, 43), want)
    assert np.array_equal(want[:, 40:], want[:, :3]) and (want > 0).sum() > 20


@pytest.mark.parametrize("which", ["pred", "pred_rle"])
@pytest.mark.parametrize("as_list", [False, True])
def test_lvis_eval_segm_matches_reference_golden(which, as_list):
    from tao_amodal_amd.evaluation.lvis_amodal import LVIS, LVISEval, LVISResults
    with gzip.open(path("f6", "lvis_segm.json.gz")) as f:
        w = json.load(f)[which]
    z = np.load(path("f6", "lvis_segm.npz"))
    gt = LVIS(path("f6", "gt.json"))
    pred = path("f6", which + ".json")
    if as_list:
        pred = json.load(open(pred))
    ev = LVISEval(gt, LVISResults(gt, pred), "segm")
    ev.run()
    assert np.array_equal(ev.eval["precision"], z[which + "_precision"])
    assert np.array_equal(ev.eval["recall"], z[which + "_recall"])
    assert [[k, float(v)] for k, v in ev.results.items()] == w["results"]
    for c in w["cells"]:
        got = ev.ious[tuple(c["key"])]
        assert np.array_equal(got, np.asarray(c["ious"], dtype=float).reshape(np.shape(got))), c["key"]
    evals = {tuple(e["key"]): e for e in w["evals"]}
    n_img, n_rng = len(ev.params.img_ids), 6
    seen = 0
    for i, e in enumerate(ev.eval_imgs):
        if e is None:
            continue
        a = (i // n_img) % n_rng
        r = evals[e["image_id"], e["category_id"], a]
        assert e["dt_ids"] == r["dt_ids"] and e["gt_ids"] == r["gt_ids"]
        assert np.array_equal(e["dt_matches"], np.asarray(r["dt_matches"]).reshape(e["dt_matches"].shape))
        assert np.array_equal(e["dt_ignore"].astype(int), np.asarray(r["dt_ignore"]).reshape(e["dt_ignore"].shape))
        assert np.array_equal(np.asarray(e["gt_ignore"]).astype(int), np.asarray(r["gt_ignore"]))
        seen += 1
    assert seen == len(evals)


def test_segm_shard_of_categories_equals_the_whole():
    """The category-partitioned multi-GPU path slices the masks with the
    cells (dist.shard_by_category)."""
    from tao_amodal_amd import dist, engine, flatten as fl
    from tao_amodal_amd.evaluation.lvis_amodal import LVIS, LVISEval, LVISResults
    gt = LVIS(path("f6", "gt.json"))
    dt = LVISResults(gt, path("f6", "pred.json"))
    ev = LVISEval(gt, dt, "segm")
    f = fl.flatten_lvis(gt.columns, dt.columns_dt)
    f.masks = ev._masks(f)
    whole = engine.evaluate_flat(f, detail=True)
    K = len(f.cat_ids)
    for k0, k1 in ((0, K // 2), (K // 2, K)):
        part = dist.shard_by_category(f, k0, k1)
        got = engine.evaluate_flat(part, detail=True)
        assert np.array_equal(got["precision"][:, :, k0:k1], whole["precision"][:, :, k0:k1])
        assert np.array_equal(got["recall"][:, k0:k1], whole["recall"][:, k0:k1])
