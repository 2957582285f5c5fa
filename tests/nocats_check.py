"""Shared check of a class-agnostic image-level run (flatten_lvis(use_cats=
False) evaluated by the C oracle or by the HIP path) against the golden
vectors of the reference LVISEval with params.use_cats = 0."""
import numpy as np

import orclib
from goldenio import load_lvis_nocats

N_THR = 10


def _bits(words, combo):
    return ((words[:, combo // 64] >> np.uint64(combo % 64)) & np.uint64(1)).astype(int)


def check(f, out, name):
    cells, eval_imgs, p, r, _ = load_lvis_nocats(name)
    assert f.cat_ids.tolist() == [-1] and not f.use_cats
    assert (f.cell_cat == 0).all() and (f.dt_cat == 0).all() and (f.gt_cat == 0).all()
    A = 6
    per = len(eval_imgs) // A
    assert per == f.n_cells
    want = {(e["image_id"], i // per): e for i, e in enumerate(eval_imgs)}
    off = orclib.iou_offsets(f)
    seen = set()
    for k in range(f.n_cells):
        im = int(f.img_ids[f.cell_unit[k]])
        d0, d1 = f.cell_dt_off[k], f.cell_dt_off[k + 1]
        g0, g1 = f.cell_gt_off[k], f.cell_gt_off[k + 1]
        D, G = d1 - d0, g1 - g0
        if D and G:
            seen.add(im)
            got = out["iou"][off[k]:off[k + 1]].reshape(D, G)
            assert np.array_equal(got, cells[im]), im
        gid = f.gt_id[g0:g1]
        for a in range(A):
            w = want[im, a]
            assert f.dt_id[d0:d1].tolist() == w["dt_ids"], (im, a)
            ig = (out["gt_rng"][g0:g1] >> np.uint32(a)) & np.uint32(1)
            perm = np.argsort(ig, kind="stable")      # ignore-last, stable
            assert gid[perm].tolist() == w["gt_ids"], (im, a)
            assert ig[perm].tolist() == w["gt_ignore"], (im, a)
            for t in range(N_THR):
                combo = a * N_THR + t
                m = out["match_gt"][d0:d1, combo]
                dt_m = np.where(m >= 0, gid[np.maximum(m, 0)] if G else 0, 0)
                wm = np.asarray(w["dt_matches"]).reshape(N_THR, -1)[t]
                wi = np.asarray(w["dt_ignore"]).reshape(N_THR, -1)[t]
                assert dt_m.tolist() == wm.tolist(), (im, a, t)
                assert _bits(out["ignored"][d0:d1], combo).tolist() == wi.tolist(), (im, a, t)
    assert seen == set(cells)
    assert np.array_equal(out["precision"].reshape(p.shape), p)
    assert np.array_equal(out["recall"].reshape(r.shape), r)
