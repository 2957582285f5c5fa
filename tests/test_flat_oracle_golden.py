"""Host flatten (product) + C oracle (test infrastructure), together, against
the golden vectors of the reference: cell contents and order, IoUs, every
match/ignore decision, the TP/FP sequences and precision/recall."""
import numpy as np
import pytest

import orclib
from goldenio import FIXTURES, INTEGER_FIXTURES, load_eval, load_inputs, load_json_gz
from tao_amodal_amd.columns import DTColumns, GTColumns
from tao_amodal_amd import flatten as fl

N_THR = 10


def _bits(words, combo):
    return ((words[:, combo // 64] >> np.uint64(combo % 64)) & np.uint64(1)).astype(int)


def _check_side(f, out, want, unit_ids, sentinel, exact_iou=True):
    n_rng = 6 if f.kind == "lvis" else 20
    cells = {(int(unit_ids[u]), int(f.cat_ids[c])): k
             for k, (u, c) in enumerate(zip(f.cell_unit, f.cell_cat))}
    want_cells = {tuple(c["key"]): c for c in want["cells"]}
    assert set(cells) == set(want_cells)
    off = orclib.iou_offsets(f)
    for key, w in want_cells.items():
        k = cells[key]
        d0, d1 = f.cell_dt_off[k], f.cell_dt_off[k + 1]
        g0, g1 = f.cell_gt_off[k], f.cell_gt_off[k + 1]
        D, G = d1 - d0, g1 - g0
        r0 = w["ranges"][0]
        assert f.dt_id[d0:d1].tolist() == r0["dt_ids"], key
        assert f.dt_score[d0:d1].tolist() == r0["dt_scores"], key
        assert sorted(f.gt_id[g0:g1].tolist()) == sorted(r0["gt_ids"]), key
        wi = np.asarray(w["ious"], dtype=float)
        if D and G:
            gi = out["iou"][off[k]:off[k + 1]].reshape(D, G)
            if exact_iou:
                assert np.array_equal(gi, wi), key
            else:
                assert np.allclose(gi, wi, rtol=0, atol=1e-12), key
        gid = f.gt_id[g0:g1]
        for r in range(n_rng):
            wr = w["ranges"][r]
            got_ig = (out["gt_rng"][g0:g1] >> np.uint32(r)) & np.uint32(1)
            # the reference sorts GT ignore-last, stably
            perm = np.argsort(got_ig, kind="stable")
            assert gid[perm].tolist() == wr["gt_ids"], (key, r)
            assert got_ig[perm].tolist() == wr["gt_ignore"], (key, r)
            for t in range(N_THR):
                combo = r * N_THR + t
                m = out["match_gt"][d0:d1, combo]
                dt_m = np.where(m >= 0, gid[np.maximum(m, 0)] if G else sentinel, sentinel)
                assert dt_m.tolist() == [int(x) for x in wr["dt_matches"][t]], (key, r, t)
                assert _bits(out["ignored"][d0:d1], combo).tolist() == \
                    [int(x) for x in wr["dt_ignore"][t]], (key, r, t)
                assert _bits(out["matched"][d0:d1], combo).tolist() == \
                    [int(x != sentinel) for x in wr["dt_matches"][t]], (key, r, t)
                gt_m = np.full(G, sentinel, dtype=np.int64)
                for d in range(D):
                    if m[d] >= 0:
                        gt_m[m[d]] = f.dt_id[d0 + d]
                assert gt_m[perm].tolist() == \
                    [int(x) for x in wr["gt_matches"][t]], (key, r, t)
    # accumulate: order and TP/FP sequences per (category, range)
    for p in want["dt_pointers"]:
        k = p["idx"][0]
        r = p["idx"][1] if n_rng == 6 else p["idx"][1] * 4 + p["idx"][2]
        sel = out["order"][f.dt_cat[out["order"]] == k]
        assert f.dt_id[sel].tolist() == p["dt_ids"], p["idx"]
        for t in range(N_THR):
            mt = _bits(out["matched"][sel], r * N_THR + t)
            ig = _bits(out["ignored"][sel], r * N_THR + t)
            assert (mt & (1 - ig)).tolist() == p["tps"][t] if sel.size else True
            assert ((1 - mt) & (1 - ig)).tolist() == p["fps"][t] if sel.size else True
    # every (k, r) with evaluated GT appears in dt_pointers
    have = {(p["idx"][0], p["idx"][1] if n_rng == 6 else p["idx"][1] * 4 + p["idx"][2])
            for p in want["dt_pointers"]}
    assert have == {(int(k), int(r)) for k, r in zip(*np.nonzero(out["num_gt"]))}


@pytest.mark.parametrize("name", FIXTURES)
def test_lvis_flatten_and_c_oracle(name):
    gtj, predj = load_inputs(name)
    want = load_json_gz(name, "lvis.json.gz")
    f = fl.flatten_lvis(GTColumns.from_json(gtj), DTColumns.from_json(predj))
    assert f.img_ids.tolist() == want["img_ids"] and f.cat_ids.tolist() == want["cat_ids"]
    out = orclib.run_flat(f)
    _check_side(f, out, want, f.img_ids, 0)
    p, r = load_eval(name)["lvis"]
    assert np.array_equal(out["precision"], p)
    assert np.array_equal(out["recall"], r)


@pytest.mark.parametrize("name", FIXTURES)
def test_tao_flatten_and_c_oracle(name):
    gtj, predj = load_inputs(name)
    want = load_json_gz(name, "tao.json.gz")
    dt = DTColumns.from_json(predj)
    dt.track_id, n = fl.make_track_ids_unique(dt)
    assert n == want["n_track_ids_changed"]
    assert dt.track_id.tolist() == want["unique_track_ids"]
    f = fl.flatten_tao(GTColumns.from_json(gtj), dt)
    assert f.vid_ids.tolist() == want["vid_ids"] and f.cat_ids.tolist() == want["cat_ids"]
    assert {str(k): v for k, v in f.track_scores.items()} == want["track_scores"]
    out = orclib.run_flat(f)
    exact = name in INTEGER_FIXTURES
    _check_side(f, out, want, f.vid_ids, -1, exact_iou=exact)
    p, r = load_eval(name)["tao"]
    assert np.array_equal(out["precision"].reshape(p.shape), p)
    assert np.array_equal(out["recall"].reshape(r.shape), r)


def check_lvis_counts(f, out, want):
    """Integer match counts of a LITE fixture: detections, TPs and FPs per
    (category, range, threshold)."""
    got = {}
    cat = f.dt_cat[out["order"]]
    for k in np.unique(cat):
        sel = out["order"][cat == k]
        for r in range(6):
            if out["num_gt"][k, r] == 0:
                continue
            tps, fps = [], []
            for t in range(N_THR):
                mt = _bits(out["matched"][sel], r * N_THR + t)
                ig = _bits(out["ignored"][sel], r * N_THR + t)
                tps.append(int((mt & (1 - ig)).sum()))
                fps.append(int(((1 - mt) & (1 - ig)).sum()))
            got[int(k), r] = [len(sel), tps, fps]
    for k, r in zip(*np.nonzero(out["num_gt"])):
        got.setdefault((int(k), int(r)), [0, [0] * N_THR, [0] * N_THR])
    assert got == {(c[0], c[1]): c[2:] for c in want["counts"]}


from goldenio import DECIMAL_SCALE_FIXTURES


@pytest.mark.parametrize("name", DECIMAL_SCALE_FIXTURES)
def test_decimal_scale_fixture_flatten_and_c_oracle(name):
    """F8 (down-scaled decimal Config 2, golden from the reference): the C
    oracle adds a pair's frames in timeline order, so its IoUs carry other last
    bits (1e-12) -- no comparison of the match sits that close on this set, and
    every decision, count and table is the reference's."""
    gtj, predj = load_inputs(name)
    gt, dt = GTColumns.from_json(gtj), DTColumns.from_json(predj)
    want = load_json_gz(name, "lvis.json.gz")
    f = fl.flatten_lvis(gt, dt)
    assert f.img_ids.tolist() == want["img_ids"] and f.cat_ids.tolist() == want["cat_ids"]
    out = orclib.run_flat(f)
    check_lvis_counts(f, out, want)
    ev = load_eval(name)
    assert np.array_equal(out["precision"], ev["lvis"][0])
    assert np.array_equal(out["recall"], ev["lvis"][1])
    want = load_json_gz(name, "tao.json.gz")
    dt.track_id, n = fl.make_track_ids_unique(dt)
    assert n == want["n_track_ids_changed"]
    f = fl.flatten_tao(gt, dt)
    out = orclib.run_flat(f)
    _check_side(f, out, want, f.vid_ids, -1, exact_iou=False)
    p, r = ev["tao"]
    assert np.array_equal(out["precision"].reshape(p.shape), p)
    assert np.array_equal(out["recall"].reshape(r.shape), r)


def test_thresholds_are_numpy_linspace_bit_for_bit():
    a, b = orclib.thresholds()
    assert np.array_equal(a, np.linspace(0.5, 0.95, 10))
    assert np.array_equal(b, np.linspace(0.0, 1.0, 101))


def test_bb_iou_against_reference_compiled_from_source():
    """oracle/_ref holds bbIou compiled from the reference's own maskApi.c."""
    import os
    if not os.path.exists(orclib.REF_SO):
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    for scale in (1.0, 0.37):
        dt = np.c_[rng.integers(-50, 500, (200, 2)), rng.integers(0, 300, (200, 2))] * scale
        gt = np.c_[rng.integers(-50, 500, (90, 2)), rng.integers(0, 300, (90, 2))] * scale
        assert np.array_equal(orclib.bb_iou(dt, gt), orclib.ref_bb_iou(dt, gt))
    assert orclib.bb_iou([[0, 0, 20, 20]], [[0, 0, 10, 10]])[0, 0] == 0.25


from goldenio import MODE_FIXTURES, MODES, load_modes


@pytest.mark.parametrize("name", MODE_FIXTURES)
@pytest.mark.parametrize("mode", list(MODES))
def test_tao_other_modes_flatten_and_c_oracle(name, mode):
    """avg_iou (canonical frame order: within 1e-12 of the reference's
    np.mean), imagenetvid (integer counts: exact) and use_cats=0."""
    gtj, predj = load_inputs(name)
    dt = DTColumns.from_json(predj)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    cfg = MODES[mode]
    f = fl.flatten_tao(GTColumns.from_json(gtj), dt, use_cats=cfg["use_cats"])
    out = orclib.run_flat(f, iou_3d_type=cfg["iou_3d_type"])
    cells, p, r, _ = load_modes(name)[mode]
    off = orclib.iou_offsets(f)
    exact = mode == "imagenetvid" or (mode == "nocats" and name in INTEGER_FIXTURES)
    for k in range(f.n_cells):
        key = (int(f.vid_ids[f.cell_unit[k]]), int(f.cat_ids[f.cell_cat[k]]))
        D = f.cell_dt_off[k + 1] - f.cell_dt_off[k]
        G = f.cell_gt_off[k + 1] - f.cell_gt_off[k]
        if D == 0 or G == 0:
            continue
        got, want = out["iou"][off[k]:off[k + 1]].reshape(D, G), cells[key]
        if exact:
            assert np.array_equal(got, want), key
        else:
            assert np.allclose(got, want, rtol=0, atol=1e-12), key
    assert np.array_equal(out["precision"].reshape(p.shape), p)
    assert np.array_equal(out["recall"].reshape(r.shape), r)


@pytest.mark.parametrize("name", MODE_FIXTURES)
def test_lvis_without_categories_flatten_and_c_oracle(name):
    """flatten_lvis(use_cats=False): one cell per image, category-major then
    score order (reference L/eval.py:147-166), against the reference's own
    LVISEval run with params.use_cats = 0."""
    import nocats_check
    gtj, predj = load_inputs(name)
    f = fl.flatten_lvis(GTColumns.from_json(gtj), DTColumns.from_json(predj),
                        use_cats=False)
    nocats_check.check(f, orclib.run_flat(f), name)


def test_oracle_threads_do_not_change_results():
    """The all-cores CPU baseline of bench.py = the same oracle with OpenMP
    over cells / categories: bit-identical to the single-thread run."""
    import orclib
    from tao_amodal_amd.synth import synth
    gt, dt = synth(seed=5, V=6, F=25, C=40, dets_per_frame=30)
    fl_ = fl.flatten_lvis(gt, dt)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    ft_ = fl.flatten_tao(gt, dt)
    try:
        for f in (fl_, ft_):
            orclib.set_threads(1)
            a = orclib.run_flat(f)
            assert orclib.set_threads(0) >= 1
            b = orclib.run_flat(f)
            for k in ("matched", "ignored", "precision", "recall", "order", "num_gt"):
                assert np.array_equal(a[k], b[k]), k
            if f.kind == "tao":
                assert np.array_equal(a["iou"], b["iou"]) and a["pairs"] == b["pairs"]
    finally:
        orclib.set_threads(1)
