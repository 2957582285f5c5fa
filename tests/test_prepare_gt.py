"""flatten_dev.prepare_gt: the ground-truth halves of the cell tables built
ahead of the predictions (the CLI builds them while prediction.json is still
being parsed) are handed over once and never to edited columns."""
import numpy as np

from tao_amodal_amd import flatten, flatten_dev
from tao_amodal_amd.synth import synth


def _same(a, b):
    assert sorted(a) == sorted(b)
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, dict):
            _same(x, y)
        elif isinstance(x, tuple):
            for p, q in zip(x, y):
                assert np.array_equal(np.asarray(p), np.asarray(q)), k
        else:
            assert np.array_equal(np.asarray(x), np.asarray(y)), k


def test_bundles_equal_a_fresh_build_and_are_single_use():
    gt, _ = synth(seed=4, V=5, F=12, C=9, dets_per_frame=6, n_present=4)
    flatten_dev.prepare_gt(gt)
    for kind in ("lvis", "tao"):
        key, made, _parts = vars(gt)["_prepared_gt"]
        stored = made[kind].result()
        got = flatten_dev._gt_ready(gt, kind)
        assert got is stored
        again = flatten_dev._gt_ready(gt, kind)     # gone: built afresh
        assert again is not stored
        _same(got, again)
    assert "_prepared_gt" not in vars(gt)
    # the halves are what the host statement computes
    G = flatten.lvis_gt_side(gt)
    R = flatten_dev._lvis_gt_ready(gt)
    assert np.array_equal(np.sort(R.g_sel), np.sort(G.g_sel))


def test_a_rebound_column_is_not_served_from_the_bundle():
    gt, _ = synth(seed=4, V=5, F=12, C=9, dets_per_frame=6, n_present=4)
    flatten_dev.prepare_gt(gt)
    stored = vars(gt)["_prepared_gt"][1]["lvis"].result()
    gt.ann_area = gt.ann_area.copy()
    gt.ann_area[:] = 0.0                    # every ground truth filtered out
    got = flatten_dev._gt_ready(gt, "lvis")
    assert got is not stored
    assert len(got.g_sel) == 0


def test_errors_wait_for_the_build_that_needs_the_bundle():
    gt, _ = synth(seed=4, V=5, F=12, C=9, dets_per_frame=6, n_present=4)
    gt.ann_trk = gt.ann_trk.copy()
    gt.ann_trk[0] = 10 ** 9                 # annotation of a track that is not listed
    flatten_dev.prepare_gt(gt)              # silent
    assert vars(gt)["_prepared_gt"][1]["tao"].result() is None
    try:
        flatten_dev._gt_ready(gt, "tao")
    except KeyError as e:
        assert e.args[0] == 10 ** 9
    else:
        raise AssertionError("KeyError expected")


def test_halves_built_in_the_background_are_waited_for():
    gt, _ = synth(seed=4, V=5, F=12, C=9, dets_per_frame=6, n_present=4)
    flatten_dev.prepare_gt(gt, wait=False)          # returns at once
    for kind in ("tao", "lvis"):
        fut = vars(gt)["_prepared_gt"][1][kind]
        got = flatten_dev._gt_ready(gt, kind)       # waits for the half
        assert fut.done() and got is fut.result()
        _same(got, flatten_dev._READY[kind](gt))
    assert "_prepared_gt" not in vars(gt)


def test_the_track_level_half_in_two_stages():
    """prepare_gt builds the track level's half as tao_gt_universe (no
    annotation looked at) and the rest on top of it: the same tables as in one
    piece, the first stage handed out ahead of the whole."""
    gt, _ = synth(seed=4, V=5, F=12, C=9, dets_per_frame=6, n_present=4)
    whole = flatten.tao_gt_side(gt)
    A = flatten.tao_gt_universe(gt)
    staged = flatten.tao_gt_side(gt, universe=A)
    assert set(whole.keys()) == set(staged.keys())
    for k in whole:
        assert np.array_equal(np.asarray(whole[k]), np.asarray(staged[k])), k
    flatten_dev.prepare_gt(gt, wait=False)
    U = flatten_dev._gt_universe(gt)
    assert U is not None and np.array_equal(U.visit_rank, A.visit_rank)
    assert flatten_dev._gt_universe(gt) is None            # single use
    R = flatten_dev._gt_ready(gt, "tao")
    _same(R, flatten_dev._tao_gt_ready(gt))
    flatten_dev._gt_ready(gt, "lvis")
    assert "_prepared_gt" not in vars(gt)
    # a rebound column: the universe of the old columns is not handed out
    flatten_dev.prepare_gt(gt, wait=False)
    gt.img_frame = gt.img_frame.copy()
    assert flatten_dev._gt_universe(gt) is None
