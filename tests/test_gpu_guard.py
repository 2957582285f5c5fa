"""-m gpu: the frame-order guard of the 3D IoU.  The kernels add a track
pair's frames in timeline order, the reference in CPython set order
(T/eval.py:83-94); pairs whose IoU a reordering could move across a comparison
of the match are listed on the device (taoamd_track_iou_near) and recomputed
there in the reference's order (taoamd_track_iou_setorder: CPython's set
restated in csrc/pyset.hpp) -- or, as a cross-check, on the host with Python's
own sets.  F7 -- golden vectors from the reference, every IoU on a threshold --
is reproduced exactly only with it; F8 -- golden vectors from the reference,
decimal boxes, tracks of hundreds of frames -- pins the device's set order at
the size real files have."""
import numpy as np
import pytest

from goldenio import (ADVERSARIAL_FIXTURES, DECIMAL_SCALE_FIXTURES, load_eval,
                      load_inputs, load_json_gz)
from test_flat_oracle_golden import _check_side, check_lvis_counts
from tao_amodal_amd import flatten as fl
from tao_amodal_amd.columns import DTColumns, GTColumns

pytestmark = pytest.mark.gpu


def _flat(name, device_build):
    gtj, predj = load_inputs(name)
    gt, dt = GTColumns.from_json(gtj), DTColumns.from_json(predj)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    if device_build:
        from tao_amodal_amd import flatten_dev
        return flatten_dev.flatten_tao_device(gt, dt, "cuda:0")
    return fl.flatten_tao(gt, dt)


@pytest.mark.parametrize("guard", ["device", "host"])
@pytest.mark.parametrize("device_build", [False, True])
@pytest.mark.parametrize("name", ADVERSARIAL_FIXTURES)
def test_guarded_pairs_reproduce_the_reference_exactly(name, device_build, guard):
    from tao_amodal_amd import engine
    f = _flat(name, device_build)
    got = engine.evaluate_flat(f, "cuda:0", detail=True, guard=guard)
    assert got["near_threshold_pairs"] > 0
    want = load_json_gz(name, "tao.json.gz")
    # (IoUs away from every comparison keep their timeline-order bits: within
    # 1e-12 of the reference; every match decision, ignore flag and the
    # precision / recall tables are the reference's)
    _check_side(f, got, want, f.vid_ids, -1, exact_iou=False)
    p, r = load_eval(name)["tao"]
    assert np.array_equal(got["precision"].reshape(p.shape), p)
    assert np.array_equal(got["recall"].reshape(r.shape), r)


@pytest.mark.parametrize("name", ADVERSARIAL_FIXTURES)
def test_without_the_guard_matches_flip(name):
    """The same tables through the bare stages (no host patch between the 3D
    IoU and the match): some matches differ from the reference's -- what the
    guard is for."""
    import torch
    from tao_amodal_amd import engine
    f = _flat(name, False)
    dp = engine.DeviceProblem(f, "cuda:0")
    ws = engine.Workspace(dp)
    assert not dp.exact_terms and dp.guard_active() and dp.guard_on_device
    for stage in (engine.stage_ranges, engine.stage_sort, engine.stage_track_iou,
                  engine.stage_match, engine.stage_accumulate):
        stage(dp, ws)
    torch.cuda.synchronize()
    p, _ = load_eval(name)["tao"]
    assert not np.array_equal(ws.precision.cpu().numpy().reshape(p.shape), p)
    # ... and the plain pass (engine.run) carries the guard
    engine.run(dp, ws)
    torch.cuda.synchronize()
    assert engine.guarded_pairs(dp, ws) > 0
    assert np.array_equal(ws.precision.cpu().numpy().reshape(p.shape), p)


def test_integer_boxes_have_nothing_to_guard():
    from tao_amodal_amd import engine
    from tao_amodal_amd.synth import synth
    gt, dt = synth(seed=3, V=3, F=12, C=8, dets_per_frame=10, n_present=4)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    f = fl.flatten_tao(gt, dt)
    dp = engine.DeviceProblem(f, "cuda:0")
    assert dp.exact_terms
    assert engine.evaluate_flat(f, "cuda:0")["near_threshold_pairs"] == 0


def test_class_api_reports_the_guarded_pairs(tmp_path):
    from goldenio import path
    from tao_amodal_amd.evaluation.tao_amodal import Tao, TaoEval, TaoResults
    gtj, predj = load_inputs("f7")
    dt = DTColumns.from_json(predj)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    gt = Tao(gtj)
    ev = TaoEval(gt, TaoResults(gt, dt))
    ev.run()
    assert ev.near_threshold_pairs > 0
    want = load_json_gz("f7", "tao.json.gz")
    res = [[k if isinstance(k, str) else list(k), float(v)] for k, v in ev.results.items()]
    assert res == want["results"]


@pytest.mark.parametrize("mode", ["category", "unit"])
def test_multi_gpu_plans_apply_the_guard(mode):
    """The plans of dist.py (one rank here: RCCL group of one) patch the
    listed pairs between the 3D IoU and the match like the single-GPU path:
    the adversarial fixture comes out as the reference computed it."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from tao_amodal_amd import dist as tdist, engine, flatten_dev
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        gtj, predj = load_inputs("f7")
        gt, dt = GTColumns.from_json(gtj), DTColumns.from_json(predj)
        f_l = flatten_dev.flatten_lvis(gt, dt, device=dev)
        dt.track_id, _ = fl.make_track_ids_unique(dt)
        f_t = flatten_dev.flatten_tao(gt, dt, device=dev)
        dpl, dpt = engine.DeviceProblem(f_l, dev), engine.DeviceProblem(f_t, dev)
        assert not dpt.exact_terms and dpt.guard_on_device
        cls = tdist.CategoryPlan if mode == "category" else tdist.ExchangePlan
        plan = cls(dpl, dpt, 0, 1, dev)
        plan.step()
        plan.step()
        torch.cuda.synchronize()
        assert engine.guarded_pairs(dpt, plan.tao.ws) > 0
        p, r = load_eval("f7")["tao"]
        assert np.array_equal(plan.tao.precision.cpu().numpy().reshape(p.shape), p)
        assert np.array_equal(plan.tao.recall.cpu().numpy().reshape(r.shape), r)
        p, r = load_eval("f7")["lvis"]
        assert np.array_equal(plan.lvis.precision.cpu().numpy().reshape(p.shape), p)
        assert np.array_equal(plan.lvis.recall.cpu().numpy().reshape(r.shape), r)
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------- F8: scale
def _setorder_all_pairs(dp, ws):
    """Recompute EVERY pair in set order (the near list filled by hand)."""
    import torch
    from tao_amodal_amd import _lib
    lib, t = _lib.load(), dp.t
    ws.near_list = torch.arange(dp.n_iou, dtype=torch.int64, device=dp.device)
    ws.near_cap = dp.n_iou
    ws.near_count.fill_(dp.n_iou)
    ptr = lambda x: x.data_ptr()
    _lib.check(lib.taoamd_track_iou_setorder(
        dp.n_cells, ptr(t["cell_dt_off"]), ptr(t["cell_gt_off"]), ptr(t["cell_iou_off"]),
        ptr(t["cell_unit"]), ptr(t["tl_vid_start"]), ptr(t["tl_image_id"]),
        ptr(t["dt_frame_off"]), ptr(t["dt_frame_pos"]), ptr(t["dt_frame_box"]),
        ptr(t["gt_frame_off"]), ptr(t["gt_frame_pos"]), ptr(t["gt_frame_box"]),
        dp.iou_mode, ptr(ws.near_count), ws.near_cap, ptr(ws.near_list), ptr(ws.iou),
        ptr(ws.guard_scratch), ws.guard_slots, ws.guard_table, ptr(ws.guard_status),
        torch.cuda.current_stream().cuda_stream), "taoamd_track_iou_setorder")
    torch.cuda.synchronize()
    assert int(ws.guard_status.item()) == 0


@pytest.mark.parametrize("device_build", [False, True])
@pytest.mark.parametrize("name", DECIMAL_SCALE_FIXTURES)
def test_decimal_scale_fixture_every_decision_is_the_references(name, device_build):
    """F8 through the plain pass: every IoU within 1e-12, every match, ignore
    flag, TP/FP sequence, precision and recall == the reference's."""
    from tao_amodal_amd import engine
    f = _flat(name, device_build)
    got = engine.evaluate_flat(f, "cuda:0", detail=True)
    want = load_json_gz(name, "tao.json.gz")
    _check_side(f, got, want, f.vid_ids, -1, exact_iou=False)
    p, r = load_eval(name)["tao"]
    assert np.array_equal(got["precision"].reshape(p.shape), p)
    assert np.array_equal(got["recall"].reshape(r.shape), r)


@pytest.mark.parametrize("name", DECIMAL_SCALE_FIXTURES)
def test_decimal_scale_fixture_image_level(name):
    from tao_amodal_amd import engine, flatten_dev
    gtj, predj = load_inputs(name)
    gt, dt = GTColumns.from_json(gtj), DTColumns.from_json(predj)
    f = flatten_dev.flatten_lvis_device(gt, dt, "cuda:0")
    got = engine.evaluate_flat(f, "cuda:0")
    check_lvis_counts(f, got, load_json_gz(name, "lvis.json.gz"))
    p, r = load_eval(name)["lvis"]
    assert np.array_equal(got["precision"], p)
    assert np.array_equal(got["recall"], r)


@pytest.mark.parametrize("name", DECIMAL_SCALE_FIXTURES + ADVERSARIAL_FIXTURES + ["f4"])
def test_device_set_order_equals_the_reference_bit_for_bit(name):
    """Every pair recomputed by taoamd_track_iou_setorder: the IoU matrix is
    the reference's to the last bit (golden `ious`: set-order sums over up to
    600 frames on F8) -- CPython's set, restated on the device."""
    import torch
    from tao_amodal_amd import engine
    f = _flat(name, False)
    dp = engine.DeviceProblem(f, "cuda:0")
    ws = engine.Workspace(dp, detail=True)
    assert dp.guard_on_device
    engine.stage_ranges(dp, ws)
    engine.stage_sort(dp, ws)
    engine.stage_track_iou(dp, ws)
    _setorder_all_pairs(dp, ws)
    got_iou = ws.iou[:dp.n_iou].cpu().numpy()
    want = {tuple(c["key"]): c for c in load_json_gz(name, "tao.json.gz")["cells"]}
    n = 0
    for k in range(f.n_cells):
        D = f.cell_dt_off[k + 1] - f.cell_dt_off[k]
        G = f.cell_gt_off[k + 1] - f.cell_gt_off[k]
        if D == 0 or G == 0:
            continue
        key = (int(f.vid_ids[f.cell_unit[k]]), int(f.cat_ids[f.cell_cat[k]]))
        o = int(dp.t["cell_iou_off"][k].item())
        wi = np.asarray(want[key]["ious"], dtype=float)
        assert np.array_equal(got_iou[o:o + D * G].reshape(D, G), wi), key
        n += D * G
    assert n == dp.n_iou and n > 0


@pytest.mark.parametrize("mode", ["3d_iou", "avg_iou"])
def test_device_guard_equals_the_host_guard(mode):
    """Synthetic decimal set with shuffled image ids, every pair listed: the
    device's recompute (pyset.hpp) == the host's (Python sets, np.mean)."""
    import torch
    from tao_amodal_amd import engine
    from tao_amodal_amd.synth import synth
    gt, dt = synth(seed=21, V=6, F=90, C=12, dets_per_frame=8, n_present=4,
                   decimal=True, shuffle_image_ids=True)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    f = fl.flatten_tao(gt, dt)
    dp = engine.DeviceProblem(f, "cuda:0", iou_3d_type=mode)
    ws = engine.Workspace(dp)
    engine.stage_track_iou(dp, ws)
    _setorder_all_pairs(dp, ws)
    got = ws.iou[:dp.n_iou].cpu().numpy()
    want = engine.set_order_iou(f, np.arange(dp.n_iou), dp.iou_mode)
    assert np.array_equal(got, want)


def test_average_iou_of_integer_boxes_is_guarded():
    """Integer boxes make the 3D IoU's sums exact, NOT the average IoU's
    per-frame ratios: the guard stays on for avg_iou (and off for the
    count-based imagenetvid IoU)."""
    from tao_amodal_amd import engine
    from tao_amodal_amd.synth import synth
    gt, dt = synth(seed=3, V=3, F=12, C=8, dets_per_frame=10, n_present=4)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    f = fl.flatten_tao(gt, dt)
    assert not engine.DeviceProblem(f, "cuda:0", "3d_iou").guard_active()
    assert engine.DeviceProblem(f, "cuda:0", "avg_iou").guard_active()
    assert not engine.DeviceProblem(f, "cuda:0", "imagenetvid").guard_active()


@pytest.mark.parametrize("mode", ["category", "unit"])
@pytest.mark.parametrize("name", DECIMAL_SCALE_FIXTURES)
def test_multi_gpu_plans_on_the_decimal_scale_fixture(name, mode):
    """F8 through both partitions of dist.py (a group of one rank): tables ==
    the reference's."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from tao_amodal_amd import dist as tdist, engine, flatten_dev
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        gtj, predj = load_inputs(name)
        gt, dt = GTColumns.from_json(gtj), DTColumns.from_json(predj)
        f_l = flatten_dev.flatten_lvis(gt, dt, device=dev)
        dt.track_id, _ = fl.make_track_ids_unique(dt)
        f_t = flatten_dev.flatten_tao(gt, dt, device=dev)
        dpl, dpt = engine.DeviceProblem(f_l, dev), engine.DeviceProblem(f_t, dev)
        assert dpt.guard_active() and dpt.guard_on_device
        cls = tdist.CategoryPlan if mode == "category" else tdist.ExchangePlan
        plan = cls(dpl, dpt, 0, 1, dev)
        plan.step()
        plan.step()
        torch.cuda.synchronize()
        ev = load_eval(name)
        for side, pl in (("tao", plan.tao), ("lvis", plan.lvis)):
            p, r = ev[side]
            assert np.array_equal(pl.precision.cpu().numpy().reshape(p.shape), p)
            assert np.array_equal(pl.recall.cpu().numpy().reshape(r.shape), r)
    finally:
        dist.destroy_process_group()
