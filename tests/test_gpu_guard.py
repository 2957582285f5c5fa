"""-m gpu: the frame-order guard of the 3D IoU.  The kernels add a track
pair's frames in timeline order, the reference in CPython set order
(T/eval.py:83-94); pairs whose IoU a last-bit difference could move across a
comparison of the match are listed on the device (taoamd_track_iou_near) and
recomputed on the host in the reference's order.  F7 -- golden vectors from the
reference, every IoU on a threshold -- is reproduced exactly only with it."""
import numpy as np
import pytest

from goldenio import ADVERSARIAL_FIXTURES, load_eval, load_inputs, load_json_gz
from test_flat_oracle_golden import _check_side
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


@pytest.mark.parametrize("device_build", [False, True])
@pytest.mark.parametrize("name", ADVERSARIAL_FIXTURES)
def test_guarded_pairs_reproduce_the_reference_exactly(name, device_build):
    from tao_amodal_amd import engine
    f = _flat(name, device_build)
    got = engine.evaluate_flat(f, "cuda:0", detail=True)
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
    engine.run(dp, ws)
    torch.cuda.synchronize()
    assert not dp.exact_terms and int(ws.near_count.item()) > 0
    p, _ = load_eval(name)["tao"]
    assert not np.array_equal(ws.precision.cpu().numpy().reshape(p.shape), p)


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
        assert not dpt.exact_terms and dpt.guard_flat is not None
        cls = tdist.CategoryPlan if mode == "category" else tdist.ExchangePlan
        plan = cls(dpl, dpt, 0, 1, dev)
        plan.step()
        plan.step()
        torch.cuda.synchronize()
        assert plan.tao.ws.guarded_pairs > 0
        p, r = load_eval("f7")["tao"]
        assert np.array_equal(plan.tao.precision.cpu().numpy().reshape(p.shape), p)
        assert np.array_equal(plan.tao.recall.cpu().numpy().reshape(r.shape), r)
        p, r = load_eval("f7")["lvis"]
        assert np.array_equal(plan.lvis.precision.cpu().numpy().reshape(p.shape), p)
        assert np.array_equal(plan.lvis.recall.cpu().numpy().reshape(r.shape), r)
    finally:
        dist.destroy_process_group()
