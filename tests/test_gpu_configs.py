"""-m gpu: the BASELINE.json configurations beyond Config 2, at full size.

    Config 3 (stand-in)  2000 videos x 300 frames x 50 dets, 1203 categories
                         (the real validation JSONs are not in the container)
    Config 4             the full-validation-scale set (2000 videos) split BY VIDEO
                         over 2 and 8 ranks, the partition BASELINE.json names;
                         and, as the fast variant, Config 2 (200 videos) over
                         2 / 4 / 8 ranks in BOTH partitions (by category, by
                         video).  RCCL when the box has that many GPUs, else the
                         ranks share GPU 0 and talk over gloo
    Config 5 (stand-in)  10 000 videos x 1 frame x 1000 dets: the top-300 cut
                         per image at scale (L/results.py:39-40, T/results.py:56-58),
                         on one GPU and split by video over 8 ranks

Every one is compared bit for bit with the C oracle (all host cores) on the
WHOLE problem, plus idempotence of a second pass."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orclib
from tao_amodal_amd import flatten as fl
from tao_amodal_amd.synth import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _whole_problem_vs_oracle(gt, dt):
    """The cell tables are built on the device (flatten_dev) and must equal the
    numpy tables the oracle evaluates; the kernels run on the device tables."""
    from tao_amodal_amd import engine, flatten_dev
    f_l = fl.flatten_lvis(gt, dt)
    d_l = flatten_dev.flatten_lvis_device(gt, dt, "cuda:0")
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    f_t = fl.flatten_tao(gt, dt)
    d_t = flatten_dev.flatten_tao_device(gt, dt, "cuda:0")
    for a, b, names in ((d_l, f_l, ("dt_row", "dt_score", "dt_flags", "dt_cell")),
                        (d_t, f_t, ("dt_id", "dt_score", "dt_area", "dt_len",
                                    "dt_frame_off", "dt_frame_pos", "dt_cell"))):
        assert np.array_equal(a.cell_dt_off, b.cell_dt_off)
        assert np.array_equal(a.cell_gt_off, b.cell_gt_off)
        for k in names:
            assert np.array_equal(np.asarray(a[k]).astype(np.asarray(b[k]).dtype),
                                  np.asarray(b[k])), k
    orclib.set_threads(0)
    try:
        for flat, dflat in ((f_l, d_l), (f_t, d_t)):
            want = orclib.run_flat(flat, detail=False)
            dp = engine.DeviceProblem(dflat, "cuda:0")
            ws = engine.Workspace(dp)
            engine.run(dp, ws)
            torch.cuda.synchronize()
            n = dp.n_dt
            dst = ws.dst[:n].long()
            assert np.array_equal(ws.matched[:n][dst].cpu().numpy().view(np.uint64),
                                  want["matched"])
            assert np.array_equal(ws.ignored[:n][dst].cpu().numpy().view(np.uint64),
                                  want["ignored"])
            if flat.kind == "tao":
                assert np.array_equal(ws.iou[:dp.n_iou].cpu().numpy(), want["iou"])
                assert int(ws.pair_frames.item()) == want["pairs"]
            p1, r1 = ws.precision.clone(), ws.recall.clone()
            assert np.array_equal(p1.cpu().numpy(), want["precision"])
            assert np.array_equal(r1.cpu().numpy(), want["recall"])
            engine.run(dp, ws)                       # idempotent
            torch.cuda.synchronize()
            assert torch.equal(ws.precision, p1) and torch.equal(ws.recall, r1)
            del dp, ws, p1, r1
            torch.cuda.empty_cache()
    finally:
        orclib.set_threads(1)
    return f_l, f_t


@pytest.mark.timeout(3000)
def test_config3_standin_full_size():
    gt, dt = synth(V=2000, F=300, C=1203, dets_per_frame=50)
    f_l, f_t = _whole_problem_vs_oracle(gt, dt)
    assert f_l.n_pairs > 2 * 10 ** 7 and len(f_l.dt_flags) > 2 * 10 ** 7


@pytest.mark.timeout(3000)
def test_config5_standin_top300_cut_at_scale():
    gt, dt = synth(seed=5, V=10000, F=1, C=1203, dets_per_frame=1000)
    assert len(dt) == 10 ** 7
    # the cut itself, against a plain restatement: per image the 300 highest
    # scores, ties to the earlier box (stable sort, L/results.py:39-40)
    keep = fl.limit_dets_per_image(dt, 300)
    order = np.lexsort((np.arange(len(dt)), -dt.score, dt.image_id))
    rank_in_img = np.arange(len(dt)) - np.searchsorted(dt.image_id[order],
                                                       dt.image_id[order], "left")
    assert np.array_equal(np.sort(keep), np.sort(order[rank_in_img < 300]))
    assert len(keep) == 300 * 10000
    _whole_problem_vs_oracle(gt, dt)


# --------------------------------------------------------------------------
# Config 4: Config 2 over 2 / 4 / 8 ranks
# --------------------------------------------------------------------------
V_TOTAL = 200


def _parts(world):
    from tao_amodal_amd.columns import DTColumns, GTColumns
    per = V_TOTAL // world
    parts = [synth(seed=99 + r, V=per, F=300, C=1203, dets_per_frame=50,
                   video_id_base=r * per) for r in range(world)]
    return parts, GTColumns, DTColumns


def _whole(world):
    parts, GTColumns, DTColumns = _parts(world)
    gt = GTColumns.concat([p[0] for p in parts])
    dt = DTColumns.concat([p[1] for p in parts])
    f_l = fl.flatten_lvis(gt, dt)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    return f_l, fl.flatten_tao(gt, dt)


def _worker(rank, world, port, mode, out):
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    n_gpu = torch.cuda.device_count()
    rccl = n_gpu >= world
    dev = torch.device("cuda", rank if rccl else 0)
    torch.cuda.set_device(dev)
    if rccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        # (gloo moves no device tensors through all_to_all: the product stages
        # the record exchange on the host then, tao_amodal_amd.dist.all_to_all)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from tao_amodal_amd import dist as tdist, engine
    if mode == "category":
        f_l, f_t = _whole(world)
        k0, k1, _ = tdist.category_block(len(f_l.cat_ids), rank, world)
        plan = tdist.CategoryPlan(
            engine.DeviceProblem(tdist.shard_by_category(f_l, k0, k1), dev),
            engine.DeviceProblem(tdist.shard_by_category(f_t, k0, k1), dev),
            rank, world, dev)
    else:
        # a rank's own videos, tables built on the device; the visiting order
        # of the track level from the image ids of all ranks
        from tao_amodal_amd import flatten_dev
        parts, _, _ = _parts(world)
        gt, dt = parts[rank]
        universe = tdist.gather_visit_universe(gt, dev)
        f_l = flatten_dev.flatten_lvis(gt, dt, device=dev)
        dt.track_id, _ = fl.make_track_ids_unique(dt)
        f_t = flatten_dev.flatten_tao(gt, dt, device=dev, visit_universe=universe)
        plan = tdist.ExchangePlan(engine.DeviceProblem(f_l, dev),
                                  engine.DeviceProblem(f_t, dev), rank, world, dev)
    plan.step()
    for ev in (plan.lvis, plan.tao):       # (rows shipped before they are matched would show)
        ev.ws.rows.fill_(-1)
    plan.step()
    torch.cuda.synchronize()
    if mode == "category":
        plan.lvis.check()
        plan.tao.check()
    torch.save({"lvis": (plan.lvis.precision.cpu().numpy(), plan.lvis.recall.cpu().numpy()),
                "tao": (plan.tao.precision.cpu().numpy(), plan.tao.recall.cpu().numpy()),
                "backend": dist.get_backend()},
               os.path.join(out, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.timeout(3000)
@pytest.mark.parametrize("mode", ["category", "unit"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_config4_ranks_reproduce_the_whole_problem(tmp_path, world, mode):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, mode, str(tmp_path)), nprocs=world, join=True)
    f_l, f_t = _whole(world)
    orclib.set_threads(0)
    try:
        want = {"lvis": orclib.run_flat(f_l, detail=False),
                "tao": orclib.run_flat(f_t, detail=False)}
    finally:
        orclib.set_threads(1)
    for rank in range(world):
        got = torch.load(os.path.join(str(tmp_path), "r%d.pt" % rank), weights_only=False)
        for k in ("lvis", "tao"):
            assert np.array_equal(got[k][0], want[k]["precision"]), (rank, k)
            assert np.array_equal(got[k][1], want[k]["recall"]), (rank, k)


# --------------------------------------------------------------------------
# Config 4 at its own size: ONE set of 2000 videos split by video (the
# `--scaling strong` semantics of bench.py) -- VERDICT r3 configs_untested
# --------------------------------------------------------------------------
V_FULL = 2000


def _full_worker(rank, world, port, src, out):
    import pickle
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    rccl = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if rccl else 0)
    torch.cuda.set_device(dev)
    if rccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from tao_amodal_amd import dist as tdist, engine, flatten_dev
    with open(os.path.join(src, "w%d_r%d.pkl" % (world, rank)), "rb") as f:
        gt, dt = pickle.load(f)
    universe = tdist.gather_visit_universe(gt, dev)
    f_l = flatten_dev.flatten_lvis(gt, dt, device=dev)
    f_t = flatten_dev.flatten_tao(gt, dt, device=dev, visit_universe=universe)
    plan = tdist.ExchangePlan(engine.DeviceProblem(f_l, dev), engine.DeviceProblem(f_t, dev),
                              rank, world, dev)
    plan.step()
    for ev in (plan.lvis, plan.tao):       # (rows shipped before they are matched would show)
        ev.ws.rows.fill_(-1)
    plan.step()
    torch.cuda.synchronize()
    plan.lvis.check()
    plan.tao.check()
    # every rank holds the whole tables: a digest each, the tensors from rank 0
    digest = [int(t.contiguous().view(torch.int64).sum().item())
              for t in (plan.lvis.precision, plan.lvis.recall, plan.tao.precision,
                        plan.tao.recall)]
    got = {"digest": digest, "backend": dist.get_backend()}
    if rank == 0:
        got.update(lvis=(plan.lvis.precision.cpu().numpy(), plan.lvis.recall.cpu().numpy()),
                   tao=(plan.tao.precision.cpu().numpy(), plan.tao.recall.cpu().numpy()))
    torch.save(got, os.path.join(out, "r%d.pt" % rank))
    dist.destroy_process_group()


def _split_by_video_vs_oracle(tmp_path, gt, dt, n_videos, worlds):
    """ONE set cut into contiguous blocks of videos over `worlds` ranks (the
    by-video plan); every rank must end with the whole precision / recall
    tables, bit for bit the C oracle's on the whole set."""
    import pickle
    import shutil
    dt.track_id, _ = fl.make_track_ids_unique(dt)       # (a statement about the whole list)
    f_l = fl.flatten_lvis(gt, dt)
    f_t = fl.flatten_tao(gt, dt)
    orclib.set_threads(0)
    try:
        want = {"lvis": orclib.run_flat(f_l, detail=False),
                "tao": orclib.run_flat(f_t, detail=False)}
    finally:
        orclib.set_threads(1)
    del f_l, f_t
    want_digest = [int(np.ascontiguousarray(want[k][f]).view(np.int64).sum())
                   for k in ("lvis", "tao") for f in ("precision", "recall")]
    for world in worlds:
        out = str(tmp_path / ("out%d" % world))
        src = str(tmp_path / ("blocks%d" % world))
        os.makedirs(out)
        os.makedirs(src)
        for r in range(world):
            keep = np.zeros(n_videos, dtype=bool)
            keep[n_videos * r // world:n_videos * (r + 1) // world] = True   # (synth: ids ascending)
            mine = gt.vid_id[keep]
            part = (gt.select_videos(keep),
                    dt.take(np.flatnonzero((dt.video_id >= mine[0]) & (dt.video_id <= mine[-1]))))
            with open(os.path.join(src, "w%d_r%d.pkl" % (world, r)), "wb") as f:
                pickle.dump(part, f, protocol=4)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_full_worker, args=(world, port, src, out), nprocs=world, join=True)
        shutil.rmtree(src)
        for rank in range(world):
            got = torch.load(os.path.join(out, "r%d.pt" % rank), weights_only=False)
            assert [d & (2 ** 64 - 1) for d in got["digest"]] == \
                [d & (2 ** 64 - 1) for d in want_digest], (world, rank)
            if rank == 0:
                for k in ("lvis", "tao"):
                    assert np.array_equal(got[k][0], want[k]["precision"]), (world, k)
                    assert np.array_equal(got[k][1], want[k]["recall"]), (world, k)


@pytest.mark.timeout(3000)
def test_config4_at_full_validation_scale_split_by_video(tmp_path):
    """2000 videos x 300 frames x 50 detections, 1203 categories (the Config 3
    stand-in) cut into contiguous blocks of videos over 2 and over 8 ranks;
    every rank ends with the whole precision / recall tables, bit for bit the C
    oracle's on the whole set."""
    gt, dt = synth(seed=20240807, V=V_FULL, F=300, C=1203, dets_per_frame=50)
    _split_by_video_vs_oracle(tmp_path, gt, dt, V_FULL, (2, 8))


@pytest.mark.timeout(3000)
def test_config5_stress_set_split_over_8_ranks(tmp_path):
    """BASELINE Config 5 as written ("10k synthetic videos, 1k dets/frame, 8
    GPUs"): the stress stand-in (10 000 one-frame videos x 1000 detections, the
    top-300 cut per image) cut by video over 8 ranks -- every rank cuts its own
    images' lists (L/results.py:39-40 is per image), one-frame tracks take the
    single-frame 3D IoU kernel, the exchange runs in its category phases."""
    gt, dt = synth(seed=5, V=10000, F=1, C=1203, dets_per_frame=1000)
    _split_by_video_vs_oracle(tmp_path, gt, dt, 10000, (8,))
