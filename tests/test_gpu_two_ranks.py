"""The category-partitioned multi-GPU evaluation with REAL kernels and more
than one rank: the ranks share the one GPU of the test box and talk over gloo
(RCCL refuses two ranks on one device), so pack -> all-gather -> unpack runs
between genuinely different processes holding different category blocks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orclib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _problem(world):
    from tao_amodal_amd import flatten
    from tao_amodal_amd.columns import DTColumns, GTColumns
    from tao_amodal_amd.synth import synth
    parts = [synth(seed=23 + r, V=4, F=20, C=40, dets_per_frame=30, n_present=5,
                   video_id_base=r * 4) for r in range(world)]
    gt = GTColumns.concat([p[0] for p in parts])
    dt = DTColumns.concat([p[1] for p in parts])
    fl = flatten.flatten_lvis(gt, dt)
    dt.track_id, _ = flatten.make_track_ids_unique(dt)
    return fl, flatten.flatten_tao(gt, dt)


def _worker(rank, world, port, out):
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tao_amodal_amd import dist as tdist, engine
    fl, ft = _problem(world)
    dev = torch.device("cuda", 0)
    k0, k1, _ = tdist.category_block(len(fl.cat_ids), rank, world)
    plan = tdist.CategoryPlan(
        engine.DeviceProblem(tdist.shard_by_category(fl, k0, k1), dev),
        engine.DeviceProblem(tdist.shard_by_category(ft, k0, k1), dev),
        rank, world, dev)
    plan.step()
    plan.step()
    torch.cuda.synchronize()
    plan.lvis.check()
    plan.tao.check()
    torch.save({"lvis": (plan.lvis.precision.cpu().numpy(), plan.lvis.recall.cpu().numpy()),
                "tao": (plan.tao.precision.cpu().numpy(), plan.tao.recall.cpu().numpy())},
               os.path.join(out, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_reproduce_the_whole_problem(tmp_path, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    fl, ft = _problem(world)
    want = {"lvis": orclib.run_flat(fl, detail=False),
            "tao": orclib.run_flat(ft, detail=False)}
    for rank in range(world):
        got = torch.load(os.path.join(str(tmp_path), "r%d.pt" % rank), weights_only=False)
        for k in ("lvis", "tao"):
            assert np.array_equal(got[k][0], want[k]["precision"]), (rank, k)
            assert np.array_equal(got[k][1], want[k]["recall"]), (rank, k)


# --------------------------------------------------------------------------
# By-video plan, a rank WITHOUT a detection in its own category block (ADVICE
# r5): the rank's own-block slices of the send buffers are empty -- a NULL data
# pointer -- while records of the other rank arrive for that very block.
# --------------------------------------------------------------------------
def _unit_parts(world):
    from tao_amodal_amd.synth import synth
    parts = [synth(seed=61 + r, V=3, F=12, C=40, dets_per_frame=25, n_present=6,
                   video_id_base=r * 3) for r in range(world)]
    # the last rank keeps only detections of the FIRST owner's categories
    gt, dt = parts[-1]
    kb = (40 + world - 1) // world
    first_block = np.sort(np.asarray(gt.cat_id))[:kb]
    parts[-1] = (gt, dt.take(np.flatnonzero(np.isin(dt.category_id, first_block))))
    return parts


def _unit_worker(rank, world, port, out):
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tao_amodal_amd import dist as tdist, engine, flatten, flatten_dev
    dev = torch.device("cuda", 0)
    gt, dt = _unit_parts(world)[rank]
    universe = tdist.gather_visit_universe(gt, dev)
    f_l = flatten_dev.flatten_lvis(gt, dt, device=dev)
    dt.track_id, _ = flatten.make_track_ids_unique(dt)
    f_t = flatten_dev.flatten_tao(gt, dt, device=dev, visit_universe=universe)
    plan = tdist.ExchangePlan(engine.DeviceProblem(f_l, dev), engine.DeviceProblem(f_t, dev),
                              rank, world, dev)
    empty_own = [ev.own_at >= ev.dp.n_dt and ev.n_recv > 0 for ev in (plan.lvis, plan.tao)]
    plan.step()
    plan.step()
    torch.cuda.synchronize()
    plan.lvis.check()
    plan.tao.check()
    torch.save({"lvis": (plan.lvis.precision.cpu().numpy(), plan.lvis.recall.cpu().numpy()),
                "tao": (plan.tao.precision.cpu().numpy(), plan.tao.recall.cpu().numpy()),
                "empty_own": empty_own},
               os.path.join(out, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_by_video_rank_without_detections_in_its_own_block(tmp_path):
    from tao_amodal_amd import flatten
    from tao_amodal_amd.columns import DTColumns, GTColumns
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_unit_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = _unit_parts(world)
    gt = GTColumns.concat([p[0] for p in parts])
    dt = DTColumns.concat([p[1] for p in parts])
    fl = flatten.flatten_lvis(gt, dt)
    dt.track_id, _ = flatten.make_track_ids_unique(dt)
    ft = flatten.flatten_tao(gt, dt)
    want = {"lvis": orclib.run_flat(fl, detail=False), "tao": orclib.run_flat(ft, detail=False)}
    for rank in range(world):
        got = torch.load(os.path.join(str(tmp_path), "r%d.pt" % rank), weights_only=False)
        if rank == world - 1:
            assert all(got["empty_own"]), got["empty_own"]     # (the case is really hit)
        for k in ("lvis", "tao"):
            assert np.array_equal(got[k][0], want[k]["precision"]), (rank, k)
            assert np.array_equal(got[k][1], want[k]["recall"]), (rank, k)
