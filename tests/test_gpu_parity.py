"""-m gpu: the HIP path (through the C ABI) against the golden vectors of the
reference and against the C oracle on seeded synthetic inputs.  Bit-exact:
integer match decisions, IoUs, precision and recall are compared with ==."""
import numpy as np
import pytest

import exchange_ref
import orclib
from goldenio import FIXTURES, INTEGER_FIXTURES, load_eval, load_inputs, load_json_gz
from test_flat_oracle_golden import _check_side
from tao_amodal_amd import flatten as fl
from tao_amodal_amd.columns import DTColumns, GTColumns
from tao_amodal_amd.synth import synth

pytestmark = pytest.mark.gpu


def _engine():
    from tao_amodal_amd import engine
    return engine


def _compare_with_oracle(f, got, detail=True):
    want = orclib.run_flat(f)
    assert np.array_equal(got["gt_rng"], want["gt_rng"])
    assert np.array_equal(got["dt_rng"], want["dt_rng"])
    assert np.array_equal(got["num_gt"], want["num_gt"])
    assert np.array_equal(got["order"], want["order"])
    if f.kind == "tao" or detail:
        assert np.array_equal(got["iou"], want["iou"])
    if f.kind == "tao":
        assert got["pairs"] == want["pairs"]
    assert np.array_equal(got["matched"], want["matched"])
    assert np.array_equal(got["ignored"], want["ignored"])
    if detail:
        assert np.array_equal(got["match_gt"], want["match_gt"])
    assert np.array_equal(got["precision"], want["precision"])
    assert np.array_equal(got["recall"], want["recall"])


@pytest.mark.parametrize("name", FIXTURES)
def test_lvis_hip_matches_reference_golden(name):
    gtj, predj = load_inputs(name)
    want = load_json_gz(name, "lvis.json.gz")
    f = fl.flatten_lvis(GTColumns.from_json(gtj), DTColumns.from_json(predj))
    got = _engine().evaluate_flat(f, detail=True)
    _check_side(f, got, want, f.img_ids, 0)
    p, r = load_eval(name)["lvis"]
    assert np.array_equal(got["precision"], p)
    assert np.array_equal(got["recall"], r)
    _compare_with_oracle(f, got)


@pytest.mark.parametrize("name", FIXTURES)
def test_tao_hip_matches_reference_golden(name):
    gtj, predj = load_inputs(name)
    want = load_json_gz(name, "tao.json.gz")
    dt = DTColumns.from_json(predj)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    f = fl.flatten_tao(GTColumns.from_json(gtj), dt)
    got = _engine().evaluate_flat(f, detail=True)
    _check_side(f, got, want, f.vid_ids, -1, exact_iou=name in INTEGER_FIXTURES)
    p, r = load_eval(name)["tao"]
    assert np.array_equal(got["precision"].reshape(p.shape), p)
    assert np.array_equal(got["recall"].reshape(r.shape), r)
    _compare_with_oracle(f, got)


@pytest.mark.parametrize("seed,V,F,C,dpf", [(1, 6, 30, 40, 25), (2, 3, 8, 1203, 60),
                                            (3, 10, 50, 7, 340)])
def test_synthetic_hip_vs_c_oracle(seed, V, F, C, dpf):
    gt, dt = synth(seed=seed, V=V, F=F, C=C, dets_per_frame=dpf,
                   n_present=min(5, C - 3))
    f = fl.flatten_lvis(gt, dt)
    _compare_with_oracle(f, _engine().evaluate_flat(f, detail=True))
    _compare_with_oracle(f, _engine().evaluate_flat(f, detail=False), detail=False)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    f = fl.flatten_tao(gt, dt)
    _compare_with_oracle(f, _engine().evaluate_flat(f, detail=True))


def _drop_frames(gt, dt, seed, keep=0.6):
    """Tracks with holes: a random subset of the boxes of both files."""
    rng = np.random.default_rng(seed)
    dt = dt.take(np.flatnonzero(rng.random(len(dt)) < keep))
    sel = np.flatnonzero(rng.random(len(gt.ann_id)) < keep)
    kw = {k: getattr(gt, k) for k in gt.FIELDS}
    for k in gt.FIELDS:
        if k.startswith("ann_"):
            kw[k] = kw[k][sel]
    return GTColumns(**kw), dt


def _check_plan(f, meta, tasks, rows, pairs, out):
    """Every (detection track, GT track) pair of every cell in exactly one
    task, each pair's rows hold its two tracks, sizes within the kernel's."""
    n_dt = len(f.dt_flags)
    seen = np.zeros(int(f.cell_iou_off[-1]), dtype=np.int32)
    cell_of = np.repeat(np.arange(f.n_cells), np.diff(f.cell_iou_off))
    for r0, nr, p0, npair in tasks:
        assert 0 < nr <= 32 and 0 < npair <= 64
        trk = rows[r0:r0 + nr]
        assert len(set(trk.tolist())) == nr
        assert (np.diff(meta[trk, 0]) >= 0).all()        # by first position
        pr, o = pairs[p0:p0 + npair], out[p0:p0 + npair]
        td, tg = trk[pr & 0xFF], trk[(pr >> 8) & 0xFF]
        assert (td < n_dt).all() and (tg >= n_dt).all()
        c = cell_of[o]
        G = (f.cell_gt_off[c + 1] - f.cell_gt_off[c]).astype(np.int64)
        want = f.cell_iou_off[c] + (td - f.cell_dt_off[c]) * G + \
            (tg - n_dt - f.cell_gt_off[c])
        assert np.array_equal(o, want)
        seen[o] += 1
    assert (seen == 1).all()


@pytest.mark.parametrize("seed,V,F,G,dpf", [
    (31, 1, 1100, 24, 40), (32, 2, 200, 40, 40), (34, 2, 200, 14, 40),
    (33, 1, 300, 8, 40), (35, 2, 150, 18, 40), (36, 1, 90, 70, 20),
    (37, 2, 37, 3, 300), (38, 3, 16, 1, 120), (39, 1, 17, 33, 30)])
def test_track_iou_tasks(seed, V, F, G, dpf):
    """Long timelines, one to seventy GT tracks per cell (GT blocks), cells
    with hundreds of detection tracks (several groups), timelines that end
    mid-chunk, tracks with holes: the task kernel, the plan-less merge kernel
    and the oracle agree bit for bit."""
    from tao_amodal_amd import engine
    gt, dt = synth(seed=seed, V=V, F=F, C=6, dets_per_frame=dpf,
                   gt_tracks_per_video=G, n_present=2, n_neg=1)
    for holes in (False, True):
        if holes:
            gt, dt = _drop_frames(gt, dt, seed)
        dt.track_id, _ = fl.make_track_ids_unique(dt)
        f = fl.flatten_tao(gt, dt)
        meta = engine.track_meta(f)[0]
        _check_plan(f, meta, *engine.track_iou_plan(f, meta))
        got = _engine().evaluate_flat(f, detail=True)
        _compare_with_oracle(f, got)
        for mode in ("avg_iou", "imagenetvid"):
            want, _ = orclib.track_iou(f, mode)
            g2 = _engine().evaluate_flat(f, iou_3d_type=mode)
            assert np.array_equal(g2["iou"], want)
        # without a plan: the two-pointer merge kernel
        dp = engine.DeviceProblem(f)
        dp.t["tasks"] = None
        ws = engine.Workspace(dp)
        engine.stage_track_iou(dp, ws)
        assert np.array_equal(ws.iou[:dp.n_iou].cpu().numpy(), got["iou"])
        assert int(ws.pair_frames.item()) == got["pairs"]


def test_cells_with_more_than_64_ground_truths():
    """Crowded cells take match_big_kernel (LDS row + LDS bitsets)."""
    gt, dt = synth(seed=9, V=2, F=3, C=4, dets_per_frame=200,
                   gt_tracks_per_video=150, n_present=1, n_neg=1)
    f = fl.flatten_lvis(gt, dt)
    assert np.diff(f.cell_gt_off).max() > 64
    _compare_with_oracle(f, _engine().evaluate_flat(f, detail=True))
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    f = fl.flatten_tao(gt, dt)
    assert np.diff(f.cell_gt_off).max() > 64
    _compare_with_oracle(f, _engine().evaluate_flat(f, detail=True))


def test_bb_iou_entry_points_match_reference_bbiou():
    """taoamd_bb_iou[_host] against the C oracle and (when present) the
    reference's own bbIou compiled from its source (oracle/_ref)."""
    import ctypes, os
    import torch
    from tao_amodal_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    dt = np.c_[rng.integers(-50, 500, (300, 2)), rng.integers(0, 300, (300, 2))] * 0.37
    gt = np.c_[rng.integers(-50, 500, (77, 2)), rng.integers(0, 300, (77, 2))] * 0.37
    dt, gt = np.ascontiguousarray(dt), np.ascontiguousarray(gt)
    o = np.zeros(300 * 77)
    _lib.check(lib.taoamd_bb_iou_host(dt.ctypes.data, gt.ctypes.data, 300, 77, None,
                                      o.ctypes.data), "bb_iou_host")
    got = o.reshape((300, 77), order="F")
    assert np.array_equal(got, orclib.bb_iou(dt, gt))
    if os.path.exists(orclib.REF_SO):
        assert np.array_equal(got, orclib.ref_bb_iou(dt, gt))
    d_dt, d_gt = torch.from_numpy(dt).cuda(), torch.from_numpy(gt).cuda()
    d_o = torch.empty(300 * 77, dtype=torch.float64, device="cuda")
    _lib.check(lib.taoamd_bb_iou(d_dt.data_ptr(), d_gt.data_ptr(), 300, 77, None,
                                 d_o.data_ptr(), None), "bb_iou")
    torch.cuda.synchronize()
    assert np.array_equal(d_o.cpu().numpy().reshape((300, 77), order="F"), got)


def test_sharded_path_on_one_gpu_equals_plain_path():
    """The multi-GPU code path (records written with a row stride, RCCL
    all_to_all / all_gather, gather_rows, compact + finalize) run with a
    one-rank RCCL group must reproduce the single-process pipeline."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from tao_amodal_amd import dist as tdist, engine
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        gt, dt = synth(seed=21, V=5, F=25, C=50, dets_per_frame=40)
        fl_ = fl.flatten_lvis(gt, dt)
        dt.track_id, _ = fl.make_track_ids_unique(dt)
        ft_ = fl.flatten_tao(gt, dt)
        for flat in (fl_, ft_):
            want = orclib.run_flat(flat, detail=False)
            dp = engine.DeviceProblem(flat, "cuda:0")
            ev = tdist.ShardedEval(dp, engine.Workspace(dp), 0, 1, tdist.HipBackend())
            ev.step()
            ev.step()
            torch.cuda.synchronize()
            assert np.array_equal(ev.num_gt.cpu().numpy(), want["num_gt"])
            assert np.array_equal(ev.precision.cpu().numpy(), want["precision"])
            assert np.array_equal(ev.recall.cpu().numpy(), want["recall"])
            # category-partitioned mode: local stages + in-place all_gather
            cv = tdist.CategoryShardedEval(dp, engine.Workspace(dp), 0, 1,
                                           tdist.HipBackend())
            cv.step()
            cv.step()
            torch.cuda.synchronize()
            assert np.array_equal(cv.precision.cpu().numpy(), want["precision"])
            assert np.array_equal(cv.recall.cpu().numpy(), want["recall"])
            # a rank of a two-rank job: its block of the tables must equal
            # the same block of the whole problem (rows of the other block
            # stay zero until the all_gather fills them)
            K = len(flat.cat_ids)
            for rank in range(2):
                k0, k1, Kb = tdist.category_block(K, rank, 2)
                shard = tdist.shard_by_category(flat, k0, k1)
                sdp = engine.DeviceProblem(shard, "cuda:0")
                half = tdist.CategoryShardedEval(sdp, engine.Workspace(sdp), 0, 1,
                                                 tdist.HipBackend())
                half.k0, half.k1 = k0, k1      # block of `rank`, group of one
                half.compute()
                torch.cuda.synchronize()
                val = exchange_ref.decode_records(half.val.cpu().numpy()[k0:k1])
                ng = want["num_gt"][k0:k1] > 0
                ref = want["precision"][:, :, k0:k1].transpose(2, 3, 0, 1)
                assert np.array_equal(val[ng], ref[ng])
        # the by-video plan's exchange primitive on RCCL (its list form and the
        # asynchronous handle only run with more than one rank otherwise):
        # pieces are views into larger buffers, an empty piece travels too
        src = torch.arange(40, dtype=torch.int64, device="cuda:0").reshape(10, 2, 2)
        dst = torch.zeros_like(src)
        work = tdist.exchange_pieces([dst[3:9]], [src[1:7]])
        side = torch.cuda.Stream("cuda:0")
        with torch.cuda.stream(side):
            work.wait()                       # the side stream waits, the host does not
            got = dst.clone()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(got[3:9], src[1:7]) and int(got[:3].abs().sum()) == 0
        tdist.exchange_pieces([dst[:0]], [src[:0]]).wait()
        plan = tdist.CategoryPlan(engine.DeviceProblem(fl_, "cuda:0"),
                                  engine.DeviceProblem(ft_, "cuda:0"), 0, 1,
                                  torch.device("cuda", 0))
        plan.step()
        plan.step()
        torch.cuda.synchronize()
        for ev, flat in ((plan.lvis, fl_), (plan.tao, ft_)):
            want = orclib.run_flat(flat, detail=False)
            assert np.array_equal(ev.precision.cpu().numpy(), want["precision"])
            assert np.array_equal(ev.recall.cpu().numpy(), want["recall"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,n_cat,quant", [(50000, 7, 50), (200000, 3, 0), (3000, 500, 5),
                                            (70000, 1, 3), (60000, 40, -1), (60000, 40, -2),
                                            (20000, 300, -1), (9000, 2, -2), (30000, 20, -3),
                                            (5000, 1, -3), (100000, 60, -4),
                                            # categories of 6..32 tiles (splitter buckets),
                                            # of more (merge-path passes)
                                            (30000, 1, 0), (60000, 1, 5), (90000, 2, -2),
                                            (45000, 1, -4), (130000, 1, -4), (95000, 1, 2),
                                            (250000, 12, -1)])
def test_both_sorts_are_the_stable_mergesort_order(n, n_cat, quant):
    """Radix sort and tile+merge sort against numpy's stable argsort, with
    heavy score ties and categories far longer than one LDS tile."""
    import torch
    from tao_amodal_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(n + n_cat)
    cat = np.sort(rng.integers(0, n_cat, n)).astype(np.int32)
    score = rng.random(n)
    if quant > 0:
        score = np.round(score * quant) / quant
    elif quant == -1:
        # few elements per high key word, different low words: the run repair
        base = np.round(rng.random(n) * (n // 3)) / (n // 3) * 0.5 + 0.25
        score = base + rng.integers(0, 1 << 20, n) * 2.0 ** -50
    elif quant in (-3, -4):
        # a few long runs of equal high words per tile (low-byte passes on the
        # run only); -4: mixed with ordinary scores and exact ties
        base = np.round(rng.random(n) * 3) / 3 * 0.5 + 0.25
        score = base + rng.integers(0, 1 << 20, n) * 2.0 ** -50
        if quant == -4:
            other = rng.random(n)
            score = np.where(rng.random(n) < 0.5, other, score)
            score[rng.random(n) < 0.1] = 0.75
    elif quant == -2:
        # long runs of equal high words with different low words: full sort
        base = np.round(rng.random(n) * 20) / 20 * 0.5 + 0.25
        score = base + rng.integers(0, 1 << 20, n) * 2.0 ** -50
    score[rng.integers(0, n, 5)] = -0.0
    score[rng.integers(0, n, 5)] = 0.0
    want = np.lexsort((np.arange(n), -score, cat))
    d_cat, d_score = torch.from_numpy(cat).cuda(), torch.from_numpy(score).cuda()
    order = torch.empty(n, dtype=torch.int32, device="cuda")
    dst = torch.empty(n, dtype=torch.int32, device="cuda")
    nb = max(lib.taoamd_sort_workspace(n), lib.taoamd_sort_segments_workspace(n))
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.taoamd_sort_by_cat_score(n, d_cat.data_ptr(), d_score.data_ptr(),
                                            order.data_ptr(), dst.data_ptr(),
                                            ws.data_ptr(), nb, None), "radix")
    torch.cuda.synchronize()
    assert np.array_equal(order.cpu().numpy(), want)
    assert np.array_equal(dst.cpu().numpy()[want], np.arange(n))
    cat_off = np.zeros(n_cat + 1, np.int32)
    np.cumsum(np.bincount(cat, minlength=n_cat), out=cat_off[1:])
    tiles = (np.diff(cat_off) + _lib.SEGMENT_TILE - 1) // _lib.SEGMENT_TILE
    tile_off = np.zeros(n_cat + 1, np.int32)
    np.cumsum(tiles, out=tile_off[1:])
    order.zero_(); dst.zero_()
    d_co, d_to = torch.from_numpy(cat_off).cuda(), torch.from_numpy(tile_off).cuda()
    _lib.check(lib.taoamd_sort_segments(
        n, n_cat, d_co.data_ptr(), d_to.data_ptr(), int(tile_off[-1]),
        int(np.diff(cat_off).max()), d_cat.data_ptr(), d_score.data_ptr(),
        order.data_ptr(), dst.data_ptr(), ws.data_ptr(), nb, None), "segments")
    torch.cuda.synchronize()
    assert np.array_equal(order.cpu().numpy(), want)
    assert np.array_equal(dst.cpu().numpy()[want], np.arange(n))
    # the sample sort (what the evaluator passes call)
    o2, d2 = _sampled_sort(cat_off, tile_off, d_score)
    assert np.array_equal(o2, want)
    assert np.array_equal(d2[want], np.arange(n))


def _sampled_sort(cat_off, tile_off, d_score, repeat=1):
    """taoamd_sort_sampled on device scores; returns order, dst (numpy)."""
    import torch
    from tao_amodal_amd import _lib, engine
    lib = _lib.load()
    n = int(cat_off[-1])
    tabs, (nc, ns, nt, nb), merge = engine.sort_plan(cat_off)
    dev = [torch.from_numpy(t).cuda() for t in tabs]
    d_co = torch.from_numpy(np.ascontiguousarray(cat_off, np.int32)).cuda()
    d_to = torch.from_numpy(np.ascontiguousarray(tile_off, np.int32)).cuda()
    order = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    dst = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    nbytes = lib.taoamd_sort_sampled_workspace(n, nb, int(merge))
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for _ in range(repeat):
        _lib.check(lib.taoamd_sort_sampled(
            n, len(cat_off) - 1, d_co.data_ptr(), d_to.data_ptr(), int(tile_off[-1]),
            int(np.diff(cat_off).max()), d_score.data_ptr(), nc, dev[0].data_ptr(), ns,
            dev[1].data_ptr(), nt, dev[2].data_ptr(), nb, dev[3].data_ptr(),
            order.data_ptr(), dst.data_ptr(), ws.data_ptr(), nbytes, None), "sampled")
    torch.cuda.synchronize()
    return order.cpu().numpy(), dst.cpu().numpy()


@pytest.mark.parametrize("limit", [0, 300, 40])
def test_sample_sort_every_kind_of_category(limit):
    """One launch sequence: empty, tiny (one wavefront, 1..16 elements per
    lane), just around the direct limit, split into buckets, exactly one chunk,
    one element more (two chunks + a merge pass), many chunks.  Scores with
    heavy exact ties, and all-equal categories (the splitters are (key, index)
    pairs).  limit > 0: buckets beyond it take the overflow path (ranking by
    counting) -- with 40 nearly every chunk does."""
    import torch
    from tao_amodal_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(31)
    chunk = 16 * _lib.SEGMENT_TILE
    sizes = [0, 1, 63, 64, 65, 129, 500, 1024, 1025, 1300, 0, 2816, 2817, 9000,
             chunk, chunk + 1, 3 * chunk + 77, 20000, 7]
    if limit:
        sizes = [0, 1, 65, 500, 1025, 1300, 2817, 5000, 7]
    n, n_cat = int(np.sum(sizes)), len(sizes)
    cat = np.repeat(np.arange(n_cat), sizes)
    score = rng.random(n)
    score[rng.random(n) < 0.3] = 0.5
    score[rng.random(n) < 0.2] = np.round(score[rng.random(n) < 0.2][:1], 2)
    score[cat == 9] = 0.125                      # one value for a whole category
    score[cat == n_cat - 2] = np.round(score[cat == n_cat - 2], 1)
    want = np.lexsort((np.arange(n), -score, cat))
    cat_off = np.zeros(n_cat + 1, np.int32)
    np.cumsum(sizes, out=cat_off[1:])
    tiles = (np.diff(cat_off) + _lib.SEGMENT_TILE - 1) // _lib.SEGMENT_TILE
    tile_off = np.zeros(n_cat + 1, np.int32)
    np.cumsum(tiles, out=tile_off[1:])
    d_score = torch.from_numpy(score).cuda()
    try:
        lib.taoamd_sort_sampled_cap_limit(limit)
        order, dst = _sampled_sort(cat_off, tile_off, d_score, repeat=2)
    finally:
        lib.taoamd_sort_sampled_cap_limit(0)
    assert np.array_equal(order, want)
    assert np.array_equal(dst[want], np.arange(n))


from goldenio import MODE_FIXTURES, MODES, load_modes


@pytest.mark.parametrize("name", MODE_FIXTURES)
@pytest.mark.parametrize("mode", list(MODES))
def test_tao_other_modes_hip_vs_oracle_and_reference(name, mode):
    """avg_iou / imagenetvid / use_cats=0 through the class API: equal to the
    C oracle bit for bit, and to the reference's precision/recall."""
    from tao_amodal_amd.evaluation.tao_amodal import Tao, TaoEval, TaoResults
    cfg = MODES[mode]
    dt = DTColumns.from_json(load_inputs(name)[1])
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    gt = Tao(load_inputs(name)[0])
    ev = TaoEval(gt, TaoResults(gt, dt), iou_3d_type=cfg["iou_3d_type"])
    ev.params.use_cats = 1 if cfg["use_cats"] else 0
    ev.run()
    _, p, r, res = load_modes(name)[mode]
    assert np.array_equal(ev.eval["precision"], p)
    assert np.array_equal(ev.eval["recall"], r)
    assert [float(v) for v in ev.results.values()] == res.tolist()
    f = fl.flatten_tao(gt.columns, dt, use_cats=cfg["use_cats"])
    want = orclib.run_flat(f, iou_3d_type=cfg["iou_3d_type"])
    got = _engine().evaluate_flat(f, detail=True, iou_3d_type=cfg["iou_3d_type"])
    assert np.array_equal(got["iou"], want["iou"])
    assert np.array_equal(got["matched"], want["matched"])
    assert np.array_equal(got["precision"], want["precision"])


@pytest.mark.parametrize("name", MODE_FIXTURES)
def test_lvis_without_categories_hip_vs_reference(name):
    """LVISEval with params.use_cats = 0 through the class API: evaluate +
    accumulate equal the reference's golden run (IoUs, every match / ignore
    decision, precision, recall); summarize() fails on the frequency groups
    like the reference's (IndexError); and the HIP results equal the C
    oracle's bit for bit."""
    import nocats_check
    from goldenio import load_lvis_nocats
    from tao_amodal_amd.evaluation.lvis_amodal import LVIS, LVISEval, LVISResults
    gtj, predj = load_inputs(name)
    gt = LVIS(gtj)
    ev = LVISEval(gt, LVISResults(gt, predj), "bbox")
    ev.params.use_cats = 0
    ev.evaluate()
    ev.accumulate()
    cells, eval_imgs, p, r, err = load_lvis_nocats(name)
    assert np.array_equal(ev.eval["precision"], p)
    assert np.array_equal(ev.eval["recall"], r)
    assert ev.eval["counts"] == [10, 101, 1, 6]
    with pytest.raises(IndexError):
        ev.summarize()
    for im, w in cells.items():
        assert np.array_equal(ev.ious[im, -1], w)
    live = [e for e in ev.eval_imgs if e is not None]
    assert len(live) == len(eval_imgs) and len(ev.eval_imgs) == 6 * len(ev.params.img_ids)
    for e, w in zip(live, eval_imgs):
        assert e["image_id"] == w["image_id"] and e["category_id"] == -1
        assert e["dt_ids"] == w["dt_ids"] and e["gt_ids"] == w["gt_ids"]
        assert np.array_equal(e["dt_matches"], np.asarray(w["dt_matches"]).reshape(e["dt_matches"].shape))
        assert np.array_equal(e["dt_ignore"].astype(int), np.asarray(w["dt_ignore"]).reshape(e["dt_ignore"].shape))
    f = fl.flatten_lvis(gt.columns, DTColumns.from_json(predj), use_cats=False)
    got = _engine().evaluate_flat(f, detail=True)
    nocats_check.check(f, got, name)
    _compare_with_oracle(f, got)


def test_lvis_without_categories_synthetic_hip_vs_c_oracle():
    gt, dt = synth(seed=77, V=4, F=12, C=30, dets_per_frame=40, n_present=6)
    f = fl.flatten_lvis(gt, dt, use_cats=False)
    assert f.n_cells <= 48 and len(f.cat_ids) == 1
    _compare_with_oracle(f, _engine().evaluate_flat(f, detail=True))


def test_other_modes_on_merge_kernel():
    gt, dt = synth(seed=41, V=1, F=1100, C=6, dets_per_frame=20,
                   gt_tracks_per_video=24, n_present=2, n_neg=1)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    f = fl.flatten_tao(gt, dt)
    for mode in ("avg_iou", "imagenetvid"):
        want = orclib.run_flat(f, iou_3d_type=mode)
        got = _engine().evaluate_flat(f, iou_3d_type=mode)
        assert np.array_equal(got["iou"], want["iou"]), mode
        assert np.array_equal(got["precision"], want["precision"]), mode


def test_overlapped_streams_give_the_same_tensors():
    import torch
    from tao_amodal_amd import engine
    gt, dt = synth(seed=51, V=6, F=40, C=60, dets_per_frame=40)
    fl_ = fl.flatten_lvis(gt, dt)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    ft_ = fl.flatten_tao(gt, dt)
    dpl, dpt = engine.DeviceProblem(fl_), engine.DeviceProblem(ft_)
    wsl, wst = engine.Workspace(dpl), engine.Workspace(dpt)
    ov = engine.Overlap("cuda")
    for _ in range(3):
        ov.run_pair(dpl, wsl, dpt, wst)
    torch.cuda.synchronize()
    wl, wt = orclib.run_flat(fl_, detail=False), orclib.run_flat(ft_, detail=False)
    assert np.array_equal(wsl.precision.cpu().numpy(), wl["precision"])
    assert np.array_equal(wsl.recall.cpu().numpy(), wl["recall"])
    assert np.array_equal(wst.precision.cpu().numpy(), wt["precision"])
    assert np.array_equal(wst.recall.cpu().numpy(), wt["recall"])


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_exchange_chunks_hip_vs_numpy_restatement(world):
    """taoamd_exchange_{sizes,pack,unpack} against tests/exchange_ref.py: the
    packed chunks byte for byte, the expanded tables against the plain
    single-GPU finalize (no collective needed: every block is packed here)."""
    import torch
    import exchange_ref
    from tao_amodal_amd import dist as tdist, engine
    be = tdist.HipBackend()
    gt, dt = synth(seed=31, V=6, F=20, C=53, dets_per_frame=40, n_present=9)
    fl_ = fl.flatten_lvis(gt, dt)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    ft_ = fl.flatten_tao(gt, dt)
    for flat in (fl_, ft_):
        dp = engine.DeviceProblem(flat, "cuda:0")
        ws = engine.Workspace(dp)
        engine.run(dp, ws)
        K, R = dp.n_cat, dp.n_rng
        val = torch.zeros((K, R, 10, 101), dtype=torch.float64, device="cuda")
        rec = torch.zeros((K, R, 10), dtype=torch.float64, device="cuda")
        be.accumulate_compact(dp.n_dt, K, R, dp.t["cat_off"], ws.matched, ws.ignored,
                              ws.num_gt, 0, K, val, rec, ws.acc_ws, ws.acc_bytes)
        _, _, Kb = tdist.category_block(K, 0, world)
        table = torch.zeros((Kb * world, R), dtype=torch.int32, device="cuda")
        table[:K] = ws.num_gt
        xws = torch.empty(be.exchange_workspace(Kb, R, world), dtype=torch.uint8,
                          device="cuda")
        totals = torch.zeros(world, dtype=torch.int64, device="cuda")
        be.exchange_sizes(Kb, R, world, table, totals, xws)
        want_sizes = exchange_ref.sizes(Kb, R, world, table.cpu().numpy())
        assert np.array_equal(totals.cpu().numpy(), want_sizes)
        cap = int(want_sizes.max())
        assert cap < Kb * R * 1010          # the run-length form is smaller
        cb = be.exchange_chunk_bytes(Kb, R, cap)
        assert cb == exchange_ref.layout(Kb, R, cap)[2]
        chunks = torch.zeros(world * cb, dtype=torch.uint8, device="cuda")
        over = torch.zeros(1, dtype=torch.int32, device="cuda")
        hv, hr, hn = val.cpu().numpy(), rec.cpu().numpy(), ws.num_gt.cpu().numpy()
        for b in range(world):
            be.exchange_pack(K, R, Kb, world, b, ws.num_gt, val, rec,
                             chunks[b * cb:(b + 1) * cb], cap, over, xws)
            want = exchange_ref.pack(K, R, Kb, b, hn, hv, hr, cap)
            got = chunks[b * cb:(b + 1) * cb].cpu().numpy()
            assert np.array_equal(got, want), (flat.kind, world, b)
        prec = torch.empty((10, 101, K, R), dtype=torch.float64, device="cuda")
        rcl = torch.empty((10, K, R), dtype=torch.float64, device="cuda")
        ng = torch.zeros((K, R), dtype=torch.int32, device="cuda")
        be.exchange_unpack(K, R, Kb, world, chunks, cap, ng, prec, rcl, over, xws)
        torch.cuda.synchronize()
        assert int(over.item()) == 0
        assert np.array_equal(ng.cpu().numpy(), hn)
        assert np.array_equal(prec.cpu().numpy(), ws.precision.cpu().numpy())
        assert np.array_equal(rcl.cpu().numpy(), ws.recall.cpu().numpy())
        n2, p2, r2 = exchange_ref.unpack(K, R, Kb, world, chunks.cpu().numpy(), cap,
                                          records=True)
        assert np.array_equal(p2, ws.precision.cpu().numpy())
        assert np.array_equal(r2, ws.recall.cpu().numpy())
        # too small a capacity is reported, not silently wrong (a smaller
        # capacity also means smaller chunks, so only block 0 is still read
        # where it was written: the check needs its levels to overflow)
        if cap > 10 and int(want_sizes[0]) > cap - 10:
            be.exchange_unpack(K, R, Kb, world, chunks, cap - 10, ng, prec, rcl,
                               over, xws)
            torch.cuda.synchronize()
            assert int(over.item()) == 1


def test_fused_and_chunked_sweeps_agree():
    """taoamd_accumulate takes the fused single-launch sweep when the host says
    every category fits one workgroup, the chunked kernels with the
    per-category scans folded into them when every category has at most 32
    chunks (hint 5000: above the fused limit, below 8192), the six-kernel
    chain otherwise (hint 0 = unknown): all must give the oracle's tables,
    also when a category straddles a limit or has ground truth but no
    detection at all."""
    import torch
    from tao_amodal_amd import engine
    for seed, V, F, C, dpf in ((5, 6, 40, 12, 60), (6, 4, 30, 30, 40), (7, 5, 30, 9, 50)):
        gt, dt = synth(seed=seed, V=V, F=F, C=C, dets_per_frame=dpf, n_present=4)
        if seed == 7:       # a category with ground truth and no detection
            victim = np.bincount(gt.ann_cat).argmax()
            dt = dt.take(np.flatnonzero(dt.category_id != victim))
        fl_ = fl.flatten_lvis(gt, dt)
        dt.track_id, _ = fl.make_track_ids_unique(dt)
        ft_ = fl.flatten_tao(gt, dt)
        for flat in (fl_, ft_):
            want = orclib.run_flat(flat, detail=False)
            if seed == 7:
                k = int(np.flatnonzero(flat.cat_ids == victim)[0])
                assert (np.asarray(flat.dt_cat) != k).all() and want["num_gt"][k].max() > 0
            for hint in ("own", 0, 5000, 1 << 30):
                dp = engine.DeviceProblem(flat, "cuda:0")
                if hint != "own":
                    dp.acc_hint = hint
                ws = engine.Workspace(dp)
                ws.precision.fill_(7.0)
                ws.recall.fill_(7.0)
                engine.run(dp, ws)
                torch.cuda.synchronize()
                assert np.array_equal(ws.precision.cpu().numpy(), want["precision"]), (flat.kind, hint)
                assert np.array_equal(ws.recall.cpu().numpy(), want["recall"]), (flat.kind, hint)


@pytest.mark.timeout(900)
def test_config2_full_size_properties():
    """BASELINE.json Config 2 (200 videos x 300 frames x 50 dets, 1203
    categories) at full size: bit-exact against the C oracle, idempotent
    (a second pass over the same workspace gives the same tensors), and
    consistent under the category partition (the two halves evaluated on their
    own reproduce the halves of the whole, block for block)."""
    import torch
    from tao_amodal_amd import dist as tdist, engine
    gt, dt = synth(V=200, F=300, C=1203, dets_per_frame=50)
    fl_ = fl.flatten_lvis(gt, dt)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    ft_ = fl.flatten_tao(gt, dt)
    for flat in (fl_, ft_):
        want = orclib.run_flat(flat, detail=False)
        dp = engine.DeviceProblem(flat, "cuda:0")
        ws = engine.Workspace(dp)
        engine.run(dp, ws)
        torch.cuda.synchronize()
        p1, r1 = ws.precision.clone(), ws.recall.clone()
        assert np.array_equal(p1.cpu().numpy(), want["precision"])
        assert np.array_equal(r1.cpu().numpy(), want["recall"])
        engine.run(dp, ws)
        torch.cuda.synchronize()
        assert torch.equal(ws.precision, p1) and torch.equal(ws.recall, r1)
        K = len(flat.cat_ids)
        for rank in range(2):
            k0, k1, _ = tdist.category_block(K, rank, 2)
            sdp = engine.DeviceProblem(tdist.shard_by_category(flat, k0, k1), "cuda:0")
            sws = engine.Workspace(sdp)
            engine.run(sdp, sws)
            torch.cuda.synchronize()
            assert torch.equal(sws.precision[:, :, k0:k1], p1[:, :, k0:k1])
            assert torch.equal(sws.recall[:, k0:k1], r1[:, k0:k1])


def test_sort_beside_the_match_gives_the_same_tables():
    """Overlap's image-level chain: the match leaves its rows in cell order while
    the sort runs beside it, the first sweep gathers them through order[]
    (taoamd_accumulate_by_order).  Same precision / recall as the chain that
    scatters the rows to their sorted place, and as the oracle."""
    import torch
    from tao_amodal_amd import engine
    gt, dt = synth(seed=21, V=8, F=40, C=60, dets_per_frame=40, n_present=5)
    f = fl.flatten_lvis(gt, dt)
    want = orclib.run_flat(f, detail=False)
    dp = engine.DeviceProblem(f, "cuda:0")
    ws = engine.Workspace(dp)
    aux = torch.cuda.Stream("cuda:0")
    for aside in (True, False, True):
        ws.precision.fill_(7.0)
        engine.run_forked(dp, ws, aux, sort_aside=aside)
        torch.cuda.synchronize()
        assert np.array_equal(ws.precision.cpu().numpy(), want["precision"]), aside
        assert np.array_equal(ws.recall.cpu().numpy(), want["recall"]), aside


def test_segment_sort_mixes_the_three_ways_of_finishing_a_category():
    """One launch sequence, four kinds of category: one tile (final from the
    tile sort), 4 tiles and 11 tiles (splitter buckets), 47 tiles (merge-path
    passes beside the buckets).  Scores with many exact ties."""
    import torch
    from tao_amodal_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    sizes = [500, 130000, 9000, 30000, 0, 2816, 2817]
    cat = np.repeat(np.arange(len(sizes)), sizes).astype(np.int32)
    n, n_cat = len(cat), len(sizes)
    score = rng.random(n)
    score[rng.random(n) < 0.3] = 0.5
    score[rng.random(n) < 0.2] = np.round(score[rng.random(n) < 0.2][:1], 2)
    want = np.lexsort((np.arange(n), -score, cat))
    cat_off = np.zeros(n_cat + 1, np.int32)
    np.cumsum(sizes, out=cat_off[1:])
    tiles = (np.diff(cat_off) + _lib.SEGMENT_TILE - 1) // _lib.SEGMENT_TILE
    tile_off = np.zeros(n_cat + 1, np.int32)
    np.cumsum(tiles, out=tile_off[1:])
    d_cat, d_score = torch.from_numpy(cat).cuda(), torch.from_numpy(score).cuda()
    d_co, d_to = torch.from_numpy(cat_off).cuda(), torch.from_numpy(tile_off).cuda()
    order = torch.zeros(n, dtype=torch.int32, device="cuda")
    dst = torch.zeros(n, dtype=torch.int32, device="cuda")
    nb = lib.taoamd_sort_segments_workspace(n)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        _lib.check(lib.taoamd_sort_segments(
            n, n_cat, d_co.data_ptr(), d_to.data_ptr(), int(tile_off[-1]), max(sizes),
            d_cat.data_ptr(), d_score.data_ptr(), order.data_ptr(), dst.data_ptr(),
            ws.data_ptr(), nb, None), "segments")
    torch.cuda.synchronize()
    assert np.array_equal(order.cpu().numpy(), want)
    assert np.array_equal(dst.cpu().numpy()[want], np.arange(n))


@pytest.mark.parametrize("mode", ["3d_iou", "avg_iou", "imagenetvid"])
@pytest.mark.parametrize("decimal", [False, True])
def test_one_frame_tracks_take_the_single_frame_kernel(mode, decimal):
    """Videos of one frame (the stress shape): every track pair is one box IoU
    or 0 (taoamd_track_iou_single), no frame sums -- equal to the oracle in
    every IoU mode, decimal boxes included, and nothing for the guard to do."""
    from tao_amodal_amd import engine
    from tao_amodal_amd.synth import synth
    gt, dt = synth(seed=8, V=60, F=1, C=15, dets_per_frame=40, n_present=5, decimal=decimal)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    f = fl.flatten_tao(gt, dt)
    dp = engine.DeviceProblem(f, "cuda:0", iou_3d_type=mode)
    assert dp.single_frame and not dp.guard_active()
    got = engine.evaluate_flat(f, "cuda:0", iou_3d_type=mode)
    want = orclib.run_flat(f, iou_3d_type=mode)
    assert got["pairs"] == want["pairs"] > 0
    for k in ("iou", "matched", "ignored", "precision", "recall"):
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("detail", [False, True])
def test_cells_whose_detections_overlap_two_ground_truths(detail):
    """The run kernel's closed form covers cells in which no detection has two
    candidate GTs; the others go through its sequential greedy loop, which --
    image level -- computes the IoUs again from the boxes instead of keeping a
    tile of them.  GTs of one (image, category) cell are made near-duplicates
    of each other so that most detections overlap two or more of them."""
    gt, dt = synth(seed=33, V=4, F=20, C=6, dets_per_frame=30, n_present=3)
    order = np.lexsort((gt.ann_cat, gt.ann_img))
    img, cat = gt.ann_img[order], gt.ann_cat[order]
    box = gt.ann_bbox.copy()
    same = np.flatnonzero((img[1:] == img[:-1]) & (cat[1:] == cat[:-1])) + 1
    assert len(same) > 50
    for k in same:                        # (in order: chains of near-duplicates)
        box[order[k]] = box[order[k - 1]] + np.array([1.0, 2.0, 0.0, 1.0])
    gt.ann_bbox = box
    gt.ann_area = box[:, 2] * box[:, 3]
    # detections: copies of ground-truth boxes of their image and category, jittered
    rng = np.random.default_rng(5)
    key_g = gt.ann_img * 10 ** 6 + gt.ann_cat
    key_d = dt.image_id * 10 ** 6 + dt.category_id
    first = {}
    for j, k in enumerate(key_g.tolist()):
        first.setdefault(k, j)
    src = np.array([first.get(k, -1) for k in key_d.tolist()])
    has = src >= 0
    dbox = dt.bbox.copy()
    dbox[has] = box[src[has]] + rng.integers(-2, 3, (int(has.sum()), 4))
    dbox[:, 2:] = np.maximum(dbox[:, 2:], 1.0)
    dt.bbox = dbox
    f = fl.flatten_lvis(gt, dt)
    want = orclib.run_flat(f)
    # the shape is there: detections with two or more IoUs >= 0.5 inside their cell
    multi, at = 0, 0
    for c in range(f.n_cells):
        D = int(f.cell_dt_off[c + 1] - f.cell_dt_off[c])
        G = int(f.cell_gt_off[c + 1] - f.cell_gt_off[c])
        if D and G > 1:
            m = want["iou"][at:at + D * G].reshape(D, G)
            multi += int(((m >= 0.5).sum(1) >= 2).sum())
        at += D * G
    assert at == len(want["iou"]) and multi > 100
    got = _engine().evaluate_flat(f, detail=detail)
    _compare_with_oracle(f, got, detail=detail)


@pytest.mark.parametrize("world,own", [(1, 0), (3, -1), (3, 1), (8, 7), (8, 0)])
def test_exchange_positions_and_place_vs_numpy(world, own):
    """taoamd_exchange_scores / _positions / _place (the by-video plan's owner
    side, round 5) against the numpy statement the gloo tests run with
    (tests/test_dist_gloo.py: stable -score sort of the sources' concatenation
    in rank order, L/eval.py:353-361): runs of equal scores across and inside
    sources, empty runs, an empty source, the rank's own rows read outside the
    wire buffer (own >= 0) or every source on the wire (own < 0), one and four
    combo words."""
    import torch
    from test_dist_gloo import OracleBackend
    from tao_amodal_amd import dist as tdist
    be, ref = tdist.HipBackend(), OracleBackend({})
    rng = np.random.default_rng(7 * world + own + 2)
    Kb = 9
    for nw in (1, 4):
        rc = rng.integers(0, 40, (world, Kb))
        rc[:, 3] = 0                                  # a category nobody sends
        if world > 2:
            rc[1] = 0                                 # a source without records
        src_base = np.zeros(world + 1, np.int64)
        np.cumsum(rc.sum(1), out=src_base[1:])
        run_off = np.zeros((world, Kb + 1), np.int64)
        np.cumsum(rc, axis=1, out=run_off[:, 1:])
        cat_base = np.zeros(Kb + 1, np.int64)
        np.cumsum(rc.sum(0), out=cat_base[1:])
        n = int(rc.sum())
        # scores: few distinct values (ties across and inside sources), each
        # source's run descending
        scores = np.zeros(n)
        for s in range(world):
            for kb in range(Kb):
                a, b = src_base[s] + run_off[s, kb], src_base[s] + run_off[s, kb + 1]
                scores[a:b] = -np.sort(-rng.integers(0, 6, b - a) / 5.0)
        rows = rng.integers(-2 ** 62, 2 ** 62, (n, nw, 2))
        own_lo, own_hi = (int(src_base[own]), int(src_base[own + 1])) if own >= 0 else (0, 0)
        keep = np.ones(n, bool)
        keep[own_lo:own_hi] = False
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
        h_wire_s, h_own_s = t(scores[keep].view(np.int64)), t(scores[own_lo:own_hi].view(np.int64))
        h_wire_r, h_own_r = t(rows[keep]), t(rows[own_lo:own_hi])
        pad = lambda x, shape: x if x.numel() else torch.zeros(shape, dtype=torch.int64)  # noqa: E731
        args_h = (t(src_base), t(run_off), t(cat_base))
        want_pos = torch.zeros(max(n, 1), dtype=torch.int32)
        ref.positions(n, world, Kb, h_wire_s, h_own_s, own, *args_h, want_pos)
        want_rows = torch.zeros((max(n, 1), nw, 2), dtype=torch.int64)
        ref.place(n, world, nw, h_wire_r, h_own_r, own, args_h[0], want_pos, want_rows)
        assert sorted(want_pos[:n].tolist()) == list(range(n))
        dev = "cuda:0"
        d = [x.to(dev) for x in args_h]
        got_pos = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
        be.positions(n, world, Kb, pad(h_wire_s, (1,)).to(dev), pad(h_own_s, (1,)).to(dev), own,
                     d[0], d[1], d[2], got_pos)
        got_rows = torch.zeros((max(n, 1), nw, 2), dtype=torch.int64, device=dev)
        be.place(n, world, nw, pad(h_wire_r, (1, nw, 2)).to(dev), pad(h_own_r, (1, nw, 2)).to(dev),
                 own, d[0], got_pos, got_rows)
        torch.cuda.synchronize()
        assert np.array_equal(got_pos.cpu().numpy()[:n], want_pos.numpy()[:n]), (world, own, nw)
        assert np.array_equal(got_rows.cpu().numpy()[:n], want_rows.numpy()[:n]), (world, own, nw)
    # the scores at their sorted place
    n = 1000
    dst = torch.from_numpy(rng.permutation(n).astype(np.int32)).to("cuda:0")
    sc = torch.from_numpy(rng.random(n)).to("cuda:0")
    out = torch.zeros(n, dtype=torch.int64, device="cuda:0")
    from tao_amodal_amd import _lib
    _lib.check(_lib.load().taoamd_exchange_scores(
        n, dst.data_ptr(), sc.data_ptr(), out.data_ptr(),
        torch.cuda.current_stream().cuda_stream), "taoamd_exchange_scores")
    torch.cuda.synchronize()
    want = np.zeros(n, np.int64)
    want[dst.cpu().numpy()] = sc.cpu().numpy().view(np.int64)
    assert np.array_equal(out.cpu().numpy(), want)
