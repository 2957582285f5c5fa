"""Every way the sweep of long categories can run (taoamd_accumulate_sweep_mode:
the chunked kernels, the one-pass sweep with the decoupled look-back, the
one-pass sweep behind a counting pass) against the reference goldens and the C
oracle -- the small fixtures are driven through the one-pass kernels by the
explicit mode --, category lengths on every boundary of the kernels' blocking,
both row layouts, one and four combo words; and the recovery of a pass whose
look-back gave up (fault injection through taoamd_accumulate_spin_limit).
Reference: lvis_amodal/eval.py:339-426, tao_amodal/eval.py:496-584."""
import ctypes as C

import numpy as np
import pytest

import orclib
from goldenio import FIXTURES
from tao_amodal_amd import _lib
from tao_amodal_amd import flatten as fl
from tao_amodal_amd.synth import synth

pytestmark = pytest.mark.gpu

N_THR, N_REC = _lib.N_THR, _lib.N_REC
MODES = ["chunked", "lookback", "twopass"]


@pytest.fixture(params=MODES)
def sweep_mode(request):
    _lib.sweep_mode(request.param)
    yield request.param
    _lib.sweep_mode("auto")


@pytest.mark.parametrize("name", FIXTURES)
def test_reference_goldens_under_every_sweep_mode(sweep_mode, name):
    import test_gpu_parity as tp
    tp.test_lvis_hip_matches_reference_golden(name)
    tp.test_tao_hip_matches_reference_golden(name)


def test_fused_chunked_and_onepass_sweeps_agree(sweep_mode):
    import test_gpu_parity as tp
    tp.test_fused_and_chunked_sweeps_agree()


@pytest.mark.parametrize("name", ["f1", "f2", "f4"])
def test_cli_text_and_class_state_under_every_sweep_mode(sweep_mode, name, tmp_path):
    """The drop-in CLI (both levels on two threads) and the class API's state
    -- eval_imgs / ious views, dt_pointers -- with the sweep forced onto each
    path: the reference's text byte for byte, its tensors and pointers."""
    import test_gpu_cli as tc
    tc.test_cli_text_is_identical_to_the_reference(name, tmp_path)
    if name in ("f1", "f2"):
        tc.test_class_api_state_matches_reference(name)
        tc.test_tao_class_api_state_matches_reference(name)


@pytest.mark.parametrize("case", ["many", "unsorted_rec"])
@pytest.mark.parametrize("name", ["f1", "f4"])
def test_edited_constants_under_every_sweep_mode(sweep_mode, name, case):
    import test_gpu_constants as tc
    assert case in tc.cases()
    tc.test_edited_constants_match_the_reference(name, case)


# ---------------------------------------------------------------------------
# rows of every shape, straight into taoamd_accumulate
# ---------------------------------------------------------------------------
SC = 2048      # rows of a super-chunk of the one-pass sweep (4 wavefronts x 512)
SIZES = [0, 1, 63, 64, 65, 511, 512, 513, SC - 1, SC, SC + 1, 0, 2 * SC, 2 * SC + 1,
         3 * SC - 1, 256, 257, 5 * SC + 700, 1024, 4 * SC - 1, 7]


def _rows(seed, sizes, n_rng):
    """Random TP / FP / ignored bits with a precision that falls along the rank
    in some categories and stays flat in others, ground-truth counts from 0 (no
    evaluated ground truth: -1 rows) to more than the TPs (thresholds never
    reached), and a category with ground truth and no row at all."""
    rng = np.random.default_rng(seed)
    K, nw = len(sizes), (n_rng * N_THR + 63) // 64
    n = int(sum(sizes))
    cat_off = np.zeros(K + 1, np.int32)
    np.cumsum(sizes, out=cat_off[1:])
    matched = np.zeros((n, nw), np.uint64)
    ignored = np.zeros((n, nw), np.uint64)
    num_gt = np.zeros((K, n_rng), np.int32)
    for k, sz in enumerate(sizes):
        rank = np.arange(sz) / max(sz, 1)
        for r in range(n_rng):
            kind = (k + r) % 4
            if kind == 3:
                num_gt[k, r] = 0
            for t in range(N_THR):
                c = r * N_THR + t
                p = {0: 0.7 - 0.5 * rank, 1: np.full(sz, 0.3), 2: 0.05 + 0.9 * (rank > 0.5),
                     3: np.full(sz, 0.5)}[kind] * (1 - 0.05 * t)
                m = rng.random(sz) < p
                ig = rng.random(sz) < (0.1 if (k + t) % 3 else 0.0)
                bit = np.uint64(1) << np.uint64(c % 64)
                matched[cat_off[k]:cat_off[k + 1], c // 64] |= np.where(m, bit, np.uint64(0))
                ignored[cat_off[k]:cat_off[k + 1], c // 64] |= np.where(ig, bit, np.uint64(0))
            if kind != 3:
                tp = int((matched[cat_off[k]:cat_off[k + 1], (r * N_THR) // 64] >>
                          np.uint64((r * N_THR) % 64) & np.uint64(1)).sum())
                num_gt[k, r] = max(1, int(tp * (0.5 + rng.random())) + int(rng.integers(0, 3)))
    return cat_off, matched, ignored, num_gt


def _oracle_tables(cat_off, matched, ignored, num_gt):
    K, n_rng = num_gt.shape
    n = int(cat_off[-1])
    cat = np.repeat(np.arange(K, dtype=np.int32), np.diff(cat_off))
    score = -np.arange(n, dtype=np.float64)          # the rows are in sorted order
    gcat, grng = [], []
    for k in range(K):
        for r in range(n_rng):
            gcat += [k] * int(num_gt[k, r])
            grng += [(~(1 << r)) & 0xffffffff] * int(num_gt[k, r])
    gcat, grng = np.asarray(gcat, np.int32), np.asarray(grng, np.uint32)
    prec = np.zeros((N_THR, N_REC, K, n_rng))
    rec = np.zeros((N_THR, K, n_rng))
    m = np.ascontiguousarray(matched if n else np.zeros((1, matched.shape[1]), np.uint64))
    i = np.ascontiguousarray(ignored if n else np.zeros((1, matched.shape[1]), np.uint64))
    p = orclib._p
    orclib.lib().orc_accumulate(C.c_int64(n), C.c_int32(K), C.c_int(n_rng), p(cat), p(score),
                                p(m), p(i), C.c_int64(len(gcat)), p(gcat), p(grng), p(prec),
                                p(rec), None, None)
    return prec, rec


def _device_tables(cat_off, matched, ignored, num_gt, layout, hint, prepared=False):
    import torch
    lib = _lib.load()
    dev = "cuda:0"
    K, n_rng = num_gt.shape
    n, nw = matched.shape
    d_off = torch.from_numpy(cat_off).to(dev)
    d_ng = torch.from_numpy(num_gt).to(dev)
    if layout == "paired":
        rows = torch.empty((max(n, 1), nw, 2), dtype=torch.int64, device=dev)
        rows[:n, :, 0] = torch.from_numpy(matched.view(np.int64)).to(dev)
        rows[:n, :, 1] = torch.from_numpy(ignored.view(np.int64)).to(dev)
        d_m, d_i = rows[..., 0], rows[..., 1]
    else:
        d_m = torch.from_numpy(np.ascontiguousarray(matched).view(np.int64)).to(dev)
        d_i = torch.from_numpy(np.ascontiguousarray(ignored).view(np.int64)).to(dev)
        if n == 0:
            d_m = torch.zeros((1, nw), dtype=torch.int64, device=dev)
            d_i = torch.zeros((1, nw), dtype=torch.int64, device=dev)
    nbytes = lib.taoamd_accumulate_workspace(n, K, n_rng)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    prec = torch.full((N_THR, N_REC, K, n_rng), 7.0, dtype=torch.float64, device=dev)
    rec = torch.full((N_THR, K, n_rng), 7.0, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    args = (n, K, n_rng, d_off.data_ptr(), d_m.data_ptr(), d_i.data_ptr(), d_ng.data_ptr(),
            hint, prec.data_ptr(), rec.data_ptr(), ws.data_ptr(), nbytes, s)
    if prepared:
        _lib.check(lib.taoamd_accumulate_prepare(n, K, n_rng, d_off.data_ptr(), hint,
                                                 ws.data_ptr(), nbytes, s), "prepare")
        for _ in range(2):          # a prepared plan serves pass after pass
            _lib.check(lib.taoamd_accumulate_prepared(*args), "prepared")
    else:
        _lib.check(lib.taoamd_accumulate(*args), "taoamd_accumulate")
    flag = C.c_int32(0)
    _lib.check(lib.taoamd_accumulate_error(ws.data_ptr(), s, C.addressof(flag)), "error")
    return prec.cpu().numpy(), rec.cpu().numpy(), flag.value


@pytest.mark.parametrize("layout", ["paired", "split"])
@pytest.mark.parametrize("n_rng", [6, 20])
def test_categories_on_every_boundary_of_the_blocking(sweep_mode, n_rng, layout):
    """Category lengths of 0, 1, k * 64 +- 1, k * 512 +- 1 and k * 2048 +- 1
    rows (a super-chunk = 4 wavefronts x 512 rows), a category that ends
    exactly on a super-chunk boundary followed by an empty one, one combo word
    (image level) and four (track level: n_words = 4 through the look-back)."""
    cat_off, m, i, ng = _rows(5 + n_rng, SIZES, n_rng)
    want_p, want_r = _oracle_tables(cat_off, m, i, ng)
    for hint in (0, int(max(SIZES))):
        for prepared in (False, True):
            got_p, got_r, flag = _device_tables(cat_off, m, i, ng, layout, hint, prepared)
            assert flag == 0
            assert np.array_equal(got_r, want_r), (sweep_mode, n_rng, layout, hint, prepared)
            assert np.array_equal(got_p, want_p), (sweep_mode, n_rng, layout, hint, prepared)


def test_long_categories_through_the_look_back(sweep_mode):
    """Categories of 40 and 70 super-chunks (the raise kernel's 64-SC rounds)
    beside short ones."""
    sizes = [40 * SC + 17, 3, 70 * SC, 0, SC]
    cat_off, m, i, ng = _rows(11, sizes, 6)
    want_p, want_r = _oracle_tables(cat_off, m, i, ng)
    got_p, got_r, flag = _device_tables(cat_off, m, i, ng, "paired", 0)
    assert flag == 0
    assert np.array_equal(got_r, want_r) and np.array_equal(got_p, want_p)


# ---------------------------------------------------------------------------
# a look-back that gives up: the pass is swept again with the chunked kernels
# ---------------------------------------------------------------------------
@pytest.fixture
def failing_look_back():
    _lib.sweep_mode("lookback", spin_limit=-1)
    yield
    _lib.sweep_mode("auto", spin_limit=0)


def test_the_flag_is_raised_and_the_chunked_entry_point_recovers(failing_look_back):
    import torch
    lib = _lib.load()
    cat_off, m, i, ng = _rows(3, SIZES, 6)
    want_p, want_r = _oracle_tables(cat_off, m, i, ng)
    # a caller-owned counter of the waits that gave up: not cleared by a pass
    counter = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    _lib.check(lib.taoamd_accumulate_giveup_counter(counter.data_ptr()), "giveup_counter")
    try:
        got_p, got_r, flag = _device_tables(cat_off, m, i, ng, "paired", 0)
        first = int(counter.item())
        _device_tables(cat_off, m, i, ng, "paired", 0)
        assert first > 0 and int(counter.item()) == 2 * first
    finally:
        _lib.check(lib.taoamd_accumulate_giveup_counter(None), "giveup_counter")
    _device_tables(cat_off, m, i, ng, "paired", 0)
    assert int(counter.item()) == 2 * first          # (deregistered)
    assert flag == 1                      # every look-back gave up at once
    assert not np.array_equal(got_p, want_p)
    # the same rows through taoamd_accumulate_chunked
    dev = "cuda:0"
    K, n_rng = ng.shape
    n, nw = m.shape
    rows = torch.empty((n, nw, 2), dtype=torch.int64, device=dev)
    rows[:, :, 0] = torch.from_numpy(m.view(np.int64)).to(dev)
    rows[:, :, 1] = torch.from_numpy(i.view(np.int64)).to(dev)
    d_off, d_ng = torch.from_numpy(cat_off).to(dev), torch.from_numpy(ng).to(dev)
    nbytes = lib.taoamd_accumulate_workspace(n, K, n_rng)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    prec = torch.empty((N_THR, N_REC, K, n_rng), dtype=torch.float64, device=dev)
    rec = torch.empty((N_THR, K, n_rng), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.taoamd_accumulate_chunked(
        n, K, n_rng, d_off.data_ptr(), rows[..., 0].data_ptr(), rows[..., 1].data_ptr(),
        d_ng.data_ptr(), 0, prec.data_ptr(), rec.data_ptr(), ws.data_ptr(), nbytes, s),
        "taoamd_accumulate_chunked")
    flag = C.c_int32(7)
    _lib.check(lib.taoamd_accumulate_error(ws.data_ptr(), s, C.addressof(flag)), "error")
    assert flag.value == 0
    assert np.array_equal(prec.cpu().numpy(), want_p) and np.array_equal(rec.cpu().numpy(), want_r)


def _long_category_problem():
    gt, dt = synth(seed=9, V=96, F=12, C=5, dets_per_frame=60, n_present=2)
    f_l = fl.flatten_lvis(gt, dt)
    dt.track_id, _ = fl.make_track_ids_unique(dt)
    f_t = fl.flatten_tao(gt, dt)
    for f in (f_l, f_t):       # several super-chunks per category at both levels
        assert int(np.bincount(np.asarray(f.dt_cat)).max()) > SC
    return gt, dt, f_l, f_t


def test_a_timed_out_pass_is_swept_again_by_the_engine(failing_look_back, caplog):
    """engine.sweep_ok (reached through guarded_pairs at the end of every pass
    the class API and evaluate_flat run): flag seen, chunked kernels, plan
    rebuilt; the tables are the oracle's, and so are those of the next pass."""
    import torch
    from tao_amodal_amd import engine
    _, _, f_l, f_t = _long_category_problem()
    for flat in (f_l, f_t):
        want = orclib.run_flat(flat, detail=False)
        dp = engine.DeviceProblem(flat, "cuda:0")
        ws = engine.Workspace(dp)
        for rep in range(2):
            ws.precision.fill_(7.0)
            engine.run_guarded(dp, ws, flat)
            assert ws.sweep_recovered == rep + 1
            assert np.array_equal(ws.precision.cpu().numpy(), want["precision"])
            assert np.array_equal(ws.recall.cpu().numpy(), want["recall"])
        # with the look-back working again the rebuilt plan serves the pass
        _lib.sweep_mode("lookback", spin_limit=0)
        ws.precision.fill_(7.0)
        engine.run_guarded(dp, ws, flat)
        assert ws.sweep_recovered == 2
        assert np.array_equal(ws.precision.cpu().numpy(), want["precision"])
        _lib.sweep_mode("lookback", spin_limit=-1)
    assert any("swept again" in r.getMessage() for r in caplog.records)


def test_a_timed_out_pass_is_swept_again_by_both_multi_gpu_plans(failing_look_back):
    """dist.ShardedEval.check / CategoryShardedEval.check read the flag of the
    workspace that was swept and repeat sweep + result exchange."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from tao_amodal_amd import dist as tdist, engine
    _, _, f_l, f_t = _long_category_problem()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        for flat in (f_l, f_t):
            want = orclib.run_flat(flat, detail=False)
            for cls in (tdist.ShardedEval, tdist.CategoryShardedEval):
                dp = engine.DeviceProblem(flat, "cuda:0")
                ws = engine.Workspace(dp)
                ev = cls(dp, ws, 0, 1, tdist.HipBackend())
                ev.step()
                torch.cuda.synchronize()
                ev.check()
                assert ev.sweep_recovered == 1, cls.__name__
                assert np.array_equal(ev.precision.cpu().numpy(), want["precision"]), cls.__name__
                assert np.array_equal(ev.recall.cpu().numpy(), want["recall"]), cls.__name__
    finally:
        if created:
            dist.destroy_process_group()
