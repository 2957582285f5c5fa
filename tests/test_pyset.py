"""csrc/pyset.hpp -- CPython's set restated for the frame-order guard -- pinned
against the interpreter's own sets: the iteration order of
``set(a) | set(b)`` (the order the reference adds a track pair's frames in,
T/eval.py:83) and, on top of it, one pair's 3D / average IoU against the
statement the reference makes with Python's sets and numpy's mean.

Host entry points of the HIP library (taoamd_*_host): no GPU needed."""
import ctypes as C

import numpy as np
import pytest

from tao_amodal_amd import _lib


def union_order(a, b):
    lib = _lib.load()
    a = np.ascontiguousarray(a, dtype=np.int64)
    b = np.ascontiguousarray(b, dtype=np.int64)
    out = np.zeros(len(a) + len(b) + 1, dtype=np.int64)
    n = C.c_int64(0)
    _lib.check(lib.taoamd_pyset_union_order_host(
        len(a), a.ctypes.data, len(b), b.ctypes.data, out.ctypes.data,
        C.addressof(n)), "taoamd_pyset_union_order_host")
    return out[:n.value].tolist()


def py_order(a, b):
    # dict keys, as the reference builds its per-track maps (T/eval.py:322-325)
    ka = {int(k): 0 for k in a}.keys()
    kb = {int(k): 0 for k in b}.keys()
    return list(set(ka) | set(kb))


def test_every_size_up_to_the_third_resize():
    """All (|a|, |b|) up to 90 x 90 with consecutive and scattered ids: every
    growth step of the three tables and the equal-size slot copy."""
    rng = np.random.default_rng(1)
    for na in list(range(0, 24)) + [31, 32, 33, 76, 77, 78, 90]:
        for nb in list(range(0, 24)) + [31, 32, 33, 76, 77, 78, 90]:
            if na + nb == 0:
                continue
            base = int(rng.integers(0, 5000))
            a = base + np.arange(na)
            b = base + int(rng.integers(0, max(na, 1) + 3)) + np.arange(nb)
            assert union_order(a, b) == py_order(a, b), (na, nb, "runs")
            pool = rng.permutation(4 * (na + nb) + 8) + base
            a, b = pool[:na], rng.permutation(pool[: na + nb])[:nb]
            assert union_order(a, b) == py_order(a, b), (na, nb, "scattered")


@pytest.mark.parametrize("seed", range(8))
def test_long_tracks_like_the_synthetic_sets(seed):
    """Hundreds of frames: shuffled image ids, contiguous ids, ids that collide
    modulo the table size, partial overlap, holes."""
    rng = np.random.default_rng(100 + seed)
    for _ in range(40):
        n = int(rng.integers(1, 1300))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            ids = int(rng.integers(0, 1 << 20)) + np.arange(2 * n)
        elif kind == 1:
            ids = rng.permutation(int(rng.integers(2 * n, 40 * n + 1)))[: 2 * n] + 1
        elif kind == 2:
            ids = (np.arange(2 * n) * int(2 ** rng.integers(3, 12))) + int(rng.integers(0, 99))
        else:
            ids = rng.integers(0, 1 << 40, 2 * n)
            ids = np.unique(ids)
            ids = rng.permutation(ids)
        m = len(ids)
        lo, hi = sorted(rng.integers(0, m + 1, 2).tolist())
        a = ids[lo:hi][rng.random(hi - lo) < rng.uniform(0.3, 1.0)]
        lo, hi = sorted(rng.integers(0, m + 1, 2).tolist())
        b = ids[lo:hi][rng.random(hi - lo) < rng.uniform(0.3, 1.0)]
        if len(a) + len(b) == 0:
            continue
        assert union_order(a, b) == py_order(a, b), (seed, kind, len(a), len(b))


def test_large_sets_cross_the_50000_rule():
    rng = np.random.default_rng(7)
    ids = rng.permutation(400000)[:130000]
    a, b = ids[:70000], ids[40000:]
    assert union_order(a, b) == py_order(a, b)


def ref_pair(tl_id, dpos, dbox, gpos, gbox, mode):
    """compute_track_box_iou / compute_avg_track_iou (T/eval.py:73-117) in the
    reference's words, with the interpreter's sets."""
    dt = {int(tl_id[p]): b for p, b in zip(dpos.tolist(), dbox.tolist())}
    gt = {int(tl_id[p]): b for p, b in zip(gpos.tolist(), gbox.tolist())}
    i = u = 0
    ious = []
    for image in set(gt.keys()) | set(dt.keys()):
        g, d = gt.get(image), dt.get(image)
        if d and g:
            w = max(min(d[0] + d[2], g[0] + g[2]) - max(d[0], g[0]), 0)
            h = max(min(d[1] + d[3], g[1] + g[3]) - max(d[1], g[1]), 0)
            i_ = w * h
            u_ = d[2] * d[3] + g[2] * g[3] - i_
            i += i_
            u += u_
            ious.append(i_ / u_ if u_ > 0 else 0)
        elif g:
            u += g[2] * g[3]
            ious.append(0)
        elif d:
            u += d[2] * d[3]
            ious.append(0)
    return (i / u if u > 0 else 0) if mode == 0 else float(np.mean(ious))


def host_pair(tl_id, dpos, dbox, gpos, gbox, mode):
    lib = _lib.load()
    tl_id = np.ascontiguousarray(tl_id, dtype=np.int64)
    dpos = np.ascontiguousarray(dpos, dtype=np.int32)
    gpos = np.ascontiguousarray(gpos, dtype=np.int32)
    dbox, gbox = np.ascontiguousarray(dbox), np.ascontiguousarray(gbox)
    out = C.c_double(0)
    _lib.check(lib.taoamd_set_order_iou_host(
        tl_id.ctypes.data, len(dpos), dpos.ctypes.data, dbox.ctypes.data, len(gpos),
        gpos.ctypes.data, gbox.ctypes.data, mode, C.addressof(out)),
        "taoamd_set_order_iou_host")
    return out.value


@pytest.mark.parametrize("mode", [0, 1])
def test_pair_iou_in_set_order_equals_the_reference_statement(mode):
    rng = np.random.default_rng(42 + mode)
    for trial in range(120):
        F = int(rng.integers(1, 900))
        tl_id = rng.permutation(3 * F)[:F] + 1 if trial % 2 else 1000 + np.arange(F)

        def track():
            lo, hi = sorted(rng.integers(0, F + 1, 2).tolist())
            if lo == hi:
                lo, hi = 0, F
            pos = np.arange(lo, hi)[rng.random(hi - lo) < rng.uniform(0.5, 1.0)]
            if len(pos) == 0:
                pos = np.array([lo])
            box = np.c_[rng.uniform(-50, 900, (len(pos), 2)), rng.uniform(1, 400, (len(pos), 2))]
            return pos, np.round(box, int(rng.integers(0, 4)))
        dpos, dbox = track()
        gpos, gbox = track()
        if trial % 3 == 0:       # overlapping boxes on the common frames
            common = np.intersect1d(dpos, gpos)
            dbox[np.searchsorted(dpos, common)] = \
                gbox[np.searchsorted(gpos, common)] + np.round(rng.uniform(-5, 5, (len(common), 4)), 2)
            dbox[:, 2:] = np.maximum(dbox[:, 2:], 0.5)
        want = ref_pair(tl_id, dpos, dbox, gpos, gbox, mode)
        got = host_pair(tl_id, dpos, dbox, gpos, gbox, mode)
        assert got == want, (trial, F, len(dpos), len(gpos))
