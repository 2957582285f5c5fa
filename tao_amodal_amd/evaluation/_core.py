"""Shared machinery of the two evaluator classes: running a flattened problem
on the GPU, the numpy-side summaries (kept in numpy so that the means are
computed by the very same pairwise summation as the reference), and the lazy
views that expose the reference's bulky per-cell state (``ious``,
``eval_imgs`` / ``eval_vids``, ``dt_pointers``) without materialising it.
"""
import datetime
from collections.abc import Mapping, Sequence

import numpy as np

N_THR, N_REC = 10, 101


def masked_mean(s):
    """``np.mean(s[s > -1])`` or -1 (reference lvis_amodal/eval.py:453-457,
    tao_amodal/eval.py:619-623)."""
    sel = s[s > -1]
    if len(sel) == 0:
        return -1
    return np.mean(sel)


def summaries(jobs):
    """``[(key, thunk), ...]`` -> yields ``(key, thunk())`` in the order given:
    the masked means of a summary are independent of each other and numpy's
    slicing, comparing and adding run without the interpreter lock, so they go
    to a few threads (46 means over 58 + 194 MB of tables: 0.08 s one after the
    other).  The caller fills its result dict as the pairs arrive -- the
    reference's order of insertion, a key written twice keeping the LAST value
    (lvis_amodal/eval.py:497-499) -- and a thunk that raises (edited params:
    fewer ranges than the labels name) raises HERE, at its place in the order,
    behind the results before it, like the reference's loop."""
    if len(jobs) < 4:
        for k, f in jobs:
            yield k, f()
        return
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=4)
    try:
        futures = [pool.submit(f) for _, f in jobs]
        for (k, _), fu in zip(jobs, futures):
            yield k, fu.result()
    finally:
        pool.shutdown(wait=True, cancel_futures=True)


TIMING = {}   # wall-clock seconds per stage, filled when TAOAMD_TIMING is set


import threading as _threading
_TIMING_LOCK = _threading.Lock()


def timed(name):
    """Context manager adding the elapsed wall-clock to TIMING[name]."""
    import contextlib
    import os
    import time

    @contextlib.contextmanager
    def cm():
        if not os.environ.get("TAOAMD_TIMING"):
            yield
            return
        t0 = time.perf_counter()
        try:
            yield
        finally:
            with _TIMING_LOCK:      # (the CLI's two levels time the same keys)
                TIMING[name] = TIMING.get(name, 0.0) + time.perf_counter() - t0
    return cm()


class GpuRun:
    """One evaluator pass on the device, split like the reference's API:
    evaluate() = ranges + sort + [track IoU] + match, accumulate() = sweep."""

    def __init__(self, flat, device=None, iou_3d_type="3d_iou", constants=None):
        """``constants`` (EvalConstants): edited ``params`` thresholds / ranges;
        None = the reference's defaults."""
        import torch
        from .. import engine
        self.constants = constants if constants is not None and not constants.default \
            else None
        if not torch.cuda.is_available():
            raise RuntimeError(
                "tao_amodal_amd evaluates on an AMD GPU through its HIP "
                "extension; no GPU is visible and there is no CPU fallback.")
        self.engine = engine
        self.torch = torch
        self.flat = flat
        self.iou_3d_type = iou_3d_type
        self.device = torch.device(device or "cuda")
        with timed("upload+plan"):
            self.dp = engine.DeviceProblem(flat, self.device, iou_3d_type)
            self.ws = engine.Workspace(self.dp)
            torch.cuda.synchronize(self.device)
        self._detail = None
        self._head_done = False      # sort + IoU matrix of this problem are in the workspace
        self.near_threshold_pairs = 0
        self.precision = self.recall = None

    def evaluate(self):
        e = self.engine
        c = self.constants
        if c is not None and not c.single:
            return         # several blocks of thresholds: every pass in accumulate()
        with timed("kernels"), applied(c):
            # (frame-order guard of the 3D IoU applied before the match, on
            # the device: engine.stage_iou_guard; its count is read later)
            e.run_guarded(self.dp, self.ws, self.flat, upto="match",
                          read_count=False)
            self._head_done = True
            import os
            if os.environ.get("TAOAMD_TIMING"):
                self.torch.cuda.synchronize(self.device)

    def accumulate(self):
        c = self.constants
        if c is not None:
            return self._accumulate_blocks(c)
        with timed("kernels"):
            self.engine.stage_accumulate(self.dp, self.ws)
            self.torch.cuda.synchronize(self.device)
            self.near_threshold_pairs = self.engine.guarded_pairs(self.dp, self.ws)
        with timed("download"):
            self.precision = self.ws.precision.cpu().numpy()
            self.recall = self.ws.recall.cpu().numpy()

    def _accumulate_blocks(self, c):
        """Edited constants: the kernels take N_THR IoU thresholds, N_REC
        recall thresholds and their own number of ranges per pass -- the
        caller's arrays are cut into such blocks (EvalConstants) and every
        block's slice of the tables is put where the caller's order has it.
        IoU thresholds are independent of each other (L/eval.py:234-277), recall
        thresholds (:406-410) and ranges (:140-145, T/eval.py:271-276) too."""
        e, dp, ws = self.engine, self.dp, self.ws
        K = dp.n_cat
        self.precision = np.empty((c.T, c.R, K, c.n_rng))
        self.recall = np.empty((c.T, K, c.n_rng))
        first = True
        for bi, (t_idx, t_val) in enumerate(c.thr_blocks):
            for bk, (_, slots) in enumerate(c.rng_blocks):
                ks = np.array([k for k, _ in slots], dtype=np.int64)
                out = np.array([i for _, i in slots], dtype=np.int64)
                for bj, (r_idx, r_val) in enumerate(c.rec_blocks):
                    with timed("kernels"), applied(c, bi, bj, bk):
                        if bj == 0 and not (c.single and first):
                            # (the sort and the IoU matrix once: they depend on
                            # neither thresholds nor ranges -- ADVICE r5)
                            e.run_guarded(dp, ws, self.flat, upto="match", read_count=False,
                                          head=not self._head_done)
                            self._head_done = True
                        first = False
                        e.stage_accumulate(dp, ws)
                        self.torch.cuda.synchronize(self.device)
                        # (per block, under the block's constants: an unprepared
                        # pass clears the flag of the pass before it)
                        e.sweep_ok(dp, ws)
                    with timed("download"):
                        p = ws.precision[:len(t_idx), :len(r_idx)].cpu().numpy()
                        self.precision[np.ix_(t_idx, r_idx, np.arange(K), out)] = p[..., ks]
                        if bj == 0:
                            rc = ws.recall[:len(t_idx)].cpu().numpy()
                            self.recall[np.ix_(t_idx, np.arange(K), out)] = rc[..., ks]
        self.near_threshold_pairs = e.guarded_pairs(dp, ws)
        if not c.rec_sorted:
            # the reference fills a row's recall thresholds IN THE CALLER'S
            # ORDER and stops at the first one the category never reaches (the
            # bare `except` of L/eval.py:406-410): what follows stays 0
            never = np.maximum.accumulate(self.precision == 0, axis=1)
            self.precision[never] = 0

    # ------------------------------------------------------ lazy detail
    def detail(self):
        """Per-detection match indices / IoUs (a second, detail-mode pass,
        only when the per-cell views are actually inspected)."""
        if self._detail is None:
            c = self.constants
            if c is not None and not c.single:
                raise NotImplementedError(
                    "the per-cell views (ious, eval_imgs / eval_vids, dt_pointers) "
                    "are kept for up to %d IoU thresholds and the reference's "
                    "number of ranges" % N_THR)
            with applied(c):
                self._detail = self.engine.evaluate_flat(
                    self.flat, self.device, detail=True,
                    iou_3d_type=self.iou_3d_type)
        return self._detail

    def thr_slots(self):
        """Place of the caller's i-th IoU threshold among the kernels' combos."""
        c = self.constants
        if c is None:
            return list(range(N_THR))
        if not c.single:
            self.detail()        # raises
        slot = np.empty(c.T, dtype=np.int64)
        slot[c.thr_blocks[0][0]] = np.arange(c.T)
        return slot.tolist()

    def pointer_tables(self):
        """What eval['dt_pointers'] is read from: the detections' ids and
        (matched, ignored) words in the sweep's order (category-major, stable
        descending score), num_gt and the categories' row offsets."""
        n = self.dp.n_dt
        ws = self.ws
        order = ws.order[:n].cpu().numpy().astype(np.int64)
        return {"ids": np.asarray(self.flat.dt_id)[order],
                "matched": ws.matched[:n].cpu().numpy().view(np.uint64),
                "ignored": ws.ignored[:n].cpu().numpy().view(np.uint64),
                "num_gt": ws.num_gt.cpu().numpy(),
                "cat_off": np.asarray(self.dp.cat_off_host, dtype=np.int64)}


def _bit(words, combo):
    return ((words[..., combo // 64] >> np.uint64(combo % 64))
            & np.uint64(1)).astype(bool)


def restrict_to_params(gt, dt, unit, unit_ids, cat_ids, use_cats):
    """The reference's ``_prepare`` (lvis_amodal/eval.py:59-105,
    tao_amodal/eval.py:178-233) for edited ``params``: the ground truth and the
    predictions of the images / videos in ``unit_ids`` only.  ``unit`` is
    "image" or "video"; ``cat_ids`` = params.cat_ids in the caller's order.
    Returns (gt columns, dt columns, cat_pos): cat_pos[i] = index of
    cat_ids[i] in the sorted category table -- the category axis of the result
    tables is taken through it (categories are independent of each other when
    use_cats = 1, so a category subset is a selection of result columns; the
    class-agnostic pool of use_cats = 0 is formed from all categories and a
    subset there raises).  Unknown ids raise KeyError like the reference's
    load_imgs / load_vids / load_cats."""
    all_units = np.unique(gt.img_id if unit == "image" else gt.vid_id)
    unit_ids = np.asarray(unit_ids, dtype=np.int64).reshape(-1)
    pos = np.searchsorted(all_units, unit_ids)
    bad = (pos >= len(all_units)) | (all_units[np.minimum(pos, len(all_units) - 1)] != unit_ids) \
        if len(all_units) else np.ones(len(unit_ids), bool)
    if bad.any():
        raise KeyError(int(unit_ids[np.flatnonzero(bad)[0]]))
    all_cats = np.unique(gt.cat_id)
    cat_ids = np.asarray(cat_ids, dtype=np.int64).reshape(-1)
    if len(cat_ids) == 0:
        raise NotImplementedError("an empty params.cat_ids is not evaluated")
    cpos = np.searchsorted(all_cats, cat_ids)
    cbad = (cpos >= len(all_cats)) | (all_cats[np.minimum(cpos, len(all_cats) - 1)] != cat_ids)
    if cbad.any():
        raise KeyError(int(cat_ids[np.flatnonzero(cbad)[0]]))
    whole_cats = len(cat_ids) == len(all_cats) and np.array_equal(cat_ids, all_cats)
    if not whole_cats and not use_cats:
        raise NotImplementedError(
            "params.cat_ids restricts the categories while use_cats = 0 pools "
            "them: not evaluated on the HIP path")
    if len(unit_ids) != len(all_units):
        if unit == "image":
            keep = np.isin(gt.img_id, unit_ids)
            gt = gt.select_images(keep)
            dt = dt.take(np.flatnonzero(np.isin(dt.image_id, unit_ids)))
        else:
            gt = gt.select_videos(np.isin(gt.vid_id, unit_ids))
            # predictions on the images of those videos (T/tao.py:224-235)
            dt = dt.take(np.flatnonzero(np.isin(dt.image_id, gt.img_id)))
    return gt, dt, (None if whole_cats else cpos)


class CellView:
    """What the reference stores per (unit, category, range) in eval_imgs /
    eval_vids, rebuilt on demand from the device results.  ``cat_pos``: index
    of the caller's i-th category (params.cat_ids) in the tables' category
    axis, when params.cat_ids is not the whole sorted table."""

    cat_pos = None

    def __init__(self, run, unit_ids, sentinel, unit_key, rng_key, rng_values):
        self.run = run
        self.flat = run.flat
        self.unit_ids = unit_ids
        self.sentinel = sentinel
        self.unit_key, self.rng_key, self.rng_values = unit_key, rng_key, rng_values
        self.K = len(self.flat.cat_ids)
        self._index = None

    def cell_of(self, unit_idx, cat_idx, table_index=False):
        if self.cat_pos is not None and not table_index:
            cat_idx = self.cat_pos[cat_idx]
        if self._index is None:     # built on first inspection only
            f = self.flat
            self._index = {int(u) * self.K + int(c): k for k, (u, c) in
                           enumerate(zip(f.cell_unit, f.cell_cat))}
        return self._index.get(int(unit_idx) * self.K + int(cat_idx))

    def iou(self, k):
        f, d = self.flat, self.run.detail()
        off = np.zeros(f.n_cells + 1, dtype=np.int64)
        np.cumsum(np.diff(f.cell_dt_off).astype(np.int64)
                  * np.diff(f.cell_gt_off), out=off[1:])
        D = f.cell_dt_off[k + 1] - f.cell_dt_off[k]
        G = f.cell_gt_off[k + 1] - f.cell_gt_off[k]
        if self.flat.kind == "lvis" and (D == 0 or G == 0):
            return []           # mask_utils.iou returns [] (_mask.pyx:203-204)
        return d["iou"][off[k]:off[k + 1]].reshape(D, G).copy()

    def entry(self, k, r):
        f, d = self.flat, self.run.detail()
        d0, d1 = f.cell_dt_off[k], f.cell_dt_off[k + 1]
        g0, g1 = f.cell_gt_off[k], f.cell_gt_off[k + 1]
        D, G = d1 - d0, g1 - g0
        gid = f.gt_id[g0:g1]
        ig = ((d["gt_rng"][g0:g1] >> np.uint32(r)) & np.uint32(1)).astype(np.int64)
        perm = np.argsort(ig, kind="mergesort")
        slots = self.run.thr_slots()
        dt_m = np.full((len(slots), D), float(self.sentinel))
        gt_m = np.full((len(slots), G), float(self.sentinel))
        dt_ig = np.zeros((len(slots), D), dtype=bool)
        for t, slot in enumerate(slots):
            combo = r * N_THR + slot
            m = d["match_gt"][d0:d1, combo]
            hit = m >= 0
            if G:
                dt_m[t, hit] = gid[m[hit]]
                for j in np.flatnonzero(hit):
                    gt_m[t, m[j]] = f.dt_id[d0 + j]
            dt_ig[t] = _bit(d["ignored"][d0:d1], combo)
        return {
            self.unit_key: int(self.unit_ids[f.cell_unit[k]]),
            "category_id": int(f.cat_ids[f.cell_cat[k]]),
            self.rng_key: self.rng_values[r],
            "dt_ids": f.dt_id[d0:d1].tolist(),
            "gt_ids": gid[perm].tolist(),
            "dt_matches": dt_m,
            "gt_matches": gt_m[:, perm],
            "dt_scores": f.dt_score[d0:d1].tolist(),
            "gt_ignore": ig[perm],
            "dt_ignore": dt_ig,
        }


class LazyIous(Mapping):
    """``self.ious``: {(unit_id, cat_id): (D, G) array or []} over ALL unit x
    category pairs, like the reference, but computed on access."""

    def __init__(self, view, unit_list, cat_list):
        self.view, self.units, self.cats = view, unit_list, cat_list
        self._pos = None

    def _positions(self):
        # (600 k images at full scale: built when somebody looks something up)
        if self._pos is None:
            self._pos = ({int(u): i for i, u in enumerate(self.view.unit_ids)},
                         {int(c): i for i, c in enumerate(self.view.flat.cat_ids)})
        return self._pos

    @property
    def upos(self):
        return self._positions()[0]

    @property
    def cpos(self):
        return self._positions()[1]

    def __getitem__(self, key):
        u, c = key
        if u not in self.upos or c not in self.cpos:
            raise KeyError(key)
        k = self.view.cell_of(self.upos[u], self.cpos[c], table_index=True)
        return [] if k is None else self.view.iou(k)

    def __iter__(self):
        return ((int(u), int(c)) for u in self.units for c in self.cats)

    def __len__(self):
        return len(self.units) * len(self.cats)


class LazyPointers(Mapping):
    """eval['dt_pointers'][cat_idx][range...] = {dt_ids, tps, fps}."""

    def __init__(self, run, n_rng, shape, cat_pos=None):
        self.run, self.n_rng, self.shape = run, n_rng, shape
        self.cat_pos = cat_pos
        self._rows = None

    def _data(self):
        if self._rows is None:
            self._rows = self.run.pointer_tables()
        return self._rows

    def leaf(self, k, r):
        d = self._data()
        matched, ignored = d["matched"], d["ignored"]
        if d["num_gt"][k, r] == 0:
            return {}
        lo, hi = int(d["cat_off"][k]), int(d["cat_off"][k + 1])
        slots = self.run.thr_slots()
        m = np.stack([_bit(matched[lo:hi], r * N_THR + t) for t in slots])
        i = np.stack([_bit(ignored[lo:hi], r * N_THR + t) for t in slots])
        return {"dt_ids": d["ids"][lo:hi], "tps": m & ~i, "fps": ~m & ~i}

    def __getitem__(self, k):
        if not 0 <= k < len(self):
            raise KeyError(k)
        if self.cat_pos is not None:
            k = int(self.cat_pos[k])
        if len(self.shape) == 1:
            return {r: self.leaf(k, r) for r in range(self.shape[0])}
        A, T = self.shape
        return {a: {t: self.leaf(k, a * T + t) for t in range(T)}
                for a in range(A)}

    def __iter__(self):
        return iter(range(len(self)))

    def __len__(self):
        return self.run.dp.n_cat if self.cat_pos is None else len(self.cat_pos)


def now():
    return datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")


class EvalConstants:
    """``params.iou_thrs`` / ``rec_thrs`` and the range tables as the kernels
    take them.  The reference reads all of them when it runs (L/eval.py:140-145,
    205,234,319-329,407; T/eval.py:271-276,385,473-477,562), so a caller may
    edit them before ``run()`` -- values, order AND number.

    The kernels take N_THR IoU thresholds and N_REC recall thresholds per pass,
    ascending, as by-value arguments, and the range VALUES likewise
    (taoamd_set_thresholds / taoamd_set_ranges); the NUMBER of ranges per pass
    is theirs (5 visibility ranges + the out-of-frame one -- the reference's
    LAST range, whatever its bounds, L/eval.py:143 --; 4 areas + the occlusion
    one -- the LAST area range, T/eval.py:272 -- x 4 durations).  Every
    threshold and every range is evaluated independently of the others, so
    the caller's arrays -- any number, any order -- are cut into blocks, a
    short block padded with copies of its last value, a pass runs per block
    and every block's slice of the tables goes where the caller's order has
    it (``rng_blocks``: the tables of a pass and, per kernel range slot, the
    caller's range index)."""

    def __init__(self, params, fresh, kind):
        iou = np.asarray(params.iou_thrs, dtype=np.float64).reshape(-1)
        rec = np.asarray(params.rec_thrs, dtype=np.float64).reshape(-1)
        self.T, self.R = len(iou), len(rec)
        if self.T == 0 or self.R == 0:
            raise NotImplementedError("empty params.iou_thrs / params.rec_thrs")
        if np.isnan(iou).any() or np.isnan(rec).any():
            raise ValueError("params.iou_thrs / params.rec_thrs hold a NaN")
        same = np.array_equal(iou, fresh.iou_thrs) and np.array_equal(rec, fresh.rec_thrs)

        def table(name):
            try:
                got = np.asarray(getattr(params, name), dtype=np.float64)
            except (TypeError, ValueError):
                got = np.zeros(0)
            if got.ndim != 2 or got.shape[1] != 2 or len(got) == 0:
                raise NotImplementedError(
                    "params.%s: a non-empty list of [lo, hi] pairs is evaluated" % name)
            return got, np.asarray(getattr(fresh, name), dtype=np.float64)

        def cut(idx, cap, values, filler):
            """Blocks of <= cap of the ranges `idx`: (indices, (cap, 2) values
            padded with the block's last range)."""
            out = []
            for i in range(0, max(len(idx), 1), cap):
                sel = idx[i:i + cap]
                v = values[sel] if len(sel) else np.zeros((0, 2))
                pad = v[-1:] if len(sel) else filler[None]
                out.append((list(sel), np.concatenate([v] + [pad] * (cap - len(sel)))))
            return out

        self.rng_blocks = []
        if kind == "lvis":
            got, want = table("visibility_rng")
            self.n_rng = len(got)
            # (the last range is the out-of-frame one, its bounds are never
            # read -- L/eval.py:143,209-217)
            same = same and got.shape == want.shape and np.array_equal(got[:-1], want[:-1])
            for b, (sel, vals) in enumerate(cut(np.arange(len(got) - 1), 5, got,
                                                 np.array([0.0, 1.0]))):
                slots = [(k, int(i)) for k, i in enumerate(sel)]
                if b == 0:
                    slots.append((5, len(got) - 1))
                self.rng_blocks.append(({"visibility_rng": vals}, slots))
            identity = len(got) == 6
        else:
            area, want_a = table("area_rng")
            time_, want_t = table("time_rng")
            A, Tm = len(area), len(time_)
            self.n_rng = A * Tm
            same = same and area.shape == want_a.shape and time_.shape == want_t.shape \
                and np.array_equal(area, want_a) and np.array_equal(time_, want_t)
            a_blocks = cut(np.arange(A - 1), 4, area, area[-1])
            t_blocks = cut(np.arange(Tm), 4, time_, time_[-1])
            for ab, (asel, avals) in enumerate(a_blocks):
                # slot 4 of the kernels' area table: the occlusion range
                atab = np.concatenate([avals, area[-1:]])
                aslots = [(k, int(i)) for k, i in enumerate(asel)]
                if ab == 0:
                    aslots.append((4, A - 1))
                for tsel, tvals in t_blocks:
                    slots = [(ka * 4 + kt, ia * Tm + int(it)) for ka, ia in aslots
                             for kt, it in enumerate(tsel)]
                    self.rng_blocks.append(({"area_rng": atab, "time_rng": tvals}, slots))
            identity = A == 5 and Tm == 4
        self.default = bool(same)
        self.thr_blocks = self._blocks(iou, N_THR)
        self.rec_blocks = self._blocks(rec, N_REC)
        # one pass whose tables are the caller's, slot for slot: the per-cell
        # views (ious, eval_imgs / eval_vids, dt_pointers) read such a pass
        self.single = len(self.thr_blocks) == 1 and len(self.rec_blocks) == 1 \
            and len(self.rng_blocks) == 1 and identity
        self.rec_sorted = bool(np.all(np.diff(rec) >= 0))
        self.ranges = self.rng_blocks[0][0]

    @staticmethod
    def _blocks(values, cap):
        order = np.argsort(values, kind="stable")
        out = []
        for i in range(0, len(values), cap):
            idx = order[i:i + cap]
            v = values[idx]
            out.append((idx, np.concatenate([v, np.full(cap - len(v), v[-1])])))
        return out


def applied(constants, thr_block=0, rec_block=0, rng_block=0):
    """Context manager: the calling thread's kernels launched inside take
    ``constants`` (None: nothing to do); the reference's defaults come back on
    the way out."""
    import contextlib
    from .. import _lib

    @contextlib.contextmanager
    def cm():
        if constants is None or constants.default:
            yield
            return
        _lib.set_constants(iou_thrs=constants.thr_blocks[thr_block][1],
                           rec_thrs=constants.rec_blocks[rec_block][1],
                           **constants.rng_blocks[rng_block][0])
        try:
            yield
        finally:
            _lib.set_constants()
    return cm()
