"""Cell-table build with the detection side on the device (csrc/flatten.hip).

``flatten.py`` is the numpy statement of the flatten stage and stays the
reference for every corner case; this module produces the SAME tables
(tests/test_gpu_flatten.py compares them field by field) with the per-box work
-- id lookups, top-300 cut, filters, the stable (cell, -score) sort, the
gathers -- done by HIP kernels on one upload of the raw prediction columns.
The ground-truth side (10^5 rows, dict / alias semantics) is taken from
``flatten.py`` unchanged, and so is everything that is O(cells).

PyTorch only allocates; every pass is a kernel of the C ABI
(``taoamd_flat_*``, ``taoamd_sort_by_cat_score``).
"""
import threading
import warnings
import weakref

import numpy as np
import torch

from . import _lib, flatten
from .flatten import DT_IGNORE_UNMATCHED, Flat, I32, MAX_DETS   # noqa: F401

class Unsupported(Exception):
    """The input is outside what the device build handles (more than 2^31
    cell keys or boxes): the caller falls back to flatten.py."""


class Rejected(Exception):
    """The input breaks a rule the reference asserts (a track in two videos or
    with two categories, results of unknown images / videos): the caller runs
    the numpy statement, which raises the reference's exception."""


class DeviceFlat(Flat):
    """Flat whose per-detection tables live on the device (``dev``); reading
    one as an attribute downloads it once (views, tests, the oracle).  ``lazy``
    holds host-side thunks for tables only the class API's views read (called
    with the table set itself: a thunk that closed over it would tie the set,
    its device tensors and the prediction columns into a reference cycle that
    only the interpreter's cycle collector undoes -- gigabytes a call, freed
    at some later call's expense)."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "dev", {})
        object.__setattr__(self, "lazy", {})

    def __getitem__(self, k):
        if dict.__contains__(self, k):
            return dict.__getitem__(self, k)
        if k in self.lazy:
            v = self.lazy[k](self)
        elif k in self.dev:
            v = self.dev[k].cpu().numpy()
        else:
            raise KeyError(k)
        dict.__setitem__(self, k, v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def to_host(self):
        """A plain Flat with every table on the host."""
        f = Flat()
        for k in list(self.dev) + list(self.lazy):
            self[k]
        for k, v in self.items():
            f[k] = v
        return f


_RAW = {}      # id(DTColumns) -> (weakref, {device: dict of tensors})
_RAW_LOCK = threading.Lock()     # (the CLI builds both levels' tables side by side)


def _column_key(dt):
    """Identity of the arrays behind a DTColumns' columns: the cached device
    copy is that of THESE arrays with THIS content (a sampled digest,
    _fingerprint) -- a caller that rebinds a column (dt.score = other) or edits
    one in place gets a fresh upload."""
    key = []
    for name in ("image_id", "category_id", "score", "bbox", "video_id", "area"):
        v = getattr(dt, name, None)
        key.append(None if v is None else
                   (id(v), v.__array_interface__["data"][0] if isinstance(v, np.ndarray)
                    else 0, getattr(v, "shape", None), _fingerprint(v)))
    return tuple(key)


from .columns import fingerprint as _fingerprint   # noqa: E402


def forget_columns(dt):
    """Drop the cached device copy of a DTColumns' columns."""
    _RAW.pop(id(dt), None)


def raw_columns(dt, device):
    """The prediction columns on the device, uploaded once per DTColumns (and
    set of column arrays) and shared by the image-level and the track-level
    build."""
    with _RAW_LOCK:
        return _raw_columns(dt, device)


def _raw_columns(dt, device):
    dev = torch.device(device)
    born = getattr(dt, "device_columns", None)
    if born is not None and getattr(dt, "area", None) is None:
        # columns the device-side reader made (columns.DeviceDTColumns): there
        # already, unless the caller has replaced or edited one since
        cols = born(dev, ("image_id", "category_id", "score", "bbox", "video_id"))
        if cols is not None:
            cols["area"] = None
            return cols
    key = id(dt)
    cols_key = _column_key(dt)
    ent = _RAW.get(key)
    if ent is None or ent[0]() is not dt or ent[2] != cols_key:
        ent = (weakref.ref(dt, lambda _r, k=key: _RAW.pop(k, None)), {}, cols_key)
        _RAW[key] = ent
    if dev not in ent[1]:
        cols = {}
        for name in ("image_id", "category_id", "score", "bbox", "video_id"):
            v = getattr(dt, name, None)
            cols[name] = None if v is None else torch.from_numpy(
                np.ascontiguousarray(v)).to(dev, non_blocking=True)
        area = getattr(dt, "area", None)
        cols["area"] = None if area is None else torch.from_numpy(
            np.ascontiguousarray(area, dtype=np.float64)).to(dev)
        ent[1][dev] = cols
    return ent[1][dev]


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Sorter:
    """taoamd_sort_by_cat_score with a workspace that is reused."""

    def __init__(self, n, dev):
        self.lib = _lib.load()
        self.bytes = int(self.lib.taoamd_sort_workspace(n))
        self.ws = torch.empty(max(self.bytes, 256), dtype=torch.uint8, device=dev)
        self.dev = dev

    def __call__(self, n, key, score):
        order = torch.empty(max(n, 1), dtype=torch.int32, device=self.dev)
        _lib.check(self.lib.taoamd_sort_by_cat_score(
            n, _ptr(key), _ptr(score), _ptr(order), None, _ptr(self.ws),
            self.bytes, _stream()), "taoamd_sort_by_cat_score")
        return order


def _cells_from_runs(lib, dev, n_keep, dt_key, gkeys_sorted, keys_g):
    """Union of the detection cells (runs of dt_key) and the ground-truth cells;
    returns (cell_keys, dt_cell tensor, d_off, cell of every ground truth,
    g_off).  Round 4: the O(cells) arithmetic -- union of the two sorted key
    lists, the runs' and the ground truths' places in it, the two offset scans;
    3.2 M cells at 2000 videos, 0.22 s of numpy on the critical path of the
    CLI -- runs on the device (torch's sort / search / scan primitives: glue
    like engine.DeviceProblem's dt_meta, not a kernel of the hot path) and the
    four host tables the launch plans read come back in one batch."""
    run_id = torch.empty(max(n_keep, 1), dtype=torch.int32, device=dev)
    run_key = torch.empty(max(n_keep, 1), dtype=torch.int32, device=dev)
    run_start = torch.empty(max(n_keep, 1), dtype=torch.int32, device=dev)
    n_runs = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = int(lib.taoamd_flat_runs_workspace(n_keep))
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
    _lib.check(lib.taoamd_flat_runs(
        n_keep, _ptr(dt_key), _ptr(run_id), _ptr(run_key), _ptr(run_start),
        _ptr(n_runs), _ptr(ws), wsb, _stream()), "taoamd_flat_runs")
    t_gk = torch.from_numpy(np.ascontiguousarray(gkeys_sorted, dtype=np.int32)).to(dev)
    t_kg = torch.from_numpy(np.ascontiguousarray(keys_g, dtype=np.int32)).to(dev)
    nr = int(n_runs.item())
    rk, rs = run_key[:nr], run_start[:nr].long()
    t_cells = torch.unique(torch.cat([t_gk, rk]))            # sorted
    n_cells = int(t_cells.numel())
    map_d = torch.searchsorted(t_cells, rk).to(torch.int32)
    cnt = torch.zeros(n_cells + 1, dtype=torch.int64, device=dev)
    if nr:
        ends = torch.cat([rs[1:], torch.tensor([n_keep], dtype=torch.int64, device=dev)])
        cnt[map_d.long() + 1] = ends - rs
    d_off = torch.cumsum(cnt, 0)
    g_cell = torch.searchsorted(t_cells, t_kg)
    g_off = torch.zeros(n_cells + 1, dtype=torch.int64, device=dev)
    if t_kg.numel():
        g_off[1:] = torch.cumsum(torch.bincount(g_cell, minlength=n_cells), 0)
    dt_cell = torch.empty(max(n_keep, 1), dtype=torch.int32, device=dev)
    if nr:
        _lib.check(lib.taoamd_flat_remap(
            n_keep, _ptr(run_id), _ptr(map_d), _ptr(dt_cell), _stream()),
            "taoamd_flat_remap")
    # (map_d stays referenced until the downloads below have synchronised)
    host = [t.cpu().numpy() for t in (t_cells, d_off, g_cell, g_off)]
    return host[0], dt_cell[:n_keep], host[1], host[2], host[3]


# ---------------------------------------------------------------------------
# ground-truth halves, buildable before the predictions are there: prepare.py
# (no torch there -- the CLI builds them while torch is still being imported)
from .prepare import (_gt_key, _gt_ready, _gt_universe, _lvis_gt_ready, _READY,   # noqa: E402,F401
                      _tao_gt_ready, prepare_gt)


def flatten_lvis_device(gt, dt, device="cuda", max_dets=MAX_DETS):
    """flatten.flatten_lvis(gt, dt, max_dets) with the detection side built on
    the device.  Raises Unsupported for inputs the kernels do not cover."""
    if len(dt) == 0:
        raise IndexError("list index out of range")  # L/results.py:42
    lib = _lib.load()
    dev = torch.device(device)
    ready = _gt_ready(gt, "lvis")
    G, keys_g, gkeys = ready.G, ready.keys_g, ready.gkeys
    img_ids, cat_ids = G.img_ids, G.cat_ids
    U, K, n = len(img_ids), len(cat_ids), len(dt)
    if U == 0 or K == 0 or K * U >= 2 ** 31 - 1 or n >= 2 ** 31 - 1:
        raise Unsupported("cell keys do not fit 31 bits")

    with torch.cuda.device(dev):
        raw = raw_columns(dt, dev)
        up = lambda a, t: torch.from_numpy(np.ascontiguousarray(a, dtype=t)).to(dev)
        d_img = torch.empty(n, dtype=torch.int32, device=dev)
        d_cat = torch.empty(n, dtype=torch.int32, device=dev)
        d_area = torch.empty(n, dtype=torch.float64, device=dev)
        img_count = torch.empty(U + 1, dtype=torch.int32, device=dev)
        img_start = torch.empty(U + 1, dtype=torch.int32, device=dev)
        status = torch.empty(4, dtype=torch.int32, device=dev)
        t_img_ids, t_cat_ids = up(img_ids, np.int64), up(cat_ids, np.int64)
        _lib.check(lib.taoamd_flat_map(
            n, _ptr(raw["image_id"]), _ptr(raw["category_id"]), _ptr(raw["bbox"]),
            _ptr(raw["area"]), U, _ptr(t_img_ids), K, _ptr(t_cat_ids),
            _ptr(d_img), _ptr(d_cat), _ptr(d_area), _ptr(img_count),
            _ptr(img_start), None, _ptr(status), _stream()), "taoamd_flat_map")
        # the uploads of the small ground-truth tables travel meanwhile
        t_gkeys = up(gkeys, np.int32)
        t_img_row = up(G.img_row, np.int32)
        t_neg_off, t_neg = up(gt.img_neg_off, np.int64), up(gt.img_neg, np.int64)
        t_nel_off, t_nel = up(gt.img_nel_off, np.int64), up(gt.img_nel, np.int64)
        st = status.cpu().numpy()
        if st[0]:
            raise AssertionError("Results do not correspond to current LVIS set.")
        sorter = _Sorter(n, dev)
        dropped = None
        if 0 <= max_dets < int(st[1]):
            # some image holds more than max_dets boxes: rank inside the image
            order1 = sorter(n, d_img, raw["score"])
            dropped = torch.empty(n, dtype=torch.uint8, device=dev)
            _lib.check(lib.taoamd_flat_rank_drop(
                n, _ptr(order1), _ptr(d_img), _ptr(img_start), max_dets,
                _ptr(dropped), _stream()), "taoamd_flat_rank_drop")
        key = torch.empty(n, dtype=torch.int32, device=dev)
        flags = torch.empty(n, dtype=torch.uint8, device=dev)
        n_keep_t = torch.empty(1, dtype=torch.int32, device=dev)
        _lib.check(lib.taoamd_flat_filter(
            n, _ptr(d_img), _ptr(d_cat), _ptr(d_area), _ptr(raw["category_id"]),
            _ptr(dropped), U, len(gkeys), _ptr(t_gkeys), _ptr(t_img_row),
            _ptr(t_neg_off), _ptr(t_neg), _ptr(t_nel_off), _ptr(t_nel), 1,
            _ptr(key), _ptr(flags), _ptr(n_keep_t), _stream()),
            "taoamd_flat_filter")
        order2 = sorter(n, key, raw["score"])
        n_keep = int(n_keep_t.item())
        m = max(n_keep, 1)
        dt_row = torch.empty(m, dtype=torch.int32, device=dev)
        dt_score = torch.empty(m, dtype=torch.float64, device=dev)
        dt_flags = torch.empty(m, dtype=torch.uint8, device=dev)
        dt_key = torch.empty(m, dtype=torch.int32, device=dev)
        dt_cat = torch.empty(m, dtype=torch.int32, device=dev)
        dt_box = torch.empty((m, 4), dtype=torch.float64, device=dev)
        _lib.check(lib.taoamd_flat_gather(
            n_keep, _ptr(order2), _ptr(raw["score"]), _ptr(flags), _ptr(key),
            _ptr(raw["bbox"]), U, _ptr(dt_row), _ptr(dt_score), _ptr(dt_flags),
            _ptr(dt_key), _ptr(dt_cat), _ptr(dt_box), _stream()),
            "taoamd_flat_gather")
        cell_keys, dt_cell, d_off, g_cell, g_off = _cells_from_runs(
            lib, dev, n_keep, dt_key, gkeys, keys_g)

    n_cells = len(cell_keys)

    f = DeviceFlat()
    f.kind = "lvis"
    f.use_cats = True
    f.img_ids, f.cat_ids = img_ids, cat_ids
    f.cat_freq = flatten._freq_of(gt, cat_ids)
    f.n_cells = n_cells
    f.cell_unit = (cell_keys % U).astype(I32)
    f.cell_cat = (cell_keys // U).astype(I32)
    f.cell_dt_off = d_off.astype(I32)
    f.cell_gt_off = g_off.astype(I32)
    f.update(ready.tables)
    f.gt_cell = g_cell.astype(I32)
    f.n_pairs = int(np.sum(np.diff(d_off) * np.diff(g_off)))
    f.dev.update(dt_box=dt_box[:n_keep], dt_score=dt_score[:n_keep],
                 dt_flags=dt_flags[:n_keep], dt_cat=dt_cat[:n_keep],
                 dt_cell=dt_cell, dt_row=dt_row[:n_keep])

    def dt_id(f):
        # id = 1 + position in the post-truncation list (L/results.py:73-84):
        # only the class API's views read it
        keep = flatten.limit_dets_per_image(dt, max_dets)
        pos = np.full(len(dt), -1, dtype=np.int64)
        pos[keep] = np.arange(len(keep))
        return pos[f.dt_row.astype(np.int64)] + 1
    f.lazy["dt_id"] = dt_id
    return f


class _Runs:
    """Runs of equal keys of key[order[.]] (taoamd_flat_runs*_by)."""

    def __init__(self, lib, dev, n, key, order, wide=False):
        m = max(n, 1)
        self.run_id = torch.empty(m, dtype=torch.int32, device=dev)
        self.run_key = torch.empty(m, dtype=torch.int64 if wide else torch.int32,
                                   device=dev)
        self.run_start = torch.empty(m, dtype=torch.int32, device=dev)
        n_runs = torch.zeros(1, dtype=torch.int32, device=dev)
        wsb = int(lib.taoamd_flat_runs_workspace(n))
        ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
        fn = lib.taoamd_flat_runs64_by if wide else lib.taoamd_flat_runs_by
        _lib.check(fn(n, _ptr(key), _ptr(order), _ptr(self.run_id),
                      _ptr(self.run_key), _ptr(self.run_start), _ptr(n_runs),
                      _ptr(ws), wsb, _stream()), "taoamd_flat_runs_by")
        self.n = int(n_runs.item())


def flatten_tao_device(gt, dt, device="cuda", max_dets=MAX_DETS,
                       visit_universe=None):
    """flatten.flatten_tao(gt, dt, max_dets) with the prediction side built on
    the device (``dt.track_id`` already unique per video).  Inputs the kernels
    do not cover raise Unsupported, inputs the reference rejects raise Rejected
    (``flatten_tao`` / ``flatten_lvis`` below fall back to flatten.py, which
    words the reference's exception)."""
    if len(dt) == 0:
        raise IndexError("list index out of range")  # T/results.py:61
    lib = _lib.load()
    dev = torch.device(device)
    # (a rank's share of the annotation file takes its visiting order from the
    # whole set: never prepared ahead)
    # Round 6: when the ground-truth half is being built in the background in
    # two stages (prepare_gt), the prediction side starts with the first --
    # ids, timeline, visiting order -- and meets the annotation part (`ready`)
    # at the federated filter, by when it is there.
    ready = None
    T = _gt_universe(gt) if visit_universe is None else None
    if T is None:
        ready = _gt_ready(gt, "tao") if visit_universe is None \
            else _tao_gt_ready(gt, visit_universe)
        T = ready.T
    img_frame = gt.img_frame[T.img_row]
    vid_ids, cat_ids, img_ids = T.vid_ids, T.cat_ids, T.img_ids
    U, K, NI, n = len(vid_ids), len(cat_ids), len(img_ids), len(dt)
    # (columns made on the device bring their track ids along: no wait for the
    # host array, no upload)
    born = getattr(dt, "device_columns", None)
    tid_dev = born(dev, ("track_id",)) if born is not None else None
    if tid_dev is not None:
        tid_dev = tid_dev["track_id"]
        with torch.cuda.device(dev):
            tid_lo, tid_hi = int(tid_dev.min()), int(tid_dev.max())
    else:
        tid_host = np.ascontiguousarray(dt.track_id, dtype=np.int64)
        tid_lo, tid_hi = int(tid_host.min()), int(tid_host.max())
    if U == 0 or K == 0 or K * U >= 2 ** 31 - 1 or n >= 2 ** 31 - 1 or \
            tid_lo < 0 or tid_hi >= 2 ** 62:
        raise Unsupported("keys do not fit")

    def reject():
        raise Rejected("predictions break a rule of TaoResults")

    I32_MAX = 2 ** 31 - 1
    with torch.cuda.device(dev):
        raw = raw_columns(dt, dev)
        if raw["video_id"] is None:
            raise Unsupported("predictions carry no video_id")
        hold = []      # uploads stay referenced until the build is over: a
        # tensor freed right after data_ptr() would be recycled by the next one

        def up(a, t):
            hold.append(torch.from_numpy(np.ascontiguousarray(a, dtype=t)).to(dev))
            return hold[-1]
        new = lambda m, t: torch.empty(max(int(m), 1), dtype=t, device=dev)
        # (not cached: make_track_ids_unique rewrites it)
        tid = tid_dev if tid_dev is not None else up(tid_host, np.int64)
        d_img, d_cat0 = new(n, torch.int32), new(n, torch.int32)
        d_area = new(n, torch.float64)
        img_count, img_start = new(NI + 1, torch.int32), new(NI + 1, torch.int32)
        img_first = new(NI, torch.int32)
        status = new(4, torch.int32)
        t_img_ids, t_cat_ids = up(img_ids, np.int64), up(cat_ids, np.int64)
        _lib.check(lib.taoamd_flat_map(
            n, _ptr(raw["image_id"]), _ptr(raw["category_id"]), _ptr(raw["bbox"]),
            _ptr(raw["area"]), NI, _ptr(t_img_ids), K, _ptr(t_cat_ids),
            _ptr(d_img), _ptr(d_cat0), _ptr(d_area), _ptr(img_count),
            _ptr(img_start), _ptr(img_first), _ptr(status), _stream()),
            "taoamd_flat_map")
        merged_id, d_cat = new(n, torch.int64), new(n, torch.int32)
        _lib.check(lib.taoamd_flat_merge_cat(
            n, _ptr(raw["category_id"]), len(T.ms), _ptr(up(T.ms, np.int64)),
            _ptr(up(T.md, np.int64)), K, _ptr(t_cat_ids), _ptr(merged_id),
            _ptr(d_cat), _stream()), "taoamd_flat_merge_cat")
        st = status.cpu().numpy()
        if st[0]:
            reject()
        max_count = int(st[1])
        sorter = _Sorter(n, dev)

        # ---- place of every box inside its image, top-max_dets cut
        sc0 = None
        if 0 <= max_dets < max_count:
            sc0 = new(n, torch.float64)
            _lib.check(lib.taoamd_flat_ordscore(
                n, _ptr(d_img), _ptr(img_count), _ptr(raw["score"]), max_dets,
                _ptr(sc0), _stream()), "taoamd_flat_ordscore")
        order0 = sorter(n, d_img, sc0)
        ordinal, dropped = new(n, torch.int32), new(n, torch.uint8)
        _lib.check(lib.taoamd_flat_ordinal(
            n, _ptr(order0), _ptr(d_img), _ptr(img_start), max_dets, _ptr(ordinal),
            _ptr(dropped), _stream()), "taoamd_flat_ordinal")

        n_bad = new(1, torch.int32)
        _lib.check(lib.taoamd_flat_count_bad(n, _ptr(raw["bbox"]), _ptr(dropped),
                                             _ptr(n_bad), _stream()),
                   "taoamd_flat_count_bad")

        # ---- tracks: runs of the boxes sorted by track id
        lo, hi = new(n, torch.int32), None
        wide = tid_hi >= 2 ** 31
        if wide:
            hi = new(n, torch.int32)
        _lib.check(lib.taoamd_flat_split64(n, _ptr(tid), None, _ptr(lo), None,
                                           _stream()), "taoamd_flat_split64")
        order_t = sorter(n, lo, None)
        if wide:
            _lib.check(lib.taoamd_flat_split64(n, _ptr(tid), _ptr(order_t), None,
                                               _ptr(hi), _stream()),
                       "taoamd_flat_split64")
            inner = sorter(n, hi, None)
            comp = new(n, torch.int32)
            _lib.check(lib.taoamd_flat_compose(n, _ptr(order_t), _ptr(inner),
                                               _ptr(comp), _stream()),
                       "taoamd_flat_compose")
            order_t = comp
        tr = _Runs(lib, dev, n, tid, order_t, wide=True)
        n_trk = tr.n
        trk = new(n, torch.int32)
        st2 = torch.full((4,), I32_MAX, dtype=torch.int32, device=dev)
        st2[1] = 0
        _lib.check(lib.taoamd_flat_track_of(
            n, _ptr(order_t), _ptr(tr.run_id), _ptr(tr.run_start),
            _ptr(raw["video_id"]), _ptr(trk), _ptr(st2), _stream()),
            "taoamd_flat_track_of")

        # ---- sort keys (host: first-seen rank of the images)
        first = img_first.cpu().numpy().astype(np.int64)
        img_rank = np.empty(NI, dtype=np.int32)
        img_rank[np.argsort(first, kind="stable")] = np.arange(NI, dtype=np.int32)
        M = float(1 << max(int(max_count), 1).bit_length())
        if NI * M >= 2.0 ** 52:
            raise Unsupported("sort keys do not fit a double")
        keep_key, visit_key = new(n, torch.float64), new(n, torch.float64)
        frame_key, pos_key = new(n, torch.float64), new(n, torch.float64)
        trk_keep, trk_sel = new(n, torch.int32), new(n, torch.int32)
        t_tl_pos = up(T.tl_pos, np.int32)
        _lib.check(lib.taoamd_flat_keys(
            n, _ptr(d_img), _ptr(ordinal), _ptr(dropped), _ptr(d_cat), _ptr(d_area),
            _ptr(trk), _ptr(up(img_rank, np.int32)), _ptr(up(T.visit_rank, np.int32)),
            _ptr(up(img_frame, np.float64)), _ptr(t_tl_pos), M, _ptr(keep_key),
            _ptr(visit_key), _ptr(frame_key), _ptr(pos_key), _ptr(trk_keep),
            _ptr(trk_sel), _stream()), "taoamd_flat_keys")

        # ---- per track over its kept boxes in list order: score, category
        order_k = sorter(n, trk_keep, keep_key)
        rk = _Runs(lib, dev, n, trk_keep, order_k)
        trk_score = torch.zeros(max(n_trk, 1), dtype=torch.float64, device=dev)
        trk_first = torch.full((max(n_trk, 1),), -1, dtype=torch.int32, device=dev)
        _lib.check(lib.taoamd_flat_track_kept(
            rk.n, _ptr(rk.run_key), _ptr(rk.run_start), n, _ptr(order_k),
            _ptr(raw["score"]), _ptr(merged_id), _ptr(trk_score), _ptr(trk_first),
            _ptr(st2), _stream()), "taoamd_flat_track_kept")

        # ---- selected boxes: visiting order, then (track, frame_index) stable
        zeros = torch.zeros(n, dtype=torch.int32, device=dev)
        order_v = sorter(n, zeros, visit_key)

        def resort(outer, key_f64):
            """stable sort of the list `outer` by (selected track, key)"""
            kt, kf = new(n, torch.int32), new(n, torch.float64)
            _lib.check(lib.taoamd_flat_gather_cols(
                n, _ptr(outer), _ptr(trk_sel), _ptr(kt), _ptr(key_f64), _ptr(kf),
                None, None, _stream()), "taoamd_flat_gather_cols")
            inner = sorter(n, kt, kf)
            out = new(n, torch.int32)
            _lib.check(lib.taoamd_flat_compose(n, _ptr(outer), _ptr(inner), _ptr(out),
                                               _stream()), "taoamd_flat_compose")
            return out
        order_s = resort(order_v, frame_key)         # frame order (T/tao.py:181-187)
        order_p = resort(order_s, pos_key)           # timeline order, last box wins
        rs = _Runs(lib, dev, n, trk_sel, order_s)
        R = rs.n
        sel_area = torch.zeros(max(R, 1), dtype=torch.float64, device=dev)
        sel_first = torch.full((max(R, 1),), -np.inf, dtype=torch.float64, device=dev)
        sel_len, sel_frames = new(R, torch.int32), new(R, torch.int32)
        _lib.check(lib.taoamd_flat_track_sel(
            R, _ptr(rs.run_key), _ptr(rs.run_start), n, _ptr(order_s), _ptr(d_area),
            _ptr(visit_key), _ptr(d_img), _ptr(sel_area), _ptr(sel_len),
            _ptr(sel_frames), _ptr(sel_first), _stream()), "taoamd_flat_track_sel")
        # distinct images are counted on the timeline-ordered list
        sel_area_p = torch.zeros(max(R, 1), dtype=torch.float64, device=dev)
        sel_first_p = torch.zeros(max(R, 1), dtype=torch.float64, device=dev)
        sel_len_p = new(R, torch.int32)
        _lib.check(lib.taoamd_flat_track_sel(
            R, _ptr(rs.run_key), _ptr(rs.run_start), n, _ptr(order_p), _ptr(d_area),
            _ptr(visit_key), _ptr(d_img), _ptr(sel_area_p), _ptr(sel_len_p),
            _ptr(sel_frames), _ptr(sel_first_p), _stream()), "taoamd_flat_track_sel")

        # ---- federated filter, track order
        if ready is None:
            ready = _gt_ready(gt, "tao")     # (the annotation part: waits for it)
            T = ready.T
        keys_g, gkeys = ready.keys_g, ready.gkeys
        key, flags = new(R, torch.int32), new(R, torch.uint8)
        n_keep_t = new(1, torch.int32)
        _lib.check(lib.taoamd_flat_track_filter(
            R, _ptr(rs.run_key), _ptr(sel_len), _ptr(trk_first), _ptr(d_cat),
            _ptr(merged_id), _ptr(raw["video_id"]), _ptr(tid), U,
            _ptr(up(vid_ids, np.int64)), len(gkeys), _ptr(up(gkeys, np.int32)),
            _ptr(up(T.vid_row, np.int32)), _ptr(up(gt.vid_neg_off, np.int64)),
            _ptr(up(gt.vid_neg, np.int64)), _ptr(up(gt.vid_nel_off, np.int64)),
            _ptr(up(gt.vid_nel, np.int64)), _ptr(key), _ptr(flags), _ptr(n_keep_t),
            _ptr(st2), _stream()), "taoamd_flat_track_filter")
        s2 = st2.cpu().numpy()
        if s2[0] != I32_MAX or s2[2] != I32_MAX or s2[3] != I32_MAX:
            import os
            if os.environ.get("TAOAMD_DEBUG"):
                print("flatten_tao_device status", s2, "R", R, "n_trk", n_trk)
            reject()
        required_average = bool(s2[1])
        neg_coords = int(n_bad.item())
        n_sel_tracks = int((sel_len[:R] > 0).sum().item()) if R else 0
        if n_sel_tracks == 0:
            raise ValueError("Found no predicted annotations for given params")
        n_keep = int(n_keep_t.item())
        tsorter = _Sorter(R, dev)
        zr = torch.zeros(max(R, 1), dtype=torch.int32, device=dev)
        order_a = tsorter(R, zr, sel_first)              # first appearance
        ka, ta, sa = new(R, torch.int32), new(R, torch.int32), new(R, torch.float64)
        _lib.check(lib.taoamd_flat_gather_cols(
            R, _ptr(order_a), _ptr(key), _ptr(ka), None, None, None, None,
            _stream()), "taoamd_flat_gather_cols")
        # score of run r = trk_score[track of run r]
        run_trk = torch.where(rs.run_key[:max(R, 1)] == I32_MAX,
                              torch.zeros_like(rs.run_key[:max(R, 1)]),
                              rs.run_key[:max(R, 1)]).contiguous()
        run_score = new(R, torch.float64)
        _lib.check(lib.taoamd_flat_gather_cols(
            R, _ptr(run_trk), None, None, _ptr(trk_score), _ptr(run_score), None,
            None, _stream()), "taoamd_flat_gather_cols")
        _lib.check(lib.taoamd_flat_gather_cols(
            R, _ptr(order_a), _ptr(run_trk), _ptr(ta), _ptr(run_score), _ptr(sa),
            None, None, _stream()), "taoamd_flat_gather_cols")
        order_b = tsorter(R, ka, sa)
        final = new(R, torch.int32)                      # final position -> run
        _lib.check(lib.taoamd_flat_compose(R, _ptr(order_a), _ptr(order_b),
                                           _ptr(final), _stream()),
                   "taoamd_flat_compose")

        # ---- tables of the kept tracks in final order
        m = max(n_keep, 1)
        dt_key, dt_len, dt_nfr = new(m, torch.int32), new(m, torch.int32), new(m + 1, torch.int32)
        dt_score, dt_area = new(m, torch.float64), new(m, torch.float64)
        dt_flags, dt_first = new(m, torch.uint8), new(m, torch.int32)
        g = lib.taoamd_flat_gather_cols
        _lib.check(g(n_keep, _ptr(final), _ptr(key), _ptr(dt_key), _ptr(run_score),
                     _ptr(dt_score), _ptr(flags), _ptr(dt_flags), _stream()), "gather")
        _lib.check(g(n_keep, _ptr(final), _ptr(sel_len), _ptr(dt_len), _ptr(sel_area),
                     _ptr(dt_area), None, None, _stream()), "gather")
        _lib.check(g(n_keep, _ptr(final), _ptr(sel_frames), _ptr(dt_nfr), None, None,
                     None, None, _stream()), "gather")
        trk_of_final = new(m, torch.int32)
        _lib.check(g(n_keep, _ptr(final), _ptr(run_trk), _ptr(trk_of_final), None, None,
                     None, None, _stream()), "gather")
        _lib.check(g(n_keep, _ptr(trk_of_final), _ptr(trk_first), _ptr(dt_first), None,
                     None, None, None, _stream()), "gather")
        frame_off = new(m + 1, torch.int32)
        scratch = new(4, torch.int32)
        _lib.check(lib.taoamd_flat_scan(n_keep, _ptr(dt_nfr), _ptr(frame_off),
                                        _ptr(scratch), _stream()), "taoamd_flat_scan")
        n_frames = int(frame_off[n_keep].item())
        frame_pos = new(n_frames, torch.int32)
        frame_box = torch.empty((max(n_frames, 1), 4), dtype=torch.float64, device=dev)
        _lib.check(lib.taoamd_flat_frames(
            n_keep, _ptr(final), _ptr(rs.run_start), R, n, _ptr(order_p), _ptr(d_img),
            _ptr(t_tl_pos), _ptr(raw["bbox"]), _ptr(frame_off), _ptr(frame_pos),
            _ptr(frame_box), _stream()), "taoamd_flat_frames")
        dt_cat = torch.div(dt_key[:n_keep], U, rounding_mode="floor").to(torch.int32)
        cell_keys, dt_cell, d_off, g_cell, g_off = _cells_from_runs(
            lib, dev, n_keep, dt_key, gkeys, keys_g)
        dt_id = tid[dt_first[:n_keep].long()]

    n_cells = len(cell_keys)
    g_fpos, g_fbox, g_foff = ready.frames

    f = DeviceFlat()
    f.kind = "tao"
    f.vid_ids, f.cat_ids = vid_ids, cat_ids
    f.required_average = required_average
    f.neg_coords = neg_coords       # kept boxes with a negative corner / empty side
    f.n_cells = n_cells
    f.use_cats = True
    f.cell_unit = (cell_keys % U).astype(I32)
    f.cell_cat = (cell_keys // U).astype(I32)
    f.cell_dt_off = d_off.astype(I32)
    f.cell_gt_off = g_off.astype(I32)
    iou_off = np.zeros(n_cells + 1, dtype=np.int64)
    np.cumsum(np.diff(d_off) * np.diff(g_off), out=iou_off[1:])
    f.cell_iou_off = iou_off
    f.update(ready.tables)
    f.gt_cell = g_cell.astype(I32)
    f.gt_frame_off, f.gt_frame_pos, f.gt_frame_box = g_foff, g_fpos, g_fbox
    f.n_pairs = int(iou_off[-1])
    f.tl_image_id, f.tl_vid_start = T.tl_image_id, T.tl_vid_start
    f.dev.update(dt_score=dt_score[:n_keep], dt_area=dt_area[:n_keep],
                 dt_len=dt_len[:n_keep], dt_flags=dt_flags[:n_keep],
                 dt_id=dt_id, dt_cat=dt_cat, dt_cell=dt_cell,
                 dt_frame_off=frame_off[:n_keep + 1],
                 dt_frame_pos=frame_pos[:n_frames], dt_frame_box=frame_box[:n_frames])

    def track_scores(f):
        # tracks with at least one box left after the top-max_dets cut
        live = (trk_first[:n_trk] >= 0).cpu().numpy()
        ids = tr.run_key[:n_trk].cpu().numpy()[live]
        return dict(zip(ids.tolist(),
                        trk_score[:n_trk].cpu().numpy()[live].tolist()))
    f.lazy["track_scores"] = track_scores

    def cell_span(f):
        span = np.zeros(n_cells, dtype=np.int64)
        for off, pos, coff in ((f.dt_frame_off, f.dt_frame_pos, d_off),
                               (g_foff, g_fpos, g_off)):
            off = np.asarray(off, dtype=np.int64)
            pos = np.asarray(pos)
            ts = np.zeros(len(off) - 1, dtype=np.int64)
            has = np.diff(off) > 0
            ts[has] = pos[off[1:][has] - 1].astype(np.int64) + 1
            ne = np.diff(coff) > 0
            cs = np.zeros(n_cells, dtype=np.int64)
            if ne.any():
                cs[ne] = np.maximum.reduceat(ts, coff[:-1][ne])
            span = np.maximum(span, cs)
        return span.astype(I32)
    f.lazy["cell_span"] = cell_span
    return f


def _cuda(device):
    if device is None:
        device = "cuda"
    dev = torch.device(device)
    return dev if dev.type == "cuda" and torch.cuda.is_available() else None


def flatten_lvis(gt, dt, max_dets=MAX_DETS, use_cats=True, device=None, share=False):
    """The image-level cell tables: built on the device when the input allows
    it, by flatten.py otherwise (class-agnostic cells, > 2^31 keys, no GPU for
    host-side tooling).  ``share``: one rank's block of a multi-GPU run, which
    may hold no prediction at all (flatten.flatten_lvis)."""
    dev = _cuda(device)
    if dev is not None and use_cats and len(dt):
        try:
            return flatten_lvis_device(gt, dt, dev, max_dets)
        except Unsupported:
            pass
    return flatten.flatten_lvis(gt, dt, max_dets, use_cats=use_cats, share=share)


def flatten_tao(gt, dt, max_dets=MAX_DETS, use_cats=True, device=None,
                visit_universe=None):
    """The track-level cell tables (see flatten_lvis)."""
    dev = _cuda(device)
    if dev is not None and use_cats and len(dt):
        try:
            return flatten_tao_device(gt, dt, dev, max_dets, visit_universe)
        except (Unsupported, Rejected):
            pass
    return flatten.flatten_tao(gt, dt, max_dets, use_cats=use_cats,
                               visit_universe=visit_universe)
