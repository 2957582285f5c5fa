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
import weakref

import numpy as np
import torch

from . import _lib, flatten
from .flatten import DT_IGNORE_UNMATCHED, Flat, I32, MAX_DETS   # noqa: F401

class Unsupported(Exception):
    """The input is outside what the device build handles (more than 2^31
    cell keys or boxes): the caller falls back to flatten.py."""


class DeviceFlat(Flat):
    """Flat whose per-detection tables live on the device (``dev``); reading
    one as an attribute downloads it once (views, tests, the oracle).  ``lazy``
    holds host-side thunks for tables only the class API's views read."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "dev", {})
        object.__setattr__(self, "lazy", {})

    def __getitem__(self, k):
        if dict.__contains__(self, k):
            return dict.__getitem__(self, k)
        if k in self.lazy:
            v = self.lazy[k]()
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


def raw_columns(dt, device):
    """The prediction columns on the device, uploaded once per DTColumns and
    shared by the image-level and the track-level build."""
    dev = torch.device(device)
    key = id(dt)
    ent = _RAW.get(key)
    if ent is None or ent[0]() is not dt:
        ent = (weakref.ref(dt, lambda _r, k=key: _RAW.pop(k, None)), {})
        _RAW[key] = ent
    if dev not in ent[1]:
        cols = {}
        for name in ("image_id", "category_id", "score", "bbox"):
            cols[name] = torch.from_numpy(
                np.ascontiguousarray(getattr(dt, name))).to(dev, non_blocking=True)
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


def _cells_from_runs(lib, dev, n_keep, dt_key, gkeys_sorted):
    """Union of the detection cells (runs of dt_key) and the ground-truth cells;
    returns (cell_keys, dt_cell tensor, d_off, cells of the GT keys)."""
    run_id = torch.empty(max(n_keep, 1), dtype=torch.int32, device=dev)
    run_key = torch.empty(max(n_keep, 1), dtype=torch.int32, device=dev)
    run_start = torch.empty(max(n_keep, 1), dtype=torch.int32, device=dev)
    n_runs = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = int(lib.taoamd_flat_runs_workspace(n_keep))
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
    _lib.check(lib.taoamd_flat_runs(
        n_keep, _ptr(dt_key), _ptr(run_id), _ptr(run_key), _ptr(run_start),
        _ptr(n_runs), _ptr(ws), wsb, _stream()), "taoamd_flat_runs")
    nr = int(n_runs.item())
    rk = run_key[:nr].cpu().numpy()
    rs = run_start[:nr].cpu().numpy().astype(np.int64)
    cell_keys = np.union1d(gkeys_sorted, rk)
    n_cells = len(cell_keys)
    map_d = np.searchsorted(cell_keys, rk).astype(np.int32)
    cnt = np.zeros(n_cells, dtype=np.int64)
    cnt[map_d] = np.diff(np.r_[rs, n_keep])
    d_off = np.zeros(n_cells + 1, dtype=np.int64)
    np.cumsum(cnt, out=d_off[1:])
    dt_cell = torch.empty(max(n_keep, 1), dtype=torch.int32, device=dev)
    if nr:
        _lib.check(lib.taoamd_flat_remap(
            n_keep, _ptr(run_id), _ptr(torch.from_numpy(map_d).to(dev)),
            _ptr(dt_cell), _stream()), "taoamd_flat_remap")
    return cell_keys, dt_cell[:n_keep], d_off


def flatten_lvis_device(gt, dt, device="cuda", max_dets=MAX_DETS):
    """flatten.flatten_lvis(gt, dt, max_dets) with the detection side built on
    the device.  Raises Unsupported for inputs the kernels do not cover."""
    if len(dt) == 0:
        raise IndexError("list index out of range")  # L/results.py:42
    lib = _lib.load()
    dev = torch.device(device)
    G = flatten.lvis_gt_side(gt)
    img_ids, cat_ids = G.img_ids, G.cat_ids
    U, K, n = len(img_ids), len(cat_ids), len(dt)
    if U == 0 or K == 0 or K * U >= 2 ** 31 - 1 or n >= 2 ** 31 - 1:
        raise Unsupported("cell keys do not fit 31 bits")
    keys_g = G.g_cat * U + G.g_img
    og = flatten.sort_key_score(keys_g)
    g_sel, keys_g = G.g_sel[og], keys_g[og]
    gkeys = np.unique(keys_g).astype(np.int32)

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
            _ptr(img_start), _ptr(status), _stream()), "taoamd_flat_map")
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
        cell_keys, dt_cell, d_off = _cells_from_runs(lib, dev, n_keep, dt_key, gkeys)

    n_cells = len(cell_keys)
    g_cell = np.searchsorted(cell_keys, keys_g)
    g_off = np.zeros(n_cells + 1, dtype=np.int64)
    np.cumsum(np.bincount(g_cell, minlength=n_cells), out=g_off[1:])

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
    flatten.lvis_gt_tables(f, gt, g_sel, keys_g, U)
    f.gt_cell = g_cell.astype(I32)
    f.n_pairs = int(np.sum(np.diff(d_off) * np.diff(g_off)))
    f.dev.update(dt_box=dt_box[:n_keep], dt_score=dt_score[:n_keep],
                 dt_flags=dt_flags[:n_keep], dt_cat=dt_cat[:n_keep],
                 dt_cell=dt_cell, dt_row=dt_row[:n_keep])

    def dt_id():
        # id = 1 + position in the post-truncation list (L/results.py:73-84):
        # only the class API's views read it
        keep = flatten.limit_dets_per_image(dt, max_dets)
        pos = np.full(len(dt), -1, dtype=np.int64)
        pos[keep] = np.arange(len(keep))
        return pos[f.dt_row.astype(np.int64)] + 1
    f.lazy["dt_id"] = dt_id
    return f
