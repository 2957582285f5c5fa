"""Host "flatten" stage: columnar inputs -> cell tables for the HIP kernels.

The reference answers every ``(image, category)`` / ``(video, category)``
query through dict-of-list maps and enumerates *all* I x C (resp. V x C) cells,
most of them empty (reference L/eval.py:132-145, T/eval.py:264-276; SURVEY.md
section 3.2).  Here only non-empty cells exist: detections, ground truths and
tracks are laid out contiguously per cell (CSR offsets), in exactly the order
the reference would visit them, so that the device kernels reproduce its
tie-breaking without any dictionary.

Order rules reproduced (L/ = tao_amodal/evaluation/lvis_amodal/, T/ =
tao_amodal/evaluation/tao_amodal/):

* top-300 per image by per-box score, stable, only when the image has more
  than 300 boxes; ids = post-truncation position + 1 (L/results.py:39-52,73-84;
  T/results.py:56-81,121-132)
* strict ``0 < area < inf`` and known-category filter (L/lvis.py:90-96,
  T/tao.py:247-253)
* federated filter: keep a detection only if its category is in the
  image's/video's ``neg_category_ids`` or has ground truth there
  (L/eval.py:99-103, T/eval.py:228-233)
* cells ordered CATEGORY-MAJOR: (sorted category id, sorted image id | sorted
  video id).  The reference walks cells unit-major while matching and
  category-major while accumulating (L/eval.py:339-346); only the second
  order is observable (it fixes the tie order of equal scores), and laying the
  cells out that way makes every category a contiguous run of detections and
  of ground truths.  Detections inside a cell by descending score, stable
  (L/eval.py:175, T/eval.py:313); ground truth in visiting order
* TAO visiting order = CPython iteration order of
  ``set(video_images) & set(video_images)`` (T/tao.py:230) -- obtained here by
  building that very set -- then first appearance of each track id
  (T/tao.py:172-188)
* category merge on the TAO side only (T/tao.py:115-118, T/results.py:47-50)
* track score = ``np.mean`` of its boxes' scores when they differ
  (T/results.py:88-98), computed with numpy itself (pairwise summation)
* track area = left-to-right ``sum(area)/len`` over annotations sorted by
  frame_index (T/tao.py:181-187)
"""
import sys

import numpy as np

from .columns import DTColumns, GTColumns

MAX_DETS = 300
I32 = np.int32

# flag bits shared with csrc/ (keep in sync with include/tao_amodal_hip.h)
GT_IGNORE = 1       # annotation/track carries a truthy "ignore"
GT_OOF = 2          # annotation is out of frame (LVIS side only)
GT_ID_HIDDEN = 4    # id equals the evaluator's "unmatched" sentinel
DT_IGNORE_UNMATCHED = 1   # unmatched detection is ignored (not exhaustive...)
DT_NO_CONSUME = 2         # id <= 0: a match does not mark the GT as taken


_HOST_LIB = None


def _host_lib():
    """libtao_amodal_ingest.so (native host helpers) or False when not built."""
    global _HOST_LIB
    if _HOST_LIB is None:
        import ctypes as C
        import os
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                          "libtao_amodal_ingest.so")
        _HOST_LIB = False
        if os.path.exists(so):
            lib = C.CDLL(so)
            vp, i64 = C.c_void_p, C.c_int64
            sigs = {
                "taoamd_host_sort_key_score": [i64, vp, vp, vp],
                "taoamd_host_lookup": [i64, vp, i64, vp, vp],
                "taoamd_host_take": [C.c_int32, i64, vp, i64, vp, vp],
                "taoamd_host_seq_mean": [i64, vp, vp, vp],
                "taoamd_host_pyset_self_and": [i64, vp, vp, vp],
                "taoamd_host_track_clash": [i64, vp, vp, vp],
                "taoamd_host_count_bad_boxes": [i64, vp, vp],
                "taoamd_host_threads": [],
                "taoamd_host_thread_cap": [C.c_int32],
            }
            # (a library of another build -- a symbol missing -- is not used at
            # all: the numpy statements of these helpers are the fallback)
            if all(hasattr(lib, name) for name in sigs):
                for name, args in sigs.items():
                    getattr(lib, name).argtypes = args
                    getattr(lib, name).restype = C.c_int
                _HOST_LIB = lib
    return _HOST_LIB


_NATIVE_MIN = 50000       # elements from which a call into the host library pays


def take(src, idx):
    """``src[idx]`` (1-D index array into the first axis) on all threads of the
    host library for the large gathers of the ground-truth halves; numpy for
    small inputs, unsupported element sizes or without the library.  Negative
    (wrapping) indices go through numpy as well."""
    n = len(idx)
    lib = _host_lib() if n >= _NATIVE_MIN else False
    if lib and isinstance(src, np.ndarray) and src.flags.c_contiguous:
        elem = src.dtype.itemsize * int(np.prod(src.shape[1:], dtype=np.int64))
        if elem in (1, 4, 8, 32) and len(src):
            ix = np.ascontiguousarray(idx, dtype=np.int64)
            out = np.empty((n,) + src.shape[1:], dtype=src.dtype)
            rc = lib.taoamd_host_take(elem, len(src), src.ctypes.data, n,
                                      ix.ctypes.data, out.ctypes.data)
            if rc == 0:
                return out
    return src[idx]


def count_bad_boxes(bbox):
    """How many (x, y, w, h) rows have ``x < 0``, ``y < 0``, ``w <= 0`` or
    ``h <= 0`` (the reference's warning, tao_amodal/tao.py:143-158): one pass of
    the host library's threads for large tables, numpy otherwise."""
    b = np.asarray(bbox)
    n = len(b)
    if n == 0:
        return 0
    lib = _host_lib() if n >= _NATIVE_MIN else False
    if lib and b.dtype == np.float64 and b.ndim == 2 and b.shape[1] == 4 \
            and b.flags.c_contiguous:
        out = np.zeros(1, dtype=np.int64)
        if lib.taoamd_host_count_bad_boxes(n, b.ctypes.data, out.ctypes.data) == 0:
            return int(out[0])
    return int(np.count_nonzero((b[:, 0] < 0) | (b[:, 1] < 0)
                                | (b[:, 2] <= 0) | (b[:, 3] <= 0)))


def sort_key_score(key, score=None):
    """``np.lexsort((arange(n), -score, key))`` (``score=None``: stable argsort
    of key) through the parallel native sort when the host library is built
    and the input is large enough to pay for the call."""
    n = len(key)
    lib = _host_lib() if n >= _NATIVE_MIN else False
    if not lib:
        if score is None:
            return np.argsort(key, kind="stable")
        return np.lexsort((np.arange(n), -np.asarray(score, dtype=np.float64), key))
    key = np.ascontiguousarray(key, dtype=np.int64)
    sc = None if score is None else np.ascontiguousarray(score, dtype=np.float64)
    order = np.empty(n, dtype=np.int64)
    rc = lib.taoamd_host_sort_key_score(n, key.ctypes.data,
                                        None if sc is None else sc.ctypes.data,
                                        order.ctypes.data)
    assert rc == 0
    return order


def _lookup(sorted_keys, values):
    """index of each value in sorted_keys (ascending, unique), -1 when absent"""
    if len(sorted_keys) == 0:
        return np.full(len(values), -1, dtype=np.int64)
    lib = _host_lib() if len(values) >= _NATIVE_MIN else False
    if lib and np.asarray(sorted_keys).dtype.kind in "iu" and \
            np.asarray(values).dtype.kind in "iu":
        k = np.ascontiguousarray(sorted_keys, dtype=np.int64)
        v = np.ascontiguousarray(values, dtype=np.int64)
        out = np.empty(len(v), dtype=np.int64)
        rc = lib.taoamd_host_lookup(len(k), k.ctypes.data, len(v), v.ctypes.data,
                                    out.ctypes.data)
        assert rc == 0
        return out
    lo, hi = int(sorted_keys[0]), int(sorted_keys[-1])
    if len(values) > 4096 and hi - lo < max(8 * len(values), 1 << 22):
        # ids in a modest range (image / category / track ids): one gather
        # through a dense table instead of a binary search per value
        table = np.full(hi - lo + 2, -1, dtype=np.int64)
        table[sorted_keys - lo] = np.arange(len(sorted_keys))
        v = np.asarray(values, dtype=np.int64) - lo
        ok = (v >= 0) & (v <= hi - lo)
        return table[np.where(ok, v, hi - lo + 1)]
    pos = np.searchsorted(sorted_keys, values)
    pos = np.minimum(pos, len(sorted_keys) - 1)
    return np.where(sorted_keys[pos] == values, pos, -1)


def _last_with_same_id(ids):
    """The reference resolves annotations through a dict keyed by id, so a
    duplicated id silently aliases the *last* annotation carrying it."""
    if len(ids) < 2 or np.all(ids[1:] > ids[:-1]) or \
            len(np.unique(ids)) == len(ids):
        return np.arange(len(ids))
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    last = np.flatnonzero(np.r_[sid[1:] != sid[:-1], True])
    grp = np.cumsum(np.r_[0, (sid[1:] != sid[:-1]).astype(np.int64)])
    out = np.empty(len(ids), dtype=np.int64)
    out[order] = order[last[grp]]
    return out


def first_inverse(ids):
    """(uniq, first, inv) of ``np.unique(ids, return_index=True,
    return_inverse=True)``.  Ids in a small non-negative range (the usual
    case: image / track / video ids) are handled with dense tables instead of
    a sort."""
    n = len(ids)
    if n and ids.min() >= 0 and ids.max() < max(4 * n, 1 << 22):
        first_of = np.full(int(ids.max()) + 1, -1, dtype=np.int64)
        first_of[ids[::-1]] = np.arange(n - 1, -1, -1)   # smallest index wins
        present = np.flatnonzero(first_of >= 0)
        rank = np.full(len(first_of), -1, dtype=np.int64)
        rank[present] = np.arange(len(present))
        return present, first_of[present], rank[ids]
    uniq, first, inv = np.unique(ids, return_index=True, return_inverse=True)
    return uniq, first, inv.reshape(-1)


def limit_dets_per_image(dt, max_dets=MAX_DETS):
    """Permutation of kept detections in post-truncation list order (images
    in first-seen order; an image with more than max_dets boxes keeps its
    best max_dets by score, stable -- reference L/results.py:73-84)."""
    n = len(dt)
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    # (the cut is reused for the same arrays: a caller that rebinds a column
    # gets a fresh one; editing a column in place after a cut is not seen)
    ident = tuple((id(c), c.__array_interface__["data"][0]) if isinstance(c, np.ndarray)
                  else id(c) for c in (dt.image_id, dt.score))
    cache = getattr(dt, "_limit_cache", None)
    if cache is not None and cache[0] == (max_dets, n, ident):
        return cache[1]
    uniq, first, inv = first_inverse(dt.image_id)
    cnt = np.bincount(inv, minlength=len(uniq))
    rank_of_img = np.empty(len(uniq), dtype=np.int64)
    rank_of_img[np.argsort(first, kind="stable")] = np.arange(len(uniq))
    img_rank = rank_of_img[inv]
    if max_dets >= 0 and (cnt > max_dets).any():
        big = (cnt > max_dets)[inv]
        order = sort_key_score(img_rank, np.where(big, dt.score, 0.0))
        r = img_rank[order]
        start = np.flatnonzero(np.r_[True, r[1:] != r[:-1]])
        pos = np.arange(n) - np.repeat(start, np.diff(np.r_[start, n]))
        order = order[pos < max_dets]
    else:
        order = sort_key_score(img_rank)
    try:
        dt._limit_cache = ((max_dets, n, ident), order)
    except AttributeError:
        pass
    return order


def make_track_ids_unique(dt):
    """tools/eval_on_tao_amodal.py:44-66 on columns; returns (new track_id
    array, number of ids that had to change)."""
    n = len(dt)
    tid, vid = dt.track_id, dt.video_id
    if n == 0:
        return tid.copy(), 0
    lib = _host_lib() if n >= _NATIVE_MIN else False
    if lib and tid.dtype == np.int64 and vid.dtype == np.int64 \
            and tid.flags.c_contiguous and vid.flags.c_contiguous:
        # the usual answer -- no id is shared between videos -- on all threads
        # (the statement below: 0.26 s of numpy at 30 M predictions); the
        # column is then returned as it is, not copied
        import ctypes as C
        clash = C.c_int64(0)
        if lib.taoamd_host_track_clash(n, tid.ctypes.data, vid.ctypes.data,
                                       C.addressof(clash)) == 0 and clash.value == 0:
            return tid, 0
    uniq, first, inv = first_inverse(tid)
    clash_t = np.bincount(inv, weights=(vid != vid[first][inv]),
                          minlength=len(uniq)) > 0
    if not clash_t.any():
        return tid.copy(), 0
    top = max(int(tid.max()), 0)
    sel = np.flatnonzero(clash_t[inv])
    pair = np.stack([tid[sel], vid[sel]], 1)
    _, pfirst, pinv = np.unique(pair, axis=0, return_index=True,
                                return_inverse=True)
    pinv = pinv.reshape(-1)
    new_of_pair = np.empty(len(pfirst), dtype=np.int64)
    new_of_pair[np.argsort(pfirst, kind="stable")] = \
        top + 1 + np.arange(len(pfirst))
    out = tid.copy()
    out[sel] = new_of_pair[pinv]
    return out, int(clash_t.sum())


class LazyRows:
    """``source[index]`` (rows of an (N, 4) float64 table), not gathered yet.

    The gathers of the kept detections' boxes are the largest single step of
    the cell-table build on the host; the device has to receive those rows
    anyway, so ``engine.DeviceProblem`` uploads the raw table once (shared by
    the image-level and the track-level problem) and gathers there.  On the
    host the object turns into the gathered array on demand (``np.asarray``)."""

    def __init__(self, source, index):
        self.source = source
        self.index = np.asarray(index, dtype=np.int64)

    @property
    def shape(self):
        return (len(self.index),) + tuple(self.source.shape[1:])

    @property
    def dtype(self):
        return self.source.dtype

    def __len__(self):
        return len(self.index)

    def __array__(self, dtype=None, copy=None):
        a = np.ascontiguousarray(self.source[self.index])
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, key):
        if isinstance(key, slice):
            return LazyRows(self.source, self.index[key])
        return np.asarray(self)[key]


class Flat(dict):
    """A bag of arrays with attribute access."""
    __setattr__ = dict.__setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


DENSE_TABLE_MAX = 1 << 28     # entries of a boolean membership table (256 MB)


def _member(keys, universe, queries):
    """queries[k] in keys, for non-negative integer keys < universe."""
    if universe <= DENSE_TABLE_MAX:
        table = np.zeros(universe + 1, dtype=bool)      # last slot: "absent"
        table[keys[(keys >= 0) & (keys < universe)]] = True
        ok = (queries >= 0) & (queries < universe)
        return table[np.where(ok, queries, universe)]
    return _lookup(np.unique(keys), queries) >= 0


def _csr_member(off, val, row, item):
    """item[k] in val[off[row[k]]:off[row[k]+1]] for every k (one gather
    through a boolean table over (row, value); a sorted key table when that
    would be too large)."""
    if len(val) == 0 or len(row) == 0:
        return np.zeros(len(row), dtype=bool)
    rows = np.repeat(np.arange(len(off) - 1), np.diff(off))
    lo = min(int(val.min()), int(item.min()))
    width = max(int(val.max()), int(item.max())) - lo + 1
    if width * (len(off) + 1) < (1 << 62):
        return _member(rows * width + (val - lo), width * (len(off) - 1),
                       row * width + (item - lo))
    keys = np.stack([rows, val], 1)
    q = np.stack([row, item], 1)
    kd = np.ascontiguousarray(keys).view([("a", np.int64), ("b", np.int64)])
    qd = np.ascontiguousarray(q).view([("a", np.int64), ("b", np.int64)])
    return np.isin(qd.reshape(-1), kd.reshape(-1))


def _cells(keys_gt, keys_dt):
    cell_keys = np.union1d(keys_gt, keys_dt)
    g_cell = np.searchsorted(cell_keys, keys_gt)
    d_cell = np.searchsorted(cell_keys, keys_dt)
    n = len(cell_keys)
    g_off = np.zeros(n + 1, dtype=np.int64)
    d_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(g_cell, minlength=n), out=g_off[1:])
    np.cumsum(np.bincount(d_cell, minlength=n), out=d_off[1:])
    return cell_keys, g_cell, d_cell, g_off, d_off


# ---------------------------------------------------------------------------
# image level (LVISEval)
# ---------------------------------------------------------------------------
def lvis_gt_side(gt):
    """Ground-truth half of the image-level tables: sorted unique image and
    category ids, the dataset row of every image (dict semantics: the last one
    with an id wins), and the selected annotations in sorted-image order,
    dataset order inside (L/lvis.py:34-61,90-96)."""
    img_ids = np.unique(gt.img_id)
    cat_ids = np.unique(gt.cat_id)
    # the reference keeps images in a dict keyed by id: last one wins
    img_row = np.full(len(img_ids), -1, dtype=np.int64)
    img_row[_lookup(img_ids, gt.img_id)] = np.arange(len(gt.img_id))
    a_img = _lookup(img_ids, gt.ann_img)
    a_cat = _lookup(cat_ids, gt.ann_cat)
    alias = _last_with_same_id(gt.ann_id)
    g_sel = np.flatnonzero(a_img >= 0)
    g_sel = g_sel[sort_key_score(take(a_img, g_sel))]       # (stable)
    area = take(gt.ann_area, g_sel)
    g_sel = g_sel[(take(a_cat, g_sel) >= 0) & (area > 0) & (area < np.inf)]
    g_sel = take(alias, g_sel)
    G = Flat()
    G.img_ids, G.cat_ids, G.img_row = img_ids, cat_ids, img_row
    G.g_sel, G.g_img, G.g_cat = g_sel, take(a_img, g_sel), take(a_cat, g_sel)
    return G


def lvis_gt_tables(f, gt, g_sel, keys_g, U):
    """Ground-truth columns of a flattened image-level problem (rows g_sel,
    already in final order)."""
    f.gt_row = g_sel                # row of dataset["annotations"]
    f.gt_box = np.ascontiguousarray(take(gt.ann_bbox, g_sel))
    f.gt_vis = np.ascontiguousarray(take(gt.ann_vis, g_sel))
    f.gt_id = take(gt.ann_id, g_sel)
    f.gt_flags = ((take(gt.ann_ignore, g_sel) != 0) * np.uint8(GT_IGNORE)
                  | (take(gt.ann_oof, g_sel) != 0) * np.uint8(GT_OOF)
                  | (f.gt_id == 0) * np.uint8(GT_ID_HIDDEN)).astype(np.uint8)
    f.gt_cat = (keys_g // U).astype(I32)


def flatten_lvis(gt: GTColumns, dt: DTColumns, max_dets=MAX_DETS,
                 use_cats=True, share=False):
    """Cell tables of LVISEval (L/eval.py:59-110).  ``use_cats=False`` builds
    the class-agnostic problem of ``params.use_cats = 0`` (L/eval.py:125-128,
    147-166): one cell per image holding every ground truth / every kept
    detection of the image, category-major (the federated filter and the
    not-exhaustive flags still use the real categories), one output category."""
    # (``share``: one rank's block of a multi-GPU run -- the reference's
    # statements about an empty list are about the WHOLE list, which
    # evaluation/_dist.shard_inputs has checked: a share may hold nothing)
    if len(dt) == 0 and not share:
        raise IndexError("list index out of range")  # L/results.py:42
    G = lvis_gt_side(gt)
    img_ids, cat_ids, img_row = G.img_ids, G.cat_ids, G.img_row
    g_sel, g_img, g_cat = G.g_sel, G.g_img, G.g_cat
    K = len(cat_ids)

    keep = limit_dets_per_image(dt, max_dets)
    # columns of the kept detections (the boxes are gathered once, at the end)
    d_image, d_catid = dt.image_id[keep], dt.category_id[keep]
    d_score = dt.score[keep]
    d_id = np.arange(1, len(keep) + 1, dtype=np.int64)
    d_img = _lookup(img_ids, d_image)
    if (d_img < 0).any():
        raise AssertionError("Results do not correspond to current LVIS set.")
    # area = w * h of the box (L/results.py:51); results that come as masks
    # only bring the mask's area instead (L/results.py:56, DTColumns.area)
    d_area = (dt.bbox[:, 2] * dt.bbox[:, 3])[keep] \
        if getattr(dt, "area", None) is None else \
        np.asarray(dt.area, dtype=np.float64)[keep]

    # ---- detection selection + federated filter
    d_cat = _lookup(cat_ids, d_catid)
    order = np.argsort(d_img, kind="stable")
    order = order[(d_cat[order] >= 0) & (d_area[order] > 0)
                  & (d_area[order] < np.inf)]
    U = len(img_ids)
    k_of = d_cat[order] * U + d_img[order]
    is_present = _member(g_cat * U + g_img, K * U, k_of)
    rows = img_row[d_img[order]]
    is_neg = _csr_member(gt.img_neg_off, gt.img_neg, rows,
                         d_catid[order])
    order = order[is_present | is_neg]

    # ---- cells
    if use_cats:
        keys_g = g_cat * U + g_img
        keys_d = d_cat[order] * U + d_img[order]
        # detections: by cell, then descending score, stable in visiting order
        o2 = sort_key_score(keys_d, d_score[order])
        og = sort_key_score(keys_g)
    else:
        keys_g = g_img.copy()
        keys_d = d_img[order].copy()
        # the image's per-category lists one after the other (ascending
        # category), then the stable sort by score of compute_iou
        o2 = np.lexsort((np.arange(len(order)), d_cat[order], -d_score[order],
                         keys_d))
        og = np.lexsort((np.arange(len(g_sel)), g_cat, keys_g))
    order = order[o2]
    keys_d = keys_d[o2]
    g_sel, keys_g = g_sel[og], keys_g[og]
    cell_keys, g_cell, d_cell, g_off, d_off = _cells(keys_g, keys_d)

    rows = img_row[d_img[order]]
    nel = _csr_member(gt.img_nel_off, gt.img_nel, rows, d_catid[order])
    area = d_area[order]
    d_flags = np.where((area < 0) | (area > 1e5 ** 2) | nel,
                       DT_IGNORE_UNMATCHED, 0).astype(np.uint8)
    g_flags = (np.where(gt.ann_ignore[g_sel] != 0, GT_IGNORE, 0)
               | np.where(gt.ann_oof[g_sel] != 0, GT_OOF, 0)
               | np.where(gt.ann_id[g_sel] == 0, GT_ID_HIDDEN, 0)
               ).astype(np.uint8)

    f = Flat()
    f.kind = "lvis"
    f.use_cats = bool(use_cats)
    f.img_ids = img_ids
    f.cat_ids = cat_ids if use_cats else np.array([-1], dtype=np.int64)
    f.cat_freq = _freq_of(gt, cat_ids)      # of the real categories, always
    f.n_cells = len(cell_keys)
    f.cell_unit = (cell_keys % U).astype(I32)       # image index
    f.cell_cat = (cell_keys // U).astype(I32)       # all zero without categories
    f.cell_dt_off = d_off.astype(I32)
    f.cell_gt_off = g_off.astype(I32)
    f.dt_box = LazyRows(dt.bbox, keep[order])
    f.dt_row = keep[order]          # row of the prediction list
    f.gt_row = g_sel                # row of dataset["annotations"]
    f.dt_score = np.ascontiguousarray(d_score[order])
    f.dt_flags = d_flags
    f.dt_id = d_id[order]
    f.dt_cat = (keys_d // U).astype(I32)
    f.dt_cell = d_cell.astype(I32)
    f.gt_box = np.ascontiguousarray(gt.ann_bbox[g_sel])
    f.gt_vis = np.ascontiguousarray(gt.ann_vis[g_sel])
    f.gt_flags = g_flags
    f.gt_id = gt.ann_id[g_sel]
    f.gt_cat = (keys_g // U).astype(I32)
    f.gt_cell = g_cell.astype(I32)
    f.n_pairs = int(np.sum(np.diff(d_off) * np.diff(g_off)))
    return f


def _freq_of(gt, cat_ids):
    # categories live in a dict keyed by id: last one wins
    row = np.full(len(cat_ids), -1, dtype=np.int64)
    row[_lookup(cat_ids, gt.cat_id)] = np.arange(len(gt.cat_id))
    return gt.cat_freq[row]


# ---------------------------------------------------------------------------
# track level (TaoEval)
# ---------------------------------------------------------------------------
def _seq_track_mean(vals, off):
    """left-to-right sum(vals)/len per CSR segment, the arithmetic of
    ``sum(x['area'] ...) / len(...)`` (T/tao.py:186-187)."""
    lens = np.diff(off)
    acc = np.zeros(len(lens))
    if len(lens) == 0:
        return acc
    lib = _host_lib() if len(vals) >= _NATIVE_MIN else False
    if lib:
        o = np.ascontiguousarray(off, dtype=np.int64)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        with np.errstate(all="ignore"):
            rc = lib.taoamd_host_seq_mean(len(lens), o.ctypes.data, v.ctypes.data,
                                          acc.ctypes.data)
        assert rc == 0
        return acc
    for s in range(int(lens.max())):
        live = np.flatnonzero(lens > s)
        acc[live] = acc[live] + vals[off[live] + s]
    return acc / lens


def _group_tracks(sel_trk, sel_frame_index):
    """Group selected annotations (in visiting order) into tracks.

    Returns (track order = first appearance, per-annotation permutation that
    lists every track's annotations contiguously sorted by frame_index
    (stable), CSR offsets)."""
    uniq, first, inv = first_inverse(sel_trk)
    t_rank = np.empty(len(uniq), dtype=np.int64)
    t_rank[np.argsort(first, kind="stable")] = np.arange(len(uniq))
    trk_of_ann = t_rank[inv]
    fi = np.asarray(sel_frame_index)
    span = 0
    if len(fi) and fi.dtype.kind in "iuf":
        # frame indices that are whole numbers (the usual case; a JSON number
        # parses as a double): (track, frame_index) is one integer key
        with np.errstate(invalid="ignore"):
            whole = fi.astype(np.int64)
        if fi.dtype.kind != "f" or np.array_equal(whole.astype(np.float64), fi):
            base = int(whole.min())
            span = int(whole.max()) - base + 1
    if span and len(uniq) * span < (1 << 32):
        perm = sort_key_score(trk_of_ann * span + (whole - base))   # (radix sort)
    else:
        perm = sort_key_score(trk_of_ann, -np.asarray(fi, dtype=np.float64))
    off = np.zeros(len(uniq) + 1, dtype=np.int64)
    np.cumsum(np.bincount(trk_of_ann, minlength=len(uniq)), out=off[1:])
    ids = np.empty(len(uniq), dtype=np.int64)
    ids[t_rank] = uniq
    return ids, perm, off


def _unique_frames(trk_of, pos_of, n_trk):
    """Per track: one box per image (the *last* annotation in frame_index
    order wins, T/eval.py:322-325), listed by ascending timeline position.
    Inputs are per-annotation arrays already grouped by track.  Returns
    (selection into the annotations, CSR offsets)."""
    n = len(trk_of)
    order = sort_key_score(trk_of, -pos_of.astype(np.float64))
    t, p = trk_of[order], pos_of[order]
    last = np.r_[(t[1:] != t[:-1]) | (p[1:] != p[:-1]), True] if n else \
        np.zeros(0, bool)
    sel = order[last]
    off = np.zeros(n_trk + 1, dtype=np.int64)
    np.cumsum(np.bincount(trk_of[sel], minlength=n_trk), out=off[1:])
    return sel, off


def track_frames(tl_pos, trk_order, trk_of_ann, aoff, ann_rows, img_idx_of_ann,
                 boxes):
    """Frame lists of the tracks `trk_order` (final, cell order).  The
    annotations are grouped by track (CSR `aoff`) in frame_index order."""
    pos = take(tl_pos, img_idx_of_ann)
    if len(pos) > 1:
        rising = pos[1:] > pos[:-1]
        rising[aoff[1:-1][aoff[1:-1] < len(pos)] - 1] = True   # track boundaries
    if len(pos) <= 1 or rising.all():
        # every track already lists distinct images in timeline order (the
        # usual case): the result is just the tracks' segments, permuted
        lens = np.diff(aoff)[trk_order]
        off = np.zeros(len(trk_order) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        rows = np.repeat(aoff[:-1][trk_order] - off[:-1], lens) + \
            np.arange(int(off[-1]))
    else:
        # renumber tracks to their final (cell) position, sort, keep the
        # last annotation of every (track, image)
        new_of_old = np.full(int(trk_of_ann.max()) + 1 if len(trk_of_ann)
                             else 0, -1, dtype=np.int64)
        new_of_old[trk_order] = np.arange(len(trk_order))
        nt = new_of_old[trk_of_ann]
        live = np.flatnonzero(nt >= 0)
        sel, off = _unique_frames(nt[live], pos[live], len(trk_order))
        rows = live[sel]
    return (take(pos, rows).astype(I32), LazyRows(boxes, take(ann_rows, rows)),
            off.astype(I32))


def _tao_select(visit_rank, a_img_idx, a_cat_idx, a_area, a_ids):
    """get_ann_ids(vid_ids, cat_ids) + load_anns (T/tao.py:203-254)"""
    sel = np.flatnonzero(a_img_idx >= 0)
    rank = take(visit_rank, take(a_img_idx, sel))
    sel, rank = sel[rank >= 0], rank[rank >= 0]
    sel = sel[sort_key_score(rank)]
    area = take(a_area, sel)
    sel = sel[(take(a_cat_idx, sel) >= 0) & (area > 0) & (area < np.inf)]
    return take(_last_with_same_id(a_ids), sel)


def visiting_order(image_ids):
    """``list(set(ids) & set(ids))``: the order the reference walks the images
    of the evaluated videos in (T/tao.py:224-230 with img_ids = video_images)
    -- CPython's set iteration order, from the interpreter itself or, for large
    inputs, from the host library's restatement of setobject.c (pinned to the
    interpreter in tests/test_host_primitives.py)."""
    import ctypes as C
    ids = np.ascontiguousarray(image_ids, dtype=np.int64)
    lib = _host_lib() if len(ids) >= _NATIVE_MIN else False
    if lib and sys.version_info[:2] <= (3, 12):
        out = np.empty(len(ids), dtype=np.int64)
        m = C.c_int64(0)
        if lib.taoamd_host_pyset_self_and(len(ids), ids.ctypes.data, out.ctypes.data,
                                          C.addressof(m)) == 0:
            return out[:m.value]
    lst = ids.tolist()
    return np.asarray(list(set(lst) & set(lst)), dtype=np.int64)


def video_images(gt):
    """Image ids in the order the reference collects them for its visiting set
    (videos in sorted-id order, dataset order inside; T/tao.py:224-230)."""
    vid_of_image_row = _lookup(np.unique(gt.vid_id), gt.img_vid)
    by_vid = np.argsort(vid_of_image_row, kind="stable")
    by_vid = by_vid[vid_of_image_row[by_vid] >= 0]
    return gt.img_id[by_vid]


def tao_gt_universe(gt, visit_universe=None):
    """The part of tao_gt_side that does not look at the annotations or the
    tracks: the category merge map, sorted unique ids and dataset rows of
    videos / categories / images, the per-video timeline and the CPython-set
    visiting order of the images.  What the prediction side of the track-level
    tables needs first -- the drop-in CLI builds it ahead of the (three times
    longer) annotation part, which the device build only meets at its
    federated filter (flatten_dev.flatten_tao_device)."""
    merge_src = gt.cat_merged[:, 0] if len(gt.cat_merged) else \
        np.zeros(0, np.int64)
    merge_dst = gt.cat_merged[:, 1] if len(gt.cat_merged) else \
        np.zeros(0, np.int64)
    # later "merged" entries overwrite earlier ones for the same source id
    m_order = np.argsort(merge_src, kind="stable")
    ms, md = merge_src[m_order], merge_dst[m_order]
    keep_last = np.r_[ms[1:] != ms[:-1], True] if len(ms) else np.zeros(0, bool)
    ms, md = ms[keep_last], md[keep_last]

    vid_ids = np.unique(gt.vid_id)
    cat_ids = np.unique(gt.cat_id)
    img_ids = np.unique(gt.img_id)
    K = len(cat_ids)
    vid_row = np.full(len(vid_ids), -1, dtype=np.int64)
    vid_row[_lookup(vid_ids, gt.vid_id)] = np.arange(len(gt.vid_id))
    img_row = np.full(len(img_ids), -1, dtype=np.int64)
    img_row[_lookup(img_ids, gt.img_id)] = np.arange(len(gt.img_id))

    # ---- per-video timeline position of every image: (frame_index, id)
    img_vid_idx = _lookup(vid_ids, gt.img_vid[img_row])
    img_frame = gt.img_frame[img_row]
    tl_order = np.lexsort((img_ids, img_frame, img_vid_idx))
    tl_pos = np.empty(len(img_ids), dtype=np.int64)
    v_sorted = img_vid_idx[tl_order]
    start = np.flatnonzero(np.r_[True, v_sorted[1:] != v_sorted[:-1]])
    tl_pos[tl_order] = np.arange(len(img_ids)) - np.repeat(
        start, np.diff(np.r_[start, len(img_ids)]))
    # image id at (video, timeline position): tl_image_id[tl_vid_start[v] + pos]
    tl_image_id = img_ids[tl_order]
    tl_vid_start = np.searchsorted(v_sorted, np.arange(len(vid_ids) + 1), "left")

    # ---- visiting order of images: CPython set iteration (T/tao.py:224-230)
    own_images = video_images(gt)
    visit = visiting_order(own_images if visit_universe is None
                           else np.asarray(visit_universe, dtype=np.int64))
    # (images of other ranks' videos drop out of the ranking)
    at = _lookup(img_ids, visit)
    visit_rank = np.full(len(img_ids), -1, dtype=np.int64)
    mine = np.flatnonzero((at >= 0) & np.isin(visit, own_images))
    visit_rank[at[mine]] = np.arange(len(mine))
    A = Flat()
    for k, v in list(locals().items()):
        if k not in ("A", "gt", "visit_universe") and not k.startswith("_"):
            A[k] = v
    return A


def tao_gt_side(gt, visit_universe=None, universe=None):
    """Ground-truth half of the track-level tables (T/tao.py:108-254): merged
    categories, sorted unique ids and dataset rows (dict semantics: last one
    wins), the per-video timeline, the CPython-set visiting order of the
    images, and the ground-truth tracks (selected annotations grouped by
    track in first-appearance order, frame order inside).

    ``visit_universe``: when ``gt`` is one rank's share of a larger ground
    truth (by-video partition), the image ids of the WHOLE ground truth in
    video_images() order -- the iteration order of a CPython set depends on
    everything in it, so the visiting order must come from the full set.
    ``universe``: tao_gt_universe(gt, visit_universe) built ahead of time."""
    A = universe if universe is not None else tao_gt_universe(gt, visit_universe)
    ms, md = A.ms, A.md
    vid_ids, cat_ids, img_ids, img_row = A.vid_ids, A.cat_ids, A.img_ids, A.img_row
    visit_rank = A.visit_rank

    def merged(c):
        j = _lookup(ms, c)
        return np.where(j >= 0, md[np.maximum(j, 0)], c) if len(ms) else c

    ann_cat = merged(gt.ann_cat)
    trk_cat = merged(gt.trk_cat)
    # T/tao.py:148-149
    trow = np.full(len(gt.trk_id), -1, dtype=np.int64)
    t_sorted = np.argsort(gt.trk_id, kind="stable")
    t_keys = gt.trk_id[t_sorted]
    # dict semantics: last track with an id wins
    t_last = np.r_[t_keys[1:] != t_keys[:-1], True] if len(t_keys) else \
        np.zeros(0, bool)
    t_keys_u, t_rows_u = t_keys[t_last], t_sorted[t_last]
    a_trow = _lookup(t_keys_u, gt.ann_trk)
    if (a_trow < 0).any():
        raise KeyError(int(gt.ann_trk[np.flatnonzero(a_trow < 0)[0]]))
    a_trow = t_rows_u[a_trow]
    assert np.array_equal(ann_cat, trk_cat[a_trow]), \
        "annotation/track category mismatch"

    a_img = _lookup(img_ids, gt.ann_img)
    g_sel = _tao_select(visit_rank, a_img, _lookup(cat_ids, ann_cat), gt.ann_area,
                        gt.ann_id)
    # (a rank's share -- it comes with the whole set's visiting universe -- may
    # hold no ground truth: the statement is shard_inputs' about the whole set)
    if len(g_sel) == 0 and visit_universe is None:
        raise ValueError("Found no groundtruth annotations for given params")
    g_ids, g_perm, g_aoff = _group_tracks(
        take(gt.ann_trk, g_sel),
        take(gt.img_frame, take(img_row, take(a_img, g_sel))))
    g_ann = take(g_sel, g_perm)                    # annotations, track-major
    g_trk_of_ann = np.repeat(np.arange(len(g_ids)), np.diff(g_aoff))
    g_area = _seq_track_mean(take(gt.ann_area, g_ann), g_aoff)
    g_len = np.diff(g_aoff)
    # annotations below 0.8 visibility per track (every track lists >= 1)
    _low = np.zeros(len(g_ann) + 1, dtype=np.int64)
    np.cumsum(take(gt.ann_vis, g_ann) < 0.8, out=_low[1:])
    g_nhp = _low[g_aoff[1:]] - _low[g_aoff[:-1]]
    g_row = t_rows_u[_lookup(t_keys_u, g_ids)]
    g_vid = _lookup(vid_ids, gt.trk_vid[g_row])
    g_cat = _lookup(cat_ids, trk_cat[g_row])
    g_ign = gt.trk_ignore[g_row]
    if (g_vid < 0).any():
        raise KeyError("track refers to an unknown video")
    T = Flat()
    T.update(A)
    for k, v in list(locals().items()):
        if k not in ("T", "A", "gt", "merged", "universe", "visit_universe") \
                and not k.startswith("_"):
            T[k] = v
    return T


def flatten_tao(gt: GTColumns, dt: DTColumns, max_dets=MAX_DETS,
                use_cats=True, visit_universe=None):
    """``dt.track_id`` must already be unique per video (the CLI runs
    make_track_ids_unique first; T/results.py:111-119 asserts it).

    ``use_cats=False`` builds the class-agnostic problem of
    ``params.use_cats = 0`` (T/eval.py:257-260,293-303): one cell per video
    holding the tracks of all categories (category-major, as the reference
    concatenates them), no federated filter, a single pseudo category -1."""
    share = visit_universe is not None       # one rank's block (see flatten_lvis)
    if len(dt) == 0 and not share:
        raise IndexError("list index out of range")  # T/results.py:61
    T = tao_gt_side(gt, visit_universe)
    ms, md = T.ms, T.md

    def merged(c):
        j = _lookup(ms, c)
        return np.where(j >= 0, md[np.maximum(j, 0)], c) if len(ms) else c

    pred_cat = merged(dt.category_id)
    vid_ids, cat_ids, img_ids, K = T.vid_ids, T.cat_ids, T.img_ids, T.K
    vid_row, img_row = T.vid_row, T.img_row
    trk_cat, tl_pos, visit_rank = T.trk_cat, T.tl_pos, T.visit_rank

    # ---- predictions: TaoResults
    tid = dt.track_id
    u, first, inv = first_inverse(tid)
    if (dt.video_id != dt.video_id[first][inv]).any():
        bad = tid[np.flatnonzero(dt.video_id != dt.video_id[first][inv])[0]]
        raise AssertionError(
            "Track id {} appears in more than one video".format(int(bad)))
    keep = limit_dets_per_image(dt, max_dets)
    # columns of the kept boxes (the boxes themselves are gathered once, when
    # the frame lists are built)
    d_track, d_image = dt.track_id[keep], dt.image_id[keep]
    d_cat_id = pred_cat[keep]
    u, first, inv = first_inverse(d_track)
    if (d_cat_id != d_cat_id[first][inv]).any():
        bad = d_track[np.flatnonzero(d_cat_id != d_cat_id[first][inv])[0]]
        raise AssertionError(
            "Annotations for track {} have multiple categories".format(
                int(bad)))
    d_img = _lookup(img_ids, d_image)
    if (d_img < 0).any():
        raise AssertionError("Results do not correspond to current Tao set.")
    # area = w * h of the box (L/results.py:51); results that come as masks
    # only bring the mask's area instead (L/results.py:56, DTColumns.area)
    d_area = (dt.bbox[:, 2] * dt.bbox[:, 3])[keep] \
        if getattr(dt, "area", None) is None else \
        np.asarray(dt.area, dtype=np.float64)[keep]
    # track score over *all* kept boxes of the track (T/results.py:88-98)
    trk_score = np.empty(len(u))
    by_trk = np.argsort(inv, kind="stable")
    t_off = np.zeros(len(u) + 1, dtype=np.int64)
    np.cumsum(np.bincount(inv, minlength=len(u)), out=t_off[1:])
    sc = dt.score[keep[by_trk]]
    seg_min = np.minimum.reduceat(sc, t_off[:-1])
    seg_max = np.maximum.reduceat(sc, t_off[:-1])
    trk_score[:] = sc[t_off[:-1]]
    required_average = False
    for k in np.flatnonzero(seg_min != seg_max):
        required_average = True
        trk_score[k] = np.mean(sc[t_off[k]:t_off[k + 1]])
    dt_trk_vid = dt.video_id[keep[first]]
    dt_trk_cat = d_cat_id[first]

    a_img, g_sel = T.a_img, T.g_sel
    d_sel = _tao_select(visit_rank, d_img, _lookup(cat_ids, d_cat_id), d_area,
                        np.arange(1, len(keep) + 1, dtype=np.int64))
    if len(d_sel) == 0 and not share:
        raise ValueError("Found no predicted annotations for given params")

    # ---- group into tracks
    g_ids, g_ann, g_aoff, g_trk_of_ann = T.g_ids, T.g_ann, T.g_aoff, T.g_trk_of_ann
    d_ids, d_perm, d_aoff = _group_tracks(
        d_track[d_sel], gt.img_frame[img_row[d_img[d_sel]]])
    d_ann = d_sel[d_perm]
    d_trk_of_ann = np.repeat(np.arange(len(d_ids)), np.diff(d_aoff))

    g_area, g_len, g_nhp = T.g_area, T.g_len, T.g_nhp
    d_area_t = _seq_track_mean(d_area[d_ann], d_aoff)
    d_len = np.diff(d_aoff)

    g_row, g_vid, g_cat, g_ign = T.g_row, T.g_vid, T.g_cat, T.g_ign
    d_trow = _lookup(u, d_ids)
    d_vid_id = dt_trk_vid[d_trow]
    d_vid = _lookup(vid_ids, d_vid_id)
    d_catid = dt_trk_cat[d_trow]
    d_cat = _lookup(cat_ids, d_catid)
    d_score = trk_score[d_trow]
    if (g_vid < 0).any() or (d_vid < 0).any():
        raise KeyError("track refers to an unknown video")

    # ---- federated filter on the video lists (T/eval.py:214-233)
    U = len(vid_ids)
    if use_cats:
        is_present = _member(g_cat * U + g_vid, K * U, d_cat * U + d_vid)
        is_neg = _csr_member(gt.vid_neg_off, gt.vid_neg, vid_row[d_vid],
                             d_catid)
        d_keep = np.flatnonzero(is_present | is_neg)
        keys_g = g_cat * U + g_vid
        keys_d = d_cat[d_keep] * U + d_vid[d_keep]
        o2 = sort_key_score(keys_d, d_score[d_keep])
        og = sort_key_score(keys_g)
    else:
        d_keep = np.arange(len(d_ids))
        keys_g = g_vid.copy()
        keys_d = d_vid.copy()
        # category-major inside the video, then by score (stable)
        o2 = np.lexsort((np.arange(len(d_keep)), d_cat, -d_score, keys_d))
        og = np.lexsort((np.arange(len(g_ids)), g_cat, keys_g))
        cat_ids = np.array([-1], dtype=np.int64)
        U = len(vid_ids)
    d_keep, keys_d = d_keep[o2], keys_d[o2]
    keys_g = keys_g[og]
    cell_keys, g_cell, d_cell, g_off, d_off = _cells(keys_g, keys_d)

    # ---- frames per track (unique images, timeline order)
    g_fpos, g_fbox, g_foff = track_frames(tl_pos, og, g_trk_of_ann, g_aoff, g_ann, a_img[g_ann],
                                    gt.ann_bbox)
    d_fpos, d_fbox, d_foff = track_frames(tl_pos, d_keep, d_trk_of_ann, d_aoff, keep[d_ann],
                                    d_img[d_ann], dt.bbox)

    # 1 + largest timeline position used by a cell (sizes the LDS tables of
    # the dense-timeline 3D-IoU kernel)
    def cell_span(fpos, foff, coff):
        # positions ascend inside a track and the tracks of a cell are
        # contiguous: last position of every track, then a segmented max
        ts = np.zeros(len(foff) - 1, dtype=np.int64)
        has = np.diff(foff) > 0
        ts[has] = fpos[foff[1:][has] - 1].astype(np.int64) + 1
        cs = np.zeros(len(coff) - 1, dtype=np.int64)
        ne = np.diff(coff) > 0
        if ne.any():
            cs[ne] = np.maximum.reduceat(ts, coff[:-1][ne])
        return cs

    span = np.maximum(cell_span(g_fpos, g_foff, g_off),
                      cell_span(d_fpos, d_foff, d_off))
    nel = _csr_member(gt.vid_nel_off, gt.vid_nel, vid_row[d_vid[d_keep]],
                      d_catid[d_keep])
    f = Flat()
    f.kind = "tao"
    f.vid_ids, f.cat_ids = vid_ids, cat_ids
    f.required_average = required_average
    f.n_cells = len(cell_keys)
    f.use_cats = bool(use_cats)
    f.cell_unit = (cell_keys % U).astype(I32)       # video index
    f.cell_cat = (cell_keys // U).astype(I32) if use_cats else \
        np.zeros(len(cell_keys), dtype=I32)
    f.cell_dt_off = d_off.astype(I32)
    f.cell_gt_off = g_off.astype(I32)
    iou_off = np.zeros(f.n_cells + 1, dtype=np.int64)
    np.cumsum(np.diff(d_off) * np.diff(g_off), out=iou_off[1:])
    f.cell_iou_off = iou_off
    f.cell_span = span.astype(I32)
    f.dt_score = np.ascontiguousarray(d_score[d_keep])
    f.dt_area = np.ascontiguousarray(d_area_t[d_keep])
    f.dt_len = d_len[d_keep].astype(I32)
    f.dt_flags = (np.where(nel, DT_IGNORE_UNMATCHED, 0)
                  | np.where(d_ids[d_keep] <= 0, DT_NO_CONSUME, 0)
                  ).astype(np.uint8)
    f.dt_id = d_ids[d_keep]
    f.dt_cat = (keys_d // U).astype(I32) if use_cats else \
        np.zeros(len(keys_d), dtype=I32)
    f.dt_cell = d_cell.astype(I32)
    f.dt_frame_off, f.dt_frame_pos, f.dt_frame_box = d_foff, d_fpos, d_fbox
    f.gt_area = np.ascontiguousarray(g_area[og])
    f.gt_len = g_len[og].astype(I32)
    f.gt_nhp = g_nhp[og].astype(I32)
    f.gt_flags = (np.where(g_ign[og] != 0, GT_IGNORE, 0)
                  | np.where(g_ids[og] == -1, GT_ID_HIDDEN, 0)
                  ).astype(np.uint8)
    f.gt_id = g_ids[og]
    f.gt_cat = (keys_g // U).astype(I32) if use_cats else \
        np.zeros(len(keys_g), dtype=I32)
    f.gt_cell = g_cell.astype(I32)
    f.gt_frame_off, f.gt_frame_pos, f.gt_frame_box = g_foff, g_fpos, g_fbox
    f.n_pairs = int(iou_off[-1])
    f.track_scores = dict(zip(u.tolist(), trk_score.tolist()))
    f.tl_image_id, f.tl_vid_start = T.tl_image_id, T.tl_vid_start
    return f
