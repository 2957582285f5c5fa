"""The ground-truth halves of the cell tables, buildable before the predictions
are there (and before torch is: this module imports numpy alone, so the drop-in
CLI runs it on the main thread while a helper thread imports torch, creates the
HIP context and loads the kernel library -- 1 s of a cold start that used to
sit in front of the first table).  ``flatten_dev`` re-exports everything here.
"""
import numpy as np

from . import flatten
from .flatten import Flat, I32


# ---------------------------------------------------------------------------
# ground-truth halves, buildable before the predictions are there
# ---------------------------------------------------------------------------
def _gt_key(gt):
    """Identity of the arrays behind a GTColumns (see _column_key)."""
    return tuple((k, id(v), v.__array_interface__["data"][0], v.shape)
                 for k, v in sorted(vars(gt).items()) if isinstance(v, np.ndarray))


def _sorted_unique(keys):
    """np.unique of an ascending array (no second sort)."""
    if len(keys) < 2:
        return keys.copy()
    return keys[np.r_[True, keys[1:] != keys[:-1]]]


def _lvis_gt_ready(gt):
    """Everything of the image-level tables that depends on the annotation
    file alone: lvis_gt_side, the cell order of the ground truths and their
    columns."""
    G = flatten.lvis_gt_side(gt)
    U = len(G.img_ids)
    keys_g = G.g_cat * U + G.g_img
    og = flatten.sort_key_score(keys_g)
    g_sel, keys_g = flatten.take(G.g_sel, og), flatten.take(keys_g, og)
    R = Flat()
    R.G, R.g_sel, R.keys_g = G, g_sel, keys_g
    R.gkeys = _sorted_unique(keys_g).astype(np.int32)
    R.tables = Flat()
    flatten.lvis_gt_tables(R.tables, gt, g_sel, keys_g, max(U, 1))
    return R


def _tao_gt_ready(gt, visit_universe=None, universe=None):
    """The same for the track level (``universe``: flatten.tao_gt_universe built
    ahead of the annotation part)."""
    T = flatten.tao_gt_side(gt, visit_universe, universe)
    U = len(T.vid_ids)
    keys_g = T.g_cat * U + T.g_vid
    og = flatten.sort_key_score(keys_g)
    keys_g = keys_g[og]
    R = Flat()
    R.T, R.og, R.keys_g = T, og, keys_g
    R.gkeys = _sorted_unique(keys_g).astype(np.int32)
    R.img_frame = gt.img_frame[T.img_row]
    R.frames = flatten.track_frames(T.tl_pos, og, T.g_trk_of_ann, T.g_aoff, T.g_ann,
                                    flatten.take(T.a_img, T.g_ann), gt.ann_bbox)
    t = R.tables = Flat()
    t.gt_area = np.ascontiguousarray(T.g_area[og])
    t.gt_len = T.g_len[og].astype(I32)
    t.gt_nhp = T.g_nhp[og].astype(I32)
    t.gt_flags = (np.where(T.g_ign[og] != 0, flatten.GT_IGNORE, 0)
                  | np.where(T.g_ids[og] == -1, flatten.GT_ID_HIDDEN, 0)
                  ).astype(np.uint8)
    t.gt_id = T.g_ids[og]
    t.gt_cat = (keys_g // max(U, 1)).astype(I32)
    return R


_READY = {"lvis": _lvis_gt_ready, "tao": _tao_gt_ready}


def prepare_gt(gt, kinds=("lvis", "tao"), wait=True):
    """Build the ground-truth halves of the cell tables ahead of time -- the
    CLI calls this while the prediction file is still being parsed (0.5 s of
    numpy at 3 M annotations that otherwise sits between the parse and the
    first kernel).  The bundles are handed to the NEXT flatten_*_device call on
    the same columns and dropped there (single use: a caller who edits the
    columns afterwards never meets a stale table).  Errors are not raised
    here: the build that needs the bundle runs into them at the place the
    reference does.  ``wait=False``: the halves are built on threads of their
    own and the call returns at once; the table build that needs one waits for
    it (round 6: the track level's half beside its constructor and the
    track-id check instead of in front of them)."""
    def build(kind):
        try:
            return _READY[kind](gt)
        except Exception:
            return None
    # (numpy's sorts, searches and gathers run without the GIL: the two
    # levels' halves side by side)
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=max(len(kinds), 1) + 1)
    made, parts = {}, {}
    for k in kinds:
        if k == "tao":
            # in two stages: what the prediction side of the device build needs
            # first (ids, timeline, visiting order: no annotation looked at),
            # then the three times longer annotation part, which that build
            # only meets at its federated filter
            def universe():
                try:
                    return flatten.tao_gt_universe(gt)
                except Exception:
                    return None
            parts["tao_universe"] = ua = pool.submit(universe)

            def rest(ua=ua):
                try:
                    return _tao_gt_ready(gt, universe=ua.result())
                except Exception:
                    return None
            made[k] = pool.submit(rest)
        else:
            made[k] = pool.submit(build, k)
    pool.shutdown(wait=False)
    vars(gt)["_prepared_gt"] = (_gt_key(gt), made, parts)
    if wait:
        for f in made.values():
            f.result()


def _gt_universe(gt):
    """flatten.tao_gt_universe(gt) if prepare_gt started it on these columns
    (None otherwise: the caller takes the whole half from _gt_ready)."""
    slot = vars(gt).get("_prepared_gt")
    if slot is None:
        return None
    key, _made, parts = slot
    fut = parts.pop("tao_universe", None)
    A = fut.result() if fut is not None else None
    return A if A is not None and key == _gt_key(gt) else None


def _gt_ready(gt, kind):
    slot = vars(gt).get("_prepared_gt")
    if slot is not None:
        key, made, _parts = slot
        R = made.pop(kind, None)
        if not made:
            vars(gt).pop("_prepared_gt", None)
        if R is not None:
            R = R.result()              # (a half still being built: wait for it)
        if R is not None and key == _gt_key(gt):
            return R
    return _READY[kind](gt)
