"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

A plain-Python/numpy restatement of the algorithm the reference runs in
``tools/eval_on_tao_amodal.py``: image-level ``LVISEval`` followed by
track-level ``TaoEval``.  It works on the JSON-shaped inputs directly (lists of
dicts), uses per-cell Python loops, and is only meant for small cases: the
tests compare the HIP path against it on seeded inputs, and it is itself
pinned to the golden vectors that ``tests/golden/make_golden.py`` produced by
running the real reference in the development container
(tests/test_oracle_golden.py).  Parity status: PINNED (fixtures F1-F5).

Every function cites the reference lines it restates (paths relative to
/root/reference; ``L/`` = tao_amodal/evaluation/lvis_amodal/, ``T/`` =
tao_amodal/evaluation/tao_amodal/, ``C/`` = the vendored pycocotools
``common/maskApi.c``).

One documented deviation is selectable: ``frame_order``.  The reference sums
the per-frame intersections/unions of a track pair in CPython set-iteration
order (T/eval.py:83-94).  ``frame_order="set"`` reproduces exactly that (it
builds the same sets); ``frame_order="timeline"`` sums in ascending
(frame_index, image_id) order, which is what the HIP kernels and the C oracle
do.  The two agree bit-for-bit whenever the per-frame products are exactly
representable (integer / dyadic coordinates, e.g. all synthetic sets).
"""
import copy
import itertools
from collections import OrderedDict, defaultdict

import numpy as np

# L/eval.py:560-575, T/eval.py:727-744
IOU_THRS = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1,
                       endpoint=True)
REC_THRS = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1,
                       endpoint=True)
MAX_DETS = 300
VIS_RNG = [[0, 1.0], [0, 0.1], [0.1, 0.8], [0.8, 1.0], [0, 0.8], [0, 1.0]]
VIS_LBL = ["all", "highly-occluded", "partially-occluded", "highly-visible",
           "highly-and-partially-occluded", "out-of-frame"]
AREA_RNG = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2],
            [96 ** 2, 1e5 ** 2], [0 ** 2, 1e5 ** 2]]
AREA_LBL = ["all", "small", "medium", "large",
            "highly-and-partially-occluded"]
TIME_RNG = [[0, 1e5], [0, 3], [3, 10], [10, 1e5]]
TIME_LBL = ["all", "short", "medium", "long"]


# --------------------------------------------------------------- arithmetic
def bb_iou(d, g):
    """One entry of bbIou with iscrowd=0 (C/maskApi.c:109-120)."""
    da = d[2] * d[3]
    ga = g[2] * g[3]
    w = min(d[2] + d[0], g[2] + g[0]) - max(d[0], g[0])
    if w <= 0:
        return 0.0
    h = min(d[3] + d[1], g[3] + g[1]) - max(d[1], g[1])
    if h <= 0:
        return 0.0
    i = w * h
    u = da + ga - i
    return i / u


def bb_iou_matrix(dts, gts):
    """mask_utils.iou for boxes: [] when either side is empty
    (_mask.pyx:171-239), else ious[d, g]."""
    if len(dts) == 0 or len(gts) == 0:
        return []
    dts = np.array(dts, dtype=np.double).tolist()
    gts = np.array(gts, dtype=np.double).tolist()
    return np.array([[bb_iou(d, g) for g in gts] for d in dts])


def bb_intersect_union(d, g):
    """T/eval.py:15-48."""
    w = max(min(d[0] + d[2], g[0] + g[2]) - max(d[0], g[0]), 0)
    h = max(min(d[1] + d[3], g[1] + g[3]) - max(d[1], g[1]), 0)
    i = w * h
    return i, d[2] * d[3] + g[2] * g[3] - i


def track_box_iou(dt_track, gt_track, frame_order, timeline):
    """3D IoU of two {image_id: bbox} maps (T/eval.py:73-96)."""
    if frame_order == "set":
        image_ids = set(gt_track.keys()) | set(dt_track.keys())
    else:
        image_ids = sorted(set(gt_track.keys()) | set(dt_track.keys()),
                           key=lambda im: timeline[im])
    i = 0
    u = 0
    for im in image_ids:
        g = gt_track.get(im)
        d = dt_track.get(im)
        if d and g:
            i_, u_ = bb_intersect_union(d, g)
            i += i_
            u += u_
        elif g:
            u += g[2] * g[3]
        elif d:
            u += d[2] * d[3]
    assert i <= u
    return i / u if u > 0 else 0


def track_avg_iou(dt_track, gt_track, frame_order, timeline):
    """compute_avg_track_iou (T/eval.py:99-117): mean over the union of frames
    of the per-frame IoU (0 where only one side has a box).  ``set`` order
    reproduces the reference (np.mean of the list in set order); ``timeline``
    is the canonical form of the kernels: left-to-right sum in ascending
    timeline order divided by the number of frames."""
    keys = set(gt_track.keys()) | set(dt_track.keys())
    if frame_order != "set":
        keys = sorted(keys, key=lambda im: timeline[im])
    ious = []
    for im in keys:
        g = gt_track.get(im)
        d = dt_track.get(im)
        if d and g:
            i_, u_ = bb_intersect_union(d, g)
            ious.append(i_ / u_ if u_ > 0 else 0)
        else:
            ious.append(0)
    if frame_order == "set":
        return np.mean(ious)
    total = 0.0
    for v in ious:
        total += v
    return total / len(ious)


def track_imagenetvid_iou(dt_track, gt_track, threshold=0.5):
    """compute_imagenetvid_iou (T/eval.py:51-70): fraction of the union's
    frames whose boxes overlap with intersection > threshold * union
    (integer counts: independent of the frame order)."""
    matched = total = 0
    for im in set(gt_track.keys()) | set(dt_track.keys()):
        g = gt_track.get(im)
        d = dt_track.get(im)
        if d and g:
            i_, u_ = bb_intersect_union(d, g)
            if i_ > threshold * u_:
                matched += 1
        total += 1
    return matched / total


# ----------------------------------------------------------- shared pieces
def make_track_ids_unique(preds):
    """tools/eval_on_tao_amodal.py:44-66 (in place); returns #ids changed."""
    first_video = {}
    clash = set()
    top = 0
    for p in preds:
        t = p["track_id"]
        first_video.setdefault(t, p["video_id"])
        if p["video_id"] != first_video[t]:
            clash.add(t)
        top = max(top, t)
    if clash:
        fresh = itertools.count(top + 1)
        new_id = {}
        for p in preds:
            if p["track_id"] in clash:
                key = (p["track_id"], p["video_id"])
                if key not in new_id:
                    new_id[key] = next(fresh)
                p["track_id"] = new_id[key]
    return len(clash)


def limit_dets_per_image(anns, max_dets=MAX_DETS):
    """L/results.py:73-84 == T/results.py:121-132."""
    per_img = OrderedDict()
    for a in anns:
        per_img.setdefault(a["image_id"], []).append(a)
    out = []
    for lst in per_img.values():
        if len(lst) > max_dets:
            lst = sorted(lst, key=lambda a: a["score"], reverse=True)
            lst = lst[:max_dets]
        out.extend(lst)
    return out


def stable_desc(scores):
    return np.argsort([-s for s in scores], kind="mergesort")


def greedy_match(ious, gt_ig, dt_ids, gt_ids, consumed):
    """The threshold x detection x ground-truth triple loop shared by
    L/eval.py:245-277 and T/eval.py:396-428.

    ious: (D, G) in the *ignore-sorted* GT order, or [] ; gt_ig: 0/1 per GT in
    that order.  ``consumed(v)`` says whether a stored gt_m value marks the GT
    as taken (both evaluators test ``> 0``).  Returns index matches:
    dt_gi[T, D] = matched GT position or -1, plus gt_val[T, G] holding the id
    of the matching detection (or None).
    """
    T, D, G = len(IOU_THRS), len(dt_ids), len(gt_ids)
    dt_gi = -np.ones((T, D), dtype=np.int64)
    gt_val = [[None] * G for _ in range(T)]
    if len(ious) == 0:
        return dt_gi, gt_val
    for t, thr in enumerate(IOU_THRS):
        for d in range(D):
            best = min([thr, 1 - 1e-10])
            m = -1
            for g in range(G):
                if gt_val[t][g] is not None and consumed(gt_val[t][g]):
                    continue
                if m > -1 and gt_ig[m] == 0 and gt_ig[g] == 1:
                    break
                if ious[d, g] < best:
                    continue
                best = ious[d, g]
                m = g
            if m == -1:
                continue
            dt_gi[t, d] = m
            gt_val[t][m] = dt_ids[d]
    return dt_gi, gt_val


def sweep(dt_scores, dt_ids, matched, ignored, gt_ig):
    """One (category, range) column of accumulate (L/eval.py:352-417,
    T/eval.py:507-573).  matched/ignored: bool (T, N) in concatenation order.
    Returns None when there is no evaluated GT."""
    dt_scores = np.asarray(dt_scores, dtype=np.float64)
    order = np.argsort(-dt_scores, kind="mergesort")
    matched = matched[:, order]
    ignored = ignored[:, order]
    num_gt = np.count_nonzero(np.asarray(gt_ig) == 0)
    if num_gt == 0:
        return None
    tps = np.logical_and(matched, np.logical_not(ignored))
    fps = np.logical_and(np.logical_not(matched), np.logical_not(ignored))
    tp_sum = np.cumsum(tps, axis=1).astype(dtype=float)
    fp_sum = np.cumsum(fps, axis=1).astype(dtype=float)
    R = len(REC_THRS)
    prec = np.zeros((len(IOU_THRS), R))
    rec = np.zeros(len(IOU_THRS))
    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
        n = len(tp)
        rc = tp / num_gt
        rec[t] = rc[-1] if n else 0
        pr = (tp / (fp + tp + np.spacing(1))).tolist()
        for i in range(n - 1, 0, -1):
            if pr[i] > pr[i - 1]:
                pr[i - 1] = pr[i]
        at = [0.0] * R
        for j, k in enumerate(np.searchsorted(rc, REC_THRS, side="left")):
            if k >= n:
                break
            at[j] = pr[k]
        prec[t] = at
    return {"precision": prec, "recall": rec,
            "dt_ids": np.asarray(dt_ids)[order] if len(order) else
            np.asarray(dt_ids), "tps": tps, "fps": fps, "num_gt": num_gt}


def masked_mean(s):
    s = s[s > -1]
    return -1 if len(s) == 0 else np.mean(s)


# ------------------------------------------------------------------ LVISEval
def lvis_eval(gt, preds, use_cats=True, iou_type="bbox"):
    """Image-level evaluation.  gt: parsed annotation dict; preds: list of
    dicts (a private deep copy is taken).  L/eval.py:59-145, L/lvis.py:38-97,
    L/results.py:10-71.  ``use_cats=False`` restates params.use_cats = 0
    (L/eval.py:125-128,147-166,314-317): one cell per image, category -1;
    the reference's summarize() then fails on the frequency groups
    (IndexError), so "results" / "printed" are None.  ``iou_type="segm"``
    (L/eval.py:54-58,70-73,179-191; L/results.py:42-62; L/lvis.py:171-193)
    compares run-length masks (oracle/rle.py) instead of boxes; the result
    then carries "gt_rle" / "dt_rle" = the compressed text of every mask
    compared."""
    from . import rle as _rle
    gt = copy.deepcopy(gt)
    preds = copy.deepcopy(preds)
    imgs = {im["id"]: im for im in gt["images"]}
    cats = {c["id"]: c for c in gt["categories"]}
    img_ids = [int(x) for x in np.unique(sorted(imgs))]
    cat_ids = sorted(cats)
    cat_set = set(cat_ids)

    preds = limit_dets_per_image(preds)
    if "bbox" in preds[0]:                       # L/results.py:42-52
        for k, p in enumerate(preds):
            x1, y1, w_, h_ = p["bbox"]
            if "segmentation" not in p:
                p["segmentation"] = [[x1, y1, x1, y1 + h_, x1 + w_, y1 + h_,
                                      x1 + w_, y1]]
            p["area"] = w_ * h_
            p["id"] = k + 1
    elif "segmentation" in preds[0]:             # L/results.py:54-62
        for k, p in enumerate(preds):
            m = _rle.fr_string(p["segmentation"]["counts"],
                               *p["segmentation"]["size"])
            p["area"] = _rle.area(m)
            if "bbox" not in p:
                p["bbox"] = _rle.to_bbox(m)
            p["id"] = k + 1
    assert set(p["image_id"] for p in preds) <= set(imgs), \
        "Results do not correspond to current LVIS set."

    def select(anns):
        """get_ann_ids + load_anns (L/lvis.py:63-97,121-130): image-major in
        sorted image order, strict area window, id -> *last* ann of that id"""
        by_img = defaultdict(list)
        by_id = {}
        for a in anns:
            by_img[a["image_id"]].append(a)
            by_id[a["id"]] = a
        inf = float("inf")
        return [by_id[a["id"]] for im in img_ids for a in by_img[im]
                if a["category_id"] in cat_set and 0 < a["area"] < inf]

    gts = select(gt["annotations"])
    dts = select(preds)
    if iou_type == "segm":                       # _to_mask, L/eval.py:54-58
        for a in gts + dts:
            im = imgs[a["image_id"]]
            a["_rle"] = _rle.ann_to_rle(a["segmentation"], im["height"],
                                        im["width"])
    elif iou_type != "bbox":
        raise ValueError("Unknown iou_type for iou computation.")
    cell_gt, cell_dt = defaultdict(list), defaultdict(list)
    present = defaultdict(set)
    for g in gts:
        g.setdefault("ignore", 0)
        cell_gt[g["image_id"], g["category_id"]].append(g)
        present[g["image_id"]].add(g["category_id"])
    for d in dts:
        im, c = d["image_id"], d["category_id"]
        if c not in imgs[im]["neg_category_ids"] and c not in present[im]:
            continue
        cell_dt[im, c].append(d)

    cells = OrderedDict()
    eval_cats = cat_ids if use_cats else [-1]
    for im in img_ids:
        for c in eval_cats:
            if use_cats:
                G, D = cell_gt.get((im, c), []), cell_dt.get((im, c), [])
            else:               # _get_gt_dt: the per-category lists, joined
                G = [g for cc in cat_ids for g in cell_gt.get((im, cc), [])]
                D = [d for cc in cat_ids for d in cell_dt.get((im, cc), [])]
            if not G and not D:
                continue
            D = [D[i] for i in stable_desc([d["score"] for d in D])]
            if iou_type == "segm":
                ious = _rle.iou_matrix([d["_rle"] for d in D],
                                       [g["_rle"] for g in G])
            else:
                ious = bb_iou_matrix([d["bbox"] for d in D],
                                     [g["bbox"] for g in G])
            nel_list = imgs[im]["not_exhaustive_category_ids"]
            dt_mask = np.array([d["area"] < 0 or d["area"] > 1e5 ** 2
                                or d["category_id"] in nel_list
                                for d in D], dtype=bool)
            ranges = []
            for a, rng in enumerate(VIS_RNG):
                if a < len(VIS_RNG) - 1:
                    ig = [1 if (g["ignore"] or g["visibility"] < rng[0]
                                or g["visibility"] > rng[1]) else 0
                          for g in G]
                else:
                    ig = [1 if (g["ignore"] or not g["out_of_frame"]) else 0
                          for g in G]
                gi = np.argsort(ig, kind="mergesort")
                Gs = [G[i] for i in gi]
                gt_ig = np.array([ig[i] for i in gi])
                io = ious[:, gi] if len(ious) > 0 else ious
                dt_gi, gt_val = greedy_match(
                    io, gt_ig, [d["id"] for d in D], [g["id"] for g in Gs],
                    lambda v: v > 0)
                T = len(IOU_THRS)
                dt_m = np.zeros((T, len(D)))
                dt_ig = np.zeros((T, len(D)))
                gt_m = np.zeros((T, len(Gs)))
                for t in range(T):
                    for d in range(len(D)):
                        m = dt_gi[t, d]
                        if m >= 0:
                            dt_m[t, d] = Gs[m]["id"]
                            dt_ig[t, d] = gt_ig[m]
                    for g in range(len(Gs)):
                        if gt_val[t][g] is not None:
                            gt_m[t, g] = gt_val[t][g]
                dt_ig = np.logical_or(dt_ig, np.logical_and(
                    dt_m == 0, dt_mask[None, :].repeat(T, 0)))
                ranges.append({
                    "dt_ids": [d["id"] for d in D],
                    "gt_ids": [g["id"] for g in Gs],
                    "dt_scores": [d["score"] for d in D],
                    "dt_matches": dt_m, "gt_matches": gt_m,
                    "dt_ignore": dt_ig, "gt_ignore": gt_ig})
            cells[im, c] = {"ious": ious, "ranges": ranges}

    # ---------------------------------------------------------- accumulate
    T, R, K, A = len(IOU_THRS), len(REC_THRS), len(eval_cats), len(VIS_RNG)
    precision = -np.ones((T, R, K, A))
    recall = -np.ones((T, K, A))
    pointers = {}
    by_cat = defaultdict(list)
    for (im, c), cell in cells.items():
        by_cat[c].append(cell)
    for k, c in enumerate(eval_cats):
        for a in range(A):
            E = [cell["ranges"][a] for cell in by_cat.get(c, [])]
            if not E:
                continue
            sc = np.concatenate([e["dt_scores"] for e in E], axis=0)
            ids = np.concatenate([e["dt_ids"] for e in E], axis=0)
            dm = np.concatenate([e["dt_matches"] for e in E], axis=1)
            di = np.concatenate([e["dt_ignore"] for e in E], axis=1)
            gi = np.concatenate([e["gt_ignore"] for e in E])
            s = sweep(sc, ids, dm != 0, di.astype(bool), gi)
            if s is None:
                continue
            precision[:, :, k, a] = s["precision"]
            recall[:, k, a] = s["recall"]
            pointers[k, a] = s

    freq_groups = [[], [], []]
    for k, c in enumerate(cat_ids):
        freq_groups["rcf".index(cats[c]["frequency"])].append(k)

    def summ(kind, thr=None, vis="all", freq=None):
        aidx = [i for i, lbl in enumerate(VIS_LBL) if lbl == vis]
        s = precision if kind == "ap" else recall
        if thr is not None:
            s = s[np.where(thr == IOU_THRS)[0]]
        if kind == "ap":
            s = (s[:, :, freq_groups[freq], aidx] if freq is not None
                 else s[:, :, :, aidx])
        else:
            s = s[:, :, aidx]
        return masked_mean(s)

    if not use_cats:
        return {"img_ids": img_ids, "cat_ids": eval_cats, "cells": cells,
                "precision": precision, "recall": recall, "pointers": pointers,
                "results": None, "freq_groups": freq_groups, "printed": None}
    res = OrderedDict()
    for suffix, vis in (("", "all"), ("-HO", "highly-occluded"),
                        ("-PO", "partially-occluded"),
                        ("-HP", "highly-and-partially-occluded"),
                        ("-HV", "highly-visible"), ("-OOF", "out-of-frame")):
        res["AP" + suffix] = summ("ap", vis=vis)
        res["AP50" + suffix] = summ("ap", thr=0.50, vis=vis)
        res["AP75" + suffix] = summ("ap", thr=0.75, vis=vis)
    for f, name in enumerate(("APr", "APc", "APf")):
        res[name] = summ("ap", freq=f)
    res["AR@{}".format(MAX_DETS)] = summ("ar")
    for vis in ("highly-occluded", "partially-occluded", "highly-visible",
                "highly-and-partially-occluded", "out-of-frame"):
        # L/eval.py:497-499 -- the key uses only the first letter of the label
        res["AR{}@{}".format(vis[0], MAX_DETS)] = summ("ar", vis=vis)
    out = {"img_ids": img_ids, "cat_ids": cat_ids, "cells": cells,
           "precision": precision, "recall": recall, "pointers": pointers,
           "results": res, "freq_groups": freq_groups,
           "printed": lvis_lines(res)}
    if iou_type == "segm":
        kept = [d for lst in cell_dt.values() for d in lst]
        out["gt_rle"] = {g["id"]: _rle.to_string(g["_rle"]) for g in gts}
        out["dt_rle"] = {d["id"]: _rle.to_string(d["_rle"]) for d in kept}
        out["dt_area"] = {p["id"]: float(p["area"]) for p in preds}
        out["dt_bbox"] = {p["id"]: [float(v) for v in p["bbox"]] for p in preds}
    return out


def lvis_lines(results):
    """L/eval.py:507-545."""
    tmpl = (" {:<18} {} @[ IoU={:<9} | visibility={:>6s} | maxDets={:>3d} "
            "catIds={:>3s}] = {:0.3f}")
    names = {"HO": "Highly Occluded (vis < 0.1)",
             "PO": "Partially Occluded (0.1 < vis < 0.8)",
             "HP": "Highly + Partially Occluded (vis < 0.8)",
             "HV": "Highly Visible (vis > 0.8)"}
    out = []
    for key, value in results.items():
        ap = "AP" in key
        if len(key) > 2 and key[2].isdigit():
            iou = "{:0.2f}".format(float(key[2:4]) / 100)
        else:
            iou = "{:0.2f}:{:0.2f}".format(IOU_THRS[0], IOU_THRS[-1])
        grp = key[2] if len(key) > 2 and key[2] in "rcf" else "all"
        if len(key) > 2 and key[-2:] in names:
            vis = names[key[-2:]]
        elif len(key) > 2 and key[-3:] == "OOF":
            vis = "Out-of-Frame"
        else:
            vis = "all"
        out.append(tmpl.format(
            "Average Precision" if ap else "Average Recall",
            "(AP)" if ap else "(AR)", iou, vis, MAX_DETS, grp, value))
    return out


# ------------------------------------------------------------------- TaoEval
def tao_eval(gt, preds, frame_order="set", iou_3d_type="3d_iou", use_cats=True):
    """Track-level evaluation (T/tao.py:112-254, T/results.py:27-109,
    T/eval.py:178-276,459-584).  ``preds`` must already have unique track ids
    (the CLI calls make_track_ids_unique first).  ``use_cats=False`` restates
    the class-agnostic mode (T/eval.py:257-260,293-303): one cell per video,
    no federated filter."""
    gt = copy.deepcopy(gt)
    preds = copy.deepcopy(preds)
    merge = {m["id"]: c["id"] for c in gt["categories"] if "merged" in c
             for m in c["merged"]}
    for x in gt["annotations"] + gt["tracks"] + preds:
        x["category_id"] = merge.get(x["category_id"], x["category_id"])
    vids = {v["id"]: v for v in gt["videos"]}
    imgs = {im["id"]: im for im in gt["images"]}
    cats = {c["id"]: c for c in gt["categories"]}
    vid_ids = [int(x) for x in np.unique(sorted(vids))]
    cat_ids = sorted(cats)
    cat_set = set(cat_ids)
    vid_imgs = defaultdict(list)
    for im in gt["images"]:
        vid_imgs[im["video_id"]].append(im["id"])
    gt_tracks = {t["id"]: t for t in gt["tracks"]}
    timeline = {}
    for v, lst in vid_imgs.items():
        for pos, im in enumerate(sorted(
                set(lst), key=lambda i: (imgs[i]["frame_index"], i))):
            timeline[im] = pos

    # ---- TaoResults
    seen = {}
    for p in preds:
        assert seen.setdefault(p["track_id"], p["video_id"]) == \
            p["video_id"], "Track id appears in more than one video"
    preds = limit_dets_per_image(preds)
    dt_tracks = OrderedDict()
    for k, p in enumerate(preds):
        t = dt_tracks.setdefault(p["track_id"], {
            "id": p["track_id"], "video_id": p["video_id"],
            "category_id": p["category_id"]})
        assert t["category_id"] == p["category_id"]
        p["area"] = p["bbox"][2] * p["bbox"][3]
        p["id"] = k + 1
    assert set(p["image_id"] for p in preds) <= set(imgs), \
        "Results do not correspond to current Tao set."
    per_track = defaultdict(list)
    for p in preds:
        per_track[p["track_id"]].append(p)
    for tid, lst in per_track.items():
        scores = [float(p["score"]) for p in lst]
        if len(set(scores)) > 1:
            avg = np.mean(scores)
            dt_tracks[tid]["score"] = avg
            for p in lst:
                p["score"] = avg
        else:
            dt_tracks[tid]["score"] = scores[0]

    # ---- get_ann_ids + group_ann_tracks (T/tao.py:172-188,203-254)
    video_images = [im for v in vid_ids for im in vid_imgs[v]]
    img_order = list(set(video_images) & set(video_images))

    def tracks_of(anns, track_table):
        by_img = defaultdict(list)
        by_id = {}
        for a in anns:
            a["bbox"] = [float(x) for x in a["bbox"]]
            by_img[a["image_id"]].append(a)
            by_id[a["id"]] = a
        inf = float("inf")
        sel = [by_id[a["id"]] for im in img_order for a in by_img[im]
               if a["category_id"] in cat_set and 0 < a["area"] < inf]
        tracks = OrderedDict()
        for a in sel:
            tr = tracks.get(a["track_id"])
            if tr is None:
                tr = dict(track_table[a["track_id"]])
                tr["annotations"] = []
                tracks[a["track_id"]] = tr
            tr["annotations"].append(a)
        for tr in tracks.values():
            tr["annotations"] = sorted(
                tr["annotations"],
                key=lambda x: imgs[x["image_id"]]["frame_index"])
            tr["area"] = (sum(x["area"] for x in tr["annotations"])
                          / len(tr["annotations"]))
        return sel, list(tracks.values())

    gsel, gts = tracks_of(gt["annotations"], gt_tracks)
    dsel, dts = tracks_of(preds, dt_tracks)
    if len(gsel) == 0:
        raise ValueError("Found no groundtruth annotations for given params")
    if len(dsel) == 0:
        raise ValueError("Found no predicted annotations for given params")
    cell_gt, cell_dt = defaultdict(list), defaultdict(list)
    present = defaultdict(set)
    for g in gts:
        g.setdefault("ignore", 0)
        cell_gt[g["video_id"], g["category_id"]].append(g)
        present[g["video_id"]].add(g["category_id"])
    for d in dts:
        v, c = d["video_id"], d["category_id"]
        if use_cats and c not in vids[v]["neg_category_ids"] \
                and c not in present[v]:
            continue
        cell_dt[v, c].append(d)

    T = len(IOU_THRS)
    cells = OrderedDict()
    eval_cats = cat_ids if use_cats else [-1]
    for v in vid_ids:
        for c in eval_cats:
            if use_cats:
                G, D = cell_gt.get((v, c), []), cell_dt.get((v, c), [])
            else:       # all categories of the video, category-major
                G = [t for k in cat_ids for t in cell_gt.get((v, k), [])]
                D = [t for k in cat_ids for t in cell_dt.get((v, k), [])]
            if not G and not D:
                continue
            D = [D[i] for i in stable_desc([d["score"] for d in D])]
            gmaps = [{a["image_id"]: a["bbox"] for a in g["annotations"]}
                     for g in G]
            dmaps = [{a["image_id"]: a["bbox"] for a in d["annotations"]}
                     for d in D]
            ious = np.zeros([len(D), len(G)])
            for i, j in np.ndindex(ious.shape):
                if iou_3d_type == "3d_iou":
                    ious[i, j] = track_box_iou(dmaps[i], gmaps[j], frame_order,
                                               timeline)
                elif iou_3d_type == "avg_iou":
                    ious[i, j] = track_avg_iou(dmaps[i], gmaps[j], frame_order,
                                               timeline)
                elif iou_3d_type == "imagenetvid":
                    ious[i, j] = track_imagenetvid_iou(dmaps[i], gmaps[j])
            nel_of = vids[v]["not_exhaustive_category_ids"]
            ranges = []
            for a, ar in enumerate(AREA_RNG):
                for tr in TIME_RNG:
                    ig = []
                    for g in G:
                        dur = len(g["annotations"])
                        bad = (g["ignore"] or g["area"] < ar[0]
                               or g["area"] > ar[1] or dur < tr[0]
                               or dur > tr[1])
                        if a == len(AREA_RNG) - 1:
                            n_hp = sum(x["visibility"] < 0.8
                                       for x in g["annotations"])
                            bad = bad or n_hp <= 5
                        ig.append(1 if bad else 0)
                    gi = np.argsort(ig, kind="mergesort")
                    Gs = [G[i] for i in gi]
                    gt_ig = np.array([ig[i] for i in gi])
                    io = ious[:, gi] if len(ious) > 0 else ious
                    dt_gi, gt_val = greedy_match(
                        io, gt_ig, [d["id"] for d in D],
                        [g["id"] for g in Gs], lambda v_: v_ > 0)
                    dt_m = np.zeros((T, len(D))) - 1
                    gt_m = np.zeros((T, len(Gs))) - 1
                    dt_ig = np.zeros((T, len(D)))
                    for t in range(T):
                        for d in range(len(D)):
                            m = dt_gi[t, d]
                            if m >= 0:
                                dt_m[t, d] = Gs[m]["id"]
                                dt_ig[t, d] = gt_ig[m]
                        for g in range(len(Gs)):
                            if gt_val[t][g] is not None:
                                gt_m[t, g] = gt_val[t][g]
                    mask = np.array([
                        d["area"] < ar[0] or d["area"] > ar[1]
                        or len(d["annotations"]) < tr[0]
                        or len(d["annotations"]) > tr[1]
                        or d["category_id"] in nel_of
                        for d in D], dtype=bool)
                    dt_ig = np.logical_or(dt_ig, np.logical_and(
                        dt_m == -1, mask[None, :].repeat(T, 0)))
                    ranges.append({
                        "dt_ids": [d["id"] for d in D],
                        "gt_ids": [g["id"] for g in Gs],
                        "dt_scores": [d["score"] for d in D],
                        "dt_matches": dt_m, "gt_matches": gt_m,
                        "dt_ignore": dt_ig, "gt_ignore": gt_ig})
            cells[v, c] = {"ious": ious, "ranges": ranges}

    R, K = len(REC_THRS), len(eval_cats)
    NA, NT = len(AREA_RNG), len(TIME_RNG)
    precision = -np.ones((T, R, K, NA, NT))
    recall = -np.ones((T, K, NA, NT))
    pointers = {}
    by_cat = defaultdict(list)
    for (v, c), cell in cells.items():
        by_cat[c].append(cell)
    for k, c in enumerate(eval_cats):
        for a in range(NA):
            for t_ in range(NT):
                E = [cell["ranges"][a * NT + t_]
                     for cell in by_cat.get(c, [])]
                if not E:
                    continue
                sc = np.concatenate([e["dt_scores"] for e in E], axis=0)
                ids = np.concatenate([e["dt_ids"] for e in E], axis=0)
                dm = np.concatenate([e["dt_matches"] for e in E], axis=1)
                di = np.concatenate([e["dt_ignore"] for e in E], axis=1)
                gi = np.concatenate([e["gt_ignore"] for e in E])
                s = sweep(sc, ids, dm != -1, di.astype(bool), gi)
                if s is None:
                    continue
                precision[:, :, k, a, t_] = s["precision"]
                recall[:, k, a, t_] = s["recall"]
                pointers[k, a, t_] = s

    def summ(kind, thr=None, area="all", time="all"):
        aidx = [i for i, lbl in enumerate(AREA_LBL) if lbl == area]
        tidx = [i for i, lbl in enumerate(TIME_LBL) if lbl == time]
        s = precision if kind == "ap" else recall
        if thr is not None:
            s = s[np.where(thr == IOU_THRS)[0]]
        s = s[:, :, :, aidx, tidx] if kind == "ap" else s[:, :, aidx, tidx]
        return masked_mean(s)

    hp = "highly-and-partially-occluded"
    res = OrderedDict()
    res["AP"] = summ("ap")
    res["AP50"] = summ("ap", thr=0.50)
    res["AP75"] = summ("ap", thr=0.75)
    res["AP-HP"] = summ("ap", area=hp)
    res["AP50-HP"] = summ("ap", area=hp, thr=0.50)
    res["AP75-HP"] = summ("ap", area=hp, thr=0.75)
    for kind, K_ in (("ap", "AP"), ("ar", "AR")):
        if kind == "ar":
            res["AR@{}".format(MAX_DETS)] = summ("ar")
        for lbl in ("small", "medium", "large"):
            res[(K_, "area", lbl, MAX_DETS)] = summ(kind, area=lbl)
        for lbl in ("short", "medium", "long"):
            res[(K_, "time", lbl, MAX_DETS)] = summ(kind, time=lbl)
    return {"vid_ids": vid_ids, "cat_ids": cat_ids, "cells": cells,
            "precision": precision, "recall": recall, "pointers": pointers,
            "results": res, "printed": tao_lines(res),
            "track_scores": {k: float(t["score"])
                             for k, t in dt_tracks.items()}}


def tao_lines(results):
    """T/eval.py:668-712."""
    tmpl = (" {:<18} {} @[ IoU={:<9} | area={:>6s} | dur={:>6s} | "
            "maxDets={:>3d} catIds={:>3s}] = {:0.3f}")
    out = []
    for key, value in results.items():
        ap = "AP" in key
        area = time = "all"
        max_dets = MAX_DETS
        if isinstance(key, tuple):
            kind, rng, max_dets = key[1:]
            if kind == "time":
                time = rng[0]
            else:
                area = rng[0]
        if len(key) > 2 and key[2].isdigit():
            iou = "{:0.2f}".format(float(key[2:4]) / 100)
        else:
            iou = "{:0.2f}:{:0.2f}".format(IOU_THRS[0], IOU_THRS[-1])
        grp = key[2] if len(key) > 2 and key[2] in ["r", "c", "f"] else "all"
        out.append(tmpl.format(
            "Average Precision" if ap else "Average Recall",
            "(AP)" if ap else "(AR)", iou, area, time, max_dets, grp, value))
    return out
