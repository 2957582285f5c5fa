"""Deterministic fixture *inputs* (ground truth + predictions as JSON-shaped
objects).  ``make_golden.py`` runs the reference evaluator on them in the
development container and commits its outputs next to them; the tests then
hold the oracle, the host flatten and the HIP path to those outputs.

F1  random mixed set (5 videos x 20 frames x 6 categories), integer boxes
F2  hand-built quirk set: every tie/boundary/ordering rule of SURVEY.md 8(a)
F3  1203-category sparse set (checks -1 handling and the r/c/f groups)
F4  F1 geometry with arbitrary decimal coordinates (documents the one
    tolerated deviation: frame-sum order of the 3D IoU)
F5  shuffled image ids, colliding track ids, merged categories, >300 dets
    per image
F7  adversarial decimal coordinates: every (detection track, GT track) pair
    has a 3D IoU that is EXACTLY one of the ten IoU thresholds in real
    arithmetic, so its fp64 value lands within a few ulp of the threshold and
    the side it falls on depends on the order the frames are added in -- the
    reference's CPython set order (T/eval.py:83-94) against timeline order
F8  a down-scaled Config 2 with DECIMAL coordinates (16 videos x 300 frames
    x 10 boxes, 100 categories): tracks of hundreds of frames whose per-frame
    terms are inexact in fp64, i.e. the shape real prediction files have.  Its
    golden 3D IoUs are the reference's set-order sums over up to 600 frames
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tao_amodal_amd.synth import synth  # noqa: E402


def f1():
    gt, dt = synth(seed=11, V=5, F=20, C=6, dets_per_frame=12,
                   gt_tracks_per_video=6, n_present=3, n_neg=1)
    return gt.to_json(), dt.to_json()


def f3():
    gt, dt = synth(seed=33, V=4, F=6, C=1203, dets_per_frame=20,
                   gt_tracks_per_video=10)
    return gt.to_json(), dt.to_json()


def f4():
    gt, dt = synth(seed=44, V=3, F=12, C=5, dets_per_frame=10,
                   gt_tracks_per_video=6, n_present=3, n_neg=1)
    rng = np.random.default_rng(4)
    gt.ann_bbox = gt.ann_bbox + np.round(rng.random(gt.ann_bbox.shape), 2)
    gt.ann_area = gt.ann_bbox[:, 2] * gt.ann_bbox[:, 3]
    dt.bbox = dt.bbox + np.round(rng.random(dt.bbox.shape), 3)
    return gt.to_json(), dt.to_json()


def f5():
    gt, dt = synth(seed=55, V=3, F=4, C=12, dets_per_frame=340,
                   gt_tracks_per_video=8, n_present=4, n_neg=2,
                   shuffle_image_ids=True, collide_track_ids=True,
                   n_merged=3)
    # quantise scores so that equal scores occur (stable-order paths)
    dt.score = np.round(dt.score, 2)
    return gt.to_json(), dt.to_json()


# --------------------------------------------------------------------------
# F2: hand-built quirks
# --------------------------------------------------------------------------
def f2():
    cats = [
        {"id": 1, "name": "a", "frequency": "f"},
        {"id": 2, "name": "b", "frequency": "c"},
        {"id": 3, "name": "c", "frequency": "r"},
        {"id": 4, "name": "d", "frequency": "f", "merged": [{"id": 5}]},
        {"id": 5, "name": "e", "frequency": "c"},
        {"id": 6, "name": "f", "frequency": "r"},
        {"id": 7, "name": "g", "frequency": "c"},
        {"id": 9, "name": "h", "frequency": "f"},
    ]
    videos = [
        {"id": 10, "name": "v10", "neg_category_ids": [3],
         "not_exhaustive_category_ids": [2]},
        {"id": 20, "name": "v20", "neg_category_ids": [6],
         "not_exhaustive_category_ids": []},
        {"id": 30, "name": "v30", "neg_category_ids": [],
         "not_exhaustive_category_ids": []},
    ]
    # image ids deliberately out of order (CPython set-iteration order leak,
    # SURVEY 8(a) a10) -- 12 frames for v10, 11 for v20, 3 for v30
    v10_imgs = [1000003, 5, 999999999, 12, 70000, 8, 33, 64, 65, 1024, 4097, 7]
    v20_imgs = [200 + 7 * k for k in range(11)]
    v30_imgs = [3001, 3000, 3002]
    images = []
    for vid, ids in ((10, v10_imgs), (20, v20_imgs), (30, v30_imgs)):
        v = [x for x in videos if x["id"] == vid][0]
        for f, i in enumerate(ids):
            images.append({
                "id": i, "video_id": vid, "frame_index": 30 * f,
                "neg_category_ids": list(v["neg_category_ids"]),
                "not_exhaustive_category_ids":
                    list(v["not_exhaustive_category_ids"])})
    # image-level lists may differ from the video-level ones
    images[3]["neg_category_ids"] = [3, 7]
    images[4]["not_exhaustive_category_ids"] = []

    tracks, anns = [], []

    def add_gt(tid, cat, vid, frames_boxes, vis=1.0, oof=False, ignore=None,
               trk_ignore=False, area=None, first_ann_id=None):
        t = {"id": tid, "category_id": cat, "video_id": vid}
        if trk_ignore:
            t["ignore"] = 1
        tracks.append(t)
        for k, (img, box) in enumerate(frames_boxes):
            a = {"id": (first_ann_id + k) if first_ann_id is not None
                 else len(anns) + 1,
                 "image_id": img, "track_id": tid, "category_id": cat,
                 "bbox": list(box),
                 "area": box[2] * box[3] if area is None else area[k],
                 "visibility": vis[k] if isinstance(vis, list) else vis,
                 "out_of_frame": oof[k] if isinstance(oof, list) else oof}
            if ignore is not None and ignore[k]:
                a["ignore"] = 1
            anns.append(a)

    # ---- video 10 -------------------------------------------------------
    box_a = [100, 100, 50, 50]
    # 101 and 102: identical boxes on every frame -> IoU tie (later GT wins)
    vis12 = [0.0, 0.05, 0.1, 0.1, 0.3, 0.8, 0.8, 0.9, 1.0, 1.0, 0.79, 0.81]
    add_gt(101, 1, 10, [(i, box_a) for i in v10_imgs], vis=vis12)
    add_gt(102, 1, 10, [(i, box_a) for i in v10_imgs], vis=vis12[::-1])
    # 103: duration exactly 3 (short AND medium), mean area exactly 1024
    add_gt(103, 2, 10, [(i, [300, 50, 32, 32]) for i in v10_imgs[:3]],
           vis=[0.8, 0.1, 1.0])
    # 104: duration exactly 10 (medium AND long), mean area exactly 9216
    add_gt(104, 2, 10, [(i, [400, 200, 96, 96]) for i in v10_imgs[:10]],
           vis=0.5)
    # 105: category 5 is merged into 4
    add_gt(105, 5, 10, [(i, [600, 300, 40, 60]) for i in v10_imgs[2:9]],
           vis=0.9, oof=[False, True, True, False, False, True, False])
    # 106: track-level ignore + some annotation-level ignore flags
    add_gt(106, 1, 10, [(i, [700, 100, 80, 80]) for i in v10_imgs[:6]],
           vis=1.0, ignore=[1, 0, 1, 0, 0, 1], trk_ignore=True)
    # 107: exactly 5 frames with visibility < 0.8 -> ignored by the HP range
    add_gt(107, 7, 10, [(i, [50, 400, 64, 64]) for i in v10_imgs[:8]],
           vis=[0.1, 0.2, 0.3, 0.4, 0.5, 0.8, 0.9, 1.0])
    # 108: exactly 6 such frames -> evaluated by the HP range
    add_gt(108, 7, 10, [(i, [250, 400, 64, 64]) for i in v10_imgs[:8]],
           vis=[0.1, 0.2, 0.3, 0.4, 0.5, 0.79, 0.9, 1.0])
    # 109: one zero-area annotation (dropped by the strict area>0 filter),
    #      negative / partly out-of-frame coordinates
    add_gt(109, 1, 10, [(v10_imgs[0], [-20, -10, 60, 40]),
                        (v10_imgs[1], [-20, -10, 0, 40]),
                        (v10_imgs[2], [1250, 700, 60, 40]),
                        (v10_imgs[3], [-20, -10, 60, 40])],
           vis=0.3, oof=True)
    # ---- video 20 -------------------------------------------------------
    add_gt(201, 1, 20, [(i, [10 + 5 * k, 20, 30, 30])
                        for k, i in enumerate(v20_imgs)], vis=0.95)
    add_gt(202, 2, 20, [(i, [500, 20 + 3 * k, 200, 100])
                        for k, i in enumerate(v20_imgs[:4])], vis=0.05,
           oof=True)
    add_gt(203, 3, 20, [(i, [900, 500, 20, 20]) for i in v20_imgs[5:7]],
           vis=0.6)
    # category 9: ground truth but never predicted (precision 0, recall 0)
    add_gt(204, 9, 20, [(i, [40, 600, 100, 100]) for i in v20_imgs[:5]],
           vis=1.0)
    # ---- video 30: id sentinels ----------------------------------------
    # annotation id 0 (LVIS "unmatched" sentinel) and track id -1 (TAO one)
    add_gt(-1, 1, 30, [(i, [100, 100, 100, 100]) for i in v30_imgs],
           vis=1.0, first_ann_id=0)
    add_gt(301, 1, 30, [(i, [400, 100, 100, 100]) for i in v30_imgs],
           vis=0.5, first_ann_id=9001)

    preds = []

    def add_dt(tid, cat, vid, frames_boxes, score):
        for k, (img, box) in enumerate(frames_boxes):
            preds.append({"image_id": img, "category_id": cat,
                          "bbox": list(box),
                          "score": score[k] if isinstance(score, list)
                          else score,
                          "track_id": tid, "video_id": vid})

    # exact copy of the tied GT pair
    add_dt(1, 1, 10, [(i, box_a) for i in v10_imgs], 0.9)
    # second detection on the pair with a smaller IoU, equal score to a third
    add_dt(2, 1, 10, [(i, [104, 104, 50, 50]) for i in v10_imgs], 0.8)
    add_dt(3, 1, 10, [(i, [96, 98, 52, 50]) for i in v10_imgs], 0.8)
    # matches only the ignored track 106
    add_dt(4, 1, 10, [(i, [702, 101, 80, 80]) for i in v10_imgs[:6]], 0.7)
    # not-exhaustive category 2: an unmatched one (ignored) and matches
    add_dt(5, 2, 10, [(i, [1000, 600, 30, 30]) for i in v10_imgs[:4]], 0.6)
    add_dt(6, 2, 10, [(i, [301, 51, 32, 32]) for i in v10_imgs[:3]],
           [0.5, 0.7, 0.9])                      # non-uniform scores -> mean
    add_dt(7, 2, 10, [(i, [398, 203, 99, 94]) for i in v10_imgs[:10]], 0.55)
    # negative category 3: plain false positives
    add_dt(8, 3, 10, [(i, [10, 10, 20, 20]) for i in v10_imgs[:5]], 0.95)
    # merged category: predicted as 5, scored as 4
    add_dt(9, 5, 10, [(i, [601, 301, 40, 60]) for i in v10_imgs[2:9]], 0.85)
    # category 7: HP tracks
    add_dt(10, 7, 10, [(i, [52, 402, 64, 64]) for i in v10_imgs[:8]], 0.75)
    add_dt(11, 7, 10, [(i, [251, 399, 64, 64]) for i in v10_imgs[:8]], 0.65)
    # category 6 in video 10: neither present nor negative -> dropped
    add_dt(12, 6, 10, [(i, [5, 5, 50, 50]) for i in v10_imgs[:3]], 0.99)
    # category 42 does not exist in the ground truth -> dropped
    add_dt(13, 42, 10, [(i, [5, 5, 50, 50]) for i in v10_imgs[:2]], 0.98)
    # zero-width detection (dropped) inside an otherwise valid track,
    # negative coordinates
    add_dt(14, 1, 10, [(v10_imgs[0], [-18, -12, 60, 40]),
                       (v10_imgs[1], [-18, -12, 0, 40]),
                       (v10_imgs[2], [1248, 698, 60, 40]),
                       (v10_imgs[3], [-20, -10, 60, 40])], 0.45)
    # track id 7 / 1 re-used in video 20 (make_track_ids_unique)
    add_dt(7, 1, 20, [(i, [12 + 5 * k, 21, 30, 30])
                      for k, i in enumerate(v20_imgs)], 0.9)
    add_dt(1, 2, 20, [(i, [480, 25 + 3 * k, 210, 100])
                      for k, i in enumerate(v20_imgs[:4])], 0.4)
    add_dt(20, 6, 20, [(i, [100, 100, 10, 10]) for i in v20_imgs[:2]], 0.3)
    add_dt(21, 3, 20, [(i, [901, 501, 20, 20]) for i in v20_imgs[4:8]], 0.35)
    # category 7 is unlisted in video 20 -> dropped
    add_dt(22, 7, 20, [(i, [300, 300, 64, 64]) for i in v20_imgs[:3]], 0.97)
    # more than 300 detections in one image (top-300 cut, equal scores)
    big = v20_imgs[9]
    for k in range(330):
        add_dt(1000 + k, 1 if k % 3 else 6, 20,
               [(big, [(k * 37) % 1200, (k * 91) % 650, 20 + k % 50,
                       20 + (k * 7) % 60])],
               round(0.05 + (k % 40) * 0.02, 2))
    # video 30: detection track id 0 (never consumes its GT) + two others
    add_dt(0, 1, 30, [(i, [100, 100, 100, 100]) for i in v30_imgs], 0.9)
    add_dt(31, 1, 30, [(i, [102, 100, 100, 100]) for i in v30_imgs], 0.8)
    add_dt(32, 1, 30, [(i, [398, 101, 100, 100]) for i in v30_imgs], 0.7)
    add_dt(33, 1, 30, [(i, [403, 100, 100, 100]) for i in v30_imgs], 0.6)

    gt = {"info": {"description": "quirks"}, "images": images,
          "videos": videos, "tracks": tracks, "annotations": anns,
          "categories": cats}
    return gt, preds


def f7():
    rng = np.random.default_rng(77)
    cats = [{"id": c, "name": "c%d" % c, "frequency": "rcf"[c % 3]}
            for c in (1, 2, 3)]
    videos, images, tracks, anns, preds = [], [], [], [], []
    thr = [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]
    next_img, next_trk, next_dt = 1, 1, 1
    for v in range(1, 5):
        videos.append({"id": v, "name": "v%d" % v, "neg_category_ids": [],
                       "not_exhaustive_category_ids": []})
        n_fr = 8 + 3 * v
        # ids spread so that the sets' slot order is not the timeline order
        ids = (next_img + rng.permutation(n_fr * 37)[:n_fr]).tolist()
        next_img += n_fr * 37
        for f, i in enumerate(ids):
            images.append({"id": int(i), "video_id": v, "frame_index": 30 * f,
                           "neg_category_ids": [],
                           "not_exhaustive_category_ids": []})
        for g in range(6):
            cat = 1 + g % 3
            lo = int(rng.integers(0, n_fr - 5))
            hi = int(rng.integers(lo + 5, n_fr + 1))
            x = rng.integers(0, 900, hi - lo) + rng.integers(0, 100, hi - lo) / 100
            y = rng.integers(0, 500, hi - lo) + rng.integers(0, 100, hi - lo) / 100
            w = rng.integers(20, 300, hi - lo) + rng.integers(0, 10, hi - lo) / 10
            # heights are multiples of 0.2: h * t has at most two decimals
            h = rng.integers(100, 1000, hi - lo) / 5.0
            tracks.append({"id": next_trk, "category_id": cat, "video_id": v})
            for k in range(hi - lo):
                box = [float(x[k]), float(y[k]), float(w[k]), float(h[k])]
                anns.append({"id": len(anns) + 1, "image_id": int(ids[lo + k]),
                             "track_id": next_trk, "category_id": cat,
                             "bbox": box, "area": box[2] * box[3],
                             "visibility": 1.0, "out_of_frame": False})
            # detection tracks covering exactly the same frames, the same
            # boxes with the height scaled by a threshold: IoU = t per frame
            for t in rng.permutation(thr)[:5]:
                score = float(rng.integers(1, 1000)) / 1000
                for k in range(hi - lo):
                    hh = float(np.round(h[k] * t, 2))
                    preds.append({"image_id": int(ids[lo + k]),
                                  "category_id": cat,
                                  "bbox": [float(x[k]), float(y[k]), float(w[k]), hh],
                                  "score": score, "track_id": next_dt,
                                  "video_id": v})
                next_dt += 1
            next_trk += 1
    gt = {"info": {"description": "threshold-straddling 3D IoUs"},
          "images": images, "videos": videos, "tracks": tracks,
          "annotations": anns, "categories": cats}
    return gt, preds


def f8():
    gt, dt = synth(seed=88, V=16, F=300, C=100, dets_per_frame=10,
                   decimal=True, shuffle_image_ids=True)
    return gt.to_json(), dt.to_json()


ALL = {"f1": f1, "f2": f2, "f3": f3, "f4": f4, "f5": f5, "f7": f7, "f8": f8}
# big fixtures: inputs stored gzipped, image level reduced to the integer
# match counts + precision / recall + results + text (make_golden.py)
LITE = {"f8"}
