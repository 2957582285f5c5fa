"""Seeded synthetic TAO-Amodal-shaped inputs (``SYNTH``).

There is no dataset in the container, so every config of BASELINE.json other
than the first is driven by this generator.  By default all box coordinates are
integers: every per-frame product and every sum of products is then exact in
fp64, so the 3D-IoU of a track pair does not depend on the order frames are
summed in (the reference sums in CPython set-iteration order, reference
tao_amodal/evaluation/tao_amodal/eval.py:83-94).  ``decimal=True`` gives
coordinates like the real files have (ground truth to 2 decimals, predictions
to 3): per-frame terms are then inexact, the sums depend on the order, and the
evaluation goes through the frame-order guard (engine.stage_iou_guard).

Output is columnar (``GTColumns`` / ``DTColumns``); call ``.to_json()`` on the
pair to obtain the ``validation_lvis_v1.json`` / ``prediction.json`` shaped
objects of the reference contract.
"""
import numpy as np

from .columns import DTColumns, GTColumns

VIS_LEVELS = np.array([0.0, 0.05, 0.1, 0.3, 0.8, 0.9, 1.0])


def _seg_cumsum(steps, track_of_box, first_box_of_track):
    """Cumulative sum of ``steps`` restarted at every track start."""
    cs = np.cumsum(steps, axis=0)
    base = cs[first_box_of_track] - steps[first_box_of_track]
    return cs - base[track_of_box]


def _walk(rng, n_trk, lens, W, H):
    """Integer random-walk boxes for n_trk tracks with the given lengths."""
    total = int(lens.sum())
    trk = np.repeat(np.arange(n_trk), lens)
    first = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    x0 = rng.integers(-64, W - 64 + 1, n_trk)
    y0 = rng.integers(-64, H - 64 + 1, n_trk)
    w0 = rng.integers(8, 401, n_trk)
    h0 = rng.integers(8, 401, n_trk)
    steps = rng.integers(-4, 5, (total, 4))
    steps[first] = 0
    d = _seg_cumsum(steps, trk, first)
    x = np.clip(x0[trk] + d[:, 0], -64, W - 64)
    y = np.clip(y0[trk] + d[:, 1], -64, H - 64)
    w = np.clip(w0[trk] + d[:, 2], 8, 400)
    h = np.clip(h0[trk] + d[:, 3], 8, 400)
    return trk, first, np.stack([x, y, w, h], 1).astype(np.float64)


def synth(seed=20240807, V=200, F=300, C=1203, dets_per_frame=50,
          gt_tracks_per_video=10, n_present=5, n_neg=2, W=1280, H=720,
          shuffle_image_ids=False, collide_track_ids=False, n_merged=0,
          video_id_base=0, decimal=False):
    """Generate one synthetic (ground truth, predictions) pair.

    ``video_id_base`` offsets every id so that independently generated shards
    (one per rank in the weak-scaling bench) can be concatenated.
    """
    rng = np.random.default_rng(seed)
    cat_id = np.arange(1, C + 1, dtype=np.int64)
    cat_freq = np.array([ord("rcf"[c % 3]) for c in cat_id], dtype=np.uint8)
    # optional merged categories: the last n_merged ids are folded into 1..n
    cat_merged = np.array([[C - k, k + 1] for k in range(n_merged)],
                          dtype=np.int64).reshape(-1, 2)

    n_pick = n_present + n_neg + 1
    assert C >= n_pick
    vid_id = np.arange(1, V + 1, dtype=np.int64) + video_id_base
    n_img = V * F
    img_ids_all = np.arange(1, n_img + 1, dtype=np.int64) + video_id_base * F
    if shuffle_image_ids:
        img_ids_all = rng.permutation(img_ids_all)

    vid_neg, vid_nel = [], []
    img_vid = np.repeat(vid_id, F)
    img_frame = np.tile(30.0 * np.arange(F), V)

    G = gt_tracks_per_video
    gt_parts = {k: [] for k in ("img", "trk", "cat", "bbox", "vis")}
    trk_id, trk_cat, trk_vid = [], [], []
    dt_parts = {k: [] for k in ("img", "cat", "bbox", "score", "trk", "vid")}
    next_gt_trk = 1 + video_id_base * G
    next_dt_trk = 1 + video_id_base * 100000

    for v in range(V):
        picks = rng.choice(C, n_pick, replace=False) + 1
        present = picks[:n_present]
        neg = picks[n_present:n_present + n_neg]
        unlisted = picks[-1:]
        vid_neg.append(neg.tolist())
        vid_nel.append([int(present[0])] if rng.random() < 0.25 else [])
        img_of_frame = img_ids_all[v * F:(v + 1) * F]

        # ---------------------------------------------------- ground truth
        g_len = rng.integers(1, F + 1, G)
        g_start = (rng.random(G) * (F - g_len + 1)).astype(np.int64)
        g_cat = present[np.arange(G) % n_present]
        g_trk, g_first, g_box = _walk(rng, G, g_len, W, H)
        g_frame = g_start[g_trk] + (np.arange(len(g_trk)) - g_first[g_trk])
        g_ids = next_gt_trk + np.arange(G)
        next_gt_trk += G
        gt_parts["img"].append(img_of_frame[g_frame])
        gt_parts["trk"].append(g_ids[g_trk])
        gt_parts["cat"].append(g_cat[g_trk])
        gt_parts["bbox"].append(g_box)
        gt_parts["vis"].append(VIS_LEVELS[rng.integers(0, len(VIS_LEVELS),
                                                       len(g_trk))])
        trk_id.append(g_ids)
        trk_cat.append(g_cat)
        trk_vid.append(np.full(G, vid_id[v]))
        # dense per-frame GT boxes, held at the end points outside the life
        dense = np.zeros((G, F, 4))
        for g in range(G):
            fr = np.clip(np.arange(F), g_start[g], g_start[g] + g_len[g] - 1)
            dense[g] = g_box[g_first[g] + fr - g_start[g]]

        # ------------------------------------------------------ predictions
        starts, lens = [], []
        for _ in range(dets_per_frame):
            f = 0
            while f < F:
                if rng.random() < 0.3:
                    ln = int(rng.integers(1, 13))
                else:
                    ln = int(rng.integers(1, F + 1))
                ln = min(ln, F - f)
                starts.append(f)
                lens.append(ln)
                f += ln
        starts = np.asarray(starts, dtype=np.int64)
        lens = np.asarray(lens, dtype=np.int64)
        T = len(lens)
        d_trk, d_first, d_box = _walk(rng, T, lens, W, H)
        d_frame = starts[d_trk] + (np.arange(len(d_trk)) - d_first[d_trk])
        is_copy = rng.random(T) < 0.7
        ref = rng.integers(0, G, T)
        pool = np.concatenate([present, neg, unlisted])
        d_cat = np.where(is_copy, g_cat[ref], pool[rng.integers(0, len(pool),
                                                                T)])
        cp = is_copy[d_trk]
        src = dense[ref[d_trk], d_frame]
        jit_xy = rng.integers(-8, 9, (len(d_trk), 2))
        scale = rng.uniform(0.8, 1.2, (len(d_trk), 2))
        copied = np.empty_like(src)
        copied[:, :2] = src[:, :2] + jit_xy
        copied[:, 2:] = np.maximum(np.rint(src[:, 2:] * scale), 1.0)
        d_box = np.where(cp[:, None], copied, d_box)
        base = rng.random(T)
        noisy = rng.random(T) < 0.5
        noise = rng.normal(0.0, 0.05, len(d_trk)) * noisy[d_trk]
        d_score = np.clip(base[d_trk] + noise, 0.0, 1.0)
        if collide_track_ids:
            d_ids = 1 + np.arange(T)
        else:
            d_ids = next_dt_trk + np.arange(T)
            next_dt_trk += T
        dt_parts["img"].append(img_of_frame[d_frame])
        dt_parts["cat"].append(d_cat[d_trk])
        dt_parts["bbox"].append(d_box)
        dt_parts["score"].append(d_score)
        dt_parts["trk"].append(d_ids[d_trk])
        dt_parts["vid"].append(np.full(len(d_trk), vid_id[v]))

    from .columns import _csr
    vneg = _csr(vid_neg)
    vnel = _csr(vid_nel)
    ineg = _csr([x for x in vid_neg for _ in range(F)])
    inel = _csr([x for x in vid_nel for _ in range(F)])
    bbox = np.concatenate(gt_parts["bbox"])
    dt_bbox = np.concatenate(dt_parts["bbox"])
    if decimal:
        # (its own stream: the integer sets of a seed stay what they were)
        drng = np.random.default_rng([seed, 0xdec])
        bbox = bbox + np.round(drng.random(bbox.shape), 2)
        dt_bbox = dt_bbox + np.round(drng.random(dt_bbox.shape), 3)
    x, y, w, h = bbox.T
    oof = ((x < 0) | (y < 0) | (x + w > W) | (y + h > H)).astype(np.uint8)
    n_ann = len(bbox)
    tid = np.concatenate(trk_id)
    gt = GTColumns(
        cat_id=cat_id, cat_freq=cat_freq, cat_merged=cat_merged,
        vid_id=vid_id, vid_neg_off=vneg[0], vid_neg=vneg[1],
        vid_nel_off=vnel[0], vid_nel=vnel[1],
        img_id=img_ids_all, img_vid=img_vid, img_frame=img_frame,
        img_neg_off=ineg[0], img_neg=ineg[1],
        img_nel_off=inel[0], img_nel=inel[1],
        trk_id=tid, trk_cat=np.concatenate(trk_cat),
        trk_vid=np.concatenate(trk_vid),
        trk_ignore=np.zeros(len(tid), dtype=np.uint8),
        ann_id=np.arange(1, n_ann + 1, dtype=np.int64) + video_id_base * 10 ** 6,
        ann_img=np.concatenate(gt_parts["img"]),
        ann_trk=np.concatenate(gt_parts["trk"]),
        ann_cat=np.concatenate(gt_parts["cat"]),
        ann_bbox=bbox, ann_area=w * h,
        ann_vis=np.concatenate(gt_parts["vis"]),
        ann_oof=oof, ann_ignore=np.zeros(n_ann, dtype=np.uint8),
    )
    dt = DTColumns(
        image_id=np.concatenate(dt_parts["img"]),
        category_id=np.concatenate(dt_parts["cat"]).astype(np.int64),
        bbox=dt_bbox,
        score=np.concatenate(dt_parts["score"]),
        track_id=np.concatenate(dt_parts["trk"]).astype(np.int64),
        video_id=np.concatenate(dt_parts["vid"]).astype(np.int64),
    )
    # predictions arrive in an arbitrary file order in practice
    perm = rng.permutation(len(dt))
    return gt, dt.take(perm)
