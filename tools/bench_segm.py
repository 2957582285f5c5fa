#!/usr/bin/env python
"""Measurement of the iou_type="segm" widening:
run-length mask IoU on the GPU (taoamd_rle_iou) next to the reference's own
rleIou (oracle/_ref, one host core) on a sample of the same cells.

    python tools/bench_segm.py [--images 2000] [--dets 50] [--gts 10]

Synthetic workload: 1280 x 720 frames, polygons of 8-16 vertices around
jittered boxes (detections are perturbed copies of ground truths, so the
tight boxes overlap and the runs are actually walked).  Prints one JSON
line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


def poly(rng, box, n):
    x, y, w, h = box
    cx, cy = x + w / 2, y + h / 2
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = rng.uniform(0.8, 1.0, n)
    return np.c_[cx + rad * w / 2 * np.cos(ang), cy + rad * h / 2 * np.sin(ang)].ravel().tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2000)
    ap.add_argument("--dets", type=int, default=50)
    ap.add_argument("--gts", type=int, default=10)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cpu-cells", type=int, default=60)
    a = ap.parse_args()
    import torch
    from tao_amodal_amd import _lib
    from tao_amodal_amd.masks import MaskBatch
    H, W = 720, 1280
    rng = np.random.default_rng(7)
    db, gb = MaskBatch(), MaskBatch()
    t0 = time.perf_counter()
    g_items, d_items = [], []
    for _ in range(a.images):
        boxes = np.c_[rng.uniform(-50, W - 100, a.gts), rng.uniform(-50, H - 100, a.gts),
                      rng.uniform(40, 400, a.gts), rng.uniform(40, 400, a.gts)]
        for b in boxes:
            g_items.append(([poly(rng, b, int(rng.integers(8, 17)))], H, W))
        for d in range(a.dets):
            b = boxes[d % a.gts] + rng.uniform(-12, 12, 4)
            b[2:] = np.maximum(b[2:], 8)
            d_items.append(([poly(rng, b, int(rng.integers(8, 17)))], H, W))
    t0 = time.perf_counter()            # the polygons exist: time the mask build only
    gb.add_many(g_items)
    db.add_many(d_items)
    dt, gt = db.arrays(), gb.arrays()
    t_build = time.perf_counter() - t0
    n_cells = a.images
    cd = np.full(n_cells, a.dets, np.int64)
    cg = np.full(n_cells, a.gts, np.int64)
    d_off = np.r_[0, np.cumsum(cd)].astype(np.int32)
    g_off = np.r_[0, np.cumsum(cg)].astype(np.int32)
    i_off = np.r_[0, np.cumsum(cd * cg)].astype(np.int64)
    dev = "cuda"
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    head = [t(d_off), t(g_off), t(i_off)]
    sides = [[t(m.off), t(m.counts.view(np.int32)), t(m.hw), t(m.bbox)] for m in (dt, gt)]
    out = torch.empty(int(i_off[-1]), dtype=torch.float64, device=dev)
    lib = _lib.load()
    nb = lib.taoamd_rle_iou_workspace(len(dt), int(dt.off[-1]), len(gt), int(gt.off[-1]))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def launch():
        assert lib.taoamd_rle_iou(
            n_cells, *[x.data_ptr() for x in head],
            len(dt), int(dt.off[-1]), *[x.data_ptr() for x in sides[0]],
            len(gt), int(gt.off[-1]), *[x.data_ptr() for x in sides[1]],
            out.data_ptr(), ws.data_ptr(), nb, s) == 0
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(a.reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    got = out.cpu().numpy()
    pairs = int(i_off[-1])
    # runs walked: every pair with overlapping tight boxes reads both lists
    kd, kg = np.diff(dt.off), np.diff(gt.off)
    ov = got > 0
    walked = 0
    for c in range(n_cells):
        o = ov[i_off[c]:i_off[c + 1]].reshape(a.dets, a.gts)
        walked += int((o * (kd[d_off[c]:d_off[c + 1], None] + kg[None, g_off[c]:g_off[c + 1]])).sum())
    once = (len(dt.counts) + len(gt.counts)) * 4 + pairs * 8 + (len(dt) + len(gt)) * 48
    # reference C on a sample of the cells, one core
    import orclib
    cells = list(range(0, n_cells, max(1, n_cells // a.cpu_cells)))[:a.cpu_cells]
    ds = [[dt.mask(i) for i in range(d_off[c], d_off[c + 1])] for c in cells]
    gs = [[gt.mask(j) for j in range(g_off[c], g_off[c + 1])] for c in cells]
    r = orclib._ref()
    import ctypes as C
    prepared = []
    for D, G in zip(ds, gs):
        Dm = (orclib._RefRLE * len(D))(*[orclib._ref_make(r, m) for m in D])
        Gm = (orclib._RefRLE * len(G))(*[orclib._ref_make(r, m) for m in G])
        prepared.append((Dm, Gm, np.zeros(len(D) * len(G)), np.zeros(len(G), np.uint8)))
    t0 = time.perf_counter()
    for Dm, Gm, o, crowd in prepared:
        r.rleIou(Dm, Gm, C.c_ulong(len(Dm)), C.c_ulong(len(Gm)),
                 crowd.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
    t_cpu = time.perf_counter() - t0
    ok = all(np.array_equal(o.reshape((len(Dm), len(Gm)), order="F").ravel(),
                            got[i_off[c]:i_off[c + 1]])
             for c, (Dm, Gm, o, _) in zip(cells, prepared))
    cpu_pairs = sum(len(Dm) * len(Gm) for Dm, Gm, _, _ in prepared)
    print(json.dumps({
        "metric": "run-length mask IoU throughput", "unit": "Mpair/s",
        "value": round(pairs / ms / 1e3, 1), "ms_per_launch": round(ms, 4),
        "pairs": pairs, "pairs_overlapping": int(ov.sum()),
        "runs_per_mask": round(float((len(dt.counts) + len(gt.counts)) / (len(dt) + len(gt))), 1),
        "roofline": {"bound": "hbm", "unit": "GB/s",
                     "achieved": round(once / ms / 1e6, 1), "peak": 8000,
                     "frac": round(once / ms / 1e6 / 8000, 4),
                     "bytes_once": once, "bytes_walked": walked * 4},
        "cpu_baseline": {"kind": "reference", "value": round(cpu_pairs / t_cpu / 1e6, 3),
                         "unit": "Mpair/s", "cores": 1,
                         "sample": "%d cells (%d pairs) through the reference's rleIou" % (len(cells), cpu_pairs),
                         "equal_to_gpu": bool(ok)},
        "host_mask_build_s": round(t_build, 2),
        "config": {"workload": "%d images x %d dets x %d gts, 1280x720 polygons" % (a.images, a.dets, a.gts)},
    }))


if __name__ == "__main__":
    main()
