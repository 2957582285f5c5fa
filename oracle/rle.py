"""CPU restatement of the run-length mask arithmetic behind
``LVISEval(iou_type="segm")`` -- TEST INFRASTRUCTURE ONLY (tests/, smoke(),
bench cpu_baseline), never imported by the product path.

The reference reaches this code through ``pycocotools.mask`` (a third-party
dependency, vendored in the reference tree under
visualization/tao/third_party/pysot/training_dataset/coco/pycocotools):
``M/`` below is its ``common/maskApi.c``, ``P/`` its ``_mask.pyx``.  Pinned
against (i) the golden vectors of tests/golden/f6 (the reference evaluator run
in the development container) and (ii) oracle/_ref, that same maskApi.c
compiled where it lies.

A mask is a dict {"h", "w", "counts": list[int]}: column-major run lengths,
starting with a run of zeros (possibly empty).
"""
import math

import numpy as np

INT_MIN = -(1 << 31)


def _trunc(x):
    """C's (int) conversion of a double: toward zero."""
    return int(x)


# ---------------------------------------------------------------- M/161-202
def fr_poly(xy, h, w):
    """Rasterise one polygon (x0, y0, x1, y1, ...) -- rleFrPoly.

    Three steps, as there: (1) walk every edge on a 5x finer integer grid,
    one point per step along the longer axis; (2) wherever consecutive points
    change column, a run boundary falls at (column, ceil(y)) after scaling
    back -- kept only for whole columns inside the frame, y clamped to
    [0, h]; (3) sort the boundaries as linear column-major offsets, take
    differences, fold zero-length runs into their neighbours."""
    k = len(xy) // 2
    scale = 5.0
    px = [_trunc(scale * xy[2 * j] + .5) for j in range(k)]
    py = [_trunc(scale * xy[2 * j + 1] + .5) for j in range(k)]
    us, vs = [], []
    for j in range(k):
        xs, ys = px[j], py[j]
        xe, ye = px[(j + 1) % k], py[(j + 1) % k]
        dx, dy = abs(xe - xs), abs(ys - ye)
        if dx == 0 and dy == 0:
            # a repeated vertex: the reference divides 0/0 here and stores an
            # undefined row for this single point, which is never read (its
            # neighbours are in the same column); any row does
            us.append(xs)
            vs.append(ys)
            continue
        along_x = dx >= dy
        flip = (along_x and xs > xe) or (not along_x and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if along_x:
            s = (ye - ys) / dx
            for d in range(dx + 1):
                t = dx - d if flip else d
                us.append(t + xs)
                vs.append(_trunc(ys + s * t + .5))
        else:
            s = (xe - xs) / dy
            for d in range(dy + 1):
                t = dy - d if flip else d
                vs.append(t + ys)
                us.append(_trunc(xs + s * t + .5))
    cuts = []
    for j in range(1, len(us)):
        if us[j] == us[j - 1]:
            continue
        xd = float(us[j] if us[j] < us[j - 1] else us[j] - 1)
        xd = (xd + .5) / scale - .5
        if math.floor(xd) != xd or xd < 0 or xd > w - 1:
            continue
        yd = float(min(vs[j], vs[j - 1]))
        yd = (yd + .5) / scale - .5
        yd = 0.0 if yd < 0 else (float(h) if yd > h else yd)
        cuts.append(int(xd) * h + int(math.ceil(yd)))
    cuts.append(h * w)
    cuts.sort()
    diffs = [cuts[0]] + [cuts[i] - cuts[i - 1] for i in range(1, len(cuts))]
    counts = [diffs[0]]
    j = 1
    while j < len(diffs):
        if diffs[j] > 0:
            counts.append(diffs[j])
            j += 1
        else:               # an empty run: the next one continues the previous
            j += 1
            if j < len(diffs):
                counts[-1] += diffs[j]
                j += 1
    return {"h": h, "w": w, "counts": counts}


def fr_bbox(bb, h, w):
    """rleFrBbox (M/148-156): the box as a four-corner polygon."""
    xs, ys = bb[0], bb[1]
    xe, ye = xs + bb[2], ys + bb[3]
    return fr_poly([xs, ys, xs, ye, xe, ye, xe, ys], h, w)


# ------------------------------------------------------------------ M/49-71
class _Runs:
    """Cursor over a mask's runs: `left` pixels remain in the current run of
    value `v`; exhausted runs are followed while there is a next one."""

    def __init__(self, counts):
        self.c, self.i = counts, 1
        self.left, self.v = counts[0], 0

    def take(self, n):
        self.left -= n
        if self.left == 0 and self.i < len(self.c):
            self.left = self.c[self.i]
            self.i += 1
            self.v ^= 1


def merge(masks, intersect=False):
    """Union (or intersection) of masks, folded left to right -- rleMerge."""
    if len(masks) == 0:
        return {"h": 0, "w": 0, "counts": []}
    h, w = masks[0]["h"], masks[0]["w"]
    acc = list(masks[0]["counts"])
    if len(masks) == 1:
        return {"h": h, "w": w, "counts": acc}
    for other in masks[1:]:
        if other["h"] != h or other["w"] != w:
            return {"h": 0, "w": 0, "counts": []}
        a, b = _Runs(acc), _Runs(other["counts"])
        out, run, v = [], 0, 0
        while True:
            n = min(a.left, b.left)
            run += n
            a.take(n)
            b.take(n)
            rest = a.left + b.left
            nv = (a.v & b.v) if intersect else (a.v | b.v)
            if nv != v or rest == 0:
                out.append(run)
                run = 0
            v = nv
            if rest == 0:
                break
        acc = out
    return {"h": h, "w": w, "counts": acc}


def area(mask):
    """rleArea (M/72-75): the odd-numbered runs are the ones."""
    return int(sum(mask["counts"][1::2]))


# ---------------------------------------------------------------- M/133-147
def to_bbox(mask):
    """Tight box [x, y, w, h] of the runs' end points -- rleToBbox.  Only an
    even number of runs is looked at; no run at all gives four zeros."""
    h, w = mask["h"], mask["w"]
    c = mask["counts"]
    m = (len(c) // 2) * 2
    if m == 0:
        return [0.0, 0.0, 0.0, 0.0]
    xs, ys, xe, ye = w, h, 0, 0
    cc = 0
    for j in range(m):
        cc += c[j]
        t = (cc - (j % 2)) & 0xFFFFFFFF      # unsigned arithmetic of the source
        y = t % h
        x = (t - y) // h
        xs, xe, ys, ye = min(xs, x), max(xe, x), min(ys, y), max(ye, y)
    return [float(xs), float(ys), float((xe - xs + 1) & 0xFFFFFFFF),
            float((ye - ys + 1) & 0xFFFFFFFF)]


# ---------------------------------------------------------------- M/203-230
def to_string(mask):
    """LEB128-like text form: 5 data bits + a continuation bit per character,
    offset 48; from the fourth run on the difference to the run two places
    back is stored -- rleToString."""
    c = mask["counts"]
    out = []
    for i, x in enumerate(c):
        x = int(x)
        if i > 2:
            x -= int(c[i - 2])
        while True:
            ch = x & 0x1f
            x >>= 5                      # arithmetic shift, as on a C long
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
            if not more:
                break
    return "".join(out)


def fr_string(s, h, w):
    """Inverse of to_string -- rleFrString."""
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts = []
    p = 0
    while p < len(s):
        x, k = 0, 0
        while True:
            ch = ord(s[p]) - 48
            x |= (ch & 0x1f) << (5 * k)
            more = ch & 0x20
            p += 1
            k += 1
            if not more:
                if ch & 0x10:
                    x |= -1 << (5 * k)
                break
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x & 0xFFFFFFFF)
    return {"h": h, "w": w, "counts": counts}


# ------------------------------------------------------------------ M/77-96
def bb_overlap(db, gb):
    """bbIou (M/109-121) is only asked whether it is positive here."""
    wd = min(db[2] + db[0], gb[2] + gb[0]) - max(db[0], gb[0])
    if wd <= 0:
        return False
    ht = min(db[3] + db[1], gb[3] + gb[1]) - max(db[1], gb[1])
    return ht > 0


def iou_pair(d, g, db=None, gb=None):
    """IoU of two masks, iscrowd = 0 -- the body of rleIou: 0 when the tight
    boxes do not overlap, -1 when the frames differ, else |d & g| / |d | g|
    with an empty intersection reported over a union of 1."""
    db = to_bbox(d) if db is None else db
    gb = to_bbox(g) if gb is None else gb
    if not bb_overlap(db, gb):
        return 0.0
    if d["h"] != g["h"] or d["w"] != g["w"]:
        return -1.0
    a, b = _Runs(d["counts"]), _Runs(g["counts"])
    inter = union = 0
    while True:
        n = min(a.left, b.left)
        if a.v or b.v:
            union += n
            if a.v and b.v:
                inter += n
        a.take(n)
        b.take(n)
        if a.left + b.left == 0:
            break
    if inter == 0:
        union = 1
    return float(inter) / float(union)


def iou_matrix(dts, gts):
    """mask_utils.iou(dt, gt, iscrowd=0) for lists of masks: [] when either
    side is empty (P/203-204), else a (D, G) array."""
    if len(dts) == 0 or len(gts) == 0:
        return []
    db = [to_bbox(d) for d in dts]
    gb = [to_bbox(g) for g in gts]
    out = np.zeros((len(dts), len(gts)))
    for i, d in enumerate(dts):
        for j, g in enumerate(gts):
            out[i, j] = iou_pair(d, g, db[i], gb[j])
    return out


# ------------------------------------------------- P/277-310, L/lvis.py:171-193
def fr_py_objects(obj, h, w):
    """mask_utils.frPyObjects for the shapes LVIS.ann_to_rle passes: a list of
    polygons (a list whose first element has exactly four numbers is taken as
    a list of boxes, P/284-285), or one uncompressed RLE dict.  Every result
    goes through the text form and back, as in _mask.pyx (_toString)."""
    def roundtrip(m):
        return fr_string(to_string(m), m["h"], m["w"])
    if isinstance(obj, list):
        if len(obj[0]) == 4:
            # taken for a list of boxes and handed to frBbox, whose argument
            # is typed as an ndarray: Cython rejects the list
            raise TypeError("Argument 'bb' has incorrect type (expected "
                            "numpy.ndarray, got list)")
        if len(obj[0]) > 4:
            return [roundtrip(fr_poly(p, h, w)) for p in obj]
        raise Exception("input type is not supported.")
    if isinstance(obj, dict) and "counts" in obj and "size" in obj:
        m = {"h": obj["size"][0], "w": obj["size"][1],
             "counts": [int(c) & 0xFFFFFFFF for c in obj["counts"]]}
        return roundtrip(m)
    raise Exception("input type is not supported.")


def ann_to_rle(segm, h, w):
    """LVIS.ann_to_rle: polygons are rasterised and united, an uncompressed
    RLE keeps its own size, a compressed one is taken as is."""
    if isinstance(segm, list):
        parts = fr_py_objects(segm, h, w)
        m = merge(parts)
        return fr_string(to_string(m), m["h"], m["w"])
    if isinstance(segm["counts"], list):
        return fr_py_objects(segm, h, w)
    return fr_string(segm["counts"], segm["size"][0], segm["size"][1])
