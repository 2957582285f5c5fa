"""ctypes access to the C oracle (oracle/_build/liboracle.so) for the tests
and bench.py's cpu_baseline leg.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmaskapi_ref.so")
N_THR, N_REC = 10, 101
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ROOT, "oracle", "tao_oracle.c")
        if (not os.path.exists(SO)
                or os.path.getmtime(SO) < os.path.getmtime(src)):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"),
                            "_build/liboracle.so"], check=True,
                           stdout=subprocess.DEVNULL)
        _lib = C.CDLL(SO)
        _lib.orc_track_iou.restype = C.c_int64
    return _lib


def host_cpus():
    """CPUs this process can actually use: the logical CPUs OpenMP sees,
    capped by the control group's CPU-time quota (cgroup v2 cpu.max, v1
    cfs_quota_us / cfs_period_us) -- threads beyond the quota are throttled,
    not run."""
    n = int(lib().orc_max_threads())
    quota = period = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota, period = int(q), int(p)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota and period and quota > 0:
        n = min(n, -(-quota // period))
    return max(1, n)


def set_threads(n):
    """OpenMP threads of the oracle's cell / category loops (1 = the scalar
    port; 0 = every CPU the host grants this process).  Returns the number now
    in use."""
    n = int(n) if n else host_cpus()
    lib().orc_set_threads(C.c_int(n))
    return n


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def thresholds():
    a, b = np.zeros(N_THR), np.zeros(N_REC)
    lib().orc_thresholds(_p(a), _p(b))
    return a, b


def bb_iou(dt, gt):
    dt, gt = _c(dt, np.float64).reshape(-1, 4), _c(gt, np.float64).reshape(-1, 4)
    m, n = len(dt), len(gt)
    o = np.zeros(m * n)
    lib().orc_bb_iou(_p(dt), _p(gt), C.c_size_t(m), C.c_size_t(n), _p(o))
    return o.reshape((m, n), order="F")


def ref_bb_iou(dt, gt):
    """The reference's own compiled bbIou (oracle/_ref), iscrowd = 0."""
    r = C.CDLL(REF_SO)
    dt, gt = _c(dt, np.float64).reshape(-1, 4), _c(gt, np.float64).reshape(-1, 4)
    m, n = len(dt), len(gt)
    o = np.zeros(m * n)
    crowd = np.zeros(max(n, 1), dtype=np.uint8)
    r.bbIou(_p(dt), _p(gt), C.c_ulong(m), C.c_ulong(n), _p(crowd), _p(o))
    return o.reshape((m, n), order="F")


def ranges(f):
    ng, nd = len(f.gt_flags), len(f.dt_flags)
    g, d = np.zeros(ng, np.uint32), np.zeros(nd, np.uint32)
    if f.kind == "lvis":
        lib().orc_lvis_ranges(C.c_int64(ng), _p(f.gt_vis), _p(f.gt_flags),
                              C.c_int64(nd), _p(f.dt_flags), _p(g), _p(d))
    else:
        lib().orc_tao_ranges(C.c_int64(ng), _p(f.gt_area), _p(f.gt_len),
                             _p(f.gt_nhp), _p(f.gt_flags), C.c_int64(nd),
                             _p(f.dt_area), _p(f.dt_len), _p(f.dt_flags),
                             _p(g), _p(d))
    return g, d


def iou_offsets(f):
    off = np.zeros(f.n_cells + 1, dtype=np.int64)
    np.cumsum(np.diff(f.cell_dt_off).astype(np.int64)
              * np.diff(f.cell_gt_off), out=off[1:])
    return off


IOU_MODES = {"3d_iou": 0, "avg_iou": 1, "imagenetvid": 2}


def track_iou(f, mode="3d_iou"):
    iou = np.zeros(int(f.cell_iou_off[-1]))
    dfb, gfb = np.ascontiguousarray(f.dt_frame_box), np.ascontiguousarray(f.gt_frame_box)
    pairs = lib().orc_track_iou(
        C.c_int64(f.n_cells), _p(f.cell_dt_off), _p(f.cell_gt_off),
        _p(f.cell_iou_off), _p(f.dt_frame_off), _p(f.dt_frame_pos),
        _p(dfb), _p(f.gt_frame_off), _p(f.gt_frame_pos),
        _p(gfb), C.c_int(IOU_MODES[mode]), _p(iou))
    return iou, int(pairs)


def match(f, gt_rng, dt_rng, iou=None, detail=True):
    n_rng = 6 if f.kind == "lvis" else 20
    n_combo = n_rng * N_THR
    n_words = (n_combo + 63) // 64
    nd = len(f.dt_flags)
    matched = np.zeros((nd, n_words), np.uint64)
    ignored = np.zeros((nd, n_words), np.uint64)
    mg = np.zeros((nd, n_combo), np.int32) if detail else None
    off = iou_offsets(f)
    ious_out = None
    if f.kind == "lvis":
        ious_out = np.zeros(int(off[-1])) if detail else None
        dbox, gbox = np.ascontiguousarray(f.dt_box), np.ascontiguousarray(f.gt_box)
        lib().orc_match(C.c_int64(f.n_cells), _p(f.cell_dt_off),
                        _p(f.cell_gt_off), _p(off), _p(dbox), _p(gbox),
                        None, C.c_int(n_rng), _p(gt_rng), _p(dt_rng),
                        _p(f.gt_flags), _p(f.dt_flags), _p(matched),
                        _p(ignored), _p(mg), _p(ious_out))
    else:
        lib().orc_match(C.c_int64(f.n_cells), _p(f.cell_dt_off),
                        _p(f.cell_gt_off), _p(off), None, None, _p(iou),
                        C.c_int(n_rng), _p(gt_rng), _p(dt_rng),
                        _p(f.gt_flags), _p(f.dt_flags), _p(matched),
                        _p(ignored), _p(mg), None)
    return matched, ignored, mg, ious_out


def accumulate(f, gt_rng, matched, ignored):
    n_rng = 6 if f.kind == "lvis" else 20
    K = len(f.cat_ids)
    nd = len(f.dt_flags)
    prec = np.zeros((N_THR, N_REC, K, n_rng))
    rec = np.zeros((N_THR, K, n_rng))
    order = np.zeros(nd, np.int64)
    num_gt = np.zeros((K, n_rng), np.int32)
    lib().orc_accumulate(C.c_int64(nd), C.c_int32(K), C.c_int(n_rng),
                         _p(f.dt_cat), _p(f.dt_score), _p(matched),
                         _p(ignored), C.c_int64(len(f.gt_flags)),
                         _p(f.gt_cat), _p(gt_rng), _p(prec), _p(rec),
                         _p(order), _p(num_gt))
    return prec, rec, order, num_gt


def run_flat(f, detail=True, iou_3d_type="3d_iou"):
    """Whole per-evaluator oracle pipeline on a flattened problem."""
    gt_rng, dt_rng = ranges(f)
    iou = pairs = None
    if f.kind == "tao":
        iou, pairs = track_iou(f, iou_3d_type)
    matched, ignored, mg, ious_out = match(f, gt_rng, dt_rng, iou, detail)
    prec, rec, order, num_gt = accumulate(f, gt_rng, matched, ignored)
    return dict(gt_rng=gt_rng, dt_rng=dt_rng,
                iou=iou if f.kind == "tao" else ious_out, pairs=pairs,
                matched=matched, ignored=ignored, match_gt=mg,
                precision=prec, recall=rec, order=order, num_gt=num_gt)


# ---------------------------------------------------------------------------
# the reference's own run-length mask code (oracle/_ref, compiled from the
# vendored maskApi.c where it lies) -- pins oracle/rle.py
# ---------------------------------------------------------------------------
class _RefRLE(C.Structure):
    _fields_ = [("h", C.c_ulong), ("w", C.c_ulong), ("m", C.c_ulong),
                ("cnts", C.POINTER(C.c_uint))]


def _ref():
    r = C.CDLL(REF_SO)
    r.rleToString.restype = C.c_void_p
    return r


def _ref_make(r, mask):
    R = _RefRLE()
    c = np.asarray(mask["counts"], dtype=np.uint32)
    r.rleInit(C.byref(R), C.c_ulong(mask["h"]), C.c_ulong(mask["w"]),
              C.c_ulong(len(c)), c.ctypes.data_as(C.POINTER(C.c_uint)))
    return R


def _ref_take(r, R):
    out = {"h": int(R.h), "w": int(R.w), "counts": [int(R.cnts[i]) for i in range(R.m)]}
    r.rleFree(C.byref(R))
    return out


def ref_rle_fr_poly(xy, h, w):
    r = _ref()
    R = _RefRLE()
    xy = np.ascontiguousarray(xy, dtype=np.float64)
    r.rleFrPoly(C.byref(R), _p(xy), C.c_ulong(len(xy) // 2), C.c_ulong(h), C.c_ulong(w))
    return _ref_take(r, R)


def ref_rle_merge(masks, intersect=False):
    r = _ref()
    arr = (_RefRLE * len(masks))(*[_ref_make(r, m) for m in masks])
    M = _RefRLE()
    r.rleMerge(arr, C.byref(M), C.c_ulong(len(masks)), C.c_int(int(intersect)))
    out = _ref_take(r, M)
    for R in arr:
        r.rleFree(C.byref(R))
    return out


def ref_rle_iou(dts, gts):
    r = _ref()
    D = (_RefRLE * len(dts))(*[_ref_make(r, m) for m in dts])
    G = (_RefRLE * len(gts))(*[_ref_make(r, m) for m in gts])
    o = np.zeros(len(dts) * len(gts))
    crowd = np.zeros(max(len(gts), 1), dtype=np.uint8)
    r.rleIou(D, G, C.c_ulong(len(dts)), C.c_ulong(len(gts)), _p(crowd), _p(o))
    for R in list(D) + list(G):
        r.rleFree(C.byref(R))
    return o.reshape((len(dts), len(gts)), order="F")


def ref_rle_to_bbox(mask):
    r = _ref()
    R = _ref_make(r, mask)
    bb = np.zeros(4)
    r.rleToBbox(C.byref(R), _p(bb), C.c_ulong(1))
    r.rleFree(C.byref(R))
    return bb.tolist()


def ref_rle_area(mask):
    r = _ref()
    R = _ref_make(r, mask)
    a = C.c_uint(0)
    r.rleArea(C.byref(R), C.c_ulong(1), C.byref(a))
    r.rleFree(C.byref(R))
    return int(a.value)


def ref_rle_to_string(mask):
    r = _ref()
    R = _ref_make(r, mask)
    ptr = r.rleToString(C.byref(R))
    s = C.string_at(ptr).decode("ascii")
    C.CDLL(None).free(C.c_void_p(ptr))
    r.rleFree(C.byref(R))
    return s


def ref_rle_fr_string(s, h, w):
    r = _ref()
    R = _RefRLE()
    r.rleFrString(C.byref(R), C.c_char_p(s.encode("ascii")), C.c_ulong(h), C.c_ulong(w))
    return _ref_take(r, R)
