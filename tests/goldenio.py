"""Load the committed golden vectors (tests/golden/<name>/)."""
import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ["f1", "f2", "f3", "f4", "f5"]
INTEGER_FIXTURES = ["f1", "f2", "f3", "f5"]
# every 3D IoU within a few ulp of a threshold: only code that adds a track
# pair's frames in the reference's CPython set order (the Python oracle with
# frame_order="set", the HIP path behind its frame-order guard) reproduces it
ADVERSARIAL_FIXTURES = ["f7"]
# a down-scaled Config 2 with decimal coordinates: tracks of hundreds of
# frames, golden 3D IoUs = the reference's set-order sums.  Inputs gzipped, the
# image level reduced to the integer match counts (fixtures.LITE)
DECIMAL_SCALE_FIXTURES = ["f8"]


def path(name, fn):
    return os.path.join(GOLDEN, name, fn)


def load_inputs(name):
    if not os.path.exists(path(name, "gt.json")):
        return (load_json_gz(name, "gt.json.gz"),
                load_json_gz(name, "pred.json.gz"))
    with open(path(name, "gt.json")) as f:
        gt = json.load(f)
    with open(path(name, "pred.json")) as f:
        pred = json.load(f)
    return gt, pred


def input_paths(name, tmp_dir):
    """Paths of plain-JSON inputs (gunzipped into tmp_dir when the fixture
    stores them compressed)."""
    out = []
    for fn in ("gt.json", "pred.json"):
        p = path(name, fn)
        if not os.path.exists(p):
            p = os.path.join(str(tmp_dir), fn)
            with gzip.open(path(name, fn + ".gz"), "rb") as src, open(p, "wb") as dst:
                dst.write(src.read())
        out.append(p)
    return out


def load_json_gz(name, fn):
    with gzip.open(path(name, fn), "rb") as f:
        return json.loads(f.read().decode())


def load_eval(name):
    z = np.load(path(name, "eval.npz"))
    out = {}
    for side in ("lvis", "tao"):
        shape = tuple(int(x) for x in z[side + "_shape"])
        k = z[side + "_valid_k"]
        p = -np.ones(shape)
        p[:, :, k] = z[side + "_precision"]
        r = -np.ones((shape[0],) + shape[2:])
        r[:, k] = z[side + "_recall"]
        out[side] = (p, r)
    return out

MODE_FIXTURES = ["f1", "f2", "f4"]
MODES = {"avg_iou": dict(iou_3d_type="avg_iou", use_cats=True),
         "imagenetvid": dict(iou_3d_type="imagenetvid", use_cats=True),
         "nocats": dict(iou_3d_type="3d_iou", use_cats=False)}


def load_modes(name):
    """Golden TaoEval outputs for the non-CLI modes: {mode: (cells, precision,
    recall, results)}."""
    z = np.load(path(name, "tao_modes.npz"))
    cells = load_json_gz(name, "tao_modes.json.gz")
    out = {}
    for m in MODES:
        shape = tuple(int(x) for x in z[m + "_shape"])
        k = z[m + "_k"]
        p = -np.ones(shape)
        p[:, :, k] = z[m + "_precision"]
        r = -np.ones((shape[0],) + shape[2:])
        r[:, k] = z[m + "_recall"]
        out[m] = ({tuple(c["key"]): np.asarray(c["ious"], dtype=float)
                   for c in cells[m]}, p, r, z[m + "_results"])
    return out


def load_lvis_nocats(name):
    """Golden LVISEval outputs with params.use_cats = 0: (cells {image_id:
    ious}, eval_imgs list, precision, recall, summarize_error)."""
    z = np.load(path(name, "lvis_nocats.npz"))
    j = load_json_gz(name, "lvis_nocats.json.gz")
    cells = {int(c["key"]): np.asarray(c["ious"], dtype=float) for c in j["cells"]}
    return cells, j["eval_imgs"], z["precision"], z["recall"], j["summarize_error"]
