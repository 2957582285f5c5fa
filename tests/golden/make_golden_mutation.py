"""Golden vectors of what the reference does to a LIST of prediction dicts
handed to its ``*Results`` constructors: it rewrites the caller's dicts in
place (lvis_amodal/results.py:39-65, tao_amodal/results.py:47-98).  Writes
tests/golden/<name>/mutated.json.gz = {"lvis": [...], "tao": [...]}, the lists
after the constructors ran.  Development container only (needs /root/reference)."""
import gzip
import json
import logging
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402
from make_golden import reference_make_track_ids_unique  # noqa: E402


def run(name):
    ref_lvis, ref_tao = refenv.import_reference()
    out = os.path.join(HERE, name)
    gt_path, pred_path = os.path.join(out, "gt.json"), os.path.join(out, "pred.json")
    lg = logging.getLogger("golden.mutation")
    lg.propagate = False
    preds_l = json.load(open(pred_path))
    ref_lvis.LVISResults(ref_lvis.LVIS(gt_path), preds_l)
    preds_t = json.load(open(pred_path))
    reference_make_track_ids_unique()(preds_t)
    ref_tao.TaoResults(ref_tao.Tao(gt_path), preds_t)
    with gzip.open(os.path.join(out, "mutated.json.gz"), "wt") as f:
        json.dump({"lvis": preds_l, "tao": preds_t}, f)
    print(name, len(preds_l), sum("id" in p for p in preds_l), sum("id" in p for p in preds_t))


if __name__ == "__main__":
    for n in sys.argv[1:] or ["f1", "f2", "f4"]:
        run(n)
